#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SEEDS="${SEEDS:-101 102 103 104 105 106 107 108 109 110 111 112 113 114 115 116}" bash tools/soak.sh 2>&1 | tee gpurun_out/soak_r03.txt
