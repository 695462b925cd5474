#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "radial_trajectories_total or config5_batched" --durations=3 2>&1 | tail -8
cat gpurun_out/full_size_parity.txt
