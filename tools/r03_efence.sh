#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# the whole GPU suite with every plan buffer ending at an unmapped page (debug allocator); the three M = 1e8 tests
# are left out (minutes of page-granular allocations), their kernels run at smaller sizes in the others
NUFFT_HIP_DEBUG_EFENCE=1 timeout 2400 python -m pytest tests -m gpu -x -q -k "not config4_full_size and not config4_total_parity_at_full" > gpurun_out/efence_r03.txt 2>&1
tail -5 gpurun_out/efence_r03.txt
