cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/cellsort3d_threshold.txt
: > $O
for tol in 1e-6 1e-4; do
for M in 3e6 1e7 2e7 3e7 4e7 6e7; do
  for t in CELLSORT3D_OFF CELLSORT3D_ON; do
    echo "$t: $(python tools/stage_times.py type_2 256,256,256 $M $tol $t 2>&1 | tail -1)" | tee -a $O
  done
done
done
