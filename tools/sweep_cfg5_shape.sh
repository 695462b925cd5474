cd $GRAFT_REPO_ROOT
for shape in 12x64 8x64 16x64; do
  echo -n "shape $shape: "
  NUFFT_HIP_W8_SHAPE=$shape NUFFT_HIP_OP_GROUP=8 python3 tools/bench_configs.py 5op 2>&1 | tail -1
done
