cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
for seed in ${SEEDS:-1 2 3 4 5 6 7 8 9 10 11 12}; do
  NUFFT_TEST_SEED=$seed timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "randomised" 2>&1 | grep -E "passed|failed|assert|Error" | tail -3 | sed "s/^/seed $seed: /"
done
