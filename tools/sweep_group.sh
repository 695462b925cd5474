#!/bin/bash
# config 5 through tfft.nufft (32 items of 512^2, M = 1e6, per-item points) by op-level
# group size (BENCH_OP_GROUP: point sets per plan call) and lane count
cd $GRAFT_REPO_ROOT
for lanes in 2 1; do
for grp in 1 4 8 16 32; do
  echo -n "group $grp lanes $lanes: "
  BENCH_OP_GROUP=$grp BENCH_OP_LANES=$lanes python3 tools/bench_configs.py 5op 2>&1 | tail -1
done
done
