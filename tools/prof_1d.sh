cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/p1d; mkdir -p $O
for cfg in "type_1 1048576 1e7 1e-6 t1_big" "type_1 4096 1e7 1e-6 t1_small" "type_2 1048576 1e7 1e-6 t2_big" "type_2 4096 1e7 1e-6 t2_small"; do
  set -- $cfg
  rm -rf $O/p_$5
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 6 $EXTRA_ARGS > $O/p_$5.log 2>&1
  echo "== $5"; python3 tools/kstats.py $O/p_$5 7
done
