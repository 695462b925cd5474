cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide; mkdir -p $O
for cfg in "type_1 1024,1024 1e7 1e-9 w10_2d" "type_1 1024,1024 1e7 1e-12 w13_2d" "type_1 128,128,128 1e7 1e-9 w10_3d" "type_1 128,128,128 1e7 1e-12 w13_3d" "type_2 1024,1024 1e7 1e-12 w13_2d_t2" "type_2 128,128,128 1e7 1e-12 w13_3d_t2"; do
  set -- $cfg
  rm -rf $O/p_$5
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 6 --double > $O/p_$5.log 2>&1
  echo "== $5"; python3 tools/kstats.py $O/p_$5 8
done
