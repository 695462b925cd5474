cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
for cfg in "type_1 256,256 2e5 1e-6 c4" "type_1 128,128,128 8e5 1e-6 c8"; do
  set -- $cfg
  rm -rf $O/p_$5
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 10 --one-call > $O/p_$5.log 2>&1
  echo "== $5"; python3 tools/kstats.py $O/p_$5 12
done
