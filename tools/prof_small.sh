cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/small; mkdir -p $O
for cfg in "type_1 256,256 2e5 1e-6 t1" "type_2 256,256 2e5 1e-6 t2"; do
  set -- $cfg
  rm -rf $O/p_$5
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 20 --one-call > $O/p_$5.log 2>&1
  echo "== $5"; python3 tools/ktimeline.py $O/p_$5 18
done
