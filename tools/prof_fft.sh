cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
for ty in type_1 type_2; do
rm -rf $O/p_$ty
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$ty -o run --output-format csv -- python3 tools/profile_run.py --steps 3 --type $ty > $O/p_$ty.log 2>&1
echo "== $ty"; python3 tools/ktrace.py $O/p_$ty fft_rotate | tail -4
done
