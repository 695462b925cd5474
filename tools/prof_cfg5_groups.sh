cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
for grp in 8 16; do
rm -rf $O/prof_g$grp
BENCH_OP_GROUP=$grp BENCH_OP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_g$grp -o run --output-format csv -- python3 tools/bench_configs.py 5op > $O/prof_g$grp.log 2>&1
echo "== group $grp"; python3 tools/kstats.py $O/prof_g$grp 14
done
