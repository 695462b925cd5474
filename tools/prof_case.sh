# rocprofv3 kernel stats of one case: bash tools/prof_case.sh "<profile_run.py args>" tag [rows]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/case; mkdir -p $O
rm -rf $O/p_$2
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$2 -o run --output-format csv -- python3 tools/profile_run.py $1 --steps 10 > $O/p_$2.log 2>&1
echo "== $2"; python3 tools/kstats.py $O/p_$2 ${3:-9}
