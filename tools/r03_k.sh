#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/phase_log_experiment.sh 2>&1 | tail -18
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_us'], d['roofline']['kernel_ms'])"; done
python3 tools/bench_configs.py 5op 5 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests -m gpu -x -q -k "golden or w8 or headline or reuse or stages" 2>&1 | tail -2
