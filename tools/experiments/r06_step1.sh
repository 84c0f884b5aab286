#!/bin/bash
# round 6, first GPU call: issue probe; HGR prologue on pair terms (parity + timing)
out=gpurun_out
mkdir -p $out
timeout 300 tools/issue_probe > $out/r06_issue_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_gpu_encode.py tests/test_gpu_configs.py -x -q -m gpu > $out/r06_step1_tests.txt 2>&1
tail -3 $out/r06_step1_tests.txt
timeout 600 python bench.py --mode HGR --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $out/r06_step1_hgr.json 2> $out/r06_step1_hgr.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r06_step1_hgr.json').read().strip().splitlines()[-1])
print('HGR', j['value'], 'greedy ms', j['roofline']['avg_launch_ms'], 'prologue ms', j['roofline_prologue']['avg_launch_ms'])
PY
cat $out/r06_issue_probe.txt
