#!/bin/bash
# round-5: new ingest kernels -- parity tests, then the ingest / e2e legs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ingest.py tests/test_gpu_transcode_tool.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/exp_r05c_tests.txt
cat gpurun_out/exp_r05c_tests.txt
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --extras emit,ingest > gpurun_out/exp_r05c_bench.json 2> gpurun_out/exp_r05c_bench.err
tail -3 gpurun_out/exp_r05c_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/exp_r05c_bench.json').read().strip().splitlines()[-1])
print('value', j['value'], 'emit', j.get('emit',{}).get('value'))
print(json.dumps(j.get('ingest'), indent=1))
PY
