#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/exp_r05j_bench.json 2> gpurun_out/exp_r05j_bench.err
tail -5 gpurun_out/exp_r05j_bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/exp_r05j_bench.json') if l.startswith('{')][-1])
def short(o, d=0):
    if isinstance(o, dict):
        return {k: short(v, d+1) for k, v in o.items() if k not in ('what','note','sample','provenance','traffic_source','reference_source','workload','frac_is','source')} if d < 3 else '...'
    if isinstance(o, float): return round(o, 4)
    return o
print(json.dumps(short(j), indent=1))
PY
