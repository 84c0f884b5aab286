#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ingest.py tests/test_gpu_bench_launcher.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/exp_r05l.txt
(python tools/ingest_probe.py 102400; python tools/ingest_probe.py 102400 HGR) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ingest_probe.txt
cat gpurun_out/r05_ingest_probe.txt >> gpurun_out/exp_r05l.txt
IIV_BENCH_REHEARSE_ON_ONE_GPU=1 python bench.py --gpus 2 --streams 3584 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r05_rehearse_2ranks_one_gpu.json 2> gpurun_out/r05_rehearse.err
IIV_BENCH_REHEARSE_ON_ONE_GPU=1 python bench.py --gpus 4 --streams 1792 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r05_rehearse_4ranks_one_gpu.json 2>> gpurun_out/r05_rehearse.err
python bench.py --gpus 2 > /dev/null 2> gpurun_out/r05_refuse_gpus2_on_one_gpu.txt; echo "exit code $?" >> gpurun_out/r05_refuse_gpus2_on_one_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
python - >> gpurun_out/exp_r05l.txt <<'PY'
import json
for f in ('r05_rehearse_2ranks_one_gpu','r05_rehearse_4ranks_one_gpu'):
    j=json.loads([l for l in open('gpurun_out/%s.json'%f) if l.startswith('{')][-1])
    print(f, j['n_gpus'], j['world_size'], j['dist_backend'], j['launcher'], j['per_rank_stream_seeds'], round(j['value']))
j=json.loads([l for l in open('gpurun_out/r05_bench.json') if l.startswith('{')][-1])
print('bench', round(j['value']), j['roofline']['bound'], j['roofline']['issue'] and {k: (round(v,3) if isinstance(v,float) else v) for k,v in j['roofline']['issue'].items() if k not in ('note','source')})
print('ingest', {k: (round(v['value']) if isinstance(v, dict) and v.get('value') else v) for k,v in j['ingest'].items() if isinstance(v, dict)}, j['ingest']['e2e'].get('vs_emit'), j['ingest']['e2e'].get('vs_emit_same_content'))
print('legs', {k: round(j[k]['value']) for k in ('hgr','img','fourth_offset','joint','emit','single_stream') if k in j})
PY
cat gpurun_out/exp_r05l.txt; cat gpurun_out/r05_refuse_gpus2_on_one_gpu.txt | tail -3
