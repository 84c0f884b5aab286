#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
{
echo "== longest-first launch order, same library, on / off"
for a in "" "--img --img-distinct 2048" "--coherent" "--mode HGR" "--mode HGR --img --img-distinct 2048"; do
  for o in "" "--no-stream-order" "" "--no-stream-order"; do
    python bench.py --steps 6 --warmup 2 --no-extras --no-emit --no-cpu-baseline $a $o 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-40s %-18s %10.0f fps  greedy %.4f ms  prologue %.4f ms  form %s' % ('$a', '$o' or 'longest first', j['value'], j['roofline']['avg_launch_ms'], j['roofline_prologue']['avg_launch_ms'], j['config'].get('greedy_form')))"
  done
done
python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "longest or bench_batch" 2>&1 | tail -3
} > gpurun_out/exp_r05h.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcd1 /tmp/pmcd2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmcd1 -- python3 $ROOT/tools/ingest_probe.py 102400 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d /tmp/pmcd2 -- python3 $ROOT/tools/ingest_probe.py 102400 > /dev/null 2>&1
python3 - >> $ROOT/gpurun_out/exp_r05h.txt <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for d in ('/tmp/pmcd1','/tmp/pmcd2'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:50]
            if 'ingest' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
print("== ingest kernels, counters per dispatch (102400 frames)")
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-24s %.5g (%d)'%(c,agg[k][c]/max(len(cnt[k][c]),1),len(cnt[k][c])))
PY
cat $ROOT/gpurun_out/exp_r05h.txt
