#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export IIV_PROBE_NOISE=1
rm -rf /tmp/pmcd1 /tmp/pmcd2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmcd1 -- python3 $ROOT/tools/ingest_probe.py 25600 > /tmp/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmcd2 -- python3 $ROOT/tools/ingest_probe.py 25600 > /tmp/p2.log 2>&1
python3 - > $ROOT/gpurun_out/exp_r05i.txt <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for d in ('/tmp/pmcd1','/tmp/pmcd2'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'ingest' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
print("== ingest kernels, counters per dispatch (25600 frames of noise)")
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-24s %.5g (%d)'%(c,agg[k][c]/max(len(cnt[k][c]),1),len(cnt[k][c])))
PY
tail -3 /tmp/p1.log >> $ROOT/gpurun_out/exp_r05i.txt
cat $ROOT/gpurun_out/exp_r05i.txt
