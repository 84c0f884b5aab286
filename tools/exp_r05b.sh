#!/bin/bash
# round-5 experiment B: which counters exist; instruction-fetch / branch / scalar-side counters of the greedy kernel
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $ROOT/gpurun_out/exp_r05b_counters.txt 2>&1
ARGS="--steps 2 --warmup 1 --no-extras --no-cpu-baseline --greedy shared"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_VALU SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/pmcb$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcb$i -- python3 $ROOT/bench.py $ARGS > /tmp/pmcb$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmcb$i.log; }
done
python3 - > $ROOT/gpurun_out/exp_r05b_pmc.txt <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for d in glob.glob('/tmp/pmcb[0-9]'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'prologue' in k or 'greedy' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-24s %.5g (%d)'%(c,agg[k][c]/max(len(cnt[k][c]),1),len(cnt[k][c])))
PY
cat $ROOT/gpurun_out/exp_r05b_pmc.txt
grep -c . $ROOT/gpurun_out/exp_r05b_counters.txt
