# quick instruction-mix counters of the encode kernels: bash tools/pmc_quick.sh [bench args...]   (two rocprofv3 --pmc passes)
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---steps 2 --warmup 1 --no-extras --no-cpu-baseline}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for d in ('/tmp/pmc1','/tmp/pmc2'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:64]
            if 'prologue' in k or 'greedy' in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-24s %.5g (%d)'%(c,agg[k][c]/max(len(cnt[k][c]),1),len(cnt[k][c])))
PY
