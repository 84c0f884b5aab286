#!/bin/bash
# quick instruction-mix counters of the encode kernels: bash tools/pmc_quick.sh [bench args...]   (two rocprofv3 --pmc passes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1 /tmp/pmc2
ARGS="${@:---steps 2 --warmup 1 --no-extras --no-cpu-baseline}"
run_pass() {   # directory, counters...
  local d=$1; shift
  if ! rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$d" -- python3 "$ROOT/bench.py" $ARGS > "$d.log" 2>&1; then
    echo "pmc_quick: bench.py under rocprofv3 failed (see $d.log)" >&2; tail -5 "$d.log" >&2; exit 1
  fi
  if ! ls "$d"/*/*counter_collection.csv > /dev/null 2>&1; then
    echo "pmc_quick: no counter CSV under $d" >&2; exit 1
  fi
}
run_pass /tmp/pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass /tmp/pmc2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
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
