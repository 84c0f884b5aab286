#!/bin/bash
# instruction-mix counters of the iiv kernels of ANY python command: bash tools/pmc_cmd.sh KERNEL_SUBSTRING script.py [args...]
# (two rocprofv3 --pmc passes with --kernel-trace only; python3 directly behind --)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcc1 /tmp/pmcc2
run_pass() {
  local d=$1; shift
  if ! rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$d" -- python3 $CMD > "$d.log" 2>&1; then
    echo "pmc_cmd: the command failed under rocprofv3 (see $d.log)" >&2; tail -5 "$d.log" >&2; exit 1
  fi
}
CMD=""
for a in "$@"; do case "$a" in /*) CMD="$CMD $a";; *) if [ -e "$ROOT/$a" ]; then CMD="$CMD $ROOT/$a"; else CMD="$CMD $a"; fi;; esac; done
run_pass /tmp/pmcc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass /tmp/pmcc2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
PAT="$pat" python3 - <<'PY'
import csv,glob,collections,os
pat=os.environ["PAT"]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for d in ('/tmp/pmcc1','/tmp/pmcc2'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:72]
            if pat in k:
                agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-24s %.5g (%d)'%(c,agg[k][c]/max(len(cnt[k][c]),1),len(cnt[k][c])))
PY
