"""Extra rocprofv3 --pmc passes (issue / fetch / L1 path) for the encoder kernels.

    python tools/pmc_extra.py OUTFILE [bench args...]

Each pass is its own rocprofv3 run with --kernel-trace only; a pass whose counters the device
does not expose is reported as failed and skipped."""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    "SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES",
    "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_CYCLES SQ_BUSY_CU_CYCLES",
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum",
    "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum",
    "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum",
    "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES",
]


def short(name):
    for k in ("greedy_wave_kernel", "greedy_lds_kernel", "greedy_team_kernel", "greedy_kernel", "prologue_kernel"):
        if k in name:
            return k + ("<DHGR>" if "<1" in name else "<HGR>" if "<0" in name else "")
    return None


def main():
    out = os.path.abspath(sys.argv[1])
    bench_args = sys.argv[2:] or ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    tmp = "/tmp/iiv_prof_extra"
    subprocess.run(["rm", "-rf", tmp])
    os.makedirs(tmp)
    bench = ["python3", os.path.join(ROOT, "bench.py")] + bench_args
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    failed = []
    for i, p in enumerate(PASSES):
        d = "%s/pmc%d" % (tmp, i)
        try:   # (a pass the device cannot schedule may hang: bound it)
            subprocess.run(["timeout", "150", "rocprofv3", "--pmc"] + p.split() +
                           ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + bench,
                           cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=170)
        except subprocess.TimeoutExpired:
            pass
        files = glob.glob(d + "/*/*counter_collection.csv")
        if not files:
            failed.append(p)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    with open(out, "w") as g:
        g.write("rocprofv3 --pmc extra passes (each its own run, --kernel-trace only); bench args: %s\n" % " ".join(bench_args))
        g.write("values are means per dispatch\n")
        for p in failed:
            g.write("FAILED PASS: %s\n" % p)
        for k in sorted(agg):
            g.write("\n%s\n" % k)
            for c in sorted(agg[k]):
                n = max(len(cnt[k][c]), 1)
                g.write("   %-40s %.5g   (%d dispatches)\n" % (c, agg[k][c] / n, n))


if __name__ == "__main__":
    main()
