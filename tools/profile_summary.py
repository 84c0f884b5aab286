"""Collect rocprofv3 kernel stats + PMC counters for bench.py and write compact
summaries (the raw traces are tens of MB and stay on the GPU box).

    python tools/profile_summary.py OUTDIR [bench args...]

Runs, each as its own rocprofv3 invocation (counters are never combined with
tracing domains other than --kernel-trace):
  1. --kernel-trace --stats
  2-5. four --pmc passes (SQ instruction mix, SQ waits, FETCH_SIZE, WRITE_SIZE/TCC)
and writes OUTDIR/kernel_stats.csv, OUTDIR/pmc_summary.txt, OUTDIR/pmc_latest.json.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM",
    "FETCH_SIZE GRBM_GUI_ACTIVE",
    "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
]


# tools/issue_probe.hip on an MI355X (profiles/r06_issue_probe.txt): shader clocks of one SIMD per wave64 instruction at
# w resident waves per SIMD -- the greedy step's 161 : 118 vector : scalar mix interleaved, and general vector instructions
# alone (v_mad / v_min / DPP / v_readlane / v_cndmask ...; v_add_u32 alone issues faster)
PROBE_MIX_CLK = {1: 4.71, 2: 2.94, 3: 1.96, 4: 1.99, 6: 1.65, 7: 1.41, 8: 1.49}
PROBE_VALU_CLK = {1: 4.14, 2: 4.08, 3: 2.72, 4: 3.04, 6: 2.69, 7: 2.31, 8: 2.52}


def build_id():
    sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
    try:
        import _iiv_native
        return _iiv_native.build_id()
    except Exception:
        return "unknown"


def run(cmd, cwd):
    subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)


def short(name):
    """`void iiv::greedy_wave_kernel<1, 8, false>(...)` -> `greedy_wave_kernel<1, 8, false>`: the template arguments stay, so that
    the one-wave kernel's two forms (W = 1: plain; W = 8 / 16: LDS-shared) and its fourth-offset variants keep their own counters
    -- pmc_latest.json must hold the form that ran, not a mixture.  First argument: 1 = DHGR, 0 = HGR."""
    import re
    m = re.search(r"(greedy_wave_kernel|greedy_team_kernel|greedy_kernel|prologue_kernel|table_kernel|store_kernel)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def mode_of(key):
    """'DHGR' / 'HGR' of a key made by short(), from its first template argument."""
    import re
    m = re.search(r"<\s*(\d)", key)
    return None if not m else ("DHGR" if m.group(1) == "1" else "HGR")


def main():
    out = os.path.abspath(sys.argv[1])
    bench_args = sys.argv[2:] or ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    os.makedirs(out, exist_ok=True)
    tmp = "/tmp/iiv_prof"
    subprocess.run(["rm", "-rf", tmp])
    os.makedirs(tmp)
    bench = ["python3", os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp + "/stats", "--"] + bench,
                       cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=False)
    bench_line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), "")
    if bench_line:
        open(os.path.join(out, "bench_under_rocprof.json"), "w").write(bench_line + "\n")
    f = glob.glob(tmp + "/stats/*/*kernel_stats.csv")
    avg_ns = {}   # short kernel name -> average duration of its launches in the --stats run
    if f:
        rows = list(csv.reader(open(f[0])))
        try:
            i_name, i_avg = rows[0].index("Name"), rows[0].index("AverageNs")
            for r in rows[1:]:
                if short(r[i_name]):
                    avg_ns[short(r[i_name])] = float(r[i_avg])
        except ValueError:
            pass
        with open(os.path.join(out, "kernel_stats.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(rows[0])
            for r in rows[1:]:
                w.writerow([r[0] if len(r[0]) <= 160 else r[0][:157] + "..."] + r[1:])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for i, p in enumerate(PASSES):
        d = "%s/pmc%d" % (tmp, i)
        run(["rocprofv3", "--pmc"] + p.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + bench, "/tmp")
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    with open(os.path.join(out, "pmc_summary.txt"), "w") as g:
        g.write("rocprofv3 --pmc, %d separate passes with --kernel-trace only; bench args: %s\n" % (len(PASSES), " ".join(bench_args)))
        g.write("values are means per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as reported\n\n")
        for k in sorted(agg):
            g.write("%s\n" % k)
            for c in sorted(agg[k]):
                n = max(len(cnt[k][c]), 1)
                g.write("   %-24s %.5g   (%d dispatches)\n" % (c, agg[k][c] / n, n))
            g.write("\n")
    # HBM bytes per greedy launch and PER STREAM, per mode, merged into OUTDIR/pmc_latest.json (bench.py multiplies
    # by its own stream count, so roofline.traffic and algorithmic_bytes_per_launch share mode and streams)
    streams, ops_per_launch = None, None
    try:
        bj = json.loads(bench_line)
        streams = int(bj["config"]["streams_per_gpu"])
        ops_per_launch = bj["roofline"]["algorithmic_bytes_per_launch"] / streams / 534.0   # opcodes per stream and greedy launch (bench.BYTES_PER_OPCODE)
    except Exception:
        pass

    def per_dispatch(k, c):
        return agg[k][c] / max(len(cnt[k][c]), 1) if c in agg[k] else None

    def issue_of(k):
        """What the SQ counters say binds the kernel: instructions per opcode and wave,
        instructions per clock and SIMD, and the ISSUE FLOOR of that instruction stream at its residency, from the
        register-only probe tools/issue_probe.hip (profiles/r06_issue_probe.txt): clocks of one SIMD per wave64
        instruction at w resident waves per SIMD -- the step's own vector : scalar mix, and vector instructions alone.
        One wave per stream; 256 CUs x 4 SIMDs; shader clock 2.4 GHz (the probe read 2.35-2.40 GHz under load)."""
        valu, salu = per_dispatch(k, "SQ_INSTS_VALU"), per_dispatch(k, "SQ_INSTS_SALU")
        if valu is None or salu is None or not streams or not ops_per_launch or k not in avg_ns:
            return None
        other = sum(per_dispatch(k, c) or 0.0 for c in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
        steps = streams * ops_per_launch
        clocks = avg_ns[k] * 1e-9 * 2.4e9
        import re
        w = int(re.search(r"<\s*\d\s*,\s*(\d+)", k).group(1)) if re.search(r"<\s*\d\s*,\s*(\d+)", k) else 1
        waves = per_dispatch(k, "SQ_WAVES")
        # persistent workgroups (W > 1): every launched wave is resident; the plain form (W = 1) launches one wave per stream, 28 resident per CU
        wps = (waves / 1024.0) if (w > 1 and waves) else 7.0
        wkey = min(PROBE_MIX_CLK, key=lambda x: abs(x - wps))
        mix_clk, valu_clk = PROBE_MIX_CLK[wkey], PROBE_VALU_CLK[wkey]
        step_clocks = clocks * 1024.0 / steps                      # clocks of one SIMD per step it retires
        floor_clocks = (valu + salu) / steps * mix_clk            # what issuing the step's vector + scalar instructions takes it
        return {
            "source": "tools/profile_summary.py: rocprofv3 --pmc SQ_* passes + --kernel-trace --stats of the same command; issue rates: "
                      "profiles/r06_issue_probe.txt (tools/issue_probe.hip, register-only loops)",
            "kernel": k, "launch_ms_under_rocprof": avg_ns[k] * 1e-6, "opcodes_per_stream_and_launch": ops_per_launch,
            "valu_per_opcode_wave": valu / steps, "salu_per_opcode_wave": salu / steps, "other_per_opcode_wave": other / steps,
            "instr_per_clk_per_simd": (valu + salu + other) / (clocks * 1024.0),
            "waves_per_dispatch": waves,
            "waves_per_simd": wps,
            "probe_waves_per_simd_used": wkey,
            "probe_clocks_per_instr_step_mix": mix_clk,
            "probe_clocks_per_instr_valu": valu_clk,
            # the vector pipe's share of the launch at the probed rate for this residency (one figure: the probe settles it)
            "valu_busy_frac": valu * valu_clk / (clocks * 1024.0),
            "clocks_per_step_and_simd": step_clocks,
            "issue_floor_clocks_per_step_and_simd": floor_clocks,
            "issue_floor_frac": floor_clocks / step_clocks,
            "reading": "issue_floor_frac of the launch is the SIMDs issuing the step's vector and scalar instructions at the rate a "
                       "register-only loop of the same mix sustains at this residency; the rest is dependent latency (DPP / SGPR "
                       "round trips, LDS and table-word waits) that the resident waves do not cover",
            "wait_any_frac_of_wave_cycles": (per_dispatch(k, "SQ_WAIT_ANY") / per_dispatch(k, "SQ_WAVE_CYCLES")) if per_dispatch(k, "SQ_WAVE_CYCLES") and per_dispatch(k, "SQ_WAIT_ANY") is not None else None,
            "clock_hz_assumed": 2.4e9,
        }
    # (IIV_PMC_LATEST_OUT: one file that several runs -- DHGR, HGR, S-img -- merge their entries into: tools/round_evidence.sh)
    latest_path = os.environ.get("IIV_PMC_LATEST_OUT") or os.path.join(out, "pmc_latest.json")
    bid = build_id()
    try:
        latest = json.load(open(latest_path))
        if not any(k.split(":")[0] in ("DHGR", "HGR") for k in latest):
            latest = {}
    except Exception:
        latest = {}
    for mode in ("DHGR", "HGR"):
        ks = [k for k in agg if k.startswith("greedy") and mode_of(k) == mode and "FETCH_SIZE" in agg[k]]
        if not ks or not streams:
            continue
        main_k = max(ks, key=lambda k: len(cnt[k]["FETCH_SIZE"]))
        a = agg[main_k]
        fetch = a["FETCH_SIZE"] / max(len(cnt[main_k]["FETCH_SIZE"]), 1)
        write = a["WRITE_SIZE"] / max(len(cnt[main_k]["WRITE_SIZE"]), 1)
        # the key names mode and synthetic input: "DHGR" / "HGR" are S-iid, "DHGR:img" is S-img (bench._pmc_entry)
        kind = "img" if "--img" in bench_args else "coherent" if "--coherent" in bench_args else "static" if "--static" in bench_args else "iid"
        latest[mode if kind == "iid" else "%s:%s" % (mode, kind)] = {
            "source": "tools/profile_summary.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
            "build_id": bid,   # iiv_version() of the library the counters were taken with: bench.py quotes them for this build only
            "kernel": main_k,
            "kernel_note": "template arguments: <mode (1 = DHGR), streams per workgroup (1 = plain form, 8 / 16 = LDS-shared form), fourth offset>; "
                           "the instantiation with the most dispatches in the profiled run",
            "bench_args": bench_args,
            "streams": streams,
            "fetch_size_kib_per_launch_raw": fetch,
            "write_size_kib_per_launch_raw": write,
            "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, "
                          "MI355X_MICROARCH.md HBM section; uncalibrated for 2-byte gathers); WRITE_SIZE as reported",
            "greedy_hbm_bytes_per_launch": (2 * fetch + write) * 1024,
            "greedy_hbm_bytes_per_launch_per_stream": (2 * fetch + write) * 1024 / streams,
            "issue": issue_of(main_k),
        }
        # the prologue kernel of the same run (same correction: its reads are wide coalesced ones)
        pk = [k for k in agg if k.startswith("prologue") and mode_of(k) == mode and "FETCH_SIZE" in agg[k]]
        if pk:
            pm = max(pk, key=lambda k: len(cnt[k]["FETCH_SIZE"]))
            pf = agg[pm]["FETCH_SIZE"] / max(len(cnt[pm]["FETCH_SIZE"]), 1)
            pw = agg[pm]["WRITE_SIZE"] / max(len(cnt[pm]["WRITE_SIZE"]), 1)
            key = mode if kind == "iid" else "%s:%s" % (mode, kind)
            latest[key].update({"prologue_kernel": pm, "prologue_fetch_size_kib_per_launch_raw": pf,
                                "prologue_write_size_kib_per_launch_raw": pw,
                                "prologue_hbm_bytes_per_launch_per_stream": (2 * pf + pw) * 1024 / streams})
    json.dump(latest, open(latest_path, "w"), indent=1)

if __name__ == "__main__":
    main()
