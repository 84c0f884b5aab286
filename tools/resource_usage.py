"""Register / scratch / LDS table of every kernel in libiivision.so, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage), one row per kernel instantiation:
    python tools/resource_usage.py > profiles/r06_resource_usage.txt
Runs on the CPU (hipcc cross-compiles gfx950); compiles each csrc/*.hip with the Makefile's flags into /tmp."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ii-vision_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None".split()
FIELDS = [("SGPRs", r"TotalSGPRs: (\d+)"), ("VGPRs", r" VGPRs: (\d+)"), ("AGPRs", r"AGPRs: (\d+)"), ("scratch B/lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
          ("occupancy waves/SIMD", r"Occupancy \[waves/SIMD\]: (\d+)"), ("SGPR spill", r"SGPRs Spill: (\d+)"), ("VGPR spill", r"VGPRs Spill: (\d+)"),
          ("LDS B/block", r"LDS Size \[bytes/block\]: (\d+)")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    res = []
    for n in out[:len(names)]:
        n = re.sub(r"\(.*$", "", n)          # arguments off
        n = re.sub(r"^void ", "", n).replace("iiv::", "")
        res.append(n)
    return res


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    extra = sys.argv[1:]
    rows = []
    for s in srcs:
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, s), "-o", "/tmp/ru_%s.o" % s],
                           capture_output=True, text=True, cwd=CSRC)
        if p.returncode:
            sys.stderr.write(p.stderr)
            raise SystemExit("compile of %s failed" % s)
        blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
        for b in blocks:
            name = b.split()[0]
            vals = []
            for _, rx in FIELDS:
                m = re.search(rx, b)
                vals.append(int(m.group(1)) if m else -1)
            rows.append((s, name, vals))
    names = demangle([r[1] for r in rows])
    print("# kernel resource usage, gfx950, flags: %s %s" % (" ".join(FLAGS), " ".join(extra)))
    print("# (compiler remarks: hipcc -Rpass-analysis=kernel-resource-usage; tools/resource_usage.py)")
    hdr = "%-16s %-64s" % ("file", "kernel") + "".join(" %8s" % h.split()[0] for h, _ in FIELDS)
    print(hdr)
    print("%-16s %-64s" % ("", "") + "".join(" %8s" % (" ".join(h.split()[1:]) or "")[:8] for h, _ in FIELDS))
    scratch = []
    for (s, _, vals), n in zip(rows, names):
        print("%-16s %-64s" % (s, n[:64]) + "".join(" %8d" % v for v in vals))
        if vals[3] > 0:
            scratch.append((n, vals[3]))
    print("# kernels with scratch: %s" % (", ".join("%s (%d B/lane)" % x for x in scratch) or "none"))


if __name__ == "__main__":
    main()
