"""Diagnostic: where a wave of greedy_wave_kernel spends its shader clocks (-DIIV_STAMPS build):
    make -C ii-vision_amd/csrc ../libiivision_stamps.so
    IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/greedy_phases.py [streams]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette
S = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
KIND = sys.argv[2] if len(sys.argv) > 2 else "iid"      # iid | coh | img
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.DHGR
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
if KIND == "img":
    fm, fa = stream_batch.synth_frames_img(S, 6, True, seed=5)
else:
    fm, fa = stream_batch.synth_frames_torch(S, 6, True, seed=5, coherent=KIND == "coh")
b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 1) for i in range(S)], dm=dm)
b.enc.set_greedy_kernel(sys.argv[3] if len(sys.argv) > 3 else True)   # True | "shared" | "plain"
b.encode_frames(fm, fa, 4)
b.enc.encode(fm, fa, [(4, 0, 1, 292)])     # the measured launch: 292 opcodes per stream
b.enc.check()
full = np.stack([b.enc.get_state(100, i) for i in range(S)]).astype(np.int64)
allst = full[::max(1, S // 512)]
rows, t0, t1, ops = allst[:, 16:20], allst[:, 24], allst[:, 25], allst[:, 26]
print("S-%s: %.1f list entries through the pipeline per opcode emitted (the rest were dead when their turn came)" % (
    KIND, allst[:, 27].sum() / max(ops.sum(), 1)))
span = t1.max() - t0.min()
life = t1 - t0
print("S=%d: wave lifetimes mean %.0f  min %.0f  max %.0f clocks; launch span %.0f -> mean lifetime / span %.2f" % (
    S, life.mean(), life.min(), life.max(), span, life.mean() / span))
print("        resident waves per CU (sum of wave lifetimes / launch span / 256 CUs, over all %d streams): %.1f" % (
    S, (full[:, 25] - full[:, 24])[full[:, 24] > 0].sum() / (full[:, 25].max() - full[:, 24][full[:, 24] > 0].min()) / 256) + "  (%d streams without stamps)" % int((full[:, 24] == 0).sum()))
print("        starts: first %.0f, median %.0f, last %.0f clocks after the first wave" % (
    0, np.median(t0 - t0.min()), (t0 - t0.min()).max()))
names = ["8 loads + take + row request", "MT19937 generation", "table-word wait + score + apply", "waiting for the row (requested two entries earlier)"]
tot = rows.sum(axis=1).mean()
print("        clocks per opcode %.0f:" % (tot / ops.mean()))
for n, v in zip(names, rows.mean(axis=0)):
    print("          %-44s %7.0f  (%4.1f %%)" % (n, v / ops.mean(), 100 * v / tot))
# initial residency: s_memtime counters are per XCD (far apart), so cluster the start times by XCD, and count the
# waves of each cluster that start before any wave of the cluster could have finished
t0a, t1a = full[:, 24], full[:, 25]
order = np.argsort(t0a)
gaps = np.nonzero(np.diff(t0a[order]) > 10 * (t1a - t0a).max())[0]
groups = np.split(order, gaps + 1)
res = []
for gidx in groups:
    first_end = t1a[gidx].min()
    res.append(int((t0a[gidx] < first_end).sum()))
print("        %d clock domains (XCDs); waves started before the first one of their XCD ended: %s -> %.1f per CU" % (
    len(groups), res, sum(res) / 256.0))
b.enc.profile(True)
b.enc.encode(fm, fa, [(5, 0, 1, 292)])
torch.cuda.synchronize()
pr = b.enc.profile_read()
real_us = (full[:, 30] - full[:, 29]).mean() / 100.0
print("        a second 292-opcode launch: greedy %.3f ms; mean wave lifetime = %.0f clocks = %.1f us (s_memrealtime) -> shader clock %.2f GHz; "
      "waves resident per CU = streams x lifetime / (launch time x 256) = %.1f" % (
          pr["greedy_ms"], life.mean(), real_us, life.mean() / real_us / 1e3, S * real_us / (pr["greedy_ms"] * 1e3 * 256)))
hw = full[:, 28]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
wave_id, simd, cu, sh, se = hwid & 0xf, (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
slots = {}
for k, sd, w in zip(cu_key, simd, wave_id):
    slots.setdefault(int(k), set()).add((int(sd), int(w)))
cnt = np.array([len(v) for v in slots.values()])
print("        HW_ID: %d CUs seen; distinct (SIMD, wave slot) pairs used per CU over the launch: min %d  median %d  max %d; streams per CU min %d max %d" % (
    len(slots), cnt.min(), np.median(cnt), cnt.max(), np.bincount(cu_key).min() if len(slots) else 0, np.bincount(cu_key).max()))
r0, r1 = full[:, 29], full[:, 30]
ev = np.concatenate([np.stack([r0, np.ones_like(r0)], 1), np.stack([r1, -np.ones_like(r1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
span_ticks = ev[-1, 0] - ev[0, 0]
print("        concurrency from s_memrealtime start / end of every wave: peak %.1f waves per CU, time-average %.1f; span %.1f us; "
      "quartiles of the span: %s waves per CU" % (conc.max() / 256.0, (conc[:-1] * dt).sum() / span_ticks / 256.0, span_ticks / 100.0,
          [round(float(conc[np.searchsorted(ev[:, 0], ev[0, 0] + q * span_ticks)]) / 256.0, 1) for q in (0.1, 0.25, 0.5, 0.75, 0.9)]))
tt = full[:, 31]
nt, nsmall, nmem = (tt & 0xfffff).sum(), ((tt >> 20) & 0xfffff).sum(), (tt >> 40).sum()
print("        ties (the nonces decide the two extra offsets): %.1f %% of the opcodes; bytes sharing the smallest delta at a tie: %.1f on average; "
      "ties with <= 2 such bytes: %.0f %%" % (100.0 * nt / max(full[:, 26].sum(), 1), nmem / max(nt, 1), 100.0 * nsmall / max(nt, 1)))
rel, single, double = full[:, 20].sum(), (full[:, 21] & 0xffffffff).sum(), (full[:, 21] >> 32).sum()
print("        what the nonces order at a tie (the bytes at the smallest delta if >= 2 share it, else those at the second): %.1f bytes on average; "
      "no lane holds two of them at %.0f %% of the ties, none holds three at another %.0f %%" % (rel / max(nt, 1), 100.0 * single / max(nt, 1), 100.0 * double / max(nt, 1)))
