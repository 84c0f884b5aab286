"""Per-phase shader-clock accounting of greedy_team_kernel (diagnostic -DIIV_STAMPS build):
    make -C ii-vision_amd/csrc ../libiivision_stamps.so && IIV_LIB=ii-vision_amd/libiivision_stamps.so python tools/team_phases.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ii-vision_amd", "transcoder")]
import torch
import _iiv_native as native
import palette
import stream_batch

KIND = sys.argv[1] if len(sys.argv) > 1 else "iid"     # iid | img
for mode in (native.DHGR, native.HGR):
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True)
    store = native.build_store_table(mode, dm)
    fm, fa = (stream_batch.synth_frames_img(1, 40, mode == native.DHGR, seed=99) if KIND == "img" else
              stream_batch.synth_frames_torch(1, 40, mode == native.DHGR, seed=99))
    b = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(1, 1)], dm=dm)
    b.enc.set_greedy_kernel("team")
    b.encode_frames(fm, fa, 40)
    torch.cuda.synchronize()
    st = b.enc.get_state(100)
    names = ["top/formation", "load+score", "wait barrier B", "prefix sums", "request ahead + commit", "wait barrier C",
             "bookkeeping + take over", "-"]
    launches, rounds, entries = int(st[28]), int(st[24]), int(st[25])
    total, real = int(st[26]), int(st[27])
    print("mode %d: %d launches, %d rounds, %d entries (%.2f per round); %.0f shader clocks per round; "
          "shader clock / 100 MHz real-time clock = %.2f" % (mode, launches, rounds, entries, entries / max(rounds, 1),
                                                            total / max(rounds, 1), total / max(real, 1)))
    print("   runs shorter than 7: ended by the window's edge %d, by the launch's budget %d, by a page already in the run %d; "
          "runs cut short behind a tie the nonces decide: %d; entries found dead: %d" % (
              int(st[13]), int(st[14]), int(st[15]), int(st[31]) >> 32, int(st[31]) & 0xffffffff))
    for i, n in enumerate(names[:7]):
        print("   %-26s %8.0f clocks per round" % (n, int(st[16 + i]) / max(rounds, 1)))
    print("   MT wave: %.0f clocks per round, %.2f blocks per round -> %.0f clocks per block" % (
        int(st[29]) / max(rounds, 1), int(st[30]) / max(rounds, 1), int(st[29]) / max(int(st[30]), 1)))
    b.close()
