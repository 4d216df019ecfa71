"""How often the double-precision replay of the row searches' single-precision screen fires on real content
(VERDICT r5 weak #1): the bench step at the headline operating point and at the six quality_sweep points, with
the experiments build's counters (odhip_exp_row_replay_stats).  Run with
ODHIP_LIB=daala_amd/lib/libdaalahip_exp.so python tools/replay_rate.py [frames=8]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D.init(0)
L = D.lib()
assert hasattr(L, "odhip_exp_row_replay_stats"), "needs the experiments build: ODHIP_LIB=daala_amd/lib/libdaalahip_exp.so"
out = (ctypes.c_ulonglong * 4)()
print("content   quality   bands (32 / 128 coefficients, with reference)   greedy pulses   replayed in double precision   "
      "rate-pass pulses (upper bound)")
for content, q in [("checker", 20), ("natural", 20)] + list(B.SWEEP_POINTS):
    B.GENERATOR = B.CONTENT[content]
    luma, chroma = B.synth_pictures(F, 1234)
    qt = D.QuantTables.load() if q == 20 else D.QuantTables.for_quality(q)
    pipe = D.Pipe(qt, F, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
    pipe.set_pictures(luma, chroma)
    pipe.step()
    pipe.flush()
    pipe.sync()
    L.odhip_exp_row_replay_stats(out, 1)
    pipe.step()
    pipe.flush()
    pipe.sync()
    L.odhip_exp_row_replay_stats(out, 1)
    print("%-9s -v %-3d   %10d bands   %12d greedy   %8d replayed (%.2e)   %12d rate-pass (%.0f %% of the pulses)" % (
        content, q, out[3], out[0], out[1], out[1] / max(1, out[0]), out[2], 100.0 * out[2] / max(1, out[0] + out[2])), flush=True)
    pipe.destroy()
