"""How often the host-libm resolve paths of the priced step are taken on real content: many
batches of DIFFERENT pictures (both synthetic generators, several operating points) through the bench step; counts
of bands listed inside the acos margin (theta recomputed on the host), of those whose theta changed, and of priced
decisions inside the log margin (re-decided on the host).   usage: margin_soak.py [batches=60] [frames=8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D.init(0)
tot = {"bands": 0, "theta_listed": 0, "theta_changed": 0, "price_listed": 0}
bands_per_frame = 509490
for q in (20, 5, 40):
    qt = D.QuantTables.load() if q == 20 else D.QuantTables.for_quality(q)
    pipe = D.Pipe(qt, F, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
    for b in range(batches):
        B.GENERATOR = B.CONTENT["natural" if b & 1 else "checker"]
        luma, chroma = B.synth_pictures(F, 9000 + 131*b + q)
        pipe.set_pictures(luma, chroma)
        pipe.step()
        pipe.flush()
        pipe.sync()
    row = {"bands": batches*F*bands_per_frame, "theta_listed": pipe.theta_listed(), "theta_changed": pipe.theta_reruns(),
           "price_listed": pipe.price_reruns()}
    print("-v %-3d %d batches of %d frames: %s" % (q, batches, F, row), flush=True)
    for k in tot:
        tot[k] += row[k]
    pipe.destroy()
print("total: %s" % tot)
print("per 10^9 bands: theta listed %.1f, theta changed %.1f, priced decisions re-decided %.1f" % (
    1e9*tot["theta_listed"]/tot["bands"], 1e9*tot["theta_changed"]/tot["bands"], 1e9*tot["price_listed"]/tot["bands"]))
