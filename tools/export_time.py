"""Round 6: what the export of a step's decisions costs (ms per 16-frame step): no export, export, and - experiments
build, ODHIP_EXPORT_DBG - its parts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402

F = 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D.init(0)
qt = D.QuantTables.load()
B.GENERATOR = B.CONTENT["checker"]
luma, chroma = B.synth_pictures(F, 1234)
pipe = D.Pipe(qt, F, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
pipe.set_pictures(luma, chroma)
hl = torch.from_numpy(luma).pin_memory()
hc = torch.from_numpy(chroma).pin_memory()


def run(feed):
    for _ in range(3):
        if feed:
            pipe.feed(hl, hc)
        pipe.step()
    pipe.flush()
    pipe.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        if feed:
            pipe.feed(hl, hc)
        pipe.step()
    pipe.flush()
    pipe.sync()
    return (time.perf_counter() - t0) / steps * 1e3


print("resident, no export: %.3f ms" % run(False))
print("fed, no export     : %.3f ms" % run(True))
host = torch.empty(pipe.export_bytes(), dtype=torch.uint8).pin_memory()
pipe.set_export(host)
print("resident + export  : %.3f ms" % run(False))
print("fed + export       : %.3f ms  (ODHIP_EXPORT_DBG=%s)" % (run(True), os.environ.get("ODHIP_EXPORT_DBG", "0")))
print("shipped bytes per step: %d" % pipe.export_shipped_bytes(host.numpy()))
pipe.set_export(None)
print("fed, no export (after): %.3f ms" % run(True))
print("resident, no export (after): %.3f ms" % run(False))
pipe.set_export(host)
print("fed + export (again): %.3f ms" % run(True))
