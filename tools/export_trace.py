"""Timeline around the export pack kernels of fed / resident steps (development aid): run under
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fed"
D.init(0)
qt = D.QuantTables.load()
B.GENERATOR = B.CONTENT["checker"]
luma, chroma = B.synth_pictures(16, 1234)
pipe = D.Pipe(qt, 16, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
pipe.set_pictures(luma, chroma)
hl = torch.from_numpy(luma).pin_memory()
hc = torch.from_numpy(chroma).pin_memory()
host = torch.empty(pipe.export_bytes(), dtype=torch.uint8).pin_memory()
pipe.set_export(host)
for _ in range(8):
    if mode == "fed":
        pipe.feed(hl, hc)
    pipe.step()
pipe.flush()
pipe.sync()
