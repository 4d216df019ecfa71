"""Band stage one job (plane set, level) at a time, for per-level kernel timings
under rocprofv3 (kernels of different jobs differ by grid size)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import daala_amd as D  # noqa: E402

D.init(0)
pipe = bench.Pipeline(D, 8, torch.device("cuda:0"))
pipe.step()
torch.cuda.synchronize()
for job in pipe.jobs:
    for _ in range(3):
        D.pvq_noref_bands_multi([job], pipe.lam)
    torch.cuda.synchronize()
for s in pipe.sets:
    for job in s["jobs"]:
        print(s["name"], "bs", job.bs, "blocks", job.nblocks)
