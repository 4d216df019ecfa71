"""tests/golden/k_hist.npz: the distribution of pulse counts K the reference's pvq_theta
settles on over a whole configs[1] frame (1080p 4:2:0, -v 20, every block of every level,
chroma with its chroma-from-luma reference) - the "empirical distribution dumped from C2"
SURVEY.md 8(d) asks the 1M-band search parity set to draw its K from.  CPU only: the compiled
reference (oracle/_ref) through tests/_pipeline_check.cpu_frame(decisions=...).

    python tools/make_golden_khist.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import _pipeline_check as C  # noqa: E402
from daala_amd.quant import QuantTables  # noqa: E402

qt = QuantTables.load()
pics = bench.picture_planes(bench.synth_frame_np(0, 1234))
dec = []
C.cpu_frame(qt, pics, 1920, 1080, chroma_cfl=True, decisions=dec)
hist = np.zeros(512, np.int64)
nbands = 0
for plane in dec:
    for y, band in plane:
        k = band[:, :, 3].ravel()
        nbands += k.size
        k = np.minimum(k[k > 0], 511)
        hist += np.bincount(k, minlength=512)
out = os.path.join(ROOT, "tests", "golden", "k_hist.npz")
np.savez_compressed(out, hist=hist, bands=np.int64(nbands))
print(out, "coded bands", int(hist.sum()), "of", nbands, "mean K %.2f" % ((hist * np.arange(512)).sum() / hist.sum()),
      "max K", int(np.nonzero(hist)[0].max()))
