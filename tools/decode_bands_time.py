#!/usr/bin/env python3
"""odhip_pvq_decode_bands timed per band size: the LDS-staged kernel against the one-band-per-lane
form it replaced (ODHIP_DECODE_LANE=1 in a child process), on random bands (with-reference and
no-reference mixed), resident in HBM.  Algorithmic bytes: 4n in (reference) + 4n in (pulses) + 16
(symbols) + 4n out + 8 (info) per band.

    python tools/decode_bands_time.py            (on an MI355X)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    import daala_amd as D
    D.init(0)
    qt = D.QuantTables.load()
    rng = np.random.RandomState(3)
    offs = [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]
    for bs, band in ((0, 0), (1, 1), (1, 3), (2, 6)):
        a, b = offs[band], offs[band + 1]
        n = b - a
        qm, qmi = qt.qm_slices(1, bs)
        q0, beta = int(qt.q_band(1, bs)[band]), int(qt.beta_band(1, bs)[band])
        nb = (1 << 20) if n <= 32 else (1 << 18)
        ref = torch.from_numpy((rng.laplace(size=(nb, n))*300).astype(np.int32)).cuda()
        noref = (rng.rand(nb) < 0.4).astype(np.int32)
        y = rng.randint(-2, 3, size=(nb, n)).astype(np.int32)
        y[noref == 0, n - 1] = 0
        sym = np.zeros((nb, 4), np.int32)
        sym[:, 0] = rng.randint(0, 10, size=nb)
        sym[:, 1] = np.where(noref == 1, -1, rng.randint(0, 8, size=nb))
        sym[:, 2] = noref
        yt, st = torch.from_numpy(y).cuda(), torch.from_numpy(sym).cuda()
        qmt = torch.from_numpy(np.ascontiguousarray(qm[a:b])).cuda()
        qit = torch.from_numpy(np.ascontiguousarray(qmi[a:b])).cuda()
        for _ in range(3):
            out, info = D.pvq_decode_bands(ref, yt, st, qmt, qit, q0, beta, 1, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps):
            out, info = D.pvq_decode_bands(ref, yt, st, qmt, qit, q0, beta, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/reps
        byt = nb*(12*n + 24)
        print("n=%-4d %8d bands  %8.1f us  %7.0f GB/s algorithmic  digest %d" % (n, nb, ms*1e3, byt/ms/1e6,
                                                                          int(out.to(torch.int64).sum().item())))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run()
        sys.exit(0)
    for label, env in (("LDS-staged (k_pvq_decode)", {}), ("one band per lane from memory (ODHIP_DECODE_LANE=1)",
                                                          {"ODHIP_DECODE_LANE": "1"})):
        print(label)
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, check=True)
