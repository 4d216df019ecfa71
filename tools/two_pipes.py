"""Round 6 experiment: the 16-frame batch of the bench step as S sub-batches on S pipes (S x 2 streams) whose steps
are issued round-robin, against one pipe of 16 frames.  Does a second chain pair in another phase keep the VALU
busy while the first one runs its HBM-bound kernels?   usage: two_pipes.py [frames=16] [steps=10]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D.init(0)
qt = D.QuantTables.load()
B.GENERATOR = B.CONTENT["checker"]
luma, chroma = B.synth_pictures(F, 1234)


def run(nsub, threads=False):
    fs = F // nsub
    pipes = []
    for i in range(nsub):
        p = D.Pipe(qt, fs, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
        sl = luma[i * fs:(i + 1) * fs]
        sc = np.ascontiguousarray(np.concatenate([chroma[i * fs:(i + 1) * fs], chroma[F + i * fs:F + (i + 1) * fs]]))
        p.set_pictures(sl, sc)
        pipes.append(p)
    for _ in range(3):
        for p in pipes:
            p.step()
    for p in pipes:
        p.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if threads:
        import threading

        def worker(p):
            for _ in range(steps):
                p.step()
            p.flush()
        ts = [threading.Thread(target=worker, args=(p,)) for p in pipes]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    else:
        for _ in range(steps):
            for p in pipes:
                p.step()
        for p in pipes:
            p.flush()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    for p in pipes:
        p.destroy()
    return dt


for nsub, th in ((1, False), (2, False), (2, True), (4, False), (4, True), (1, False)):
    print("%d pipe(s) x %2d frames%s: %.3f ms per %d-frame step" % (nsub, F // nsub, " (a host thread each)" if th else "", run(nsub, th), F), flush=True)
