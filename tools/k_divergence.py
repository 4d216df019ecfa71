"""what the no-reference (luma) searches lose to pulse-count divergence - they walk the blocks in natural
order, and a wavefront's pulse loops run to the largest K among its bands.  From the decisions of the bench step (K of the
winner of every luma band), per kernel: mean K, mean of the wavefront maximum, and the useful fraction of pulse iterations
sum(K) / sum(lanes x max K of the wavefront); also what sorting inside a workgroup of W bands would give.
usage: k_divergence.py [frames=2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench as B          # noqa: E402
import daala_amd as D      # noqa: E402
import _pipeline_check as C  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D.init(0)


def useful(k, per_wave):
    n = len(k) // per_wave * per_wave
    w = k[:n].reshape(-1, per_wave)
    mx = w.max(axis=1)
    return k[:n].sum() / max(1, (mx * per_wave).sum()), mx.mean()


for content in ("checker", "natural"):
    B.GENERATOR = B.CONTENT[content]
    luma, chroma = B.synth_pictures(F, 1234)
    for q in (20, 5):
        qt = D.QuantTables.load() if q == 20 else D.QuantTables.for_quality(q)
        pipe = D.Pipe(qt, F, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
        pipe.set_pictures(luma, chroma)
        pipe.step()
        pipe.flush()
        pipe.sync()
        dec = C.gpu_decisions(D, pipe)
        print("%s -v %d (luma, K of the chosen candidate; the searches place up to the larger of the two candidates' K)" % (content, q))
        groups = {"k_decide_corner<0> (band 0, 64 bands per wavefront)": [], "k_decide_corner<1> (8-coefficient bands, 64)": [],
                  "k_decide_lane32 (32 per wavefront)": [], "k_decide_pair128 (32 per wavefront)": []}
        for bs in range(5):
            _, band, _ = dec[(0, bs)]
            nb, offs, _ = D.pvq_band_layout(bs)
            for i in range(nb):
                n = offs[i + 1] - offs[i]
                k = band[:, i, 3].astype(np.int64)
                key = [x for x in groups if ("corner<0>" in x and i == 0) or ("corner<1>" in x and n == 8)
                       or ("lane32" in x and n == 32) or ("pair128" in x and n == 128)][0]
                groups[key].append(k)
        for name, ks in groups.items():
            per_wave = 64 if "64" in name else 32
            tot = 0
            den = 0
            mxs = []
            allk = np.concatenate(ks)
            for k in ks:
                n = len(k) // per_wave * per_wave
                w = k[:n].reshape(-1, per_wave)
                tot += k[:n].sum()
                den += (w.max(axis=1) * per_wave).sum()
                mxs.append(w.max(axis=1))
            line = "  %-52s mean K %6.2f  mean wavefront max %6.2f  useful %.2f" % (name, allk.mean(), np.concatenate(mxs).mean(), tot / max(1, den))
            # sorted inside workgroups of 1024 bands (descending), then cut into wavefronts
            tot2 = 0
            den2 = 0
            for k in ks:
                n = len(k) // 1024 * 1024
                if n == 0:
                    continue
                w = -np.sort(-k[:n].reshape(-1, 1024), axis=1)
                w = w.reshape(-1, per_wave)
                tot2 += k[:n].sum()
                den2 += (w.max(axis=1) * per_wave).sum()
            if den2:
                line += "   sorted within 1024 bands: %.2f" % (tot2 / den2)
            print(line, flush=True)
        pipe.destroy()
