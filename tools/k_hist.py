"""K / pulse statistics of the bench workload per band size (GPU): how many
pulse iterations each band class runs, to direct the band-stage optimisation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import daala_amd as D  # noqa: E402

D.init(0)
pipe = bench.Pipeline(D, 2, torch.device("cuda:0"))
pipe.step()
torch.cuda.synchronize()
tot = {}
for s in pipe.sets:
    for job in s["jobs"]:
        nb, offs, _ = D.pvq_band_layout(job.bs)
        c = D.unpack_cands(job.cands)
        k = c["k"].astype(np.int64)
        fl = (c["flags"] == 1).astype(np.int64)
        for b in range(nb):
            n = offs[b + 1] - offs[b]
            kk = (k[:, b, :] * (fl[:, b, :] != 0))
            # second candidate resumes from the first when k grows
            first = kk[:, 0]
            second = np.where((first > 0) & (kk[:, 1] >= first), kk[:, 1] - first, kk[:, 1])
            pulses = first + second
            e = tot.setdefault(n, dict(bands=0, pulses=0, kmax=0, searched=0, ksum=0, hist=np.zeros(8, np.int64)))
            e["bands"] += kk.shape[0]
            e["pulses"] += int(pulses.sum())
            e["kmax"] = max(e["kmax"], int(kk.max()))
            e["searched"] += int((fl[:, b, :] != 0).sum())
            e["ksum"] += int(kk.sum())
            e["hist"] += np.histogram(kk.max(axis=1), bins=[0, 1, 2, 4, 8, 16, 32, 64, 1 << 30])[0]
            if n in (32, 128):
                # wave = 4 consecutive blocks of the same band: pulses run to the wave max
                m = pulses[: (len(pulses) // 4) * 4].reshape(-1, 4)
                e.setdefault("wave_pulses", 0)
                e["wave_pulses"] += int(m.max(axis=1).sum()) * 4
print("n  bands  searched/band  mean_k(searched)  pulses/band  kmax  wave-max pulses/band   hist of max k [0,1,2-3,4-7,8-15,16-31,32-63,64+]")
for n in sorted(tot):
    e = tot[n]
    print(n, e["bands"], round(e["searched"] / e["bands"], 2), round(e["ksum"] / max(1, e["searched"]), 2),
          round(e["pulses"] / e["bands"], 2), e["kmax"],
          round(e.get("wave_pulses", 0) / e["bands"], 2), (e["hist"] / e["bands"]).round(3).tolist())
