import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daala_amd as D, bench
D.init(0)
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e3
pipe=bench.Pipeline(D, 8, torch.device('cuda',0))
pipe.step(); torch.cuda.synchronize()
print("bands all jobs      %8.1f us" % timeit(lambda: D.pvq_noref_bands_multi(pipe.jobs, pipe.lam)))
for j in pipe.jobs:
    nm = "bs=%d nblocks=%d" % (j.bs, j.nblocks)
    print("bands %-24s %8.1f us" % (nm, timeit(lambda: D.pvq_noref_bands_multi([j], pipe.lam))))
# huge quantiser: every candidate pruned -> front end + output only
saved=[(j, list(j.q_band)) for j in pipe.jobs]
for j in pipe.jobs:
    for i in range(12): j.q_band[i] = max(1, j.q_band[i])*200
print("bands all jobs, q*200 (all pruned) %8.1f us" % timeit(lambda: D.pvq_noref_bands_multi(pipe.jobs, pipe.lam)))
for j,q in saved:
    for i in range(12): j.q_band[i]=q[i]
D.pvq_noref_bands_multi(pipe.jobs, pipe.lam); torch.cuda.synchronize()
c=pipe.jobs[2].cands
print("k hist 16x16 luma:", torch.bincount(c["k"][...,1].flatten().clamp(0,63))[:40].tolist())
print("flags mean per job:", [round(float(j.cands["flags"].float().mean()),3) for j in pipe.jobs])
for j in pipe.jobs[:5]:
    f=j.cands["flags"].float().mean(0).cpu().numpy()
    kk=(j.cands["k"].float()*j.cands["flags"].float()).sum(0)/j.cands["flags"].float().sum(0).clamp(min=1)
    print("bs",j.bs,"searched frac per band [slot0,slot1]:", np.round(f,2).tolist(), "mean k:", np.round(kk.cpu().numpy(),1).tolist())
