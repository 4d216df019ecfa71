import sys, os, torch
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
for s in pipe.sets:
    for job in s["jobs"]:
        job.dq = torch.empty_like(job.coef)
        D.pvq_select_synth_noref_multi([job], pipe.lam)
        t1 = timeit(lambda: D.inverse_level(job.dq, s["dec"], job.bs, 1920, 1080, out=s["recon"]))
        t2 = timeit(lambda: D.inverse_level_pvq(job, s["dec"], 1920, 1080, out=s["recon"]))
        t3 = timeit(lambda: D.pvq_select_synth_noref_multi([job], pipe.lam))
        print("%-6s bs=%d  inverse(dq) %6.1f us   inverse_pvq %6.1f us   select_synth %6.1f us" % (s["name"], job.bs, t1, t2, t3))
        job.dq = None
