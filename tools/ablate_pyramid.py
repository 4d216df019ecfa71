import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daala_amd as D
D.init(0)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e3
F=int(sys.argv[1]) if len(sys.argv)>1 else 8
W,H=1920,1088
luma=torch.randint(0,256,(F,H,W),dtype=torch.uint8,device='cuda')
full=D.forward_pyramid(luma,0,1920,1080)
for want in (None, set(), {0},{1},{2},{3},{4},{0,1,2},{3,4}):
    lv=[full[i] if (want is None or i in want) else None for i in range(5)]
    t=timeit(lambda: D.forward_pyramid(luma,0,1920,1080,levels=lv))
    nb = F*H*W*(1+4*(5 if want is None else len(want)))
    print("want=%-12s %8.1f us  %7.1f GB/s (bytes actually moved)"%(str(want), t, nb/t/1e3))
a=torch.empty(F*H*W*5, dtype=torch.int32, device='cuda')
t=timeit(lambda: a.fill_(3)); print("fill same bytes %.1f us %.1f GB/s"%(t, a.numel()*4/t/1e3))
