"""BASELINE.json configs[2] and configs[3] as standalone microbenchmarks
(development/documentation aid; the headline metric is bench.py).

configs[2]: N x N od_bin_fdct/idct on working sets ABOVE the 256 MiB Infinity
            Cache (so the rate is an HBM rate), 8 B algorithmic per coefficient.
configs[3]: pvq_search_rdo_double on 1M random 16-dim bands."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daala_amd as D
D.init(0)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


print("# configs[2]: standalone 2-D transforms, 1 GiB in + 1 GiB out per launch (> Infinity Cache)")
a = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print("copy 1 GiB -> 1 GiB: %.0f GB/s (read + write), the practical ceiling" % (2 * a.numel() * 4 / t / 1e9))
del a, b
for ln in range(5):
    n = 4 << ln
    nb = (1 << 28) // (n * n)
    x = torch.randint(-4080, 4081, (nb, n, n), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    for ex in (0, 1):
        tf = timeit(lambda: D.fdct2d_batch(ln, x, exact32=ex, out=y))
        ti = timeit(lambda: D.idct2d_batch(ln, x, exact32=ex, out=y))
        gb = x.numel() * 8 / 1e9
        print("%2dx%-2d exact32=%d  fdct %6.0f GB/s (%4.1f%% of 8 TB/s)  idct %6.0f GB/s (%4.1f%%)  %.2f G blocks/s" % (
            n, n, ex, gb / tf, 100 * gb / tf / 8000, gb / ti, 100 * gb / ti / 8000, nb / tf / 1e9))
    del x, y
print()
print("# configs[3]: pvq_search_rdo_double, 1,048,576 bands, x ~ U[-1000,1000], g2 = 1, lambda = 0.147")
nb = 1 << 20
for n in (16, 15, 8):
    x = torch.randint(-1000, 1001, (nb, n), dtype=torch.int16, device="cuda")
    g2 = torch.ones(nb, dtype=torch.float64, device="cuda")
    for kk in (1, 2, 4, 8, 16):
        k = torch.full((nb,), kk, dtype=torch.int32, device="cuda")
        y, c = D.pvq_search_batch(x, k, g2, 0.147)
        t = timeit(lambda: D.pvq_search_batch(x, k, g2, 0.147, y=y, cos=c), iters=5, warm=1)
        print("n=%-3d k=%-2d  %7.1f us  %7.0f M bands/s  %6.0f GB/s algorithmic (6n+32 B/band)" % (
            n, kk, t * 1e6, nb / t / 1e6, nb * (6 * n + 32) / t / 1e9))
