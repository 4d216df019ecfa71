"""Ad-hoc kernel timing on the GPU box (development aid, not the bench)."""
import sys
import time
import numpy as np
import torch
import daala_amd as D

D.init(0)
dev = "cuda"


def timeit(fn, iters=20, warm=3):
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


F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1088
luma = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=dev)
chroma = torch.randint(0, 256, (2 * F, H // 2, W // 2), dtype=torch.uint8, device=dev)
lv = D.forward_pyramid(luma, 0, 1920, 1080)
lc = D.forward_pyramid(chroma, 1, 1920, 1080)
t = timeit(lambda: D.forward_pyramid(luma, 0, 1920, 1080, levels=lv))
by = F * H * W * 21
print("pyramid luma  : %.3f ms  %.1f GB/s algorithmic (%d frames)" % (t * 1e3, by / t / 1e9, F))
t = timeit(lambda: D.forward_pyramid(chroma, 1, 1920, 1080, levels=lc))
by = 2 * F * (H // 2) * (W // 2) * 17
print("pyramid chroma: %.3f ms  %.1f GB/s algorithmic" % (t * 1e3, by / t / 1e9))
for leaf in range(5):
    out = D.inverse_level(lv[leaf], 0, leaf, 1920, 1080)
    t = timeit(lambda: D.inverse_level(lv[leaf], 0, leaf, 1920, 1080, out=out))
    print("inverse luma leaf %d: %.3f ms  %.1f GB/s algorithmic (5 B/px)" % (leaf, t * 1e3, F * H * W * 5 / t / 1e9))
# copy ceiling
a = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print("copy 256MiB: %.1f GB/s (r+w)" % (2 * a.numel() * 4 / t / 1e9))
t = timeit(lambda: a.fill_(1))
print("fill 256MiB: %.1f GB/s (w)" % (a.numel() * 4 / t / 1e9))
for ln in range(5):
    n = 4 << ln
    nb = (1 << 24) // (n * n)
    x = torch.randint(-4000, 4000, (nb, n, n), dtype=torch.int32, device=dev)
    y = torch.empty_like(x)
    for ex in (0, 1):
        t = timeit(lambda: D.fdct2d_batch(ln, x, exact32=ex, out=y))
        ti = timeit(lambda: D.idct2d_batch(ln, x, exact32=ex, out=y))
        print("dct %2dx%-2d exact32=%d: fwd %.1f GB/s inv %.1f GB/s (8 B/coef)" % (n, n, ex, x.numel() * 8 / t / 1e9, x.numel() * 8 / ti / 1e9))
nb = 1 << 20
for n, kk in ((16, 4), (16, 16), (128, 16)):
    x = torch.randint(-1000, 1001, (nb, n), dtype=torch.int16, device=dev)
    k = torch.full((nb,), kk, dtype=torch.int32, device=dev)
    g2 = torch.ones(nb, dtype=torch.float64, device=dev)
    y, c = D.pvq_search_batch(x, k, g2, 0.147)
    t = timeit(lambda: D.pvq_search_batch(x, k, g2, 0.147, y=y, cos=c), iters=5, warm=1)
    print("pvq n=%d k=%d: %.3f ms  %.1f M bands/s  %.1f GB/s" % (n, kk, t * 1e3, nb / t / 1e6, nb * (6 * n + 32) / t / 1e9))
