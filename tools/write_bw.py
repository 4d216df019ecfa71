#!/usr/bin/env python3
"""What the box's HBM sustains for WRITE-dominated streams (development aid for DESIGN.md par. 4): the
luma pyramid writes 20 bytes for every byte it reads, so its attainable roof is the write rate, not the
8 TB/s read+write spec.  Times torch fill_ (write only), copy_ (1:1) and a 1:20 read:write stream of the
pyramid's own launch size (686 MB) and of 2 GiB, 20 launches each after 3 warm-ups."""
import torch


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    dev = torch.device("cuda:0")
    for nbytes in (668467200, 2 << 30):
        a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        b = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        t = timed(lambda: a.fill_(7))
        print("fill_  %5.0f MB written            : %7.1f us = %.2f TB/s written" % (nbytes / 1e6, t * 1e6, nbytes / t / 1e12))
        t = timed(lambda: a.zero_())
        print("zero_  %5.0f MB written            : %7.1f us = %.2f TB/s written" % (nbytes / 1e6, t * 1e6, nbytes / t / 1e12))
        t = timed(lambda: b.copy_(a))
        print("copy_  %5.0f MB read + as many written: %7.1f us = %.2f TB/s written, %.2f TB/s total"
              % (nbytes / 1e6, t * 1e6, nbytes / t / 1e12, 2 * nbytes / t / 1e12))
        # 1 byte read per 20 written: a uint8 source expanded to five int32 planes
        src = torch.randint(0, 255, (nbytes // 20,), dtype=torch.uint8, device=dev)
        out = a[: 5 * src.numel()].view(5, -1)
        t = timed(lambda: torch.add(src.unsqueeze(0), 1, out=out))
        print("expand %5.0f MB read, %5.0f MB written : %7.1f us = %.2f TB/s written"
              % (src.numel() / 1e6, out.numel() * 4 / 1e6, t * 1e6, out.numel() * 4 / t / 1e12))
        del a, b, src, out


if __name__ == "__main__":
    main()
