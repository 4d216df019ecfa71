#!/usr/bin/env python3
"""Development: what the luma forward pyramid's time follows (DESIGN.md par. 4).  16 frames of
1080p luma per launch, microseconds per launch:
  (a) with the global stores of some or all levels skipped (NULL level pointers: the arithmetic
      and the LDS traffic stay) - what the 20 B/px of stores cost on top of the arithmetic;
  (b) with extra dynamic LDS per workgroup (ODHIP_PYR_LDS_PAD) so that only 2 or 1 workgroups
      fit a CU instead of 3 - how the time follows occupancy (latency-bound: ~1/occupancy)."""
import os
# (round 5) ODHIP_PYR_* exist in the experiments build of the library only
os.environ.setdefault("ODHIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                "daala_amd", "lib", "libdaalahip_exp.so"))
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    sys.path.insert(0, ROOT)
    import daala_amd as D
    D.init(0)
    g = torch.Generator(device="cuda").manual_seed(7)
    F = 16
    luma = torch.randint(0, 256, (F, 1088, 1920), dtype=torch.uint8, device="cuda", generator=g)
    full = D.forward_pyramid(luma, 0, 1920, 1080)
    for name, want in (("all five levels stored", (0, 1, 2, 3, 4)), ("no level stored", ()),
                       ("only 64-point stored", (4,)), ("only 32-point stored", (3,)),
                       ("only 4-point stored", (0,)), ("all but 64-point", (0, 1, 2, 3)),
                       ("64 + 32", (4, 3)), ("64 + 4", (4, 0)), ("64 + 32 + 16", (4, 3, 2)),
                       ("aliased", (0, 1, 2, 3, 4)), ("one allocation", (0, 1, 2, 3, 4)), ("all but 32-point", (0, 1, 2, 4)), ("all but 4-point", (1, 2, 3, 4)),
                       ("all but 16-point", (0, 1, 3, 4)), ("8 + 4", (0, 1)), ("32 + 16 + 8", (1, 2, 3))):
        if os.environ.get("ODHIP_PYR_LDS_PAD", "0") != "0" and len(name) < 18 and name != "no level stored":
            continue
        lv = [full[b] if b in want else None for b in range(5)]
        if name == "aliased":
            lv = [full[0]] * 5           # five levels' stores into ONE plane set: a fifth of the footprint
        if name == "one allocation":
            big = torch.empty((5, F, 1088, 1920), dtype=torch.int32, device="cuda")
            lv = [big[b] for b in range(5)]
        for _ in range(3):
            D.forward_pyramid(luma, 0, 1920, 1080, levels=lv)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            D.forward_pyramid(luma, 0, 1920, 1080, levels=lv)
        b.record()
        torch.cuda.synchronize()
        print("  %-26s %7.1f us" % (name, a.elapsed_time(b) / 20 * 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    for pad, what in ((0, "3 workgroups (12 waves) per CU"), (14000, "2 workgroups per CU"),
                      (30000, "1 workgroup per CU")):
        print("LDS pad %d: %s" % (pad, what), flush=True)
        env = dict(os.environ, ODHIP_PYR_LDS_PAD=str(pad))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
