"""Development: time the luma forward pyramid variants (ODHIP_PYR_VARIANT) on 16 frames of
1080p and check each against variant 0 bit for bit.  One child process per variant."""
import hashlib
import os
# (round 5) ODHIP_PYR_* exist in the experiments build of the library only
os.environ.setdefault("ODHIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                "daala_amd", "lib", "libdaalahip_exp.so"))
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import daala_amd as D
    D.init(0)
    F = int(os.environ.get("F", "16"))
    g = torch.Generator(device="cuda").manual_seed(7)
    luma = torch.randint(0, 256, (F, 1088, 1920), dtype=torch.uint8, device="cuda", generator=g)
    chroma = torch.randint(0, 256, (2 * F, 544, 960), dtype=torch.uint8, device="cuda", generator=g)
    lv = D.forward_pyramid(luma, 0, 1920, 1080)
    cv = D.forward_pyramid(chroma, 1, 1920, 1080)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in lv + cv:
        h.update(t.cpu().numpy().tobytes())
    res = []
    for px, dec, lev in ((luma, 0, lv), (chroma, 1, cv)):
        for _ in range(3):
            D.forward_pyramid(px, dec, 1920, 1080, levels=lev)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            D.forward_pyramid(px, dec, 1920, 1080, levels=lev)
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    print("RESULT %s %.1f %.1f" % (h.hexdigest()[:16], res[0], res[1]))
    sys.exit(0)

base = None
for v in sys.argv[1:] or ["0", "2", "1", "3", "4", "6"]:
    e = dict(os.environ)
    e["ODHIP_PYR_VARIANT"] = v
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True,
                       env=e, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print("variant", v, "FAILED", p.stderr[-600:])
        continue
    _, dig, luma_us, chroma_us = line[0].split()
    base = base or dig
    frac = 16 * 1920 * 1088 * 21 / (float(luma_us) * 1e-6) / 8e12
    print("variant %s  luma %7.1f us (%.3f of 8 TB/s)  chroma %6.1f us  %s" % (
        v, float(luma_us), frac, float(chroma_us), "== variant 0" if dig == base else "MISMATCH"))
