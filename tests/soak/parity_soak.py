#!/usr/bin/env python3
"""Randomised parity soak of the priced frame-batch step (odhip_pipe_step, price = 1) against the compiled
reference (oracle/_ref: padding, forward pyramid, pvq_theta with the closed-form rate on every block of every
level, inverse): random picture sizes (ragged ones included), quantisers, masking / QM switches, content
generators with random amplitude and offset, keyframes with and without the chroma-from-luma reference, inter
frames against a random prediction, batches of two or three different pictures per step, full-precision
references (8 / 10 / 12-bit pictures).  Every reconstructed pixel of every level and gain / theta / K / pulses of
every band must be equal.  TEST INFRASTRUCTURE (it links the oracle), a longer companion of
tests/test_gpu_pipeline.py.   usage: [SOAK_BIG=1] parity_soak.py [cases=40] [seed0=0] [max_seconds=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B               # noqa: E402
import daala_amd as D           # noqa: E402
import _pipeline_check as C     # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 0
D.init(0)
t_start = time.time()
done = 0
pixels = 0
bands = 0
reruns_total = 0
refused = 0
for case in range(seed0, seed0 + cases):
    if max_seconds and time.time() - t_start > max_seconds:
        break
    rng = np.random.RandomState(77000 + case)
    # sizes: even, from 16 up to 1080p-ish widths; one in three a multiple of 64
    if os.environ.get("SOAK_BIG") == "1":
        # large pictures: around and up to 1920x1080 (the generators' size)
        pw, ph = 2*rng.randint(500, 961), 2*rng.randint(300, 541)
    elif case % 3 == 0:
        pw, ph = 64*rng.randint(1, 12), 64*rng.randint(1, 8)
    else:
        pw, ph = 2*rng.randint(16, 500), 2*rng.randint(16, 300)
    # the coded quantiser, log-uniform over the reference's range (-v 1 .. 511 gives 12 .. 6574); the base
    # quantiser selects the QM interpolation point (od_interp_qm) - any consistent pair is a valid input
    quantizer = int(round(np.exp(rng.uniform(np.log(8), np.log(6574)))))
    base = min(8176, max(16, int(round(quantizer*rng.uniform(1.1, 1.4)))))
    masking = int(rng.rand() < 0.8)
    hvs = int(rng.rand() < 0.8)
    cfl = bool(rng.rand() < 0.8)
    inter = bool(rng.rand() < 0.25)
    def noise_frame_np(fi, seed):
        """White noise over a gradient: bands with no structure at all (every coefficient about equal -
        many near ties in the searches), at an amplitude drawn per picture."""
        r2 = np.random.RandomState(seed + fi)
        amp = [2, 8, 30, 127][r2.randint(4)]
        yy, xx = np.mgrid[0:1080, 0:1920]
        base = 128 + 60*np.sin(xx/97.)*np.cos(yy/61.)
        return [np.clip(base[::d, ::d] + r2.randint(-amp, amp + 1, size=(1080//d, 1920//d)), 0, 255).astype(np.uint8)
                for d in (1, 2, 2)]

    u = rng.rand()
    gen = B.natural_like_frame_np if u < 0.4 else B.synth_frame_np if u < 0.8 else noise_frame_np
    qt = D.QuantTables(base, quantizer, masking, hvs)

    def picture(fi, seed):
        full = gen(fi, seed) if gen is noise_frame_np else B.picture_planes(gen(fi, seed))
        oy, ox = 2*rng.randint(0, (1080 - ph)//2 + 1), 2*rng.randint(0, (1920 - pw)//2 + 1)
        pl = [full[0][oy:oy + ph, ox:ox + pw], full[1][oy//2:(oy + ph)//2, ox//2:(ox + pw)//2],
              full[2][oy//2:(oy + ph)//2, ox//2:(ox + pw)//2]]
        # random contrast / brightness so that gains and K spread beyond the generators' own range
        gain = [0.15, 0.5, 1.0, 1.0, 1.6][rng.randint(5)]
        off = rng.randint(-40, 41)
        return [np.ascontiguousarray(np.clip((p.astype(np.float64) - 128)*gain + 128 + off, 0, 255).astype(np.uint8))
                for p in pl]

    # one case in four: full-precision references (8 / 10 / 12-bit pictures, 12-bit planes, the xstride-2 conversions)
    bits = [8, 10, 12][rng.randint(3)] if rng.rand() < 0.25 else 0
    # one keyframe case in three: a batch of 2 or 3 DIFFERENT pictures per step (frame i must not see frame j)
    F = int(rng.randint(2, 4)) if (not inter and not bits and rng.rand() < 0.33) else 1

    def deepen(pl):
        if bits in (0, 8):
            return pl
        return [((p.astype(np.int32) << (bits - 8)) + rng.randint(0, 1 << (bits - 8), size=p.shape)).astype(np.int16)
                for p in pl]

    frames = [deepen(picture(int(rng.randint(0, 50)), int(rng.randint(1, 1 << 20)))) for _ in range(F)]
    pics = frames[0]
    pred = None
    if inter:
        # a prediction = the picture itself plus noise and a small shift: correlated, as motion compensation is
        pred = []
        for p in pics:
            q = np.roll(p, (int(rng.randint(-1, 2)), int(rng.randint(-2, 3))), axis=(0, 1)).astype(np.int32)
            q = q + rng.randint(-6, 7, size=q.shape)*(1 << max(0, bits - 8))
            pred.append(np.clip(q, 0, (1 << (bits or 8)) - 1).astype(p.dtype))
    t0 = time.time()
    want = []
    try:
        if inter:
            # (the decision dump of the checker covers keyframes; an inter frame is compared by its pixels)
            cpu, blocks, _ = C.cpu_frame(qt, pics, pw, ph, inter_pred=pred, fpr_bits=bits)
            t_cpu = time.time() - t0
            gpu, reruns = C.gpu_device_priced(D, qt, pics, pw, ph, inter_pred=pred, steps=3, fpr_bits=bits)
            bad = C.compare_frame(gpu, cpu)
            badd = []
        elif bits:
            cpu, blocks, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=cfl, fpr_bits=bits)
            t_cpu = time.time() - t0
            gpu, reruns = C.gpu_device_priced(D, qt, pics, pw, ph, chroma_cfl=cfl, fpr_bits=bits)
            bad = C.compare_frame(gpu, cpu)
            badd = []
        else:
            stacked = [np.stack([f[p] for f in frames]) for p in range(3)]
            gpu, reruns, dec = C.gpu_device_priced(D, qt, stacked, pw, ph, chroma_cfl=cfl, frames=F, decisions=True,
                                                   steps=2 + (F > 1))
            bad, badd, blocks, t_cpu = [], [], 0, 0.
            for i in range(F):
                t0 = time.time()
                w_i = []
                cpu, b_i, _ = C.cpu_frame(qt, frames[i], pw, ph, chroma_cfl=cfl, decisions=w_i)
                t_cpu += time.time() - t0
                blocks += b_i
                bad += C.compare_frame(gpu, cpu, frame=i, frames=F)
                badd += C.compare_decisions(dec, w_i, frame=i, frames=F)
                want += w_i
                if i:
                    pixels += sum(int(v.size) for pl in cpu for v in pl)
    except D.PulseRangeError:
        # a band needs more than 32767 pulses (ODHIP_PVQ_MAX_K): the library refuses loudly instead of differing
        refused += 1
        print("case %3d %dx%d q %d/%d: refused - a pulse count above 32767 (reported by odhip_pipe_sync)"
              % (case, pw, ph, quantizer, base), flush=True)
        continue
    tag = "%dx%d q %d/%d masking %d hvs %d cfl %d %s %s%s%s" % (
        pw, ph, quantizer, base, masking, hvs, cfl, "inter" if inter else "key", gen.__name__,
        " fpr%d" % bits if bits else "", " x%d frames" % F if F > 1 else "")
    if bad or badd:
        print("case %d %s: MISMATCH %s %s" % (case, tag, bad[:3], badd[:3]), flush=True)
        sys.exit(1)
    done += 1
    pixels += sum(int(v.size) for pl in cpu for v in pl)
    bands += sum(int(w[1].shape[0]*w[1].shape[1]) for pl in want for w in pl)
    reruns_total += reruns
    print("case %3d %-78s equal (%d blocks, reference C %.1f s, host-libm re-decisions %d)" % (case, tag, blocks, t_cpu,
                                                                                                reruns), flush=True)
print("parity soak: %d cases equal, %d reconstructed pixels and %d bands compared, %d host-libm re-decisions, %d refused "
      "(pulse count above 32767), %.0f s" % (done, pixels, bands, reruns_total, refused, time.time() - t_start))
