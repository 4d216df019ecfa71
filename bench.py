#!/usr/bin/env python3
"""bench.py - 1080p all-intra transform blocks/s (filter + DCT + PVQ) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--chroma-noref] [--content natural]

One STEP = one pass of the block-transform hot path over a batch of F synthetic
1920x1080 4:2:0 frames (coded size 1920x1088) already resident in HBM, per GPU,
driven by ONE C call (odhip_pipe_step, include/daala_hip.h):

  luma chain   (own odhip_ctx, stream A)       chroma chain (own odhip_ctx, stream B)
    od_img_plane_copy_pad                        od_img_plane_copy_pad
    forward pyramid: 64..4, 5 levels             forward pyramid, 4 levels
    PVQ band stage, no reference                 [waits for this step's references]
    choice                                       PVQ band stage WITH the chroma-from-luma
    chroma-from-luma references ------------>      reference (pvq_theta's theta path)
    dequantise + inverse, 5 levels               choice, dequantise + inverse, 4 levels

i.e. every block the reference's block-size RDO would evaluate goes through
prefilter + fDCT + PVQ + dequantisation + iDCT + postfilter exactly once:
260 610 transform blocks per frame (173 910 luma + 2 x 43 350 chroma).  The metric
counts those blocks.  The two chains are software-pipelined over steps.

The choice between PVQ candidates inside the timed step is pvq_theta's: distortion +
lambda * od_pvq_rate, the rate in its closed form (speed > 0, src/pvq_encoder.c:250-264,
what the reference's block-size RDO prices with below complexity 5) evaluated ON THE
DEVICE in the choice kernels; a decision too close for the device's log to settle is
left to the host libm (`price_margin_reruns`).  `verified` compares every reconstructed
pixel of every level of frame 0 AS THE TIMED PIPELINE LEFT IT with the cpu_baseline leg
(the reference's own C functions: pvq_theta with the same closed-form pricing);
`pipelined_equals_serial` replays the timed pictures on one stream and compares every
plane and choice record.  --no-price: the choice on distortion alone (round 1's step;
`verified` then runs the host-priced flow stage by stage).  The live-entropy-coder rate
(speed 0) stays host state: INTEGRATION.md section 7.

N > 1: frames are sharded over ranks (independent all-intra frames, no data-path
collective) -> weak scaling, F frames per GPU; `sharded_encode_check` then encodes
one 1080p frame per rank with the real encoder + the batched GPU stage and gathers
the packets over RCCL (outside the timed region).

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the largest
exclusive duration (fp64 VALU issue is its roof); `roofline_filter_dct` for the
filter + DCT stage against HBM; `kernels` lists every stage.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

# A pipe that feeds pictures AND exports decisions drives five HIP streams (two chains, picture copies, export, the
# default one); the runtime maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams on one
# queue serialise - the H2D copy of the next pictures behind a chain cost 0.6 ms per step (profiles/r6_export.txt).
# Read once, when the HIP runtime initialises: set before torch is imported.  No effect on the resident step.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, PIC_W, PIC_H = 1920, 1088, 1920, 1080
DEFAULT_FRAMES = 16   # per GPU per step; profiles/ holds the rocprofv3 runs of this default command
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def blocks_per_frame():
    luma = sum((W // (4 << bs)) * (H // (4 << bs)) for bs in range(5))
    chroma = sum(((W // 2) // (4 << bs)) * ((H // 2) // (4 << bs)) for bs in range(4))
    return luma + 2 * chroma


def synth_frame_np(index, seed):
    """One synthetic 1920x1088 4:2:0 frame (Y, Cb, Cr uint8 planes): smooth
    texture + 32-pixel checker edges + uniform noise.  No media exists in the
    reference tree or this image.  The SAME generator feeds the GPU step and the
    CPU baseline."""
    rng = np.random.RandomState(seed + 7919 * index)
    planes = []
    for (w, h) in ((W, H), (W // 2, H // 2), (W // 2, H // 2)):
        yy, xx = np.mgrid[0:h, 0:w]
        base = 128 + 60 * np.sin((xx + 7 * index) * (6.283 / 97.0)) * np.cos(yy * (6.283 / 61.0))
        checker = ((((xx + 3 * index) // 32) + (yy // 32)) % 2) * 40 - 20
        noise = rng.randint(-12, 13, size=(h, w))
        planes.append(np.clip(base + checker + noise, 0, 255).astype(np.uint8))
    return planes


def natural_like_frame_np(index, seed):
    """The second kind of synthetic content SURVEY.md 8(d) asks for: a sum of 2-D cosines
    plus separable AR(1) noise (rho = 0.95, the model of the reference's dcttest,
    src/dct.c:4968); chroma = subsampled luma with its own gain plus a little independent
    noise, so that luma-derived predictions correlate with chroma as in natural video."""
    rng = np.random.RandomState(seed + 7919 * index)
    yy, xx = np.mgrid[0:H, 0:W]
    img = 40 * np.cos((xx + 5 * index) * 0.013 + yy * 0.007) + 25 * np.cos(xx * 0.041 - yy * 0.029) \
        + 12 * np.cos(xx * 0.11 + 1.0) * np.cos(yy * 0.09)
    e = rng.normal(size=(H, W)) * 6
    for ax in (0, 1):
        e = np.moveaxis(e, ax, 0)
        for i in range(1, e.shape[0]):
            e[i] += 0.95 * e[i - 1]
        e = np.moveaxis(e, 0, ax) * 0.31
    luma = np.clip(128 + img + e * 4, 0, 255)
    sub = luma.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
    cb = np.clip(128 + 0.5 * (sub - 128) + rng.normal(size=sub.shape) * 2, 0, 255)
    cr = np.clip(128 - 0.35 * (sub - 128) + rng.normal(size=sub.shape) * 2, 0, 255)
    return [luma.astype(np.uint8), cb.astype(np.uint8), cr.astype(np.uint8)]


CONTENT = {"checker": synth_frame_np, "natural": natural_like_frame_np}
GENERATOR = synth_frame_np


def picture_planes(planes):
    """The 1920x1080 picture of a generated frame (the generator fills the coded
    1920x1088 size; the encoder's input is the picture, the rest is padding)."""
    return [planes[0][:PIC_H], planes[1][:PIC_H // 2], planes[2][:PIC_H // 2]]


def synth_pictures(nframes, seed):
    """F pictures as the pipe takes them: luma [F,1080,1920], chroma [2F,540,960] (all Cb,
    then all Cr), host arrays; uploaded once, resident in HBM for every step."""
    fr = [picture_planes(GENERATOR(i, seed)) for i in range(nframes)]
    luma = np.stack([f[0] for f in fr])
    chroma = np.stack([f[1] for f in fr] + [f[2] for f in fr])
    return np.ascontiguousarray(luma), np.ascontiguousarray(chroma)


def copy_ceiling_gbs(D, n=10):
    """Same-run practical HBM ceiling: the library's own 16-byte-vector copy kernel over
    1 GiB (odhip_copy_ceiling, best of 2 / 4 / 8 vectors in flight per lane), read + written bytes per
    second.  Round 5 used a torch copy_ here (4.8 TB/s), which flattered every `frac_of_copy`."""
    return D.copy_ceiling(1 << 30, n)


def ref128_bytes(D, pipe):
    """Algorithmic bytes and band count of one k_refb_lean_row<8,16> launch (the decided
    with-reference stage: nothing per candidate reaches memory), counted from the choice
    records the last step left: per 128-coefficient band the 64 B preparation record, the
    254 B reflected vector and the 256 B QM-scaled vector in (an upper bound: a band reads
    one or both), 4 B of sort index, the 64 B choice record out, and 256 B of pulses when the
    winner places any."""
    total = 0
    bands = 0
    for bs in range(4):
        nb, offs, _ = D.pvq_band_layout(bs)
        which = [b for b in range(nb) if offs[b + 1] - offs[b] == 128]
        if not which:
            continue
        B = pipe.nblocks(1, bs)
        ch = pipe.read(D.BUF_CHOICE, 1, bs, dtype=np.int32).reshape(B, nb, 16)
        for b in which:
            stored = (ch[:, b, 6] == 0) & (ch[:, b, 9] >= 0)
            total += B * (64 + 254 + 256 + 4 + 64) + 256 * int(stored.sum())
            bands += B
    return total, bands


def algorithmic_bytes(F):
    """SURVEY.md 8(d) per-unit figures x units per launch."""
    luma_px = F * W * H
    chroma_px = 2 * F * (W // 2) * (H // 2)
    # PVQ band stage, per band of n coefficients: 4n B coefficients in, two
    # candidates x 2n B signed pulses (int16) + one 64 B band record out
    # (2n + 4n + 32 B for the bare search).  The x16 scratch
    # (2n B written by the preparation pass, read by the search) and the sort
    # keys / indices are implementation traffic, not counted.  All nine (plane
    # set, level) jobs of a step run in one multi-job launch group.
    nbands = [1, 4, 7, 9, 9]
    coded = [15, 63, 255, 511, 511]
    bands_b = 0
    synth_b = 0
    for (w, h, planes, top) in ((W, H, F, 4), (W // 2, H // 2, 2 * F, 3)):
        for bs in range(top + 1):
            nblk = planes * (w // (4 << bs)) * (h // (4 << bs))
            bands_b += nblk * (8 * coded[bs] + 64 * nbands[bs])
            synth_b += nblk * (8 * coded[bs] + 64 * nbands[bs])
    return {
        # od_img_plane_copy_pad: picture read, padded plane written
        "image_copy_pad_luma": F * (PIC_W * PIC_H + W * H),
        "image_copy_pad_chroma": 2 * F * ((PIC_W // 2) * (PIC_H // 2) + (W // 2) * (H // 2)),
        "forward_pyramid_luma": luma_px * 21,      # 1 B read + 5 levels x 4 B written
        "forward_pyramid_chroma": chroma_px * 17,  # 1 B read + 4 levels x 4 B written
        # dequantise-on-load inverse: per level 4 B per CODED coefficient read
        # (all of them below 32x32, 512 per block above) + 1 B/px written; the
        # figure below is the 4 B/px + 1 B/px upper bound of SURVEY 8(d).
        "dequant_inverse_luma": luma_px * 5 * 5,      # five levels per launch group
        "dequant_inverse_chroma": chroma_px * 5 * 4,  # four levels
        "pvq_noref_bands": bands_b,    # one multi-job launch group per step
    }


FILTER_DCT_STAGES = ("image_copy_pad_luma", "image_copy_pad_chroma", "forward_pyramid_luma",
                     "forward_pyramid_chroma", "dequant_inverse_luma", "dequant_inverse_chroma")


def stage_roofline(name, kernels_note, ab_bytes, ms_alone, in_step, pmc, pmc_keys):
    """roofline object of one filter + DCT stage against HBM: SURVEY 8(d) bytes / the stage timed alone."""
    gbs = ab_bytes / (ms_alone * 1e-3) / 1e9
    out = {"kernel": kernels_note, "stage": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
           "avg_ms_per_launch": round(ms_alone, 4), "launches": 10, "algorithmic_bytes_per_launch": ab_bytes}
    if in_step is not None:
        out["in_step"] = {"avg_ms_per_launch": in_step["avg_ms_per_launch"],
                          "frac": in_step.get("frac_of_hbm_peak")}
    tr = [pmc[k]["hbm_bytes_per_launch"] for k in pmc if any(k.startswith(q) for q in pmc_keys)
          and pmc[k].get("hbm_bytes_per_launch")]
    if tr:
        out["traffic"] = int(sum(tr))
    return out


def host_cpu_quota():
    """CPUs this process may use: the affinity mask capped by the cgroup CPU quota (cpu.max), which
    `nproc` and the affinity mask do not show - the GPU boxes of this pool expose 256 logical CPUs
    under a quota of 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return n


def _pin(core):
    try:
        os.sched_setaffinity(0, {core})
        return True
    except (AttributeError, OSError):
        return False


def cpu_leg(qt, chroma_cfl, first_pictures, min_frames=5, min_seconds=10.0, max_frames=24, lib=None,
            more_pictures=()):
    """The same per-block work on ONE pinned host core with the reference's own C
    functions (oracle/_ref): whole pictures of the bench generator, one timing per
    picture, until >= min_frames pictures and >= min_seconds of CPU work.  The first
    picture is the GPU batch's frame 0 (its reconstruction is what `verified` compares);
    `more_pictures` (other frames of the GPU batch) come next and their reconstructions are
    kept too.  Returns (per-picture rates, blocks per picture, recon of the first picture,
    seconds, recons of more_pictures)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _pipeline_check as C
    rates = []
    busy = 0.0
    first = None
    more = []
    blocks = 0
    n = 0
    while n < max_frames and (n < min_frames or busy < min_seconds or n <= len(more_pictures)):
        pics = first_pictures if n == 0 else more_pictures[n - 1] if n <= len(more_pictures) \
            else picture_planes(GENERATOR(1000 + n, 1234))
        recon, blocks, dt = C.cpu_frame(qt, pics, PIC_W, PIC_H, chroma_cfl=chroma_cfl, lib=lib)
        if n == 0:
            first = recon
        elif n <= len(more_pictures):
            more.append(recon)
        rates.append(blocks / dt)
        busy += dt
        n += 1
    return rates, blocks, first, busy, more


def cpu_worker(args):
    """`bench.py --cpu-worker CORE`: one pinned process of the all-cores CPU figure."""
    import daala_amd as D
    global GENERATOR
    GENERATOR = CONTENT[args.content]
    _pin(args.cpu_worker)
    qt = D.QuantTables.load()
    first = picture_planes(GENERATOR(2000 + args.cpu_worker, 1234))
    rates, blocks, _, busy, _ = cpu_leg(qt, not args.chroma_noref, first, min_frames=2, min_seconds=0.0,
                                        max_frames=2)
    print(json.dumps({"blocks": blocks * len(rates), "seconds": busy}))


def cpu_baseline(D, qt, chroma_cfl, args, gpu_frame0, timed_recon=None, timed_dec=None, more_frames=()):
    """cpu_baseline (one pinned core, median of >= 5 pictures), the all-cores figure
    (one pinned process per core, independent pictures - all-intra frames are
    independent), and the whole-frame verification of the GPU path against it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import ref
    import _pipeline_check as C
    if ref() is None:
        # oracle/_ref (the compiled reference, prebuilt by build()) did not travel
        return None, None, {"verified": None, "why": "oracle/_ref/libdaalaref.so absent"}
    prev = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = _pin(0)
    # more_frames: [(index in the batch, pictures)] - further frames of the GPU batch, verified like frame 0
    rates, blocks, recon0, busy, recon_more = cpu_leg(qt, chroma_cfl, gpu_frame0,
                                                      more_pictures=[p_ for _, p_ in more_frames])
    # the x86-intrinsics build of the reference (BASELINE.md section 3 asks for both): same
    # pinned core, median of 5 pictures; its reconstruction of frame 0 must equal the C build's
    simd = None
    from _libs import ref_simd
    if ref_simd() is not None and ref_simd().ref_stage_simd() > 0:
        srates, _, srecon0, sbusy, _ = cpu_leg(qt, chroma_cfl, gpu_frame0, min_frames=5, min_seconds=0.0,
                                               max_frames=5, lib=ref_simd())
        same = all(np.array_equal(a, b) for pa, pb in zip(recon0, srecon0) for a, b in zip(pa, pb))
        simd = {"value": float(np.median(srates)), "unit": "blocks/s", "cores": 1, "kind": "reference",
                "runs": len(srates), "min": float(min(srates)), "max": float(max(srates)),
                "isa": {1: "SSE2", 2: "SSE4.1", 3: "SSE4.1 + AVX2"}[ref_simd().ref_stage_simd()],
                "equals_plain_c": bool(same),
                "sample": "the same stage with the reference's x86 intrinsics compiled in "
                          "(oracle/_ref/libdaalaref_simd.so): they exist for the 4x4 and 8x8 transforms only "
                          "(src/x86/x86state.c:66-90), so the figure moves by the share of those transforms; "
                          "median of %d pictures, %.1f s, same pinned core" % (len(srates), sbusy)}
    if prev is not None:
        os.sched_setaffinity(0, prev)
    what = ("padding + forward pyramid + pvq_theta with closed-form pricing (luma: no-reference "
            "bands; chroma: WITH the chroma-from-luma reference) + inverse" if chroma_cfl else
            "padding + forward pyramid + pvq_theta noref bands (closed-form pricing) + inverse")
    base = {"value": float(np.median(rates)), "unit": "blocks/s", "cores": 1, "kind": "reference",
            "pinned_to_core_0": pinned, "runs": len(rates),
            "min": float(min(rates)), "max": float(max(rates)),
            "scope": "the transform/PVQ stage only (the whole reference encoder, with entropy coding "
                     "and block-size RDO, runs 7.2e4 blocks/s on the survey host, SURVEY.md section 6)",
            "sample": "median of %d synthetic 1920x1080 4:2:0 pictures of the bench generator (%d blocks "
                      "each, %.1f s in all): %s of every block at every level, reference C functions "
                      "(plain C build; its x86 intrinsics cover only the 4x4 / 8x8 transforms of this "
                      "stage), single pinned thread" % (len(rates), blocks, busy, what)}
    # all cores: one pinned process per core, two pictures each
    host = None
    # N independent encoder contexts on N = 8 cores (the survey host's
    # core count); boxes that expose more logical CPUs than their quota allows are not loaded
    # beyond that
    ncores = min(8, len(prev) if prev else (os.cpu_count() or 1))
    if ncores > 1 and not args.no_cpu_allcores:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--content", args.content]
        if args.chroma_noref:
            cmd.append("--chroma-noref")
        cores = sorted(prev)[:ncores] if prev else list(range(ncores))
        t0 = time.perf_counter()
        procs = [subprocess.Popen(cmd + ["--cpu-worker", str(c)], stdout=subprocess.PIPE, text=True)
                 for c in cores]
        done = []
        for pr in procs:
            out, _ = pr.communicate()
            try:
                done.append(json.loads(out.strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
        wall = time.perf_counter() - t0
        if len(done) == len(procs):
            host = {"value": sum(d["blocks"] / d["seconds"] for d in done), "unit": "blocks/s",
                    "cores": len(done), "kind": "reference",
                    "sample": "%d pinned processes x 2 pictures each (independent all-intra frames), "
                              "sum of the per-process rates; %.1f s wall incl. start-up" % (len(done), wall)}
    # whole-frame verification: the GPU stages with the host pricing every candidate in
    # between (closed form, the reference's own od_pvq_rate) must reproduce the CPU leg's
    # reconstruction of the batch's frame 0 at every level of all three planes
    if timed_recon is not None:
        # price = 1: what the TIMED pipeline left in its buffers after the last step
        bad = C.compare_frame(timed_recon, recon0, frame=0, frames=args.frames)
        for (fi, _), rec in zip(more_frames, recon_more):
            bad += [(fi,) + b for b in C.compare_frame(timed_recon, rec, frame=fi, frames=args.frames)]
        what = ("frames %s of the batch as the timed pipeline reconstructed them in its last step (device-"
                "priced choice), whole frames: every reconstructed pixel of every partition level of Y, "
                "Cb, Cr == the reference C functions' (cpu_baseline leg)" % ([0] + [fi for fi, _ in more_frames]))
    else:
        gpu = C.gpu_priced_frame(D, qt, gpu_frame0, PIC_W, PIC_H, chroma_cfl=chroma_cfl)
        bad = C.compare_frame(gpu, recon0)
        what = ("frame 0 of the batch, whole frame: every reconstructed pixel of every partition "
                "level of Y, Cb, Cr from the GPU stages (host-priced choice) == the reference C "
                "functions' (cpu_baseline leg)")
    ver = {"verified": not bad, "what": what, "frames_compared": [0] + [fi for fi, _ in more_frames],
           "planes_levels_compared": 13*(1 + len(more_frames) if timed_recon is not None else 1), "mismatches": bad}
    if timed_dec is not None:
        # north_star's "coefficients and PVQ pulse vectors": the coded gain index, itheta,
        # max_theta, K and the pulse vector of EVERY band of every block of every level of
        # frame 0 as the TIMED pipeline left them, against the reference's pvq_theta
        want = []
        C.cpu_frame(qt, gpu_frame0, PIC_W, PIC_H, chroma_cfl=chroma_cfl, decisions=want)
        dbad = C.compare_decisions(timed_dec, want, frame=0, frames=args.frames)
        nbands = int(sum(b.shape[0] * b.shape[1] for plane in want for (_, b) in plane))
        for fi, pics in more_frames:
            want = []
            C.cpu_frame(qt, pics, PIC_W, PIC_H, chroma_cfl=chroma_cfl, decisions=want)
            dbad += [(fi,) + b for b in C.compare_decisions(timed_dec, want, frame=fi, frames=args.frames)]
            nbands += int(sum(b.shape[0] * b.shape[1] for plane in want for (_, b) in plane))
        ver["decisions"] = {"verified": not dbad, "bands_compared": nbands, "mismatches": dbad,
                            "frames_compared": [0] + [fi for fi, _ in more_frames],
                            "what": "gain index, itheta, max_theta, K and pulse vector of every band of those frames "
                                    "(timed pipeline) == the reference's pvq_theta (ref_stage_set_dump)"}
        ver["verified"] = bool(ver["verified"] and not dbad)
    # the partition the reference encoder really codes for this picture (report
    # the coded blocks beside the evaluated ones): the whole encoder, -v 20, complexity 7
    coded = None
    try:
        r = ref()
        frame = np.concatenate([np.ascontiguousarray(p).ravel() for p in gpu_frame0])
        out = np.zeros(4 << 20, np.uint8)
        pk = np.zeros(8, np.int64)
        t0 = time.perf_counter()
        n = r.ref_encode_yuv420(frame.ctypes.data_as(ctypes.c_void_p), PIC_W, PIC_H, 1, 20, 7, 0,
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(out.size),
                                pk.ctypes.data_as(ctypes.c_void_p))
        dt = time.perf_counter() - t0
        cb = (ctypes.c_long * 2)()
        r.ref_last_coded_blocks(cb)
        if n == 1:
            coded = {"luma": int(cb[0]), "chroma_per_plane": int(cb[1]), "total": int(cb[0] + 2 * cb[1]),
                     "packet_bytes": int(pk[0]), "encoder_seconds": dt,
                     "note": "final partition of the unmodified reference encoder (state.bsize after the "
                             "block-size RDO) on frame 0 of the batch; the metric counts EVALUATED blocks "
                             "(every block of every level, what the RDO prices), of which these are coded"}
    except Exception as e:      # noqa: BLE001 - an auxiliary figure must not take the line down
        coded = {"error": repr(e)}
    ver["coded_blocks_frame0"] = coded
    if simd is not None:
        base["simd_build"] = simd
    return base, host, ver


SWEEP_POINTS = (("checker", 5), ("checker", 10), ("checker", 40), ("natural", 5), ("natural", 10), ("natural", 40))


def quality_sweep(D, torch, frames, steps, local_rank, cfl, seed, verify=True):
    """Auxiliary entries (never `value`): the same priced step at other operating points than
    `-v 20` - quantisers 5, 10 and 40 on both content types (K from ~1 to tens of pulses per band) -
    each one timed like the headline (warm-up, `steps` steps, flush, sync) and VERIFIED like it on
    frame 0 of its batch: pixels of every partition level and gain / theta / K / pulses of every band
    against the compiled reference's pvq_theta with that quantiser set-up."""
    global GENERATOR
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _pipeline_check as C
    from _libs import ref
    saved = GENERATOR
    pics = {}
    out = []
    try:
        for content, q in SWEEP_POINTS:
            GENERATOR = CONTENT[content]
            if content not in pics:
                pics[content] = synth_pictures(frames, seed)
            luma_pic, chroma_pic = pics[content]
            qt = D.QuantTables.for_quality(q)
            pipe = D.Pipe(qt, frames, PIC_W, PIC_H, chroma_cfl=cfl, device=local_rank, price=True)
            pipe.set_pictures(luma_pic, chroma_pic)
            for _ in range(2):
                pipe.step()
            pipe.flush()
            pipe.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe.step()
            pipe.flush()
            pipe.sync()
            dt = time.perf_counter() - t0
            ent = {"content": content, "quality": "-v %d" % q, "quantizer": int(qt.quantizer),
                   "ms_per_step": dt / steps * 1e3, "steps": steps,
                   "value": frames * steps * blocks_per_frame() / dt, "unit": "blocks/s",
                   "theta_margin_reruns": pipe.theta_reruns(), "price_margin_reruns": pipe.price_reruns(),
                   "verified": None}
            if verify and ref() is not None:
                F = frames
                recon = [[pipe.read(D.BUF_RECON, 0, bs).reshape(F, H, W) for bs in range(5)],
                         [pipe.read(D.BUF_RECON, 1, bs).reshape(2 * F, H // 2, W // 2) for bs in range(4)]]
                dec = C.gpu_decisions(D, pipe)
                want = []
                frame0 = [luma_pic[0], chroma_pic[0], chroma_pic[F]]
                c0 = time.perf_counter()
                cpu, blocks, _ = C.cpu_frame(qt, frame0, PIC_W, PIC_H, chroma_cfl=cfl, decisions=want)
                cdt = time.perf_counter() - c0
                bad = C.compare_frame(recon, cpu, frame=0, frames=F)
                dbad = C.compare_decisions(dec, want, frame=0, frames=F)
                ks = [int(b[..., 3].sum()) for plane in want for (_, b) in plane]
                nbands = int(sum(b.shape[0] * b.shape[1] for plane in want for (_, b) in plane))
                ent.update({"verified": not bad and not dbad, "bands_compared": nbands,
                            "mean_k_per_band": round(sum(ks) / max(1, nbands), 3),
                            "pixel_mismatches": bad[:4], "decision_mismatches": dbad[:4],
                            "reference_c_blocks_per_s_one_core": blocks / cdt})
            out.append(ent)
            pipe.destroy()
    finally:
        GENERATOR = saved
    return out


def sharded_encode_check(rank, world, local_rank, dist, torch):
    """--gpus N > 1, when the compiled reference travelled with the snapshot: BASELINE
    configs[4] in miniature on the N GPUs - one 1080p frame per rank through the real
    encoder with the batched GPU stage bound (tests/_shard_encode.py), packets gathered to
    rank 0 over RCCL (daala_amd.shard.gather_packets) and compared with the plain C encoder
    run sequentially.  A CHECK outside the timed region (the encoder's host half is the
    reference's C, i.e. test infrastructure here), reported separately from `value`."""
    import encode_job as S
    if not S.reference_available():
        return {"ran": False, "why": "the reference encoder (oracle/_ref/libdaalaref.so) or shim/libdaalahipglue.so absent"}
    from daala_amd.shard import frames_of_rank, gather_packets
    nframes = world
    err = None
    local = {}
    ipo = None
    dist.barrier()
    t0 = time.perf_counter()
    try:
        r, ipo = S.load_batched_encoder(PIC_W, PIC_H, device=local_rank)
        local = S.encode_owned(r, frames_of_rank(nframes, rank, world), PIC_W, PIC_H)
    except Exception as e:      # noqa: BLE001 - a failed check must not take the bench line down
        err = "encode on rank %d: %r" % (rank, e)
    t_enc = time.perf_counter() - t0
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    got = None
    try:
        got = gather_packets(local, nframes)      # symmetric collectives on every rank
    except Exception as e:      # noqa: BLE001
        err = err or "gather: %r" % (e,)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0
    tt = torch.tensor([t_enc, t_gather], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0 and (err or got is None):
        return {"ran": False, "why": err or "gather returned nothing"}
    if rank != 0:
        return None
    try:
        want = sequential_digest(nframes)
    except Exception as e:      # noqa: BLE001
        return {"ran": False, "why": "sequential C encoder: %r" % (e,)}
    st = S.band_stats(ipo)
    return {"ran": True, "frames": nframes, "frames_per_rank": 1,
            "packets_equal_sequential_c_encoder": S.digest(got) == want,
            "packet_bytes": sum(len(p) for p in got),
            "encode_s_max_over_ranks": float(tt[0].item()), "gather_ms": float(tt[1].item()) * 1e3,
            "bands_from_batch_rank0": st[0], "bands_left_to_reference_rank0": st[1] + st[2],
            "note": "one 1920x1080 keyframe per rank, -v 20 -z 7, real reference encoder (host entropy "
                    "coding / pricing) with the batched pyramid + luma PVQ band stage behind it (shim/); gather = "
                    "all_gather of sizes + padded all_gather of bytes over RCCL"}


def pipeline_digest(D, pipe):
    """SHA-256 over every reconstructed plane and choice record of the pipe."""
    import hashlib
    h = hashlib.sha256()
    for set_ in (0, 1):
        for bs in range(5 - set_):
            h.update(pipe.read(D.BUF_RECON, set_, bs).tobytes())
            h.update(pipe.read(D.BUF_CHOICE, set_, bs).tobytes())
    return h.hexdigest()


def source_hash():
    """SHA-256 over the kernel sources (daala_amd/csrc, sorted): the committed PMC counters
    are properties of THESE sources - tools/pmc_summary.py stores the hash beside them and
    the counters are dropped when it differs."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "daala_amd", "csrc")
    for dirpath, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if f.endswith((".hip", ".cuh", ".h")):
                h.update(f.encode())
                with open(os.path.join(dirpath, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc(tag_order=("r6", "r5", "r4", "r3")):
    """(per-kernel counters, source file, stale?) of the committed rocprofv3 --pmc passes."""
    for tag in tag_order:
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        if doc.get("source_hash") != source_hash():
            return {}, "profiles/%s_pmc_traffic.json" % tag, True
        return doc["kernels"], "profiles/%s_pmc_traffic.json" % tag, False
    return {}, None, False


# gfx950: 256 CUs x 4 SIMDs; one fp64 VALU wave-instruction issues in about 4 cycles per
# SIMD at the 2.4 GHz peak clock (tools/ubench/fp64_rate.hip measures 4.9-5.2 for
# v_add/mul/fma_f64) -> the issue-rate roof used for the search kernels
VALU_PEAK_GINSTR = 1024 * 2.4 / 4.0


def step_counters(pmc, src, stale, step_ms):
    """Whole-step figures from the committed counter passes: VALU issue utilisation = SUM
    SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz / step, and the HBM bytes a step moves."""
    if stale:
        return {"stale": True, "why": "%s was collected from other kernel sources (source_hash differs): "
                                      "dropped" % src}
    if not pmc:
        return None
    # kernels of the profiled command that are NOT part of a step: the copy-ceiling measurement, one-time table fills
    step = {k: e for k, e in pmc.items() if not k.startswith(("k_copy16", "k_export_", "k_rsq_", "k_rate_", "k_nrate_"))}
    valu = sum(e.get("valu_wave_instructions") or 0 for e in step.values())
    hbm = sum(e.get("hbm_bytes_per_launch") or 0 for e in step.values())
    return {"valu_wave_instructions_per_step": valu,
            "valu_issue_utilisation": round(valu / (VALU_PEAK_GINSTR * 1e9) / (step_ms * 1e-3), 4),
            "hbm_bytes_per_step": hbm,
            "hbm_GBs_average": round(hbm / (step_ms * 1e-3) / 1e9, 1),
            "source": "%s (every kernel of a step is launched once; per-launch counters summed)" % src}


def search_roofline(name, kernel_prefix, ms_excl, ms_in_step, launches, alg_bytes, bands, step_ms, pmc, src):
    """roofline object of a K-pulse search kernel: fp64 VALU issue is its roof (the HBM
    figure SURVEY 8(d) asks for is kept as `hbm`)."""
    hbm = alg_bytes / (ms_excl * 1e-3) / 1e9
    out = {"kernel": name, "bound": "valu-fp64", "achieved": None, "peak": round(VALU_PEAK_GINSTR, 1),
           "unit": "G wave-instructions/s", "frac": None, "traffic": None,
           "avg_ms_per_launch": round(ms_excl, 4), "avg_ms_per_launch_in_step": round(ms_in_step, 4),
           "launches": launches, "share_of_step": round(ms_in_step / step_ms, 4),
           "algorithmic_bytes_per_launch": alg_bytes, "bands_per_launch": bands,
           "hbm": {"achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(hbm / HBM_PEAK_GBS, 4)},
           "timing": "HIP events on the stream the kernel is launched on; avg_ms_per_launch = exclusive "
                     "(a serial replay of the same step after the timed region), _in_step = inside the "
                     "timed, two-stream step"}
    key = [k for k in pmc if k.startswith(kernel_prefix)]
    if key:
        ent = pmc[key[0]]
        out["traffic"] = ent.get("hbm_bytes_per_launch")
        out["traffic_source"] = "%s (rocprofv3 --pmc, same command)" % src
        vi = ent.get("valu_wave_instructions")
        if vi:
            g = vi / (ms_excl * 1e-3) / 1e9
            out["achieved"] = round(g, 1)
            out["frac"] = round(g / VALU_PEAK_GINSTR, 4)
            out["valu_wave_instructions_per_launch"] = vi
            out["valu_source"] = "%s (SQ_INSTS_VALU per launch, workload-deterministic)" % src
    if out["frac"] is None:
        # no counter file for this workload: fall back to the HBM figure
        out.update({"bound": "hbm", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm / HBM_PEAK_GBS, 4)})
    return out



def write_y4m(path, nframes, w=PIC_W, h=PIC_H):
    """The bench generator's frames 0 .. nframes-1 as a YUV4MPEG2 file (what
    encoder_example is fed, examples/encoder_example.c:89-160)."""
    import encode_job as S
    with open(path, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (w, h)).encode())
        for i in range(nframes):
            f.write(b"FRAME\n")
            f.write(S.frame_yuv(i, w, h).tobytes())


def sequential_digest(nframes):
    """Packet digest of the plain C reference encoder run sequentially on the first frames of the
    bench generator (a child process: this one has the shim bound)."""
    import subprocess
    path = "/tmp/odhip_seq_%d_%d.y4m" % (os.getpid(), nframes)
    write_y4m(path, nframes)
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    try:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--seq-encode", path, str(nframes)],
                            capture_output=True, text=True, timeout=3600, env=e)
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    if pr.returncode != 0:
        raise RuntimeError(pr.stderr[-500:])
    return json.loads(pr.stdout.strip().splitlines()[-1])["digest"]


def seq_encode_worker(args):
    """Child process of encode_mode: the plain C reference encoder (nothing bound) on frames of the
    Y4M file, one after the other on one core.  `--seq-encode path N`: the first N frames;
    `--seq-encode path i,j,k`: exactly those global frame indices.  Prints the digest of the packets
    in index order and the SHA-256 of every packet."""
    import hashlib
    import encode_job as S
    import daala_amd as D
    path, spec = args.seq_encode[0], args.seq_encode[1]
    if args.seq_core >= 0:
        _pin(args.seq_core)
    if "," in spec or spec.startswith("="):
        want = sorted(int(x) for x in spec.lstrip("=").split(",") if x != "")
    else:
        want = list(range(int(spec)))
    last = (want[-1] + 1) if want else 0
    wanted = set(want)
    # every frame up to the last wanted one is walked (read or skipped) by the library's reader
    frames, w, h, got = S.read_y4m_frames_set(D, path, wanted, last)
    idx = [i for i in want if i in frames]
    r = ctypes.CDLL(S.REFERENCE_LIB)
    t0 = time.perf_counter()
    packets = S.encode_frames(r, idx, [frames[i] for i in idx], w, h)
    dt = time.perf_counter() - t0
    print(json.dumps({"digest": S.digest([packets[i] for i in idx]), "frames": len(idx), "seconds": dt,
                      "indices": idx, "sha256": [hashlib.sha256(packets[i]).hexdigest() for i in idx]}))
    return 0


def sequential_c_check(path, indices, packets, ncores):
    """The plain C reference encoder on `indices` (global frame indices), compared packet by packet
    with `packets` (the job's gathered output).  All-intra frames are independent, so the C side is
    spread over `ncores` child processes (frame list j of ncores, each one sequential on its own
    core); the frames/s of ONE such process is the plain-C-on-one-core figure."""
    import hashlib
    import subprocess
    indices = sorted(set(int(i) for i in indices))
    if not indices:
        return None
    ncores = max(1, min(ncores, len(indices)))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    t0 = time.perf_counter()
    procs = []
    for j in range(ncores):
        part = indices[j::ncores]
        errf = open("/tmp/odhip_seqcheck_%d_%d.err" % (os.getpid(), j), "w+")
        procs.append((part, errf, subprocess.Popen(
            [sys.executable, os.path.abspath(__file__), "--seq-encode", path, "=" + ",".join(map(str, part)),
             "--seq-core", str(cores[j % len(cores)])], stdout=subprocess.PIPE, stderr=errf, text=True, env=e)))
    equal = True
    differing = []
    checked = 0
    rates = []
    error = None
    for part, errf, pr in procs:
        out, _ = pr.communicate()
        errf.seek(0)
        err = errf.read()
        errf.close()
        try:
            os.remove(errf.name)
        except OSError:
            pass
        if pr.returncode != 0:
            error = err[-500:]
            continue
        ref_out = json.loads(out.strip().splitlines()[-1])
        rates.append(ref_out["frames"] / max(ref_out["seconds"], 1e-9))
        for i, want in zip(ref_out["indices"], ref_out["sha256"]):
            checked += 1
            if hashlib.sha256(packets[i]).hexdigest() != want:
                equal = False
                differing.append(i)
    if error is not None:
        return {"error": error}
    return {"frames": checked, "packets_equal_sequential_c_encoder": equal and checked == len(indices),
            "differing_frames": differing[:16], "frame_indices": indices if len(indices) <= 64 else
            "%d frames: %d..%d" % (len(indices), indices[0], indices[-1]),
            "c_encoder_processes": ncores, "seconds": round(time.perf_counter() - t0, 2),
            "c_encoder_frames_per_s_one_core": float(np.median(rates)) if rates else None}


def encode_mode(args, D, torch, dist, rank, world, local_rank):
    """BASELINE configs[4]: an N-frame 1080p all-intra encode, frames sharded over the ranks and,
    inside a rank, over P encoder processes that share the rank's GPU (frame i -> encoder
    i mod (world * P); all-intra frames are independent, src/encode.c:303-308, :3029, :3080).
    Every encoder process is the reference encoder (its own host C: entropy coding, pricing,
    block-size RDO - the build of the reference's sources under oracle/_ref) with the batched GPU
    stage bound behind it through the shim (shim/libdaalahipglue.so, explicit configuration: one
    batched pyramid per plane, the PVQ band stage of keyframe luma, the deringing level search),
    input through the library's Y4M reader, coded packets gathered to rank 0 (all_gather of sizes
    + padded all_gather of bytes over RCCL).  Rank 0 checks a prefix against the plain C encoder
    run sequentially and prints one JSON line: frames per second of the whole job.  No 1 -> 8
    GPU curve exists until a multi-GPU box runs this."""
    import subprocess
    import encode_job as S
    from daala_amd.shard import gather_packets
    if not S.reference_available():
        raise SystemExit("--encode-frames needs the reference encoder (oracle/_ref) and shim/libdaalahipglue.so, "
                         "both built by build()")
    nframes = args.encode_frames
    P = max(1, args.procs_per_gpu)
    # encoder threads per process: by default what the host grants - the CPU quota of this box over
    # the ranks that share it (one rank per GPU) and the encoder processes of a rank.  One encoder
    # context is bound by the reference's sequential host chain at ~0.9 frames/s per core while its
    # GPU passes take ~10 ms of a frame, so the job scales with host cores: 16 encoders per GPU is
    # where one MI355X was measured (profiles/r4_encode_mode_300frames.json), fewer means the rank is
    # host-starved (and says so).
    quota = host_cpu_quota()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) or 1
    T_auto = max(1, min(32, quota // max(1, local_world * P)))
    T = args.threads_per_proc if args.threads_per_proc > 0 else T_auto
    if rank == 0 and P * T < 16:
        print("bench.py --encode-frames: %d encoder(s) per GPU (host quota %d CPUs / %d rank(s) on this host / "
              "%d process(es) per rank): below the 16 per GPU one MI355X was measured with - this run is bound "
              "by the host cores it was granted, not by the GPUs" % (P * T, quota, local_world, P),
              file=sys.stderr, flush=True)
    path = args.y4m
    made = False
    if path is None:
        path = "/tmp/odhip_bench_%d.y4m" % nframes
        if rank == 0:
            write_y4m(path, nframes)
        made = True
    if dist is not None:
        dist.barrier()
    ncores = os.cpu_count() or 1
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(ncores))
    # P encoder processes of this rank: encoder e = rank * P + p takes frames e, e + world * P, ...
    t0 = time.perf_counter()
    outs = ["/tmp/odhip_job_%d_%d_%d.npz" % (os.getpid(), rank, p) for p in range(P)]
    procs = []
    errfiles = []
    for p in range(P):
        e_idx = rank * P + p
        cmd = [sys.executable, os.path.join(ROOT, "encode_job.py"), "--worker", "--y4m", path, "--frames", str(nframes),
               "--stride", str(world * P), "--offset", str(e_idx), "--device", str(local_rank),
               "--core", "-1" if T > 1 else str(cores[e_idx % len(cores)]), "--out", outs[p],
               "--gpu-lock", str(int(args.gpu_lock)), "--threads", str(T), "--selfcheck", str(int(args.encode_selfcheck))]
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        # stderr to a file: a child that logs more than a pipe holds before READY would block forever
        errf = open("/tmp/odhip_job_%d_%d_%d.err" % (os.getpid(), rank, p), "w+")
        errfiles.append(errf)
        procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=errf,
                                      text=True, env=env))

    def child_stderr(k):
        errfiles[k].seek(0)
        return errfiles[k].read()[-800:]

    def drop_children(reason):
        # a failed start: the others would see EOF on stdin and encode the whole job for nothing
        for q in procs:
            if q.poll() is None:
                q.kill()
        for q in procs:
            q.wait()
        for f in errfiles:
            f.close()
            try:
                os.remove(f.name)
            except OSError:
                pass
        raise SystemExit(reason)

    for k, pr in enumerate(procs):        # every encoder has read its frames and coded one (untimed)
        line = pr.stdout.readline()
        if line.strip() != "READY":
            drop_children("encoder process failed to start: %s %s" % (line, child_stderr(k)))
    t_start = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for pr in procs:
        pr.stdin.write("go\n")
        pr.stdin.flush()
    stats = []
    for k, pr in enumerate(procs):
        out, _ = pr.communicate()
        if pr.returncode != 0:
            drop_children("encoder process failed: %s" % child_stderr(k))
        stats.append(json.loads(out.strip().splitlines()[-1]))
    for f in errfiles:
        f.close()
        try:
            os.remove(f.name)
        except OSError:
            pass
    t_enc = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    t_all = time.perf_counter() - t0
    local = {}
    for o in outs:
        z = np.load(o)
        pos = 0
        for i, n in zip(z["indices"].tolist(), z["sizes"].tolist()):
            local[int(i)] = bytes(z["bytes"][pos:pos + n])
            pos += n
        os.remove(o)
    nframes = min(nframes, stats[0]["frames_in_file"])
    w, h = stats[0]["w"], stats[0]["h"]
    t1 = time.perf_counter()
    packets = gather_packets(local, nframes) if dist is not None else [local[i] for i in range(nframes)]
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t1
    tt = torch.tensor([t_all, t_enc, t_gather, t_start], dtype=torch.float64,
                      device="cuda" if dist is not None and dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank != 0:
        return 0
    # Which frames the plain C encoder re-encodes (rank 0, after the timed job, spread over the host
    # quota): --encode-check N >= the job -> every frame; N > 0 -> the first N frames AND the LAST frame
    # of every encoder context (encoder e = frames e, e + E, ...; thread t of a process the t-th of
    # them): steady-state frames of every thread, not only the ones coded in the untimed warm-up;
    # -1 (default) -> that per-encoder sample alone; 0 -> no check.
    E = world * P
    n_check = args.encode_check
    check = None
    if n_check != 0:
        if n_check >= nframes:
            idx = list(range(nframes))
        else:
            idx = set(range(max(0, n_check)))
            for e_i in range(E):
                owned = list(range(e_i, nframes, E))
                for t in range(T):
                    mine = owned[t::T]
                    if mine:
                        idx.add(mine[-1])
            idx = sorted(idx)
        check = sequential_c_check(path, idx, packets, quota)
        if check is not None and "error" not in check:
            check["covers"] = "every frame of the job" if len(idx) == nframes else \
                "the first %d frames and the last frame of each of the %d encoder contexts" % (max(0, n_check), E * T)
    t_job = float(tt[0].item())
    nf0 = sum(st["frames"] for st in stats)
    batch_ms = sum(st["batch_ms"] for st in stats)
    line = {
        "metric": "1080p all-intra encode frames/s (frame-sharded over the GPUs of one node)",
        "value": nframes / t_job, "unit": "frames/s", "n_gpus": world, "frames": nframes,
        "encoder_processes_per_gpu": P, "encoder_threads_per_process": T,
        "encoders_per_gpu": P * T,
        "encoder_threads_source": "--threads-per-proc" if args.threads_per_proc > 0 else
        "host_cpu_quota %d // (%d rank(s) on this host x %d process(es))" % (quota, local_world, P),
        "host_starved": P * T < 16, "encoder_selfchecks": int(args.encode_selfcheck),
        "host_cores_available": len(cores), "host_cpu_quota": quota, "gpu_pass_lock": bool(args.gpu_lock),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
        "dtype": "int32 (lifting DCT/filters) + f64 (PVQ search)",
        "config": {"workload": "configs[4]: %d-frame %dx%d all-intra encode (-v 20, complexity 7), frame i -> "
                               "encoder i mod %d (%d GPU(s) x %d encoder processes sharing each GPU), Y4M in "
                               "(odhip_y4m_*), packets gathered to rank 0" % (nframes, w, h, world * P, world, P),
                   "encoder": "the reference's own encoder (host C: entropy coding, od_pvq_rate on the live "
                              "adaptive state, block-size RDO) with the batched GPU stage bound behind it "
                              "through shim/libdaalahipglue.so: pyramids of all planes, the PVQ band stage of "
                              "keyframe luma with batched speed-0 pricing, the deringing level search "
                              "(INTEGRATION.md section 7)"},
        "seconds": {"job_max_over_ranks": t_job, "encode_max_over_ranks": float(tt[1].item()),
                    "gather": float(tt[2].item()), "start_up_untimed_max_over_ranks": float(tt[3].item())},
        "packet_bytes": int(sum(len(p) for p in packets)),
        "rank0": {"frames": nf0, "bands_from_batch": sum(st["bands_from_batch"] for st in stats),
                  "bands_left_to_reference": sum(st["bands_left_to_reference"] for st in stats),
                  "searches_saved": sum(st["searches_saved"] for st in stats),
                  "batched_gpu_pass_ms_per_frame": batch_ms / max(1, nf0),
                  "dering_cache_ms_per_frame": sum(st.get("dering_ms", 0) for st in stats) / max(1, nf0),
                  "served_pvq_theta_ms_per_frame": sum(st.get("theta_ms", 0) for st in stats) / max(1, nf0),
                  "od_compute_dist_served_per_frame": sum(st.get("dist_served", 0) for st in stats) / max(1, nf0),
                  "dering_served_per_frame": sum(st.get("dering_served", 0) for st in stats) / max(1, nf0),
                  "encoder_seconds_per_process": [round(st["seconds"], 2) for st in stats],
                  "stage_blocks_per_s": blocks_per_frame() * nf0 / max(batch_ms * 1e-3, 1e-9)},
        "prefix_check": check,
        "note": "frames/s of the whole job (barrier to barrier, max over ranks); every encoder context is bound "
                "by the reference's sequential entropy coder at ~0.9 frames/s per host core, so a GPU serves many "
                "of them (processes and / or threads): the batched GPU passes take %.0f ms of each frame "
                "(rank0.batched_gpu_pass_ms_per_frame); the job scales with the host cores the box grants "
                "(host_cpu_quota: the cgroup quota, not the 256 logical CPUs it shows)" % (batch_ms / max(1, nf0)),
    }
    print(json.dumps(line))
    if made:
        try:
            os.remove(path)
        except OSError:
            pass
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=DEFAULT_FRAMES, help="1080p frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-allcores", action="store_true")
    ap.add_argument("--no-replay", action="store_true",
                    help="skip the serial replay after the timed region (profiling runs: no extra launches)")
    ap.add_argument("--no-streaming", action="store_true",
                    help="skip the auxiliary run that feeds every step's pictures from pinned host memory")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the auxiliary operating-point sweep (-v 5 / 10 / 40 on both content types, each "
                         "verified on frame 0 against the compiled reference: ~1 minute of host time)")
    ap.add_argument("--no-price", action="store_true",
                    help="choose on distortion alone inside the step (no od_pvq_rate)")
    ap.add_argument("--no-shard-check", action="store_true",
                    help="N > 1: skip the sharded real-encoder check (frames over ranks, RCCL gather)")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--encode-frames", type=int, default=0,
                    help="BASELINE configs[4] instead of the stage benchmark: encode this many 1080p frames "
                         "(300 in the config), frame-sharded over the ranks, with the batched GPU stage behind "
                         "the reference encoder; prints frames/s of the whole job")
    ap.add_argument("--y4m", default=None, help="--encode-frames: a YUV4MPEG2 file to encode (default: the "
                                                "bench generator's frames, written to /tmp)")
    ap.add_argument("--procs-per-gpu", type=int, default=1,
                    help="--encode-frames: encoder processes per GPU (they share the rank's device; the host "
                         "chain of one encoder is sequential, ~1 frame/s, while its GPU passes take ~10 ms)")
    ap.add_argument("--threads-per-proc", type=int, default=0,
                    help="--encode-frames: encoder contexts (host threads) per encoder process; they share the "
                         "process's HIP context, so P x T encoders share a GPU with only P processes on it.  "
                         "Default 0: the host CPU quota // (ranks on this host x processes per rank), at most 32; "
                         "a warning is printed when that leaves fewer than 16 encoders per GPU")
    ap.add_argument("--gpu-lock", type=int, default=0,
                    help="--encode-frames: the batched GPU pass of a frame under a cross-process lock "
                         "(odhip_glue_config.gpu_pass_lock); measured slower than letting the passes "
                         "overlap (profiles/r4_encode_mode_300frames.json), default off")
    ap.add_argument("--encode-check", type=int, default=-1,
                    help="--encode-frames: frames rank 0 re-encodes with the plain C encoder (child processes "
                         "spread over the host CPU quota, each sequential on one core) and compares packet by "
                         "packet: N >= the job = every frame; N > 0 = the first N and the last frame of every "
                         "encoder context; -1 (default) = the last frame of every encoder context; 0 = none")
    ap.add_argument("--encode-selfcheck", type=int, default=0,
                    help="--encode-frames: the shim's cross-checks inside every encoder (bit mask: 1 = every batched "
                         "od_pvq_rate against od_pvq_rate, 2 = every served od_dering superblock against od_dering, "
                         "4 = every served od_compute_dist against the C function; a difference aborts the encoder)")
    ap.add_argument("--seq-encode", nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--seq-core", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--content", choices=sorted(CONTENT), default="checker",
                    help="synthetic picture generator: 'checker' (smooth texture + 32-pixel checker "
                         "edges + uniform noise, independent chroma) or 'natural' (cosines + AR(1) "
                         "noise, chroma correlated with luma)")
    ap.add_argument("--chroma-noref", action="store_true",
                    help="chroma through pvq_theta's no-reference path (default: with the "
                         "chroma-from-luma reference, as the reference encoder codes keyframes)")
    ap.add_argument("--chroma-cfl", action="store_true", help="(default; kept for old command lines)")
    args = ap.parse_args()
    if args.cpu_worker is not None:
        return cpu_worker(args)
    if args.seq_encode is not None:
        return seq_encode_worker(args)

    import torch
    import daala_amd as D

    global GENERATOR
    GENERATOR = CONTENT[args.content]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    # ODHIP_BENCH_ONE_GPU=1 / ODHIP_BENCH_BACKEND=gloo: test hooks - every rank on cuda:0 and the
    # control plane over gloo, so that the N > 1 flow can be exercised on a one-GPU box
    # (tests/test_gpu_bench_multi.py); the driver's runs use one GPU per rank and RCCL
    if os.environ.get("ODHIP_BENCH_ONE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("ODHIP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    D.init(local_rank)
    dist = None
    # ODHIP_BENCH_FORCE_DIST=1: a process group even for ONE rank, so that the RCCL branch of the
    # gather (device tensors, backend "nccl") executes on a one-GPU box (tests/test_gpu_bench_multi.py)
    if world > 1 or os.environ.get("ODHIP_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    device = torch.device("cuda", local_rank)
    if args.encode_frames > 0:
        rc = encode_mode(args, D, torch, dist, rank, world, local_rank)
        if dist is not None:
            dist.destroy_process_group()
        return rc

    cfl = not args.chroma_noref
    qt = D.QuantTables.load()
    luma_pic, chroma_pic = synth_pictures(args.frames, 1234 + rank)
    price = not args.no_price
    pipe = D.Pipe(qt, args.frames, PIC_W, PIC_H, chroma_cfl=cfl, device=local_rank,
                  serial=bool(os.environ.get("ODHIP_PVQ_SERIAL")), price=price)
    pipe.set_pictures(luma_pic, chroma_pic)
    for _ in range(args.warmup):
        pipe.step()
    pipe.flush()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every stage and around the dominant kernel of each band stage,
    # on the streams they are launched on (odhip_pipe_record)
    pipe.record(True)
    barrier()
    wait0 = pipe.host_wait_ms()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        pipe.step()
        host_s += time.perf_counter() - h0
    h0 = time.perf_counter()
    pipe.flush()
    host_wait_s = time.perf_counter() - h0
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    wait_ms = pipe.host_wait_ms() - wait0
    kms = pipe.timings()
    search_ms = pipe.search_timings(False)
    ref_search_ms = pipe.search_timings(True)
    pipe.record(False)
    # The filter + DCT kernel on its own (after the timed steps): in the step it shares the
    # GPU with the other stream's kernels, which stretches its duration.
    fd_alone = pipe.time_pyramid(10)
    # ... and every other stage of the filter + DCT path the same way (odhip_pipe_time_stage: the stage
    # launched 10 times over the buffers the last step left, HIP events on its stream)
    stage_alone = None if args.no_replay else {st: pipe.time_stage(st, 10) for st in FILTER_DCT_STAGES}
    copy_gbs = copy_ceiling_gbs(D)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # what the timed steps left in the reconstruction buffers (for `verified`), before
    # anything else runs on the pipe
    timed_recon = None
    timed_dec = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and price:
        F = args.frames
        timed_recon = [[pipe.read(D.BUF_RECON, 0, bs).reshape(F, H, W) for bs in range(5)],
                       [pipe.read(D.BUF_RECON, 1, bs).reshape(2 * F, H // 2, W // 2) for bs in range(4)]]
        # ... and what it decided: gain index, theta, K and pulse vector of every band
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _pipeline_check as C0
        timed_dec = C0.gpu_decisions(D, pipe)
    # configs[1] says "single 1920x1080 frame": the same step with ONE picture per step
    # (latency-bound) - throughput of pipelined one-picture steps and the
    # latency of one step from enqueue to pixels
    single = None
    if rank == 0 and not args.no_streaming:
        p1 = D.Pipe(qt, 1, PIC_W, PIC_H, chroma_cfl=cfl, device=local_rank, price=price)
        p1.set_pictures(luma_pic[:1], np.ascontiguousarray(chroma_pic[[0, args.frames]]))
        for _ in range(3):
            p1.step()
        p1.flush()
        p1.sync()
        s0 = time.perf_counter()
        for _ in range(20):
            p1.step()
        p1.flush()
        p1.sync()
        thr = (time.perf_counter() - s0) / 20
        s0 = time.perf_counter()
        for _ in range(10):
            p1.step()
            p1.flush()
            p1.sync()
        lat = (time.perf_counter() - s0) / 10
        p1.destroy()
        single = {"frames_per_step": 1, "ms_per_step_pipelined": thr * 1e3, "blocks_per_s": blocks_per_frame() / thr,
                  "latency_ms_enqueue_to_pixels": lat * 1e3,
                  "note": "one 1920x1080 picture per step: ~60 kernel launches over 260 610 blocks - launch / "
                          "latency bound, which is why the metric batches frames"}
    # A stream of pictures instead of resident ones (auxiliary, never `value`): every step
    # codes pictures that arrive from pinned host memory through odhip_pipe_feed, on the
    # pipe's copy stream, while the previous step computes.
    streaming = None
    if rank == 0 and not args.no_streaming:
        hl = torch.from_numpy(luma_pic).pin_memory()
        hc = torch.from_numpy(chroma_pic).pin_memory()
        for _ in range(2):
            pipe.feed(hl, hc)
            pipe.step()
        pipe.flush()
        pipe.sync()
        s0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.feed(hl, hc)
            pipe.step()
        pipe.flush()
        pipe.sync()
        sdt = time.perf_counter() - s0
        h2d = hl.numel() + hc.numel()
        streaming = {"value": args.frames * args.steps * blocks_per_frame() / sdt, "unit": "blocks/s",
                     "ms_per_step": sdt / args.steps * 1e3, "h2d_bytes_per_step": int(h2d),
                     "h2d_GBs": h2d * args.steps / sdt / 1e9,
                     "note": "the same steps with the pictures of every step copied from pinned host memory "
                             "(odhip_pipe_feed: own copy stream, double-buffered, overlapped with the previous "
                             "step) - the input side of the PCIe-inclusive rate; this rank only; the outputs "
                             "stay on the device (DESIGN.md section 5b for what exporting candidates costs)"}
    # ... and the OUTPUT side too (VERDICT r4 missing #5): the same fed steps with what a host entropy
    # coder consumes - the choice record and pulse vector of every band - copied back to pinned host
    # memory on a third stream, overlapped (odhip_pipe_set_export).  Never `value`.
    streaming_io = None
    if rank == 0 and not args.no_streaming and cfl and price:
        nbytes = pipe.export_bytes()
        if nbytes > 0:
            hout = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            pipe.set_export(hout)
            for _ in range(2):
                pipe.feed(hl, hc)
                pipe.step()
            pipe.flush()
            pipe.sync()
            io_steps = max(5, args.steps)
            s0 = time.perf_counter()
            for _ in range(io_steps):
                pipe.feed(hl, hc)
                pipe.step()
            pipe.flush()
            pipe.sync()
            sdt = time.perf_counter() - s0
            shipped = pipe.export_shipped_bytes(hout.numpy())
            stale = pipe.export_stale()
            pipe.set_export(None)
            streaming_io = {"value": args.frames * io_steps * blocks_per_frame() / sdt, "unit": "blocks/s",
                            "ms_per_step": sdt / io_steps * 1e3, "steps": io_steps,
                            "h2d_bytes_per_step": int(h2d), "d2h_bytes_per_step": int(shipped),
                            "d2h_GBs": shipped * io_steps / sdt / 1e9,
                            "d2h_bytes_per_frame": int(shipped // args.frames),
                            "dense_bytes_per_frame": int(sum(
                                2 * pipe.nblocks(s_, b_) * D.pvq_band_layout(b_)[2] for s_ in (0, 1) for b_ in range(5 - s_))
                                // args.frames),
                            "export_buffer_bytes": int(nbytes), "stale_exports": int(stale),
                            "note": "streaming_input plus the decisions of every step - a 4- / 8-byte record (coded gain "
                                    "index, theta and its range, skip / no-reference flags) and the pulses of every "
                                    "band of every level as 16-bit (position, count) words, what a host entropy coder "
                                    "consumes - compacted on the device and shipped to pinned host memory on a third "
                                    "stream behind the stage that produced them (odhip_pipe_set_export, "
                                    "export_kernels.hip; decoded back to the dense buffers in "
                                    "tests/test_gpu_pipeline.py): the PCIe-inclusive rate of the stage with both "
                                    "directions counted.  The step EVALUATES every block of every level; an encoder "
                                    "that exports only the partition it codes moves a fraction of this"}
            del hout
    shard_check = None
    if dist is not None and not args.no_shard_check:
        shard_check = sharded_encode_check(rank, world, local_rank, dist, torch)

    if rank == 0:
        # exclusive durations and the pipelined == serial check: the same steps replayed on
        # ONE stream (own pipe, same pictures), after the timed region
        r_bytes, r_bands = ref128_bytes(D, pipe) if cfl else (0, 0)
        if args.no_replay:
            digest, serial_digest = None, ""
            excl, search_excl, ref_search_excl = kms, search_ms, ref_search_ms
        else:
            digest = pipeline_digest(D, pipe)
            serial = D.Pipe(qt, args.frames, PIC_W, PIC_H, chroma_cfl=cfl, device=local_rank, serial=True,
                            price=price)
            serial.set_pictures(luma_pic, chroma_pic)
            serial.step()
            serial.flush()
            serial.record(True)
            for _ in range(3):
                serial.step()
            serial.flush()
            serial.sync()
            excl = serial.timings()
            search_excl = serial.search_timings(False)
            ref_search_excl = serial.search_timings(True)
            serial_digest = pipeline_digest(D, serial)
            serial.destroy()

        if stage_alone is None:      # profiling runs (--no-replay): no extra launches, the in-step durations
            stage_alone = {st: kms[st][0] for st in FILTER_DCT_STAGES}
        bpf = blocks_per_frame()
        total_blocks = world * args.frames * args.steps * bpf
        step_ms = dt / args.steps * 1e3
        ab = algorithmic_bytes(args.frames)
        kernels = {}
        for key, (ms, count) in kms.items():
            ent = {"avg_ms_per_launch": round(ms, 4), "launches": count,
                   "share_of_step": round(ms * count / args.steps / step_ms, 4)}
            if key in excl:
                ent["exclusive_avg_ms"] = round(excl[key][0], 4)
            if key in ab:
                ent["algorithmic_bytes_per_launch"] = ab[key]
                ent["achieved_GBs"] = round(ab[key] / (ms * 1e-3) / 1e9, 1)
                ent["frac_of_hbm_peak"] = round(ent["achieved_GBs"] / HBM_PEAK_GBS, 4)
            kernels[key] = ent
        pmc, pmc_src, pmc_stale = load_pmc()
        if args.frames != DEFAULT_FRAMES or not cfl or args.content != "checker" or not price:
            pmc, pmc_src = {}, None       # the committed counters are of the default command
        # roofline = the single kernel with the largest EXCLUSIVE time among the kernels
        # the library brackets: the two K-pulse searches of the 128-coefficient bands.
        n128 = 0
        for (w, h, planes, top) in ((W, H, args.frames, 4), (W // 2, H // 2, 2 * args.frames, 3)):
            if cfl and top == 3:
                continue     # chroma goes through the with-reference stage
            for bs in range(top + 1):
                n128 += planes * (w // (4 << bs)) * (h // (4 << bs)) * [0, 0, 1, 3, 3][bs]
        # priced step: k_decide_pair128 (the band prepared, searched and decided by a lane pair):
        # per band 4n B coefficients in, 2n B pulses (chosen candidate) + 16 B choice out = 784 B;
        # --no-price: k_search<128,2,1>: 2n B x16 + 32 B record head in, 2 x 2n B pulses + 32 B
        # record tail out = 6n + 64 = 832 B
        if price:
            roof_noref = search_roofline(
                "k_decide_pair128 (the 128-coefficient luma bands, no reference: prepared, searched and "
                "decided by a lane pair)",
                "k_decide_pair128", float(np.mean(search_excl)), float(np.mean(search_ms)), len(search_ms),
                n128 * (6 * 128 + 16), n128, step_ms, pmc, pmc_src)
        else:
            roof_noref = search_roofline(
                "k_search<128,2,1> (PVQ search of the 128-coefficient luma bands, no reference)",
                "k_search<128", float(np.mean(search_excl)), float(np.mean(search_ms)), len(search_ms),
                n128 * (6 * 128 + 64), n128, step_ms, pmc, pmc_src)
        roof = roof_noref
        roof_ref = None
        if cfl and ref_search_ms:
            roof_ref = search_roofline(
                "k_refb_lean_row<8,16> (with-reference candidate chains of the 128-coefficient chroma "
                "bands, one band per 16-lane row, the band decided inside the search)",
                "k_refb_lean_row<8, 16>" if price else "k_refb_search_row<8, 16>",
                float(np.mean(ref_search_excl)), float(np.mean(ref_search_ms)), len(ref_search_ms),
                r_bytes, r_bands, step_ms, pmc, pmc_src)
            if roof_ref["avg_ms_per_launch"] >= roof_noref["avg_ms_per_launch"]:
                roof = roof_ref
        roof["note"] = ("the kernel with the largest exclusive duration of the step.  The K-pulse search "
                        "is fp64 VALU-issue bound (13-14 VALU instructions per candidate per pulse, "
                        "DESIGN.md): achieved = SQ_INSTS_VALU per launch (rocprofv3 --pmc, committed "
                        "profile of this command) / exclusive duration measured live; `hbm` is the "
                        "algorithmic-bytes figure SURVEY 8(d) asks for.  roofline_filter_dct is the stage "
                        "the north star prices at >= 60 % of HBM")
        fd = kernels["forward_pyramid_luma"]
        fd_gbs = ab["forward_pyramid_luma"] / (fd_alone * 1e-3) / 1e9
        roof_fd = {"kernel": "k_forward_pyramid64x2 (forward_pyramid_luma)", "bound": "hbm",
                   "achieved": round(fd_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(fd_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                   "avg_ms_per_launch": round(fd_alone, 4), "launches": 10,
                   "algorithmic_bytes_per_launch": ab["forward_pyramid_luma"],
                   "in_step": {"avg_ms_per_launch": fd["avg_ms_per_launch"],
                               "achieved_GBs": fd["achieved_GBs"], "frac": fd["frac_of_hbm_peak"]},
                   "copy_1GiB_GBs": round(copy_gbs, 1),
                   "frac_of_copy": round(fd_gbs / copy_gbs, 4),
                   "note": "copy_1GiB_GBs = the same run's device-to-device copy of 1 GiB by the library's own "
                           "16-byte-vector kernel (odhip_copy_ceiling; read + written), the practical ceiling "
                           "SURVEY 8(d) asks for beside the 8 TB/s spec; "
                           "timed alone after the steps (HIP events, 10 launches); in_step = the same "
                           "launch inside the step, where it shares the GPU with the other stream"}
        key = [k_ for k_ in pmc if k_.startswith("k_forward_pyramid64x2")]
        if key:
            roof_fd["traffic"] = pmc[key[0]]["hbm_bytes_per_launch"]
        # the rest of the filter + DCT stage the north star prices (od_postfilter_split + iDCT,
        # src/filter.c:1485-1527, src/dct.c:4890-4920; chroma; the padding in front)
        roof_fd_chroma = stage_roofline(
            "forward_pyramid_chroma", "k_forward_pyramid<32> (4:2:0 chroma, 4 levels)", ab["forward_pyramid_chroma"],
            stage_alone["forward_pyramid_chroma"], kernels.get("forward_pyramid_chroma"), pmc, ("k_forward_pyramid<32",))
        roof_inv_luma = stage_roofline(
            "dequant_inverse_luma", "k_inverse_walk<64,1,256,1,0,2> + k_inverse_sb_top2 + k_edge_rows/cols "
            "(dequantise on load + iDCT + od_postfilter_split + superblock edges + pixels, 5 levels)",
            ab["dequant_inverse_luma"], stage_alone["dequant_inverse_luma"], kernels.get("dequant_inverse_luma"),
            pmc, ("k_inverse_walk<64", "k_inverse_sb_top2", "k_inverse_sb<64",
                  "k_edge_rows grid %d" % (1280 * 14 * 5 * args.frames), "k_edge_cols grid %d" % (2048 * 16 * 5 * args.frames)))
        roof_inv_chroma = stage_roofline(
            "dequant_inverse_chroma", "k_inverse_walk<32,2,128,...> + k_edge_rows/cols (4 levels, with-reference "
            "synthesis on load)" if cfl else "k_inverse_walk<32,2,128,1,...> + k_edge_rows/cols (4 levels)",
            ab["dequant_inverse_chroma"], stage_alone["dequant_inverse_chroma"], kernels.get("dequant_inverse_chroma"),
            pmc, ("k_inverse_walk<32", "k_inverse_sb<32",
                  "k_edge_rows grid %d" % (768 * 4 * 8 * args.frames), "k_edge_cols grid %d" % (1024 * 16 * 8 * args.frames)))
        st_bytes = sum(ab[st] for st in FILTER_DCT_STAGES)
        st_ms = sum(stage_alone[st] for st in FILTER_DCT_STAGES)
        roof_stage = {"stage": "padding + forward pyramids + dequantise / inverse / post-filter / edges, luma and "
                               "chroma, every level", "bound": "hbm",
                      "achieved": round(st_bytes / (st_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(st_bytes / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "algorithmic_bytes_per_step": st_bytes, "ms_per_step_alone": round(st_ms, 4),
                      "ms": {st: round(stage_alone[st], 4) for st in FILTER_DCT_STAGES},
                      "note": "SURVEY 8(d) bytes of the six stages / the sum of their durations, each timed alone "
                              "(10 launch groups, HIP events on the stage's stream)"}
        line = {
            "metric": "1080p all-intra transform blocks/s (filter+DCT+PVQ)",
            "value": total_blocks / dt,
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            # the host's share: one C call per step (odhip_pipe_step).  launch = enqueuing the
            # ~100 kernel launches of a step; wait = blocked on the count of bands inside the
            # device-acos margin of the PREVIOUS step's chroma band stage and, with pricing,
            # on the counts of priced decisions inside the device-log margin (this step's luma
            # choice, the previous step's chroma choice) - the host never computes in a step
            # unless a count is non-zero
            "host_launch_ms_per_step": (host_s * 1e3 - (wait_ms - host_wait_s * 1e3)) / args.steps,
            "host_wait_ms_per_step": wait_ms / args.steps,
            "host_note": "launch = host time enqueuing a step (the host's share of the step: it must stay well "
                         "below ms_per_step); wait = host idle, blocked until the GPU has finished the previous "
                         "step's chroma band stage (it runs one step ahead by design and is otherwise free)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (lifting DCT/filters) + f64 (PVQ search)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 1920x1080 4:2:0 all-intra frames, full "
                       "4/8/16/32/64 lapped-DCT pyramid + PVQ band stages + inverse, every block of "
                       "every level" + (
                           "; luma through pvq_theta's no-reference path, chroma through the "
                           "with-reference (chroma-from-luma) path"
                           if cfl else "; chroma through the no-reference PVQ path"),
                       "frames_per_gpu_per_step": args.frames, "content": args.content,
                       "blocks_per_frame": bpf, "quality": "-v 20 (quantizer 243)",
                       "choice": ("pvq_theta's: distortion + lambda * od_pvq_rate, closed-form rate (speed > 0, "
                                  "src/pvq_encoder.c:250-264) evaluated on the device inside the timed step"
                                  if price else
                                  "on distortion alone inside the timed step; `verified` runs the "
                                  "host-priced choice"),
                       "inputs": "resident in HBM (PCIe-inclusive rates: DESIGN.md section 5)",
                       "driver": "odhip_pipe_step: one C call per step, two streams, one odhip_ctx "
                                 "per chain",
                       "sharding": "frames over ranks, no data-path collective"},
            "roofline": roof,
            "roofline_filter_dct": roof_fd,
            "roofline_filter_dct_chroma": roof_fd_chroma,
            "roofline_inverse_luma": roof_inv_luma,
            "roofline_inverse_chroma": roof_inv_chroma,
            "roofline_filter_dct_stage": roof_stage,
            "roofline_noref_search": roof_noref if roof is not roof_noref else None,
            "roofline_ref_search": roof_ref if (roof_ref is not None and roof is not roof_ref) else None,
            "pipelined_equals_serial": None if digest is None else digest == serial_digest,
            "theta_margin_reruns": pipe.theta_reruns(),
            "price_margin_reruns": pipe.price_reruns() if price else None,
            "streaming_input": streaming,
            "streaming_io": streaming_io,
            "single_frame_step": single,
            "step_counters": step_counters(pmc, pmc_src, pmc_stale, step_ms),
            "kernels": kernels,
        }
        if shard_check is not None:
            line["sharded_encode_check"] = shard_check
        if world == 1 and not args.no_cpu_baseline:
            frame0 = [luma_pic[0], chroma_pic[0], chroma_pic[args.frames]]
            # the middle and the last frame of the batch are verified like frame 0 (the
            # other frames were only covered by the pipelined == serial self-comparison)
            F = args.frames
            more = [(fi, [luma_pic[fi], chroma_pic[fi], chroma_pic[F + fi]])
                    for fi in sorted({F // 2 - (1 if F % 2 == 0 and F > 2 else 0), F - 1} - {0})] if price else []
            base, host, ver = cpu_baseline(D, qt, cfl, args, frame0, timed_recon, timed_dec, more)
            if base is not None:
                line["cpu_baseline"] = base
                line["speedup_vs_cpu_baseline"] = line["value"] / base["value"]
            if host is not None:
                line["cpu_baseline_all_cores"] = host
                line["speedup_vs_all_host_cores"] = line["value"] / host["value"]
            line["verified"] = ver["verified"]
            line["verification"] = ver
            if price and cfl and not args.no_sweep:
                # other operating points: -v 5 / 10 / 40 on both content types
                line["quality_sweep"] = quality_sweep(D, torch, args.frames, max(3, args.steps // 2), local_rank,
                                                      cfl, 1234 + rank)
            line["coded_blocks_per_frame"] = ver.pop("coded_blocks_frame0", None)
        print(json.dumps(line))
    pipe.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
