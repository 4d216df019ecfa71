#!/usr/bin/env python3
"""bench.py - 1080p all-intra transform blocks/s (filter + DCT + PVQ) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--chroma-noref]

One STEP = one pass of the block-transform hot path over a batch of F synthetic
1920x1080 4:2:0 frames (coded size 1920x1088) already resident
in HBM, per GPU:

  0. odhip_image_planes_copy_pad   the 1920x1080 pictures -> the padded 1920x1088
                             planes the encoder codes (od_img_plane_copy_pad)
  1. odhip_forward_pyramid   pixels -> coefficients, superblock-edge lapping,
                             and for EVERY block size 64..4 the split
                             pre-filter + 2-D fDCT of every block (luma 5
                             levels, chroma 4)
  2. luma: odhip_pvq_noref_bands_multi   every level: QM scaling, gain, both gain
                             candidates (K, pruning, K-pulse search, distortion)
           odhip_pvq_choose_multi        choice, od_gain_expand, synthesis scale
     chroma: odhip_pvq_ref_bands_multi (+ odhip_pvq_ref_resolve) - pvq_theta WITH
                             the chroma-from-luma reference, as the reference
                             encoder codes keyframe chroma: sign flip,
                             Householder reflection, theta, up to 12 (gain,
                             theta) candidates + 2 no-reference candidates per
                             band with chained K-pulse searches
             odhip_pvq_ref_choose_multi         choice, skip rules, band-wide part of
                             the synthesis
  3. for every level: inverse (dequantisation of the chosen pulses while the
                             tiles are loaded - with and without reference; iDCT,
                             split post-filters, superblock-edge post-filter,
                             coefficient -> pixel)

i.e. every block the reference's block-size RDO would evaluate goes through
prefilter + fDCT + PVQ + dequantisation + iDCT + postfilter exactly once:
260 610 transform blocks per frame (173 910 luma + 2 x 43 350 chroma).  The
metric counts those blocks.  Entropy coding / rate pricing stay on the host in
the reference's own C (SURVEY.md hard part 1) and are not part of the step; the
choice between PVQ candidates is therefore made on distortion alone.

The chroma-from-luma reference planes of a step are produced inside the step from
that step's luma band stage (odhip_cfl_refs_from_luma = od_resample_luma_coeffs,
src/intra.c:97-108, on the chosen luma candidates); the chroma chain waits for them
on its own stream while the luma chain of the NEXT step already runs (two reference
buffers; software pipelining over steps, as a frame-parallel encoder would run).
--chroma-noref runs chroma through the no-reference path instead (the workload of the
first round-1 bench lines).

N > 1: frames are sharded over ranks (independent all-intra frames, no
data-path collective) -> weak scaling, F frames per GPU.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that dominates the
step; `kernels` lists every kernel class of the step the same way.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

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


def synth_frames(nframes, seed, device):
    """F pictures resident in HBM: luma [F,1080,1920], chroma [2F,540,960] (all Cb,
    then all Cr)."""
    import torch
    fr = [picture_planes(GENERATOR(i, seed)) for i in range(nframes)]
    luma = torch.from_numpy(np.stack([f[0] for f in fr])).to(device)
    chroma = torch.from_numpy(np.stack([f[1] for f in fr] + [f[2] for f in fr])).to(device)
    return luma.contiguous(), chroma.contiguous()


class Pipeline:
    """The GPU step.  All buffers are allocated once; a step only launches
    kernels: 2 pyramid launches, the multi-job PVQ band stage and choice covering
    all nine (plane set, level) jobs, and one inverse launch group per plane set
    (all its levels)."""

    def __init__(self, D, nframes, device, chroma_cfl=False):
        import torch
        self.D = D
        self.chroma_cfl = chroma_cfl
        self.torch = torch
        self.F = nframes
        self.qt = D.QuantTables.load()
        self.lam = D.OD_PVQ_LAMBDA
        self.luma_pic, self.chroma_pic = synth_frames(nframes, 1234 + int(os.environ.get("RANK", 0)),
                                                      device)
        # padded planes the encoder codes (od_img_plane_copy_pad, done on the GPU every step)
        self.luma = D.image_planes_copy_pad(self.luma_pic, W, H)
        self.chroma = D.image_planes_copy_pad(self.chroma_pic, W // 2, H // 2)
        self.sets = []
        self.jobs = []
        for name, px, pic, dec, pli in (("luma", self.luma, self.luma_pic, 0, 0),
                                        ("chroma", self.chroma, self.chroma_pic, 1, 1)):
            levels = D.forward_pyramid(px, dec, PIC_W, PIC_H)
            s = dict(name=name, px=px, pic=pic, dec=dec, pli=pli, levels=levels, jobs=[],
                     recon=[torch.empty_like(px) for _ in range(5 - dec)])
            for bs in range(5 - dec):
                qm, qmi = self.qt.qm_slices(pli, bs)
                job = D.PvqJob(levels[bs], bs, torch.from_numpy(qm).to(device),
                               torch.from_numpy(qmi).to(device), self.qt.q_band(pli, bs),
                               self.qt.beta_band(pli, bs))
                s["jobs"].append(job)
                self.jobs.append(job)
            self.sets.append(s)
        self.timers = {}
        self.refjobs = []
        if chroma_cfl:
            self._setup_chroma_cfl(device)

    def _setup_chroma_cfl(self, device):
        """Keyframe chroma goes through pvq_theta's WITH-reference path, as in the
        reference encoder (chroma-from-luma, src/encode.c:1680-1687).  The reference
        planes of a step come from THAT step's luma band stage
        (odhip_cfl_refs_from_luma = od_resample_luma_coeffs on the chosen luma
        candidates), in two alternating buffers so that the luma chain of step i+1
        can run while the chroma chain of step i still reads its references."""
        D, torch = self.D, self.torch
        luma, chroma = self.sets
        self.noref_jobs = luma["jobs"]
        # ODHIP_PVQ_SERIAL=1 (profiling: exclusive kernel durations) keeps everything on one stream
        self.side = (torch.cuda.current_stream() if os.environ.get("ODHIP_PVQ_SERIAL")
                     else torch.cuda.Stream(device=device))
        D.pvq_noref_bands_multi(luma["jobs"], self.lam)
        D.pvq_choose_multi(luma["jobs"], self.lam)
        self.refs = [D.cfl_refs_from_luma(luma["jobs"][1:], copies=2) for _ in range(2)]
        self.refjobs = [[], []]
        for bs in range(4):
            cj = chroma["jobs"][bs]
            first = D.PvqRefJob(cj.coef, self.refs[0][bs], bs, cj.qm, cj.qm_inv,
                                self.qt.q_band(1, bs), self.qt.beta_band(1, bs), 1, 1)
            self.refjobs[0].append(first)
            self.refjobs[1].append(D.PvqRefJob(cj.coef, self.refs[1][bs], bs, cj.qm, cj.qm_inv,
                                               self.qt.q_band(1, bs), self.qt.beta_band(1, bs), 1, 1,
                                               share=first))
        self.ev_refs = [torch.cuda.Event() for _ in range(2)]     # references of parity p written
        self.ev_used = [torch.cuda.Event() for _ in range(2)]     # ... and no longer read
        self.nstep = 0
        self.pending = None
        torch.cuda.synchronize()

    def _timed(self, key, fn, record):
        if not record:
            return fn()
        t = self.torch
        a = t.cuda.Event(enable_timing=True)
        b = t.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.timers.setdefault(key, []).append((a, b))
        return r

    def step(self, record=False):
        D = self.D
        if self.chroma_cfl:
            return self._step_cfl(record)
        for s in self.sets:
            self._timed("image_copy_pad_" + s["name"],
                        lambda: D.image_planes_copy_pad(s["pic"], W >> s["dec"], H >> s["dec"],
                                                        out=s["px"]), record)
            self._timed("forward_pyramid_" + s["name"],
                        lambda: D.forward_pyramid(s["px"], s["dec"], PIC_W, PIC_H,
                                                  levels=s["levels"]), record)
        self._timed("pvq_noref_bands", lambda: D.pvq_noref_bands_multi(self.jobs, self.lam),
                    record)
        self._timed("pvq_choose", lambda: D.pvq_choose_multi(self.jobs, self.lam), record)
        for s in self.sets:
            # every level of the plane set in one set of launches, one
            # reconstruction per level (what a block-size decision compares)
            self._timed("dequant_inverse_" + s["name"],
                        lambda: D.inverse_levels_pvq(s["jobs"], s["dec"], PIC_W, PIC_H,
                                                     outs=s["recon"]), record)

    def _chroma_tail(self, jobs, record):
        D = self.D
        chroma = self.sets[1]
        # choice, then the inverse of all four levels with the chosen candidates dequantised
        # while the tiles are loaded (no dequantised chroma plane in HBM)
        self._timed("pvq_ref_choose", lambda: D.pvq_ref_choose_multi(jobs, self.lam), record)
        self._timed("dequant_inverse_chroma",
                    lambda: D.inverse_levels_pvq_ref(jobs, 1, PIC_W, PIC_H, outs=chroma["recon"]), record)

    def _finish_pending(self, record):
        """The count of bands inside the device-acos margin of the previous step's
        with-reference stage: checked one step late, so the host never waits inside a
        step; a listed band whose theta the host corrects (never seen outside the
        forced tests) repeats what consumed it."""
        if self.pending is None:
            return
        jobs, self.pending = self.pending, None
        with self.torch.cuda.stream(self.side):
            if self.D.pvq_ref_resolve_finish(jobs, self.lam) > 0:
                self._chroma_tail(jobs, record)

    def flush(self):
        if self.chroma_cfl:
            self._finish_pending(False)

    def _step_cfl(self, record):
        """Software-pipelined over steps: luma (no-reference path) on the current stream,
        chroma (with-reference path) on a side stream.  The chroma chain of a step waits
        for that step's chroma-from-luma references (written by the luma chain after its
        choice); the luma chain of the NEXT step overlaps with it."""
        D, torch = self.D, self.torch
        luma, chroma = self.sets
        main = torch.cuda.current_stream()
        par = self.nstep & 1
        self.nstep += 1
        jobs = self.refjobs[par]
        self._timed("image_copy_pad_luma",
                    lambda: D.image_planes_copy_pad(luma["pic"], W, H, out=luma["px"]), record)
        self._timed("forward_pyramid_luma",
                    lambda: D.forward_pyramid(luma["px"], 0, PIC_W, PIC_H, levels=luma["levels"]),
                    record)
        self._timed("pvq_noref_bands", lambda: D.pvq_noref_bands_multi(self.noref_jobs, self.lam),
                    record)
        self._timed("pvq_choose", lambda: D.pvq_choose_multi(self.noref_jobs, self.lam), record)
        main.wait_event(self.ev_used[par])      # step i-2 no longer reads this reference buffer
        self._timed("cfl_refs_from_luma",
                    lambda: D.cfl_refs_from_luma(self.noref_jobs[1:], refs=self.refs[par], copies=2),
                    record)
        self.ev_refs[par].record(main)
        self._timed("dequant_inverse_luma",
                    lambda: D.inverse_levels_pvq(luma["jobs"], 0, PIC_W, PIC_H, outs=luma["recon"]),
                    record)
        self._finish_pending(record)
        with torch.cuda.stream(self.side):
            self._timed("image_copy_pad_chroma",
                        lambda: D.image_planes_copy_pad(chroma["pic"], W // 2, H // 2,
                                                        out=chroma["px"]), record)
            self._timed("forward_pyramid_chroma",
                        lambda: D.forward_pyramid(chroma["px"], 1, PIC_W, PIC_H,
                                                  levels=chroma["levels"]), record)
            self.side.wait_event(self.ev_refs[par])
            self._timed("pvq_ref_bands",
                        lambda: D.pvq_ref_bands_multi(jobs, self.lam, resolve="async"), record)
            self._chroma_tail(jobs, record)
            self.ev_used[par].record(self.side)
        self.pending = jobs

    def pyramid_alone_ms(self, n=10):
        """Average duration of the luma forward pyramid launched on an otherwise idle GPU."""
        t, D = self.torch, self.D
        luma = self.sets[0]
        t.cuda.synchronize()
        a = t.cuda.Event(enable_timing=True)
        b = t.cuda.Event(enable_timing=True)
        D.forward_pyramid(luma["px"], 0, PIC_W, PIC_H, levels=luma["levels"])
        a.record()
        for _ in range(n):
            D.forward_pyramid(luma["px"], 0, PIC_W, PIC_H, levels=luma["levels"])
        b.record()
        t.cuda.synchronize()
        return a.elapsed_time(b) / n

    def copy_ceiling_gbs(self, n=10):
        """Same-run practical HBM ceiling: a 1 GiB device-to-device copy,
        read + written bytes per second."""
        t = self.torch
        a = t.empty(1 << 30, dtype=t.uint8, device=self.luma.device)
        b = t.empty_like(a)
        b.copy_(a)
        t.cuda.synchronize()
        e0 = t.cuda.Event(enable_timing=True)
        e1 = t.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            b.copy_(a)
        e1.record()
        t.cuda.synchronize()
        return 2.0 * (1 << 30) * n / (e0.elapsed_time(e1) * 1e-3) / 1e9

    def ref128_bytes(self):
        """Algorithmic bytes and band count of one k_refb_search_row<8> launch,
        counted from the records and candidate vectors the last step left."""
        t = self.torch
        total = 0
        bands = 0
        for job in self.refjobs[0]:
            for b in range(job.nb):
                if job.offsets[b + 1] - job.offsets[b] != 128:
                    continue
                rec = job.band[:, b, :].contiguous().view(t.int32)       # [B][16]
                nitems = rec[:, 10].to(t.int64)
                ntheta = rec[:, 11].to(t.int64)
                tail = job.items[1, b].contiguous().view(t.int32)        # [slot][B][4]
                slot = t.arange(tail.shape[0], device=tail.device).view(-1, 1)
                valid = slot < nitems.view(1, -1)
                searched = valid & ((tail[:, :, 2] & 1) != 0)
                stored = searched & (tail[:, :, 3] == slot)
                total += int((64 + 254 * (ntheta > 0) + 256 * (nitems > ntheta) + 32 * nitems).sum())
                total += int(16 * searched.sum() + 256 * stored.sum())
                bands += rec.shape[0]
        return total, bands

    def kernel_ms(self):
        """Average milliseconds per launch group and groups per run, per class."""
        out = {}
        for key, evs in self.timers.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = (float(np.mean(ms)), len(ms))
        return out


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


def cpu_baseline(qt, chroma_cfl, min_seconds=12.0, max_frames=32):
    """The same per-block work on ONE host core with the reference's own C
    functions (oracle/_ref, kind 'reference') or, when that library is absent,
    the oracle port (no-reference chroma only).  Bounded sample: whole frames of
    the bench generator until at least `min_seconds` of CPU work (about 10-30 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import P, oracle, ref
    r = ref()
    kind = "reference" if r is not None else "port"
    degraded = r is None and chroma_cfl
    if degraded:
        # oracle/_ref (the compiled reference, prebuilt by build()) did not travel: time the
        # oracle port, which has the no-reference stage only, and say so in `sample`
        chroma_cfl = False
    tables = []
    for p in (0, 1):
        qm_off = (ctypes.c_int * 5)(*[int(qt.qm_offset[bs][p]) for bs in range(5)])
        qb = (ctypes.c_int * 60)()
        bb = (ctypes.c_int * 60)()
        for bs in range(5):
            for i, v in enumerate(qt.q_band(p, bs)):
                qb[bs * 12 + i] = v
            for i, v in enumerate(qt.beta_band(p, bs)):
                bb[bs * 12 + i] = v
        tables.append((qm_off, qb, bb))
    qm = np.ascontiguousarray(qt.qm)
    qmi = np.ascontiguousarray(qt.qm_inv)
    lam = ctypes.c_double(0.147)
    blocks = 0
    nframes = 0
    busy = 0.0
    if r is not None:
        r.ref_stage_plane.restype = ctypes.c_long
        r.ref_stage_plane_cfl.restype = ctypes.c_long
    while busy < min_seconds and nframes < max_frames:
        pics = picture_planes(GENERATOR(1000 + nframes, 1234))  # generation is not timed
        ldq = [np.zeros((H, W), np.int32) for _ in range(5)] if chroma_cfl else None
        refs = None
        for pli, pic, dec in ((0, pics[0], 0), (1, pics[1], 1), (2, pics[2], 1)):
            p = 1 if pli else 0
            h, w = H >> dec, W >> dec
            qm_off, qb, bb = tables[p]
            pic = np.ascontiguousarray(pic)
            px = np.zeros((h, w), np.uint8)
            recon = np.zeros_like(px)
            t0 = time.perf_counter()
            # od_img_plane_copy_pad (static in the reference's encode.c: the pinned restatement)
            oracle().odo_img_plane_copy_pad(P(px), w, w, h, P(pic), pic.shape[1], pic.shape[1],
                                            pic.shape[0])
            if r is None:
                o = oracle()
                o.odo_stage_plane.restype = ctypes.c_long
                blocks += o.odo_stage_plane(P(px), w, w, h, dec, PIC_W, PIC_H, p, P(qm), P(qmi),
                                            qm_off, qb, bb, lam, 1, P(recon))
            elif not chroma_cfl:
                blocks += r.ref_stage_plane(P(px), w, w, h, dec, PIC_W, PIC_H, p, P(qm), P(qmi),
                                            qm_off, qb, bb, lam, P(recon))
            elif pli == 0:
                arr = (ctypes.c_void_p * 5)(*[a.ctypes.data for a in ldq])
                blocks += r.ref_stage_plane_cfl(P(px), w, w, h, 0, PIC_W, PIC_H, 0, P(qm), P(qmi),
                                                qm_off, qb, bb, lam, P(recon), arr, None)
            else:
                arr = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in refs] + [None]))
                blocks += r.ref_stage_plane_cfl(P(px), w, w, h, 1, PIC_W, PIC_H, 1, P(qm), P(qmi),
                                                qm_off, qb, bb, lam, P(recon), None, arr)
            busy += time.perf_counter() - t0
            if chroma_cfl and pli == 0:
                # chroma-from-luma predictions: od_resample_luma_coeffs for luma blocks one
                # size up (src/intra.c:97-108, a strided copy), timed like the GPU's kernel
                t0 = time.perf_counter()
                refs = []
                for bs in range(4):
                    n = 4 << bs
                    c = ldq[bs + 1].reshape(H // (2 * n), 2 * n, W // (2 * n), 2 * n)[:, :n, :, :n]
                    refs.append(np.ascontiguousarray(c.reshape(H // 2, W // 2)))
                busy += time.perf_counter() - t0
        nframes += 1
    if degraded:
        what = ("padding + forward pyramid + pvq_theta noref bands (oracle/_ref absent: chroma through "
                "the no-reference port, less work than the GPU step) + inverse")
    else:
        what = None
    what = what or ("padding + forward pyramid + pvq_theta (luma: no-reference bands; chroma: WITH the "
            "chroma-from-luma reference) + inverse" if chroma_cfl else
            "padding + forward pyramid + pvq_theta noref bands + inverse")
    return {"value": blocks / busy, "unit": "blocks/s", "cores": 1, "kind": kind,
            "sample": "%d synthetic 1920x1080 4:2:0 pictures of the bench generator (%d blocks) in "
                      "%.1f s: %s of every block at every level, %s, single "
                      "thread" % (nframes, blocks, busy, what,
                                  "reference C functions" if kind == "reference" else "oracle port")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=DEFAULT_FRAMES, help="1080p frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--content", choices=sorted(CONTENT), default="checker",
                    help="synthetic picture generator: 'checker' (smooth texture + 32-pixel checker "
                         "edges + uniform noise, independent chroma) or 'natural' (cosines + AR(1) "
                         "noise, chroma correlated with luma)")
    ap.add_argument("--chroma-noref", action="store_true",
                    help="chroma through pvq_theta's no-reference path (default: with the "
                         "chroma-from-luma reference, as the reference encoder codes keyframes)")
    ap.add_argument("--chroma-cfl", action="store_true", help="(default; kept for old command lines)")
    args = ap.parse_args()

    import torch
    import daala_amd as D

    global GENERATOR
    GENERATOR = CONTENT[args.content]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    D.init(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)

    cfl = not args.chroma_noref
    pipe = Pipeline(D, args.frames, device, chroma_cfl=cfl)
    for _ in range(args.warmup):
        pipe.step()
    pipe.flush()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around the dominant kernel, on the stream it is launched on
    D.pvq_profile(True)
    D.pvq_ref_profile(True)
    barrier()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        pipe.step(record=True)
        host_s += time.perf_counter() - h0
    pipe.flush()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    search_ms = D.pvq_profile_read()
    D.pvq_profile(False)
    # The filter + DCT kernel on its own (after the timed steps): in the step it shares the
    # GPU with the other stream's kernels, which stretches its duration.
    fd_alone = pipe.pyramid_alone_ms()
    copy_gbs = pipe.copy_ceiling_gbs()
    ref_search_ms = D.pvq_ref_profile_read()
    D.pvq_ref_profile(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        bpf = blocks_per_frame()
        total_blocks = world * args.frames * args.steps * bpf
        kms = pipe.kernel_ms()
        ab = algorithmic_bytes(args.frames)
        kernels = {}
        for key, (ms, count) in kms.items():
            ent = {"avg_ms_per_launch": round(ms, 4), "launches": count,
                   "share_of_step": round(ms * count / args.steps / (dt / args.steps * 1e3), 4)}
            if key in ab:
                ent["algorithmic_bytes_per_launch"] = ab[key]
                ent["achieved_GBs"] = round(ab[key] / (ms * 1e-3) / 1e9, 1)
                ent["frac_of_hbm_peak"] = round(ent["achieved_GBs"] / HBM_PEAK_GBS, 4)
            kernels[key] = ent
        # roofline = the single kernel with the largest share of the step: the
        # search of the 128-coefficient PVQ bands.  Its duration is measured by
        # the library with HIP events on the stream the kernel is launched on
        # (odhip_pvq_profile); the rocprofv3 summary under profiles/ shows the same
        # kernel.  Algorithmic bytes per band: 2n B x16 in + 32 B record head in,
        # 2 x 2n B pulses + 32 B record tail out = 6n + 64 = 832 B.
        n128 = 0
        for (w, h, planes, top) in ((W, H, args.frames, 4), (W // 2, H // 2, 2 * args.frames, 3)):
            if cfl and top == 3:
                continue     # chroma goes through the with-reference stage
            for bs in range(top + 1):
                n128 += planes * (w // (4 << bs)) * (h // (4 << bs)) * [0, 0, 1, 3, 3][bs]
        s_ms = float(np.mean(search_ms)) if search_ms else float("nan")
        s_bytes = n128 * (6 * 128 + 64)
        roof = {"kernel": "k_search<128,2,1> (PVQ search of the 128-coefficient bands)",
                "bound": "hbm", "achieved": round(s_bytes / (s_ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_ms_per_launch": round(s_ms, 4), "launches": len(search_ms),
                "share_of_step": round(s_ms / (dt / args.steps * 1e3), 4),
                "algorithmic_bytes_per_launch": s_bytes, "bands_per_launch": n128}
        # HBM traffic of that kernel from the committed PMC run of this command
        # (tools/profile_round.sh; bench.py cannot collect counters itself)
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                tr = json.load(f)["kernels"]
            key = [k_ for k_ in tr if k_.startswith("k_search<128")]
            # the committed PMC run is of the default command (DEFAULT_FRAMES, chroma with reference)
            if key and args.frames == DEFAULT_FRAMES and cfl:
                roof["traffic"] = tr[key[0]]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r1_pmc_traffic.json (rocprofv3 --pmc, same command)"
                vi = tr[key[0]].get("valu_wave_instructions")
                ex = tr[key[0]].get("exclusive_avg_us")
                if vi and ex:
                    # secondary number SURVEY 8(d) asks for: achieved VALU issue.  1024 SIMDs,
                    # one wave-instruction per 4 cycles at the 2.4 GHz peak clock.
                    roof["valu"] = {"wave_instructions_per_launch": vi, "exclusive_ms": round(ex / 1e3, 4),
                                    "issue_frac_of_peak": round(vi * 4 / (1024 * 2.4e9 * ex * 1e-6), 3),
                                    "source": "same PMC run (SQ_INSTS_VALU) and its serialised kernel trace"}
        except (OSError, ValueError, KeyError):
            pass
        roof_noref = roof
        if cfl and ref_search_ms:
            # The with-reference stage's search of the 128-coefficient bands (one band
            # per 16-lane row), timed the same way (odhip_pvq_ref_profile).  Algorithmic
            # bytes per launch are counted from what the launch produced: per band the
            # 64 B record, the 254 B reflected vector (+ 256 B x16 when the no-reference
            # candidates run), per candidate 16 B in + 16 B out (+ 16 B result when
            # searched), and 256 B of pulses per search that stored its vector.
            r_ms = float(np.mean(ref_search_ms))
            r_bytes, r_bands = pipe.ref128_bytes()
            roof_ref = {"kernel": "k_refb_search_row<8,16> (with-reference candidate chains of the "
                                  "128-coefficient chroma bands, one band per 16-lane row)",
                        "bound": "hbm", "achieved": round(r_bytes / (r_ms * 1e-3) / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_ms_per_launch": round(r_ms, 4),
                        "launches": len(ref_search_ms),
                        "share_of_step": round(r_ms / (dt / args.steps * 1e3), 4),
                        "algorithmic_bytes_per_launch": r_bytes, "bands_per_launch": r_bands}
            try:
                with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                    tr = json.load(f)["kernels"]
                key = [k_ for k_ in tr if k_.startswith("k_refb_search_row<8, 16>")]
                if key and args.frames == DEFAULT_FRAMES:
                    roof_ref["traffic"] = tr[key[0]]["hbm_bytes_per_launch"]
                    roof_ref["traffic_source"] = ("profiles/r1_pmc_traffic.json (rocprofv3 --pmc, "
                                                  "same command)")
                    vi = tr[key[0]].get("valu_wave_instructions")
                    ex = tr[key[0]].get("exclusive_avg_us")
                    if vi and ex:
                        roof_ref["valu"] = {"wave_instructions_per_launch": vi,
                                            "exclusive_ms": round(ex / 1e3, 4),
                                            "issue_frac_of_peak": round(vi * 4 / (1024 * 2.4e9 * ex * 1e-6), 3),
                                            "source": "same PMC run (SQ_INSTS_VALU) and its serialised "
                                                      "kernel trace"}
            except (OSError, ValueError, KeyError):
                pass
            if r_ms >= s_ms:
                roof = roof_ref
        fd = kernels["forward_pyramid_luma"]
        roof["note"] = ("largest single kernel of the step; it overlaps with the other band-size "
                        "searches on forked streams, so its share is of wall time, not exclusive. "
                        "The K-pulse search is fp64 VALU-issue bound (13 VALU instructions per "
                        "candidate per pulse, DESIGN.md), HBM is quoted because SURVEY 8(d) prices "
                        "it against HBM; see roofline_filter_dct for the stage the north star "
                        "prices at >= 60 % of HBM")
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
                   "note": "copy_1GiB_GBs = the same run's device-to-device copy of 1 GiB (read + "
                           "written), the practical ceiling SURVEY 8(d) asks for beside the 8 TB/s "
                           "spec; "
                           "timed alone after the steps (HIP events, 10 launches); in_step = the same "
                           "launch inside the step, where it shares the GPU with the other stream"}
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                tr = json.load(f)["kernels"]
            key = [k_ for k_ in tr if k_.startswith("k_forward_pyramid64x2")]
            if key and args.frames == DEFAULT_FRAMES:
                roof_fd["traffic"] = tr[key[0]]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "1080p all-intra transform blocks/s (filter+DCT+PVQ)",
            "value": total_blocks / dt,
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # time the host spends enqueuing a step (it runs ahead of the GPU unless this
            # approaches ms_per_step)
            "host_enqueue_ms_per_step": host_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (lifting DCT/filters) + f64 (PVQ search)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 1920x1080 4:2:0 all-intra frames, full "
                       "4/8/16/32/64 lapped-DCT pyramid + PVQ noref bands + inverse, "
                       "every block of every level" + (
                           "; chroma through the with-reference (chroma-from-luma) PVQ path"
                           if cfl else "; chroma through the no-reference PVQ path"),
                       "frames_per_gpu_per_step": args.frames, "content": args.content,
                       "blocks_per_frame": bpf, "quality": "-v 20 (quantizer 243)",
                       "sharding": "frames over ranks, no data-path collective"},
            "roofline": roof,
            "roofline_filter_dct": roof_fd,
            "roofline_noref_search": roof_noref if roof is not roof_noref else None,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pipe.qt, cfl)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
