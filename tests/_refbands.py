"""Shared checker of the with-reference band stage (odhip_pvq_ref_bands_multi /
odhip_pvq_ref_select_synth_multi) against the CPU oracle, band by band.  Used by
tests/test_gpu_pvq_refbands.py (asserts no mismatch) and tools/refbands_check.py
(prints a summary without stopping at the first one)."""
import ctypes
import math

import numpy as np

from _libs import P, oracle

cd = ctypes.c_double
MAXN = 128
THETA_SCALE = 32768 * 2. / math.pi


class Cand(ctypes.Structure):
    _fields_ = [("with_ref", ctypes.c_int32), ("gain", ctypes.c_int32),
                ("theta", ctypes.c_int32), ("ts", ctypes.c_int32), ("k", ctypes.c_int32),
                ("qcg", ctypes.c_int32), ("qtheta", ctypes.c_int32),
                ("searched", ctypes.c_int32), ("cos_dist", ctypes.c_double),
                ("dist", ctypes.c_double), ("y", ctypes.c_int32 * MAXN)]


class Trace(ctypes.Structure):
    _fields_ = [("xshift", ctypes.c_int32), ("rshift", ctypes.c_int32),
                ("g", ctypes.c_int32), ("gr", ctypes.c_int32), ("cg", ctypes.c_int32),
                ("cgr", ctypes.c_int32), ("icgr", ctypes.c_int32),
                ("gain_offset", ctypes.c_int32), ("m", ctypes.c_int32), ("s", ctypes.c_int32),
                ("theta", ctypes.c_int32), ("corr", ctypes.c_double),
                ("dist0", ctypes.c_double), ("skip_dist", ctypes.c_double),
                ("x16", ctypes.c_int16 * MAXN), ("r16", ctypes.c_int16 * MAXN),
                ("ncands", ctypes.c_int32), ("cands", Cand * 24)]


def make_planes(rng, nplanes, h, w, bs, zero_ref_frac=0.05):
    """Coefficient planes with the magnitudes of real levels (scale 2^4) and
    reference planes of per-block varying correlation and sign."""
    n = 4 << bs
    bh, bw = h // n, w // n
    up = lambda a: np.kron(a, np.ones((n, n)))  # noqa: E731
    x = np.zeros((nplanes, h, w), np.int32)
    r = np.zeros((nplanes, h, w), np.int32)
    for p in range(nplanes):
        amp = up(rng.choice([20, 120, 700, 4000], size=(bh, bw)))
        mix = up(rng.choice([0.05, 0.3, 1.0, 3.0], size=(bh, bw)))
        sign = up(rng.choice([1, 1, 1, -1], size=(bh, bw)))
        scale = up(rng.choice([0.02, 0.5, 1.0, 1.6], size=(bh, bw)))
        keep = up((rng.rand(bh, bw) >= zero_ref_frac).astype(np.float64))
        # energy decays with frequency inside a block, as in transform coefficients
        v, u = np.mgrid[0:h, 0:w]
        decay = 1. / (1. + 0.35 * ((v % n) + (u % n)))
        xf = rng.laplace(size=(h, w)) * amp * decay
        rf = (sign * scale * xf + rng.laplace(size=(h, w)) * amp * decay * mix) * keep
        x[p] = np.clip(xf, -(1 << 21), 1 << 21).astype(np.int32)
        r[p] = np.clip(rf, -(1 << 21), 1 << 21).astype(np.int32)
    return x, r


def _coding(o, plane, by, bx, n):
    blk = np.ascontiguousarray(plane[by * n:(by + 1) * n, bx * n:(bx + 1) * n])
    out = np.zeros(n * n, np.int32)
    o.odo_raster_to_coding_order(P(out), n, P(blk), n)
    return out


class Mismatch:
    def __init__(self):
        self.counts = {}
        self.first = {}
        self.checked = {}

    def check(self, what, ok, where):
        self.checked[what] = self.checked.get(what, 0) + 1
        if not ok:
            self.counts[what] = self.counts.get(what, 0) + 1
            self.first.setdefault(what, where)

    def total(self):
        return sum(self.counts.values())

    def summary(self):
        lines = []
        for k in sorted(self.checked):
            lines.append("%-14s checked %7d  bad %6d  first %s" % (k, self.checked[k],
                                                                   self.counts.get(k, 0),
                                                                   self.first.get(k)))
        return "\n".join(lines)


def oracle_traces(coef, ref, bs, qm, qmi, q_band, beta_band, is_keyframe, pli, lam):
    """Runs the oracle on every block and band.  Returns per (blk, band) a dict
    with the trace, the oracle's outputs and the flip decision."""
    o = oracle()
    o.odo_pvq_rate_speed1.restype = ctypes.c_double
    nplanes, h, w = coef.shape
    n = 4 << bs
    offs = (ctypes.c_int * 13)()
    nb = o.odo_band_offsets(bs, offs)
    offs = [offs[i] for i in range(nb + 1)]
    res = {}
    blk = 0
    for p in range(nplanes):
        for by in range(h // n):
            for bx in range(w // n):
                x = _coding(o, coef[p], by, bx, n)
                r = _coding(o, ref[p], by, bx, n)
                flip = 0
                if is_keyframe and pli != 0:
                    flip = o.odo_cfl_flip(P(r), P(x), P(qm), bs)
                for b in range(nb):
                    off, nn = offs[b], offs[b + 1] - offs[b]
                    out = np.zeros(nn, np.int32)
                    y = np.zeros(nn, np.int32)
                    i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    sd = cd(0)
                    tr = Trace()
                    x0 = np.ascontiguousarray(x[off:off + nn])
                    r0 = np.ascontiguousarray(r[off:off + nn])
                    qmb = np.ascontiguousarray(qm[off:off + nn])
                    qmib = np.ascontiguousarray(qmi[off:off + nn])
                    ret = o.odo_pvq_theta(P(out), P(x0), P(r0), nn, int(q_band[b]), P(y),
                                          ctypes.byref(i1), ctypes.byref(i2), ctypes.byref(i3),
                                          int(beta_band[b]), ctypes.byref(sd), 1, is_keyframe, pli,
                                          P(qmb), P(qmib), cd(lam), 1, ctypes.byref(tr))
                    res[(blk, b)] = dict(tr=tr, out=out, y=y, itheta=i1.value, max_theta=i2.value,
                                         vk=i3.value, ret=ret, flip=flip, r_null=int(not r0.any()),
                                         x0=x0, r0=r0, off=off, n=nn)
                blk += 1
    return res, offs


def compare_bands(hip, job, traces, mm):
    """Record, work vectors, candidate lists, searches of one job vs the oracle."""
    o = oracle()
    u = job.unpack()
    rec, items, y = u["rec"], u["items"], u["y"]
    for (blk, b), t in traces.items():
        tr, off, n = t["tr"], t["off"], t["n"]
        w = (job.bs, blk, b)
        rc = rec[blk, b]
        for f in ("xshift", "rshift", "g", "gr", "cg", "cgr", "icgr", "gain_offset"):
            mm.check("rec." + f, int(rc[f]) == getattr(tr, f), w)
        mm.check("rec.corr", float(rc["corr"]) == tr.corr, w)
        mm.check("rec.dist0", float(rc["dist0"]) == tr.dist0, w)
        flags = int(rc["flags"])
        ran = (not t["r_null"]) and tr.corr > 0
        mm.check("flag.r_null", bool(flags & hip.REFBAND_R_NULL) == bool(t["r_null"]), w)
        mm.check("flag.theta", bool(flags & hip.REFBAND_THETA) == bool(ran), w)
        mm.check("flag.flip", bool(flags & hip.REFBAND_FLIP) == bool(t["flip"]), w)
        mm.check("x16", np.array_equal(u["x16"][blk, off:off + n], np.array(tr.x16[:n], np.int16)), w)
        mm.check("r16", np.array_equal(u["r16"][blk, off:off + n], np.array(tr.r16[:n], np.int16)), w)
        mm.check("rec.m", int(rc["m"]) == tr.m, w)
        mm.check("rec.s", int(rc["s"]) == tr.s, w)
        if ran:
            theta = int(math.floor(.5 + THETA_SCALE * math.acos(tr.corr)))
            mm.check("rec.theta", int(rc["theta"]) == theta, w)
            want = np.zeros(n, np.int16)
            o.odo_apply_householder(P(want), P(np.array(tr.x16[:n], np.int16)),
                                    P(np.array(tr.r16[:n], np.int16)), n)
            mm.check("xr", np.array_equal(u["xr"][blk, off:off + n - 1], np.delete(want, tr.m)), w)
        cands = [tr.cands[i] for i in range(tr.ncands)]
        wr = [c for c in cands if c.with_ref]
        nr = [c for c in cands if not c.with_ref]
        mm.check("ntheta", int(rc["ntheta"]) == len(wr), w)
        mm.check("nitems", int(rc["nitems"]) == len(cands), w)
        mm.check("flag.noref", bool(flags & hip.REFBAND_NOREF) == bool(nr), w)
        if int(rc["ntheta"]) != len(wr) or int(rc["nitems"]) != len(cands):
            continue
        for i, c in enumerate(wr + nr):
            it = items[blk, b, i]
            wi = w + (i,)
            for f in ("gain", "theta", "ts", "k", "qcg", "qtheta"):
                mm.check("item." + f, int(it[f]) == getattr(c, f), wi)
            fl = int(it["flags"])
            mm.check("item.searched", bool(fl & hip.REFITEM_SEARCHED) == bool(c.searched), wi)
            mm.check("item.with_ref", bool(fl & hip.REFITEM_WITH_REF) == bool(c.with_ref), wi)
            if not c.searched or not (fl & hip.REFITEM_SEARCHED):
                continue
            mm.check("item.cos_dist", float(it["cos_dist"]) == c.cos_dist, wi)
            mm.check("item.dist", float(it["dist"]) == c.dist, wi)
            nn = n - 1 if c.with_ref else n
            wy = np.array(c.y[:nn], np.int32)
            slot = int(it["yslot"])
            if slot < 0:
                mm.check("item.y", not wy.any(), wi)
            else:
                mm.check("item.y", np.array_equal(y[slot, blk, off:off + nn].astype(np.int32), wy), wi)
            # the centre-of-mass sum od_pvq_rate prices with, kept above the flag bits
            mm.check("item.moment", fl >> hip.REFITEM_MOMENT_SHIFT == int((np.arange(nn) * np.abs(wy)).sum()), wi)


def host_rates(job, traces, is_keyframe, pli):
    """The host's part: od_pvq_rate (closed form, speed > 0) for every candidate
    the GPU searched -> [B][nb][REF_SLOTS + 1]."""
    o = oracle()
    o.odo_pvq_rate_speed1.restype = ctypes.c_double
    u = job.unpack()
    rec, items, y = u["rec"], u["items"], u["y"]
    B, nb = rec.shape
    rate = np.zeros((B, nb, 17), np.float64)
    for (blk, b), t in traces.items():
        off, n = t["off"], t["n"]
        icgr = int(rec[blk, b]["icgr"])
        if is_keyframe:
            rate[blk, b, 0] = o.odo_pvq_rate_speed1(0, 0, -1, 0, None, 0, n, is_keyframe, pli)
        else:
            rate[blk, b, 0] = o.odo_pvq_rate_speed1(0, icgr, 0, 0, None, 0, n, is_keyframe, pli)
        for i in range(int(rec[blk, b]["nitems"])):
            it = items[blk, b, i]
            if not (int(it["flags"]) & 1):
                continue
            with_ref = bool(int(it["flags"]) & 2)
            nn = n - 1 if with_ref else n
            slot = int(it["yslot"])
            yv = np.zeros(n, np.int32)
            if slot >= 0:
                yv[:nn] = y[slot, blk, off:off + nn]
            if with_ref:
                rate[blk, b, 1 + i] = o.odo_pvq_rate_speed1(int(it["gain"]), icgr, int(it["theta"]),
                                                            int(it["ts"]), P(yv), int(it["k"]), n,
                                                            is_keyframe, pli)
            else:
                rate[blk, b, 1 + i] = o.odo_pvq_rate_speed1(int(it["gain"]), 0, -1, 0, P(yv),
                                                            int(it["k"]), n, is_keyframe, pli)
    return rate


def compare_choice(job, traces, mm):
    """Choice records and the dequantised plane vs the oracle's pvq_theta outputs
    (valid when the job's rate table came from host_rates)."""
    o = oracle()
    ch = job.choice.cpu().numpy()
    dq = job.dq.cpu().numpy()
    coef = job.coef.cpu().numpy()
    nplanes, h, w = dq.shape
    n = 4 << job.bs
    bw, bh = w // n, h // n
    cache = {}
    for (blk, b), t in traces.items():
        if blk not in cache:
            p, rem = divmod(blk, bw * bh)
            by, bx = divmod(rem, bw)
            cache = {blk: (_coding(o, dq[p], by, bx, n), _coding(o, coef[p], by, bx, n),
                           dq[p, by * n:(by + 1) * n, bx * n:(bx + 1) * n], p, by, bx)}
        got, src, raster, p, by, bx = cache[blk]
        off, nn = t["off"], t["n"]
        wz = (job.bs, blk, b)
        mm.check("choice.ret", int(ch[blk, b, 7]) == t["ret"], wz)
        mm.check("choice.itheta", int(ch[blk, b, 3]) == t["itheta"], wz)
        mm.check("choice.maxth", int(ch[blk, b, 4]) == t["max_theta"], wz)
        mm.check("choice.k", int(ch[blk, b, 5]) == t["vk"], wz)
        mm.check("dq", np.array_equal(got[off:off + nn], t["out"]), wz)
        if b == 0:
            mm.check("dq.dc", got[0] == src[0], wz)
            if n * n > 512:
                # positions PVQ never codes (od_init_skipped_coeffs, src/state.c:1347-1366): zero on
                # a keyframe, the prediction's own coefficients on an inter frame
                want_r = (np.zeros((n, n), np.int32) if job.is_keyframe else
                          np.ascontiguousarray(job.ref.cpu().numpy()[p, by * n:(by + 1) * n, bx * n:(bx + 1) * n]))
                o.odo_coding_order_to_raster(P(want_r), n, P(np.ascontiguousarray(got)), n)
                mm.check("dq.tail", np.array_equal(want_r, raster), wz)
