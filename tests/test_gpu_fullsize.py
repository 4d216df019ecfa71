"""GPU, BASELINE.json full sizes: size-independent properties on whole 1080p
frames (configs[1]/[4]) and on 1M PVQ bands (configs[3]), plus oracle parity on
a random subset."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle

pytestmark = pytest.mark.gpu
W, H = 1920, 1088


@pytest.fixture(scope="module")
def hip():
    import torch
    import daala_amd
    assert torch.cuda.is_available()
    daala_amd.init(0)
    return daala_amd


def test_1080p_pyramid_inverse_round_trip_is_lossless(hip):
    """fDCT pyramid -> inverse at EVERY partition level reproduces the pixels
    exactly (the lapped transform is reversible), luma and chroma, 2 frames."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    for dec, shape in ((0, (2, H, W)), (1, (4, H // 2, W // 2))):
        px = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
        levels = hip.forward_pyramid(px, dec, 1920, 1080)
        for bs, lv in enumerate(levels):
            rec = hip.inverse_level(lv, dec, bs, 1920, 1080)
            assert torch.equal(rec, px), (dec, bs)
        # energy sanity: orthonormal-ish transform, DC of a flat plane
        flat = torch.full(shape, 200, dtype=torch.uint8, device="cuda")
        lf = hip.forward_pyramid(flat, dec, 1920, 1080)
        for bs, lv in enumerate(lf):
            n = 4 << bs
            blk = lv[0, :n, :n]
            # integer lifting: a flat block is DC plus a few LSBs of rounding residue
            assert int(blk[0, 0].abs()) > 64*n and int(blk.abs().sum() - blk[0, 0].abs()) < 4*n


def test_1080p_pyramid_random_blocks_match_oracle(hip):
    """Spot-check full-size output against the oracle: the top-left and the
    bottom-right superblock regions (picture-edge gating) of a 1080p luma plane."""
    import torch
    rng = np.random.RandomState(9)
    px = rng.randint(0, 256, size=(H, W)).astype(np.uint8)
    got = hip.forward_pyramid(torch.from_numpy(px[None]).cuda(), 0, 1920, 1080)
    want = [np.zeros((H, W), np.int32) for _ in range(5)]
    arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in want])
    c = np.zeros((H, W), np.int32)
    oracle().odo_forward_pyramid_plane(arr, P(c), P(px), W, W, H, 0, 1920, 1080)
    for bs in range(5):
        assert np.array_equal(got[bs][0].cpu().numpy(), want[bs]), bs


def _oracle_search(xs, ks, gs, lam=0.147):
    """pvq_search_rdo_double (src/pvq_encoder.c:93-224) band by band in the CPU oracle."""
    m, n = xs.shape
    yo = np.zeros((m, n), np.int32)
    co = np.zeros(m)
    oracle().odo_pvq_search_batch(P(np.ascontiguousarray(xs)), n, P(np.ascontiguousarray(ks)), P(yo),
                                  P(np.ascontiguousarray(gs)), ctypes.c_double(lam), None, P(co),
                                  ctypes.c_long(m))
    return yo, co


@pytest.mark.parametrize("n,k", [(16, 1), (16, 2), (16, 4), (16, 8), (16, 16), (15, 1), (8, 1), (128, 24)])
def test_pvq_search_1m_bands_all_match_oracle(hip, n, k):
    """configs[3]: 1M random bands (n = 16, x ~ U[-1000, 1000], k in
    {1, 2, 4, 8, 16}, g2 = 1, lambda = 0.147, prev_k = 0; plus the two lengths
    pvq_search_rdo_double special-cases at k = 1 and the longest band).  EVERY band: the
    pulse vector exactly and the returned cosine as int64 bits, against the oracle; and the
    properties of a PVQ codeword (sum |y| = k, no pulse opposes its coefficient)."""
    import torch
    nb = 1 << 20 if n <= 16 else 1 << 17
    g = torch.Generator(device="cuda").manual_seed(n * 131 + k)
    x = torch.randint(-1000, 1001, (nb, n), dtype=torch.int16, device="cuda", generator=g)
    kk = torch.full((nb,), k, dtype=torch.int32, device="cuda")
    g2 = torch.ones(nb, dtype=torch.float64, device="cuda")
    y, cos = hip.pvq_search_batch(x, kk, g2, 0.147)
    nz = x.abs().sum(1) > 0
    assert torch.equal(y.abs().sum(1)[nz], kk[nz].long() if y.dtype == torch.int64 else kk[nz])
    assert bool((y.long()*x.long() >= 0).all()), "a pulse never opposes the sign of its coefficient"
    assert float(cos.min()) >= 0.0 and float(cos.max()) <= 1.0 + 1e-12
    yo, co = _oracle_search(x.cpu().numpy(), kk.cpu().numpy(), g2.cpu().numpy())
    assert np.array_equal(y.cpu().numpy(), yo)
    assert np.array_equal(cos.cpu().numpy().view(np.int64), co.view(np.int64))


def test_pvq_search_1m_laplacian_bands_empirical_k_all_match_oracle(hip):
    """SURVEY 8(d) C4's second set: 1M bands, n = 16, Laplacian x (scale 200), K drawn from
    the pulse counts the encoder really asks for (tests/golden/k_hist.npz: the histogram of K
    over every search of the configs[1] frame, dumped by tools/k_hist.py; geometric fallback
    with a similar mean when the fixture is absent), g2 from the band's own energy.  Every
    band: pulses exact, cosine bit for bit."""
    import os
    import torch
    nb, n = 1 << 20, 16
    rng = np.random.RandomState(77)
    x = np.clip(np.rint(rng.laplace(scale=200.0, size=(nb, n))), -32000, 32000).astype(np.int16)
    x[::4097] = 0                                   # all-zero bands take the early return
    fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "k_hist.npz")
    if os.path.exists(fix):
        h = np.load(fix)["hist"].astype(np.float64)
        ks = rng.choice(len(h), size=nb, p=h / h.sum()).astype(np.int32)
    else:
        ks = np.minimum(rng.geometric(1.0 / 6.0, size=nb), 350).astype(np.int32)
    ks[:64] = np.arange(64, dtype=np.int32)         # k = 0 .. 63 each at least once
    g2 = (x.astype(np.float64) ** 2).sum(1) / 4096.0 + 1e-3
    y, cos = hip.pvq_search_batch(torch.from_numpy(x).cuda(), torch.from_numpy(ks).cuda(),
                                  torch.from_numpy(g2).cuda(), 0.147)
    yo, co = _oracle_search(x, ks, g2)
    assert np.array_equal(y.cpu().numpy(), yo)
    got = cos.cpu().numpy().view(np.int64)
    assert np.array_equal(got, co.view(np.int64))


def test_ragged_and_minimum_sizes(hip):
    """One superblock (64x64 luma / 32x32 chroma), a 1-row-of-superblocks plane,
    and a batch that does not fill the last workgroup."""
    import torch
    rng = np.random.RandomState(3)
    for dec, (h, w) in ((0, (64, 64)), (0, (64, 320)), (1, (32, 32)), (1, (96, 32))):
        px = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        got = hip.forward_pyramid(torch.from_numpy(px[None]).cuda(), dec, w << dec, h << dec)
        want = [np.zeros((h, w), np.int32) for _ in range(5 - dec)]
        arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in want])
        c = np.zeros((h, w), np.int32)
        oracle().odo_forward_pyramid_plane(arr, P(c), P(px), w, w, h, dec, w << dec, h << dec)
        for bs in range(5 - dec):
            assert np.array_equal(got[bs][0].cpu().numpy(), want[bs]), (dec, h, w, bs)
    x = torch.from_numpy(rng.randint(-500, 500, size=(65, 32)).astype(np.int16)).cuda()
    k = torch.full((65,), 7, dtype=torch.int32, device="cuda")
    g2 = torch.ones(65, dtype=torch.float64, device="cuda")
    y, _ = hip.pvq_search_batch(x, k, g2, 0.147)
    assert bool((y.abs().sum(1) == 7).all())


def test_1080p_band_stage_properties_and_oracle_subset(hip):
    """The PVQ band stage on whole 1080p frames (all nine (plane set, level) jobs of
    the bench step in one multi-job call): size-independent properties of every
    band, run-to-run identity (the counting sort uses atomics, the results must not
    depend on the order), and oracle parity on a random subset of bands."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from test_gpu_pvq_bands import Trace
    y_, cb, cr = bench.synth_frame_np(0, 99)
    qt = hip.QuantTables.load()
    lam = hip.OD_PVQ_LAMBDA
    sets = []
    jobs = []
    for px, dec, pli in ((y_[None], 0, 0), (np.stack([cb, cr]), 1, 1)):
        tpx = torch.from_numpy(np.ascontiguousarray(px)).cuda()
        levels = hip.forward_pyramid(tpx, dec, 1920, 1080)
        for bs in range(5 - dec):
            qm, qmi = qt.qm_slices(pli, bs)
            job = hip.PvqJob(levels[bs], bs, torch.from_numpy(qm).cuda(), torch.from_numpy(qmi).cuda(),
                             qt.q_band(pli, bs), qt.beta_band(pli, bs))
            jobs.append(job)
            sets.append((pli, dec, bs, qm, qmi))
    hip.pvq_noref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    first = [(j.cands["band"].clone(), j.cands["y"].clone()) for j in jobs]
    hip.pvq_noref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    o = oracle()
    rng = np.random.RandomState(1)
    cd = ctypes.c_double
    nsearched = 0
    for job, (band0, y0), (pli, dec, bs, qm, qmi) in zip(jobs, first, sets):
        assert torch.equal(job.cands["band"], band0) and torch.equal(job.cands["y"], y0), (pli, bs)
        c = hip.unpack_cands(job.cands)
        nb, offs, ln = hip.pvq_band_layout(bs)
        y = c["y"].astype(np.int64)
        for band in range(nb):
            a, b = offs[band], offs[band + 1]
            for slot in range(2):
                fl = c["flags"][:, band, slot]
                ys = y[slot][:, a:b]
                k = c["k"][:, band, slot].astype(np.int64)
                on = fl == 1
                assert (np.abs(ys).sum(1)[on] == k[on]).all(), (pli, bs, band, slot)
                assert ((ys * ys).sum(1)[on] == c["yy"][:, band, slot][on]).all()
                assert (ys[~on] == 0).all() and (c["yy"][:, band, slot][~on] == 0).all()
                assert (c["dist"][:, band, slot][on] >= 0).all()
                nsearched += int(on.sum())
            assert (c["dist0"][:, band] >= 0).all() and (c["cg"][:, band] >= 0).all()
        # oracle parity on 40 random (block, band) pairs of this job
        n = 4 << bs
        coef = job.coef.cpu().numpy()
        nplanes, h, w = coef.shape
        qb, bb = qt.q_band(pli, bs), qt.beta_band(pli, bs)
        for _ in range(40):
            blk = int(rng.randint(job.nblocks))
            band = int(rng.randint(nb))
            p, rem = divmod(blk, (h // n) * (w // n))
            by, bx = divmod(rem, w // n)
            vec = np.zeros(n * n, np.int32)
            tile = np.ascontiguousarray(coef[p, by * n:(by + 1) * n, bx * n:(bx + 1) * n])
            o.odo_raster_to_coding_order(P(vec), n, P(tile), n)
            a, b = offs[band], offs[band + 1]
            m = b - a
            x0 = np.ascontiguousarray(vec[a:b])
            r0 = np.zeros(m, np.int32)
            out = np.zeros(m, np.int32)
            yv = np.zeros(m, np.int32)
            i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            sd = cd(0)
            tr = Trace()
            o.odo_pvq_theta(P(out), P(x0), P(r0), m, qb[band], P(yv), ctypes.byref(i1), ctypes.byref(i2),
                            ctypes.byref(i3), bb[band], ctypes.byref(sd), 1, 1, 0,
                            P(np.ascontiguousarray(qm[a:b])), P(np.ascontiguousarray(qmi[a:b])), cd(lam), 1,
                            ctypes.byref(tr))
            assert c["cg"][blk, band] == tr.cg and c["dist0"][blk, band] == tr.dist0
            nr = [tr.cands[i] for i in range(tr.ncands) if not tr.cands[i].with_ref]
            for slot, cnd in enumerate(nr):
                assert c["gain"][blk, band, slot] == cnd.gain and c["k"][blk, band, slot] == cnd.k
                assert c["flags"][blk, band, slot] == cnd.searched
                if cnd.searched:
                    assert c["dist"][blk, band, slot] == cnd.dist
                    assert np.array_equal(c["y"][slot, blk, a:b], np.array(cnd.y[:m], np.int32))
    assert nsearched > 500000


def _natural_like_frame(seed):
    """A 1920x1088 4:2:0 frame of the second kind SURVEY.md 8(d) asks for: a sum of 2-D
    cosines plus separable AR(1) noise (rho = 0.95, the model of the reference's
    dcttest, src/dct.c:4968), chroma = smoothed, subsampled luma with its own gain and
    a little independent noise (so that luma-derived predictions really correlate)."""
    rng = np.random.RandomState(seed)
    h, w = 1088, 1920
    yy, xx = np.mgrid[0:h, 0:w]
    img = 40 * np.cos(xx * 0.013 + yy * 0.007) + 25 * np.cos(xx * 0.041 - yy * 0.029) \
        + 12 * np.cos(xx * 0.11 + 1.0) * np.cos(yy * 0.09)
    e = rng.normal(size=(h, w)) * 6
    for ax in (0, 1):               # separable AR(1), rho = 0.95
        e = np.moveaxis(e, ax, 0)
        for i in range(1, e.shape[0]):
            e[i] += 0.95 * e[i - 1]
        e = np.moveaxis(e, 0, ax) * 0.31
    luma = np.clip(128 + img + e * 4, 0, 255)
    sub = luma.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    cb = np.clip(128 + 0.5 * (sub - 128) + rng.normal(size=sub.shape) * 2, 0, 255)
    cr = np.clip(128 - 0.35 * (sub - 128) + rng.normal(size=sub.shape) * 2, 0, 255)
    return luma.astype(np.uint8), cb.astype(np.uint8), cr.astype(np.uint8)


@pytest.mark.parametrize("content", ["bench", "natural"])
def test_1080p_with_reference_stage_properties_and_oracle_subset(hip, content):
    """The with-reference stage on the chroma planes of a whole 1080p frame, with the
    chroma-from-luma references the bench builds (upper-left quarter of the luma
    reconstruction's dequantised coefficients, one level up): properties of every
    candidate of every band, run-to-run identity (uncertainty list and counting
    sort use atomics; results must not depend on the order), the chosen synthesis
    against an independent per-band recomputation, and oracle parity of complete
    bands (record, candidates, pulses) on a random subset."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from _refbands import Mismatch, compare_bands, oracle_traces
    y_, cb, cr = bench.synth_frame_np(0, 77) if content == "bench" else _natural_like_frame(5)
    qt = hip.QuantTables.load()
    lam = hip.OD_PVQ_LAMBDA
    W, H = 1920, 1088
    # luma reconstruction coefficients
    luma = hip.forward_pyramid(torch.from_numpy(y_[None]).cuda(), 0, 1920, 1080)
    ljobs = []
    for bs in range(5):
        qm, qmi = qt.qm_slices(0, bs)
        ljobs.append(hip.PvqJob(luma[bs], bs, torch.from_numpy(qm).cuda(), torch.from_numpy(qmi).cuda(),
                                qt.q_band(0, bs), qt.beta_band(0, bs), dq=torch.zeros_like(luma[bs])))
    hip.pvq_noref_bands_multi(ljobs, lam)
    hip.pvq_select_synth_noref_multi(ljobs, lam)
    chroma = hip.forward_pyramid(torch.from_numpy(np.stack([cb, cr])).cuda(), 1, 1920, 1080)
    jobs = []
    for bs in range(4):
        n = 4 << bs
        dq = ljobs[bs + 1].dq
        corner = dq.view(1, H // (2 * n), 2 * n, W // (2 * n), 2 * n)[:, :, :n, :, :n]
        ref = corner.reshape(1, H // 2, W // 2).contiguous()
        ref = torch.cat([ref, ref], dim=0).contiguous()
        qm, qmi = qt.qm_slices(1, bs)
        jobs.append(hip.PvqRefJob(chroma[bs], ref, bs, torch.from_numpy(qm).cuda(),
                                  torch.from_numpy(qmi).cuda(), qt.q_band(1, bs), qt.beta_band(1, bs),
                                  1, 1))
    hip.pvq_ref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    first = [(j.band.clone(), j.items.clone(), j.y.clone()) for j in jobs]
    hip.pvq_ref_bands_multi(jobs, lam)
    hip.pvq_ref_select_synth_multi(jobs, lam)
    torch.cuda.synchronize()
    rng = np.random.RandomState(3)
    nsearched = 0
    nflip = 0
    for bs, (job, (band0, items0, y0)) in enumerate(zip(jobs, first)):
        assert torch.equal(job.band, band0), bs
        u = job.unpack()
        rec, items, y = u["rec"], u["items"], u["y"].astype(np.int64)
        B, nb = rec.shape
        nitems = rec["nitems"]
        ntheta = rec["ntheta"]
        assert (nitems <= 14).all() and (ntheta <= 12).all() and (ntheta <= nitems).all()
        ran = (rec["flags"] & hip.REFBAND_THETA) != 0
        assert ((ntheta > 0) <= ran).all()
        assert ((rec["corr"] > 0) | ~ran).all()
        assert (np.abs(rec["corr"]) <= 1).all() and (rec["dist0"] >= 0).all()
        nflip += int(((rec["flags"][:, 0] & hip.REFBAND_FLIP) != 0).sum())
        # the flip is a property of the block: identical over its bands
        fl = (rec["flags"] & hip.REFBAND_FLIP) != 0
        assert (fl == fl[:, :1]).all()
        offs = job.offsets
        for b in range(nb):
            a, e = offs[b], offs[b + 1]
            for s in range(14):
                it = items[:, b, s]
                valid = s < nitems[:, b]
                searched = valid & ((it["flags"] & 1) != 0)
                with_ref = (it["flags"] & 2) != 0
                assert (with_ref[valid] == (s < ntheta[:, b])[valid]).all()
                own = searched & (it["yslot"] == s)
                nn = np.where(with_ref, e - a - 1, e - a)
                ys = y[s][:, a:e]
                # a search that stored its vector placed exactly k pulses
                for width in (e - a - 1, e - a):
                    sel = own & (nn == width)
                    assert (np.abs(ys[sel][:, :width]).sum(1) == it["k"][sel]).all(), (bs, b, s)
                assert (it["dist"][searched] >= 0).all()
                assert (np.abs(it["cos_dist"][searched]) <= 1 + 1e-12).all()
                shared = searched & (it["yslot"] >= 0) & (it["yslot"] != s)
                assert (it["yslot"][shared] < s).all()
                nsearched += int(searched.sum())
        # run-to-run identity of everything a consumer reads (slots beyond nitems and the
        # result fields of unsearched candidates are not written)
        job.items, keep = items0, job.items
        items_first = job.unpack()["items"]
        job.items = keep
        for s in range(14):
            valid = s < nitems
            for f in ("gain", "theta", "ts", "k"):
                assert (items[f][:, :, s][valid] == items_first[f][:, :, s][valid]).all()
            done = valid & ((items["flags"][:, :, s] & 1) != 0)
            for f in ("qcg", "qtheta", "flags", "yslot", "cos_dist", "dist"):
                assert (items[f][:, :, s][done] == items_first[f][:, :, s][done]).all(), (bs, s, f)
        y_first = y0.cpu().numpy()
        for b in range(nb):
            a, e = offs[b], offs[b + 1]
            for s in range(14):
                own = (s < nitems[:, b]) & ((items["flags"][:, b, s] & 1) != 0) & (items["yslot"][:, b, s] == s)
                assert np.array_equal(u["y"][s][own][:, a:e - 1], y_first[s][own][:, a:e - 1]), (bs, b, s)
    assert nsearched > 300000 and nflip > 1000
    # oracle parity of whole bands on random blocks of every level
    mm = Mismatch()
    for bs, job in enumerate(jobs):
        n = 4 << bs
        coef = job.coef.cpu().numpy()
        ref = job.ref.cpu().numpy()
        qm, qmi = qt.qm_slices(1, bs)
        nplanes, h, w = coef.shape
        bw = w // n
        for _ in range(12):
            p = int(rng.randint(nplanes))
            by = int(rng.randint(h // n))
            bx = int(rng.randint(bw))
            sub_c = np.ascontiguousarray(coef[p:p + 1, by * n:(by + 1) * n, bx * n:(bx + 1) * n])
            sub_r = np.ascontiguousarray(ref[p:p + 1, by * n:(by + 1) * n, bx * n:(bx + 1) * n])
            traces, _ = oracle_traces(sub_c, sub_r, bs, qm, qmi, qt.q_band(1, bs), qt.beta_band(1, bs),
                                      1, 1, lam)
            blk = (p * (h // n) + by) * bw + bx
            compare_bands(hip, _OneBlock(job, blk), traces, mm)
    assert mm.total() == 0, mm.summary()
    assert mm.checked["item.y"] > 500
    # dequantised planes: DC passed through, everything finite and bounded
    for job in jobs:
        dq = job.dq.cpu().numpy()
        coef = job.coef.cpu().numpy()
        n = 4 << job.bs
        assert np.array_equal(dq[:, ::n, ::n], coef[:, ::n, ::n])
        assert np.abs(dq.astype(np.int64)).max() < 1 << 24


class _OneBlock:
    """View of a PvqRefJob restricted to one block, for _refbands.compare_bands."""

    def __init__(self, job, blk):
        self.job, self.blk, self.bs = job, blk, job.bs

    def unpack(self):
        if not hasattr(self.job, "_unpacked"):
            self.job._unpacked = self.job.unpack()
        u = self.job._unpacked
        b = self.blk
        return {"rec": u["rec"][b:b + 1], "items": u["items"][b:b + 1], "y": u["y"][:, b:b + 1],
                "choice": u["choice"][b:b + 1], "r16": u["r16"][b:b + 1], "x16": u["x16"][b:b + 1],
                "xr": u["xr"][b:b + 1]}
