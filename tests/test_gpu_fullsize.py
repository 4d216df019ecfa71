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


@pytest.mark.parametrize("n,k", [(16, 1), (16, 4), (16, 16), (15, 1), (8, 1), (128, 24)])
def test_pvq_search_1m_bands_properties(hip, n, k):
    """configs[3]: 1M random bands.  Properties of a PVQ codeword: sum |y| = k,
    y_i*x_i >= 0 (a pulse may sit on a zero coefficient), 0 <= cos <= 1 (+rounding); and the
    first 4096 bands bit-exact against the oracle."""
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
    m = 4096
    xs = x[:m].cpu().numpy()
    ks = kk[:m].cpu().numpy()
    gs = g2[:m].cpu().numpy()
    yo = np.zeros((m, n), np.int32)
    co = np.zeros(m)
    oracle().odo_pvq_search_batch(P(xs), n, P(ks), P(yo), P(gs), ctypes.c_double(0.147), None,
                                  P(co), ctypes.c_long(m))
    assert np.array_equal(y[:m].cpu().numpy(), yo)
    assert np.array_equal(cos[:m].cpu().numpy().view(np.int64), co.view(np.int64))


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
