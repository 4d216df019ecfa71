"""GPU: the decoder-side arithmetic of PVQ bands (odhip_pvq_decode_bands =
pvq_decode_partition after its entropy-decoder reads, src/pvq_decoder.c:122-298) against the
oracle on random bands of every band size and mode, and at full size: every band of every
block of every level of a whole 1080p 4:2:0 keyframe, from the symbols the reference's
pvq_theta hands to the entropy coder, must decode to the coefficients pvq_theta synthesised
"like the decoder would" (src/pvq_encoder.c:623-633)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _libs import P, oracle, ref  # noqa: E402

pytestmark = pytest.mark.gpu
OFFS = [1, 16, 24, 32, 64, 96, 128, 256, 384, 512]


@pytest.fixture(scope="module")
def hip():
    import torch
    import daala_amd
    assert torch.cuda.is_available()
    daala_amd.init(0)
    return daala_amd


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("is_keyframe,pli", [(1, 0), (1, 1), (0, 0), (0, 2)])
def test_decode_bands_match_oracle(hip, is_keyframe, pli):
    o = oracle()
    rng = np.random.RandomState(7 + 2 * is_keyframe + pli)
    qt = hip.QuantTables.load()
    for bs, band in ((0, 0), (1, 1), (1, 2), (1, 3), (2, 6), (3, 8)):
        qm, qmi = qt.qm_slices(pli, bs)
        a, b = OFFS[band], OFFS[band + 1]
        n = b - a
        q0, beta = qt.q_band(pli, bs)[band], qt.beta_band(pli, bs)[band]
        bqm, bqmi = np.ascontiguousarray(qm[a:b]), np.ascontiguousarray(qmi[a:b])
        nb = 3000
        decay = 1.0 / (1.0 + 0.35 * np.arange(n))
        amp = rng.choice([20, 120, 700, 4000], size=(nb, 1))
        refv = np.clip(rng.laplace(size=(nb, n)) * amp * decay, -(1 << 21), 1 << 21).astype(np.int32)
        refv[::17] = 0                                    # all-zero references
        noref = (rng.rand(nb) < 0.4).astype(np.int32)
        if is_keyframe and pli == 0:
            noref[::3] = 1
        itheta = np.where(noref == 1, -1, rng.randint(0, 9, size=nb)).astype(np.int32)
        qg = rng.randint(0, 12, size=nb).astype(np.int32)
        k = rng.randint(0, 20, size=nb)
        y = np.zeros((nb, n), np.int32)
        for i in range(nb):
            nn = n - (1 - noref[i])
            pos = rng.randint(0, nn, size=k[i])
            np.add.at(y[i], pos, rng.choice([-1, 1], size=k[i]))
        sym = np.stack([qg, itheta, noref, np.zeros(nb, np.int32)], axis=1).astype(np.int32)
        out, info = hip.pvq_decode_bands(_cuda(refv), _cuda(y), _cuda(sym), _cuda(bqm), _cuda(bqmi), q0, beta,
                                         is_keyframe, pli)
        out = out.cpu().numpy()
        info = info.cpu().numpy()
        want = np.zeros(n, np.int32)
        for i in range(nb):
            kk = ctypes.c_int()
            sk = o.odo_pvq_decode_band(P(want), P(np.ascontiguousarray(refv[i])), P(np.ascontiguousarray(y[i])), n,
                                       q0, beta, int(qg[i]), int(itheta[i]), int(noref[i]), is_keyframe, pli,
                                       P(bqm), P(bqmi), ctypes.byref(kk))
            assert np.array_equal(out[i], want), (bs, band, i, qg[i], itheta[i], noref[i])
            assert info[i, 0] == kk.value and info[i, 1] == sk, (bs, band, i)


def _wrap32(a):
    return a.astype(np.int64).astype(np.uint32).astype(np.int32)


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
def test_whole_1080p_keyframe_decodes_to_the_reference_synthesis(hip):
    import bench
    import _pipeline_check as C
    qt = hip.QuantTables.load()
    pics = bench.picture_planes(bench.natural_like_frame_np(2, 77))
    dec, planes = [], []
    C.cpu_frame(qt, pics, 1920, 1080, chroma_cfl=True, decisions=dec, planes_out=planes)
    o = oracle()
    nbands = 0
    ncoded = 0
    nflip = 0
    for pli in range(3):
        xlev = None
        if pli:
            # the chroma coefficients (for the chroma-from-luma sign the encoder derives from
            # band 0 of every block, src/pvq_encoder.c:846-872: the decoder reads it as a bit)
            h, w = 544, 960
            px = np.zeros((h, w), np.uint8)
            C._pad(px, np.ascontiguousarray(pics[pli]), 0)
            xlev = [np.zeros((h, w), np.int32) for _ in range(4)]
            o.odo_forward_pyramid_plane((ctypes.c_void_p * 5)(*([a.ctypes.data for a in xlev] + [None])),
                                        P(np.zeros((h, w), np.int32)), P(px), w, w, h, 1, 1920, 1080)
        for bs, (yall, band) in enumerate(dec[pli]):
            n = 4 << bs
            dq = planes[pli]["dq"][bs]
            h, w = dq.shape
            bh, bw = h // n, w // n
            B = bh * bw
            ln = min(n * n, 512)
            # raster planes -> coding-order vectors [B][ln] (od_raster_to_coding_order)
            tmp = np.zeros(n * n, np.int32)
            idx = np.arange(n * n, dtype=np.int32).reshape(n, n)
            o.odo_raster_to_coding_order(P(tmp), n, P(np.ascontiguousarray(idx)), n)
            order = tmp[:ln]

            def coding(plane):
                blocks = plane.reshape(bh, n, bw, n).transpose(0, 2, 1, 3).reshape(B, n * n)
                return np.ascontiguousarray(blocks[:, order])

            want = coding(dq)
            qm, qmi = qt.qm_slices(pli, bs)
            qb, bb = qt.q_band(pli, bs), qt.beta_band(pli, bs)
            refc = np.zeros((B, ln), np.int32)
            if pli:
                refc = coding(planes[pli]["ref"][bs])
                xc = coding(xlev[bs])
                q0 = qm[1:16].astype(np.int64)
                rq = _wrap32(refc[:, 1:16].astype(np.int64) * q0).astype(np.int64)
                inq = _wrap32(xc[:, 1:16].astype(np.int64) * q0).astype(np.int64)
                xy = _wrap32(((rq * inq) >> 30).sum(axis=1))
                flip = xy < 0
                refc[flip] = -refc[flip]
                nflip += int(flip.sum())
            for i in range(len(qb)):
                a, b = OFFS[i], OFFS[i + 1]
                itheta = band[:, i, 1]
                noref = (itheta == -1).astype(np.int32)
                sym = np.stack([band[:, i, 0], itheta, noref, np.zeros(B, np.int32)], axis=1).astype(np.int32)
                out, info = hip.pvq_decode_bands(_cuda(refc[:, a:b]), _cuda(yall[:, a:b]), _cuda(sym), _cuda(qm[a:b]),
                                                 _cuda(qmi[a:b]), qb[i], bb[i], 1, pli)
                good = (out.cpu().numpy() == want[:, a:b]).all(axis=1)
                assert good.all(), (pli, bs, i, int((~good).sum()))
                assert np.array_equal(info.cpu().numpy()[:, 0], band[:, i, 3]), (pli, bs, i, "K")
                nbands += B
                ncoded += int((band[:, i, 3] > 0).sum())
    assert nbands == 509490 and ncoded > 100000 and nflip > 100


def test_lds_staged_and_per_lane_kernels_agree():
    """odhip_pvq_decode_bands runs the LDS-staged kernel; the one-band-per-lane form it replaced lives
    in the experiments build of the library (-DODHIP_EXPERIMENTS, lib/libdaalahip_exp.so) behind
    ODHIP_DECODE_LANE=1.  Same random bands of four band sizes through both, in child processes (the
    default library / the experiments library with the switch): the digests
    tools/decode_bands_time.py prints must agree."""
    import subprocess
    import daala_amd
    tool = os.path.join(ROOT, "tools", "decode_bands_time.py")
    outs = []
    for extra in ({}, {"ODHIP_DECODE_LANE": "1", "ODHIP_LIB": daala_amd.EXPERIMENTS_LIB}):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([sys.executable, tool, "--child"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests = [ln.split("digest")[1].strip() for ln in r.stdout.splitlines() if "digest" in ln]
        assert len(digests) == 4, r.stdout
        outs.append(digests)
    assert outs[0] == outs[1]
