#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref).

Dev-container only (needs /root/reference compiled by `make -C oracle ref`).
The fixtures pin the CPU oracle - and through it the HIP kernels - to the
reference's own outputs on seeded inputs, and carry the host-initialised
quantiser tables the kernels consume as data.  Deterministic: re-running it
reproduces the committed files bit for bit.
"""
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import GOLDEN, P, ref, synth_frame  # noqa: E402

cd = ctypes.c_double


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    r = ref()
    assert r is not None, "build oracle/_ref first: make -C oracle ref"
    os.makedirs(GOLDEN, exist_ok=True)

    # ---- quantiser tables at encoder_example -v 20 (host init, row a17) ----
    qn = r.ref_qm_buffer_size()
    q = ctypes.c_int()
    pq = np.zeros(3 * 30, np.uint8)
    qm = np.zeros(qn, np.int16)
    qmi = np.zeros(qn, np.int16)
    nq = r.ref_dump_quant_tables(20, ctypes.byref(q), P(pq), P(qm), P(qmi))
    assert nq == 30
    qm_flat = np.zeros(qn, np.int16)
    qmi_flat = np.zeros(qn, np.int16)
    r.ref_init_qm(P(qm_flat), P(qmi_flat), 1)
    qm_off = np.array([[r.ref_qm_offset(bs, d) for d in range(2)] for bs in range(5)], np.int32)
    qm_index = np.array([[r.ref_qm_get_index(bs, b) for b in range(13)] for bs in range(5)], np.int32)
    nbands = [1, 4, 7, 9, 9]
    beta = np.zeros((2, 3, 5, 9), np.int32)
    for m in range(2):
        for pli in range(3):
            for bs in range(5):
                for b in range(nbands[bs]):
                    beta[m, pli, bs, b] = r.ref_pvq_beta(m, pli, bs, b)
    np.savez_compressed(os.path.join(GOLDEN, "quant_v20.npz"), quantizer=np.int32(q.value),
                        pvq_qm_q4=pq.reshape(3, 30), qm=qm, qm_inv=qmi, qm_flat=qm_flat,
                        qm_inv_flat=qmi_flat, qm_offset=qm_off, qm_index=qm_index, beta=beta)

    # ---- transforms: IEEE-1180 style seeded blocks (reference dct.c:8277) ----
    rng = np.random.RandomState(1180)
    d = {}
    for ln in range(5):
        n = 4 << ln
        x = np.concatenate([(rng.randint(0, 511, size=(6, n, n)) - 255) * 16,
                            (rng.randint(0, 11, size=(3, n, n)) - 5) * 16,
                            (rng.randint(0, 601, size=(3, n, n)) - 300) * 16]).astype(np.int32)
        y = np.zeros_like(x)
        r.ref_fdct_2d_batch(ln, P(y), P(x), ctypes.c_long(len(x)))
        back = np.zeros_like(x)
        r.ref_idct_2d_batch(ln, P(back), P(y), ctypes.c_long(len(x)))
        assert np.array_equal(back, x)
        d["x%d" % n] = x
        d["y%d" % n] = y
        v = ((rng.randint(0, 511, size=(8, n)) - 255) * 16).astype(np.int32)
        o = np.zeros_like(v)
        for i in range(len(v)):
            r.ref_fdct_1d(ln, P(o[i]), P(v[i]), 1)
        d["v%d" % n] = v
        d["o%d" % n] = o
    np.savez_compressed(os.path.join(GOLDEN, "dct.npz"), **d)

    # ---- 4-point filter: random vectors + the reference's own +-676 sweep ----
    x = rng.randint(-4096, 4096, size=(256, 4)).astype(np.int32)
    pre = np.zeros_like(x)
    post = np.zeros_like(x)
    for i in range(len(x)):
        r.ref_pre_filter(0, P(pre[i]), P(x[i]))
        r.ref_post_filter(0, P(post[i]), P(x[i]))
    sweep = np.array([[676 if (i >> j) & 1 else -676 for j in range(4)] for i in range(16)], np.int32)
    ys = np.zeros_like(sweep)
    for i in range(16):
        r.ref_pre_filter(0, P(ys[i]), P(sweep[i]))
    # the known answers printed by `gcc -DTEST src/filter.c`
    assert ys.min(0).tolist() == [-1003, -1198, -1198, -1003]
    assert ys.max(0).tolist() == [1003, 1198, 1198, 1003]
    np.savez_compressed(os.path.join(GOLDEN, "filter4.npz"), x=x, pre=pre, post=post,
                        sweep=sweep, sweep_pre=ys)

    # ---- lapped pyramid + inverse: hashes of every level ---------------------
    W, H = 192, 128
    planes = synth_frame(W, H, seed=2024)
    out = {}
    for dec, idx in ((0, 0), (1, 1)):
        px = planes[idx]
        h, w = px.shape
        top = 4 - dec
        for pic in ((W, H), (W - 8, H - 24)):
            lv = [np.zeros((h, w), np.int32) for _ in range(top + 1)]
            arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in lv])
            c = np.zeros((h, w), np.int32)
            r.ref_forward_pyramid_plane(arr, P(c), P(px), w, w, h, dec, pic[0], pic[1])
            tag = "d%d_%dx%d" % (dec, pic[0], pic[1])
            for bs in range(top + 1):
                out["%s_L%d" % (tag, bs)] = np.frombuffer(bytes.fromhex(sha(lv[bs])), np.uint8)
            out["%s_c" % tag] = np.frombuffer(bytes.fromhex(sha(c)), np.uint8)
            if pic == (W, H):
                out["%s_L1_full" % tag] = lv[1]
    np.savez_compressed(os.path.join(GOLDEN, "pyramid.npz"), **out)

    # ---- PVQ search ------------------------------------------------------------
    d = {}
    for n in (8, 15, 16, 32, 128):
        nb = 200
        x = np.where(rng.rand(nb, 1) < .5, rng.randint(-1000, 1001, size=(nb, n)),
                     rng.laplace(0, 200, size=(nb, n))).astype(np.int16)
        k = rng.choice([1, 2, 3, 4, 8, 16, 33], size=nb).astype(np.int32)
        g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
        y = np.zeros((nb, n), np.int32)
        cos = np.zeros(nb, np.float64)
        r.ref_pvq_search_batch(P(x), n, P(k), P(y), P(g2), cd(0.147), None, P(cos),
                               ctypes.c_long(nb))
        k2 = (k + rng.randint(0, 6, size=nb)).astype(np.int32)
        y2 = y.copy()
        cos2 = np.zeros(nb, np.float64)
        r.ref_pvq_search_batch(P(x), n, P(k2), P(y2), P(g2), cd(0.147), P(k), P(cos2),
                               ctypes.c_long(nb))
        for name, v in (("x", x), ("k", k), ("g2", g2), ("y", y), ("cos", cos), ("k2", k2),
                        ("y2", y2), ("cos2", cos2)):
            d["%s_n%d" % (name, n)] = v
    np.savez_compressed(os.path.join(GOLDEN, "pvq_search.npz"), **d)

    # ---- pvq_theta (speed = 1: closed-form rate, no entropy-coder state) -------
    offs = {}
    for bs in range(5):
        o = (ctypes.c_int * 16)()
        nb = r.ref_band_offsets(bs, o)
        offs[bs] = [o[i] for i in range(nb + 1)]
    rows = []
    blobs = []
    for it in range(400):
        bs = int(rng.randint(0, 5))
        pli = int(rng.randint(0, 2))
        kf = int(rng.randint(0, 2))
        band = int(rng.randint(0, len(offs[bs]) - 1))
        a, b = offs[bs][band], offs[bs][band + 1]
        n = b - a
        qoff = int(qm_off[bs, 1 if pli else 0])
        scale = float(rng.choice([30, 200, 1500, 8000]))
        x0 = rng.laplace(0, scale, size=n).astype(np.int32)
        mode = it % 4
        if mode == 0:
            r0 = np.zeros(n, np.int32)
        elif mode == 1:
            r0 = (x0 + rng.laplace(0, scale / 3, size=n)).astype(np.int32)
        elif mode == 2:
            r0 = rng.laplace(0, scale, size=n).astype(np.int32)
        else:
            r0 = (x0 * 0.9 + rng.laplace(0, scale / 10, size=n)).astype(np.int32)
        q0 = int(rng.choice([5, 31, 80, 300]))
        bt = int(beta[1, pli, bs, band])
        outv = np.zeros(n, np.int32)
        y = np.zeros(n, np.int32)
        it_ = ctypes.c_int()
        mt = ctypes.c_int()
        vk = ctypes.c_int()
        sd = cd(0.25)
        qq = np.ascontiguousarray(qm[qoff + a:qoff + b])
        qi = np.ascontiguousarray(qmi[qoff + a:qoff + b])
        g = r.ref_pvq_theta(P(outv), P(x0), P(r0), n, q0, P(y), ctypes.byref(it_),
                            ctypes.byref(mt), ctypes.byref(vk), bt, ctypes.byref(sd), 1, kf, pli,
                            P(qq), P(qi), cd(0.147), 1)
        rows.append([bs, pli, kf, band, n, q0, bt, g, it_.value, mt.value, vk.value])
        blobs.append((x0, r0, outv, y, np.float64(sd.value)))
    np.savez_compressed(
        os.path.join(GOLDEN, "pvq_theta.npz"), meta=np.array(rows, np.int32),
        x0=np.concatenate([b[0] for b in blobs]), r0=np.concatenate([b[1] for b in blobs]),
        out=np.concatenate([b[2] for b in blobs]), y=np.concatenate([b[3] for b in blobs]),
        skip_diff=np.array([b[4] for b in blobs]))
    total = sum(os.path.getsize(os.path.join(GOLDEN, f)) for f in os.listdir(GOLDEN))
    print("golden fixtures written: %d bytes" % total)


if __name__ == "__main__":
    main()
