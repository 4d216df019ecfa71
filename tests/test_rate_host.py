"""od_pvq_rate at the default complexity (speed == 0, src/pvq_encoder.c:247-287: the real
codeword coder on a scratch range coder against a copy of the live adaptive context) as the
library's batched host routine odhip_pvq_rate_batch - a rate-only range coder and
copy-on-touch CDF rows - against the reference's own function on LIVE contexts: freshly
reset and after coding thousands of codewords, every band size, K from 0 to hundreds,
with and without theta terms.  The doubles must be identical bit for bit, and the live
context must come back untouched.  CPU only."""
import ctypes
import time

import numpy as np
import pytest

import daala_amd
from _libs import P, ref

cd = ctypes.c_double


def _codeword(rng, n, k):
    y = np.zeros(n, np.int32)
    if k:
        # pulses concentrated at low indices, like real bands
        p = 1.0 / (1.0 + np.arange(n)) ** rng.choice([0.3, 1.0, 2.0])
        pos = rng.choice(n, size=k, p=p / p.sum())
        np.add.at(y, pos, 1)
        y *= rng.choice([-1, 1], size=n)
    return y


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
def test_batched_rate_equals_od_pvq_rate_on_live_contexts():
    r = ref()
    L = daala_amd.lib()
    r.ref_adapt_new.restype = ctypes.c_void_p
    r.ref_pvq_rate0.restype = ctypes.c_double
    assert r.ref_codeword_ctx_size() == 2120       # the layout include/daala_hip.h mirrors
    rng = np.random.RandomState(5)
    total = 0
    t_ref = t_ours = 0.0
    for is_keyframe in (1, 0):
        a = ctypes.c_void_p(r.ref_adapt_new(is_keyframe))
        for phase in range(6):
            # adapt the live context the way real coding does
            for _ in range(0 if phase == 0 else 700):
                n = int(rng.choice([7, 8, 14, 15, 31, 32, 127, 128]))
                k = int(rng.choice([1, 1, 2, 3, 5, 9, 20, 60]))
                r.ref_adapt_code(a, P(_codeword(rng, n, k)), n, k)
            snap = ctypes.string_at(a, 2120)
            for n in (8, 15, 32, 128, 16, 2):
                ncand = 16
                ks = [0, 1, 1, 2, 3, 4, 7, 12, 25, 1, 2, 60, 150, 350, 5, 1][:ncand]
                thetas = [-1, -1, 0, 3, -1, 2, -1, 0, 5, 1, -1, -1, 4, -1, 7, 0]
                qgs = [0, 1, 2, 0, 3, 4, 1, 2, 2, 5, 1, 9, 3, 6, 2, 2]
                tss = [0, 0, 3, 7, 0, 4, 0, 1, 9, 2, 0, 0, 12, 0, 16, 5]
                pli = int(rng.randint(3))
                icgr = 2
                ys = []
                for c in range(ncand):
                    nn = n - (thetas[c] != -1)
                    ys.append(_codeword(rng, n, ks[c]) if nn == n else np.concatenate(
                        [_codeword(rng, nn, ks[c]), np.zeros(1, np.int32)]) if nn > 0 else np.zeros(n, np.int32))
                    if nn < 1:
                        ks[c] = 0
                t0 = time.perf_counter()
                want = [r.ref_pvq_rate0(a, qgs[c], icgr, thetas[c], tss[c], P(ys[c]), ks[c], n, is_keyframe, pli)
                        for c in range(ncand)]
                t_ref += time.perf_counter() - t0
                yptr = (ctypes.c_void_p * ncand)(*[y.ctypes.data for y in ys])
                got = np.zeros(ncand)
                arr = lambda v: (ctypes.c_int * ncand)(*v)      # noqa: E731
                t0 = time.perf_counter()
                rc = L.odhip_pvq_rate_batch(P(got), a, ncand, yptr, arr(ks), arr(qgs), arr(thetas), arr(tss), n, icgr,
                                            is_keyframe, pli)
                t_ours += time.perf_counter() - t0
                assert rc == 0
                assert np.array_equal(got.view(np.int64), np.array(want).view(np.int64)), (is_keyframe, phase, n)
                # int16 pulses: the form the band stages export
                y16 = [y.astype(np.int16) for y in ys]
                yptr16 = (ctypes.c_void_p * ncand)(*[y.ctypes.data for y in y16])
                got16 = np.zeros(ncand)
                assert L.odhip_pvq_rate_batch16(P(got16), a, ncand, yptr16, arr(ks), arr(qgs), arr(thetas), arr(tss), n,
                                                icgr, is_keyframe, pli) == 0
                assert np.array_equal(got16.view(np.int64), got.view(np.int64))
                total += ncand
            assert ctypes.string_at(a, 2120) == snap, "the live context is only read"
        r.ref_adapt_free(a)
    assert total >= 1000
    print("priced %d candidates: reference %.1f ms, batched routine %.1f ms (both through ctypes)"
          % (total, t_ref * 1e3, t_ours * 1e3))


def test_rate_batch_argument_validation():
    L = daala_amd.lib()
    got = np.zeros(1)
    assert L.odhip_pvq_rate_batch(P(got), None, 1, None, None, None, None, None, 8, 0, 1, 0) != 0
