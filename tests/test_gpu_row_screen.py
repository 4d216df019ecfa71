"""GPU: the single-precision screen of the row searches' greedy pulses (daala_amd/csrc/pvq_row.cuh, round 5) under
DIRECTED near ties.

pvq_search_rdo_double's greedy argmax (src/pvq_encoder.c:165-187) compares candidates a_j/b_j, a_j = (xy + |x_j|)^2,
b_j = yy + 2*y_j + 1, by cross-multiplied doubles, scanning j upwards.  The row form screens them in single precision:
a candidate within a relative 2^-17 of the best key must be an exact duplicate of it (same |x_j|, same y_j), otherwise
the row replays the pulse with the literal double-precision scan.  Real frames rarely get near that margin; these bands
are built to sit on both sides of it:

  * all |x_j| distinct CONSECUTIVE integers around X (shuffled): no pulse is placed by the projection (K < n), and at
    every greedy pulse the two best candidates have equal b and numerators (xy + X')^2, (xy + X' - 1)^2 - a relative
    difference of 2/(xy + X'), which falls through 2^-16 ... 2^-20 as xy grows by ~X per pulse;
  * X <= 600 keeps the difference above 2^-16 for every pulse (the screen must vouch for all of them: 0 replays),
    X >= 20000 and K >= 24 takes it below 2^-18 (the replay must fire);
  * exact duplicates in different lanes / quads (all |x_j| equal, or pairs of equal values): never replayed, and the
    lowest index wins as in the reference's scan.

Checked against the compiled reference (oracle/_ref) when present, else the oracle: pulses and cosine bit-exact;
odhip_pvq_search_row_batch reports how many pulses of each band were replayed."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle, ref

pytestmark = pytest.mark.gpu
LAM = 0.147


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


def _reference_search(x, k, g2):
    nb, n = x.shape
    y = np.zeros((nb, n), np.int32)
    cos = np.zeros(nb, np.float64)
    r = ref()
    if r is not None:
        r.ref_pvq_search_batch(P(x), n, P(k), P(y), P(g2), ctypes.c_double(LAM), None, P(cos), ctypes.c_long(nb))
    else:
        oracle().odo_pvq_search_batch(P(x), n, P(k), P(y), P(g2), ctypes.c_double(LAM), None, P(cos),
                                      ctypes.c_long(nb))
    return y, cos


def _greedy_pulses(x, k):
    """Greedy pulses the search places after its projection (src/pvq_encoder.c:121-146, :165)."""
    out = np.zeros(len(k), np.int64)
    for b in range(len(k)):
        kk = int(k[b])
        ax = np.abs(x[b].astype(np.int64))
        placed = 0
        if kk > 2:
            l1 = max(float(ax.sum()), 1e-100)
            placed = int(np.floor(kk * ax.astype(np.float64) * (1.0 / l1)).clip(min=0).sum())
        out[b] = max(0, kk - (1 + kk // 4) - placed)
    return out


def _ladder(rng, nb, n, xlo, xhi, step=1):
    """Bands whose magnitudes are the integers X, X - step, X - 2*step, ... in random positions with random signs."""
    x = np.zeros((nb, n), np.int16)
    for b in range(nb):
        top = int(rng.randint(xlo, xhi + 1))
        vals = top - step * np.arange(n)
        assert vals.min() > 0
        rng.shuffle(vals)
        x[b] = vals * rng.choice([-1, 1], size=n)
    return x


def _check(hip, x, k, g2, tag):
    yo, co = _reference_search(x, k, g2)
    yg, cg, rep = hip.pvq_search_row_batch(_cuda(x), _cuda(k), _cuda(g2), LAM)
    assert np.array_equal(yg.cpu().numpy(), yo), tag + ": pulses"
    assert np.array_equal(cg.cpu().numpy().view(np.int64), co.view(np.int64)), tag + ": cosine bits"
    # the literal scan for every pulse gives the same answer and counts every greedy pulse
    yf, cf, repf = hip.pvq_search_row_batch(_cuda(x), _cuda(k), _cuda(g2), LAM, force_scan=True)
    assert np.array_equal(yf.cpu().numpy(), yo), tag + ": pulses (force_scan)"
    assert np.array_equal(cf.cpu().numpy().view(np.int64), co.view(np.int64)), tag
    assert np.array_equal(repf.cpu().numpy().astype(np.int64), _greedy_pulses(x, k)), tag + ": force_scan replays every greedy pulse"
    return rep.cpu().numpy(), yo


@pytest.mark.parametrize("n", [32, 31, 128, 127])
def test_near_ties_inside_the_margin_are_replayed_and_match(hip, n):
    rng = np.random.RandomState(600 + n)
    nb = 512
    kmax = n - 2          # K < n: the projection places nothing, every pulse but the last 1 + K/4 is greedy
    x = _ladder(rng, nb, n, 20000, 32000)
    k = rng.randint(max(24, kmax // 2), kmax + 1, size=nb).astype(np.int32)
    g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
    rep, y = _check(hip, x, k, g2, "inside n=%d" % n)
    greedy = _greedy_pulses(x, k)
    assert (greedy >= 17).all()
    # after 9 pulses xy + X > 9*20000 + 20000 = 2^17.6, after 13 it is beyond 2^18: the two best candidates (equal b,
    # numerators one apart in |x|) are closer than 2^-17 and are no duplicates -> the screen must hand over
    assert (rep > 0).all(), "a band built inside the margin was not replayed"
    assert (rep <= greedy).all()
    # ... and early pulses, where 2/(xy + X) > 2^-15, must NOT have needed it
    assert (rep <= greedy - 2).all()
    print("n=%d inside: %.1f of %.1f greedy pulses per band replayed" % (n, rep.mean(), greedy.mean()))


@pytest.mark.parametrize("n", [32, 31, 128, 127])
def test_clear_margins_are_not_replayed(hip, n):
    rng = np.random.RandomState(700 + n)
    nb = 512
    kmax = n - 2
    # X <= 600 and K <= 0.7 n: the projection places nothing, xy + X <= 0.53 n * 600 < 2^15.4 - candidates without a
    # pulse differ by more than 2^-15 from each other, and one that has its pulse (b = yy + 3 against yy + 1) could tie
    # only with |x| about X larger than the best one's, which a ladder 127 wide around X >= 400 does not hold
    x = _ladder(rng, nb, n, 400, 600)
    k = rng.randint(3, int(0.7 * n) + 1, size=nb).astype(np.int32)
    assert (_greedy_pulses(x, k) == k - (1 + k // 4)).all()
    g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
    rep, _ = _check(hip, x, k, g2, "outside n=%d" % n)
    assert (rep == 0).all(), "the screen replayed %d pulses it should have vouched for" % int(rep.sum())


@pytest.mark.parametrize("n", [32, 31, 128, 127])
def test_band_crossing_the_margin(hip, n):
    """Steps of 1, 2 and 3 between neighbouring magnitudes and X in between: relative differences from 2^-14 down to
    2^-20 within one search; replays only where the margin asks for them, results equal either way."""
    rng = np.random.RandomState(800 + n)
    nb = 768
    kmax = n - 2
    x = np.concatenate([_ladder(rng, nb // 3, n, 3000, 32000, step=s) for s in (1, 2, 3)])
    k = rng.randint(3, kmax + 1, size=nb).astype(np.int32)
    g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
    rep, _ = _check(hip, x, k, g2, "crossing n=%d" % n)
    greedy = _greedy_pulses(x, k)
    assert (rep <= greedy).all()
    assert 0 < rep.sum() < greedy.sum()


@pytest.mark.parametrize("n", [32, 31, 128, 127])
def test_exact_duplicates_in_different_lanes_need_no_replay(hip, n):
    rng = np.random.RandomState(900 + n)
    nb = 384
    kmax = n - 2
    x = np.zeros((nb, n), np.int16)
    for b in range(nb):
        if b % 3 == 0:          # every magnitude equal: every candidate a duplicate of the proposal
            x[b] = int(rng.randint(1, 32001)) * rng.choice([-1, 1], size=n)
        elif b % 3 == 1:        # pairs of equal magnitudes far apart (different lanes / quads), pairs well separated
            half = (n + 1) // 2
            vals = 30000 - 400 * np.arange(half)
            v = np.concatenate([vals, vals])[:n]
            x[b] = v * rng.choice([-1, 1], size=n)
        else:                   # two equal maxima in random positions over a small floor
            x[b] = rng.randint(-40, 41, size=n)
            p = rng.choice(n, size=2, replace=False)
            x[b, p] = 25000
    k = rng.randint(3, kmax + 1, size=nb).astype(np.int32)
    g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
    rep, y = _check(hip, x, k, g2, "duplicates n=%d" % n)
    assert (rep[0::3] == 0).all(), "exact duplicates were replayed (%d pulses)" % int(rep[0::3].sum())
    # the other two kinds hold duplicates AND unequal magnitudes: a candidate that has its pulse can come within
    # 2^-17 of the best one without it by coincidence (~1e-3 per pulse); anything beyond that rate is the screen's fault
    greedy = _greedy_pulses(x, k)
    assert rep.sum() <= 0.01 * greedy.sum(), (int(rep.sum()), int(greedy.sum()))
    # all-equal bands: the reference's scan keeps the FIRST maximum, so pulses fill from index 0 upwards
    for b in range(0, nb, 3):
        ay = np.abs(y[b])
        assert (np.diff(ay) <= 0).all(), "lowest index must win among duplicates"


def test_row_batch_rejects_other_sizes(hip):
    import torch
    x = torch.zeros((4, 16), dtype=torch.int16, device="cuda")
    k = torch.ones(4, dtype=torch.int32, device="cuda")
    g2 = torch.ones(4, dtype=torch.float64, device="cuda")
    with pytest.raises(hip.DaalaHipError):
        hip.pvq_search_row_batch(x, k, g2, LAM)
