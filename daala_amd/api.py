"""ctypes binding of libdaalahip.so + torch-tensor convenience wrappers.

Names and argument meaning mirror the C ABI (include/daala_hip.h), which in
turn mirrors the reference's surfaces for this path (od_dct_func_2d,
od_*filter*_split / _frame_sbs, pvq_search_rdo_double).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DaalaHipError(RuntimeError):
    pass


# The experiments build (-DODHIP_EXPERIMENTS: superseded kernel generations and ablations behind
# their ODHIP_* switches; daala_amd/build.py): ODHIP_LIB=EXPERIMENTS_LIB selects it for a process.
EXPERIMENTS_LIB = os.path.join(_HERE, "lib", "libdaalahip_exp.so")


def lib_path():
    # ODHIP_LIB: another build of the same library (kernel A/B experiments, tools/)
    return os.environ.get("ODHIP_LIB") or os.path.join(_HERE, "lib", "libdaalahip.so")


def lib():
    """The loaded C-ABI library.  Fails loudly when it is missing."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise DaalaHipError(
                "%s not built: run `python -m daala_amd.build` "
                "(there is no CPU fallback)" % path)
        # PyTorch-ROCm ships its own copy of the HIP runtime; when it is going to be used in
        # this process (device tensors, streams) it has to be the one the process binds
        # first - the library loaded first would otherwise bring /opt/rocm's copy and
        # torch.cuda then finds no device.  A pure-C consumer never gets here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(path)
        L.odhip_version.restype = ctypes.c_char_p
        L.od_pvq_search_rdo_double_hip.restype = ctypes.c_double
        _LIB = L
    return _LIB


ERANGE = -24      # ODHIP_ERANGE: a band needs more pulses than ODHIP_PVQ_MAX_K (include/daala_hip.h)


class PulseRangeError(DaalaHipError):
    """A band's reference candidate has more than 32767 pulses (quantiser too fine for the int16 pulse vectors): the
    results are NOT the reference's."""


def _check(rc, what):
    if rc == ERANGE:
        raise PulseRangeError("%s: a band needs more pulses than ODHIP_PVQ_MAX_K = 32767; the results since the last "
                              "sync are not the reference's" % what)
    if rc != 0:
        raise DaalaHipError("%s failed with code %d" % (what, rc))


def pvq_k_range_take():
    """(no-reference bands, with-reference bands) whose candidate exceeded ODHIP_PVQ_MAX_K since the last call; cleared.
    Synchronise the stream of the band stage first."""
    a = ctypes.c_uint()
    b = ctypes.c_uint()
    rc = lib().odhip_pvq_k_range_take(ctypes.byref(a), ctypes.byref(b))
    if rc not in (0, ERANGE):
        _check(rc, "odhip_pvq_k_range_take")
    return a.value, b.value


def init(device=0):
    _check(lib().odhip_init(int(device)), "odhip_init")


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _need(t, dtype, what):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype
            and t.is_contiguous()):
        raise DaalaHipError("%s must be a contiguous CUDA tensor of %s" % (what, dtype))


# ---- batched transforms ---------------------------------------------------
def _dct_batch(fn, ln, x, exact32, out):
    import torch
    n = 4 << ln
    _need(x, torch.int32, "x")
    if x.shape[-2:] != (n, n):
        raise DaalaHipError("x must be [..., %d, %d]" % (n, n))
    nblocks = x.numel() // (n * n)
    if out is None:
        out = torch.empty_like(x)
    _check(fn(int(ln), _p(out), _p(x), ctypes.c_long(nblocks), int(bool(exact32)),
              _stream()), fn.__name__)
    return out


def fdct2d_batch(ln, x, exact32=False, out=None):
    """od_bin_fdctNxN (N = 4 << ln) of every [N, N] tile of x (int32, CUDA)."""
    return _dct_batch(lib().odhip_fdct2d_batch, ln, x, exact32, out)


def idct2d_batch(ln, y, exact32=False, out=None):
    return _dct_batch(lib().odhip_idct2d_batch, ln, y, exact32, out)


def _dct_plane(fn, ln, x, exact32):
    import torch
    _need(x, torch.int32, "plane")
    h, w = x.shape
    out = torch.empty_like(x)
    _check(fn(int(ln), _p(out), w, _p(x), w, w, h, int(bool(exact32)), _stream()),
           fn.__name__)
    return out


def fdct2d_plane(ln, x, exact32=False):
    return _dct_plane(lib().odhip_fdct2d_plane, ln, x, exact32)


def idct2d_plane(ln, y, exact32=False):
    return _dct_plane(lib().odhip_idct2d_plane, ln, y, exact32)


def filter_batch(f, x, inverse=False, out=None):
    """(4 << f)-tap lapping filter (od_pre_filterN / od_post_filterN) of every row
    of x: int32 [count, 4 << f] CUDA."""
    import torch
    _need(x, torch.int32, "x")
    count, n = x.shape
    if n != 4 << f:
        raise DaalaHipError("x must be [count, %d]" % (4 << f))
    if out is None:
        out = torch.empty_like(x)
    _check(lib().odhip_filter_batch(int(f), int(bool(inverse)), _p(out), _p(x),
                                    ctypes.c_long(count), _stream()), "odhip_filter_batch")
    return out


def dering_planes(x, xdec, dirs, pli, bskip, thresholds, overlap=1, coeff_shift=4):
    """od_dering on every superblock of every plane for every candidate threshold.
    x int16 [nplanes, h, w]; dirs int32 [nplanes, h/(8>>xdec)... see daala_hip.h] =
    [nplanes, nvsb*8, nhsb*8]; bskip uint8 [nplanes, rows, skip_stride];
    thresholds int32 [nplanes, ncand, nvsb*nhsb].  Returns y int16
    [nplanes, ncand, h, w]."""
    import torch
    _need(x, torch.int16, "x")
    _need(dirs, torch.int32, "dirs")
    _need(bskip, torch.uint8, "bskip")
    _need(thresholds, torch.int32, "thresholds")
    nplanes, h, w = x.shape
    n = 64 >> xdec
    nhsb, nvsb = w // n, h // n
    ncand = thresholds.shape[1]
    y = torch.empty((nplanes, ncand, h, w), dtype=torch.int16, device=x.device)
    _check(lib().odhip_dering_planes(_p(y), _p(x), w, nhsb, nvsb, int(xdec), nplanes, _p(dirs),
                                     int(pli), _p(bskip), bskip.shape[2],
                                     ctypes.c_long(bskip.shape[1] * bskip.shape[2]), _p(thresholds),
                                     ncand, int(overlap), int(coeff_shift), _stream()),
           "odhip_dering_planes")
    return y


# ---- fused lapped stage -----------------------------------------------------
def px_dtype():
    """torch dtype of picture planes in the calling thread's current context: uint8, or int16
    samples at 12 bits when the context is in full-precision-references mode
    (Context.set_fpr, odhip_ctx_set_fpr)."""
    import torch
    L = lib()
    L.odhip_get_current.restype = ctypes.c_void_p
    cur = L.odhip_get_current()
    return torch.int16 if cur and L.odhip_ctx_get_fpr(ctypes.c_void_p(cur)) == 1 else torch.uint8


def forward_pyramid(px, dec, pic_w, pic_h, levels=None, want=None):
    """px: uint8 (int16 in a full-precision-references context) [nplanes, h, w] CUDA.
    Returns a list of int32 [nplanes, h, w] tensors, index bs = 0..4-dec (None where
    `want` excludes a level)."""
    import torch
    _need(px, px_dtype(), "px")
    nplanes, h, w = px.shape
    top = 4 - dec
    if levels is None:
        # ONE allocation, the planes a non-power-of-two distance apart: separate torch allocations
        # of 127.5 MiB (16 frames of 1080p) land exactly 128 MiB apart, and five streams written
        # concurrently at addresses equal modulo 2^27 share HBM channels and banks - 216 us per
        # launch instead of 173 (tools/pyr_skew.py, DESIGN.md par. 4)
        n = nplanes * h * w
        stride = ((n + 1023) & ~1023) + (68 << 10)          # int32 elements; +272 KiB
        ks = [bs for bs in range(top + 1) if (want is None or bs in want)]
        levels = [None] * (top + 1)
        if len(ks) == 1:
            # a single level: its own allocation (a view of a shared block would pin the whole
            # block for as long as the caller keeps that one level)
            levels[ks[0]] = torch.empty((nplanes, h, w), dtype=torch.int32, device=px.device)
        else:
            big = torch.empty((max(len(ks), 1) * stride,), dtype=torch.int32, device=px.device)
            for i, bs in enumerate(ks):
                levels[bs] = big[i * stride: i * stride + n].view(nplanes, h, w)
    arr = (ctypes.c_void_p * 5)()
    for bs in range(5):
        t = levels[bs] if bs <= top else None
        arr[bs] = ctypes.c_void_p(t.data_ptr() if t is not None else None)
    _check(lib().odhip_forward_pyramid(arr, _p(px), w, ctypes.c_long(h * w), nplanes,
                                       w, h, int(dec), int(pic_w), int(pic_h),
                                       _stream()), "odhip_forward_pyramid")
    return levels


def inverse_level(coef, dec, leaf_bs, pic_w, pic_h, out=None):
    """coef: int32 [nplanes, h, w] at uniform partition level leaf_bs ->
    uint8 [nplanes, h, w] reconstructed pixels."""
    import torch
    _need(coef, torch.int32, "coef")
    nplanes, h, w = coef.shape
    if out is None:
        out = torch.empty((nplanes, h, w), dtype=px_dtype(), device=coef.device)
    _check(lib().odhip_inverse_level(_p(out), w, ctypes.c_long(h * w), _p(coef), nplanes,
                                     w, h, int(dec), int(leaf_bs), int(pic_w), int(pic_h),
                                     _stream()), "odhip_inverse_level")
    return out


# ---- PVQ ---------------------------------------------------------------------
def pvq_search_batch(x, k, g2, pvq_norm_lambda, prev_k=None, y=None, cos=None):
    """Batched pvq_search_rdo_double.  x int16 [nbands, n], k int32 [nbands],
    g2 float64 [nbands]; returns (y int32 [nbands, n], cos float64 [nbands])."""
    import torch
    _need(x, torch.int16, "x")
    _need(k, torch.int32, "k")
    _need(g2, torch.float64, "g2")
    nbands, n = x.shape
    if y is None:
        y = torch.zeros((nbands, n), dtype=torch.int32, device=x.device)
    if cos is None:
        cos = torch.empty(nbands, dtype=torch.float64, device=x.device)
    pk = ctypes.c_void_p(None)
    if prev_k is not None:
        _need(prev_k, torch.int32, "prev_k")
        pk = _p(prev_k)
    _check(lib().odhip_pvq_search_batch(_p(x), int(n), _p(k), _p(y), _p(g2),
                                        ctypes.c_double(pvq_norm_lambda), pk, _p(cos),
                                        ctypes.c_long(nbands), _stream()),
           "odhip_pvq_search_batch")
    return y, cos


def copy_ceiling(nbytes=1 << 30, n=10):
    """odhip_copy_ceiling: the best of the library's three 16-byte-vector copy kernels, GB/s read + written."""
    best = 0.0
    for variant in range(3):
        g = ctypes.c_double(0)
        _check(lib().odhip_copy_ceiling(ctypes.c_size_t(nbytes), int(n), variant, ctypes.byref(g), _stream()),
               "odhip_copy_ceiling")
        best = max(best, g.value)
    return best


def pvq_search_row_batch(x, k, g2, pvq_norm_lambda, force_scan=False):
    """odhip_pvq_search_row_batch: pvq_search_rdo_double one band per quad / 16-lane row (n = 31, 32,
    127, 128).  Returns (y int32 [nbands, n], cos float64 [nbands], replays int32 [nbands])."""
    import torch
    _need(x, torch.int16, "x")
    _need(k, torch.int32, "k")
    _need(g2, torch.float64, "g2")
    nbands, n = x.shape
    y = torch.zeros((nbands, n), dtype=torch.int32, device=x.device)
    cos = torch.empty(nbands, dtype=torch.float64, device=x.device)
    rep = torch.zeros(nbands, dtype=torch.int32, device=x.device)
    _check(lib().odhip_pvq_search_row_batch(_p(x), int(n), _p(k), _p(y), _p(g2), ctypes.c_double(pvq_norm_lambda),
                                            int(bool(force_scan)), _p(cos), _p(rep), ctypes.c_long(nbands), _stream()),
           "odhip_pvq_search_row_batch")
    return y, cos, rep


# ---- per-call host-pointer surfaces (numpy) ---------------------------------
class _Host:
    """The reference-signature entry points, callable on numpy arrays."""

    def dct2d(self, ln, x, inverse=False):
        n = 4 << ln
        x = np.ascontiguousarray(x, dtype=np.int32)
        assert x.shape == (n, n)
        out = np.zeros_like(x)
        name = "od_bin_%sdct%dx%d_hip" % ("i" if inverse else "f", n, n)
        getattr(lib(), name)(out.ctypes.data_as(ctypes.c_void_p), n,
                             x.ctypes.data_as(ctypes.c_void_p), n)
        return out

    def pvq_search(self, x, k, g2, lam, prev_k=0, y=None):
        x = np.ascontiguousarray(x, dtype=np.int16)
        n = x.shape[0]
        y = np.zeros(n, np.int32) if y is None else np.ascontiguousarray(y, np.int32)
        c = lib().od_pvq_search_rdo_double_hip(
            x.ctypes.data_as(ctypes.c_void_p), n, int(k),
            y.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(g2),
            ctypes.c_double(lam), int(prev_k))
        return y, c


host = _Host()


# ---- PVQ band stage -----------------------------------------------------------
_CAND_FIELDS = ("band", "y", "choice", "cos_dist")

# odhip_pvq_band (include/daala_hip.h): one 64-byte record per (block, band)
BAND_RECORD = np.dtype([("cg", "<i4"), ("gain", "<i4", (2,)), ("k", "<i2", (2,)),
                        ("flags", "u1", (2,)), ("reserved0", "u1", (6,)), ("dist0", "<f8"),
                        ("yy", "<i4", (2,)), ("dist", "<f8", (2,)), ("moment", "<i4", (2,))])
assert BAND_RECORD.itemsize == 64


class _Cands(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in _CAND_FIELDS]


class _Job(ctypes.Structure):
    _fields_ = [("d_coef", ctypes.c_void_p), ("nplanes", ctypes.c_int), ("w", ctypes.c_int),
                ("h", ctypes.c_int), ("bs", ctypes.c_int), ("d_qm", ctypes.c_void_p),
                ("d_qm_inv", ctypes.c_void_p), ("q_band", ctypes.POINTER(ctypes.c_int32)),
                ("beta_band", ctypes.POINTER(ctypes.c_int32)), ("cands", _Cands),
                ("d_dq", ctypes.c_void_p), ("d_rate", ctypes.c_void_p),
                ("d_qg", ctypes.c_void_p), ("q_band2", ctypes.POINTER(ctypes.c_int32)),
                ("plane_split", ctypes.c_int)]


# ---- with-reference (theta / Householder) building blocks -------------------------
REFPREP_RECORD = np.dtype([("xshift", "<i4"), ("rshift", "<i4"), ("g", "<i4"), ("gr", "<i4"),
                           ("cg", "<i4"), ("cgr", "<i4"), ("icgr", "<i4"), ("gain_offset", "<i4"),
                           ("m", "<i4"), ("s", "<i4"), ("r_null", "<i4"), ("reserved", "<i4"),
                           ("corr", "<f8"), ("reserved2", "<f8")])
REFCAND_RECORD = np.dtype([("gain", "<i4"), ("theta", "<i4"), ("ts", "<i4"), ("k", "<i4"),
                           ("qcg", "<i4"), ("qtheta", "<i4")])
MAX_REFCANDS = 24
assert REFPREP_RECORD.itemsize == 64 and REFCAND_RECORD.itemsize == 24


def pvq_ref_prepare(x0, r0, qm, q0, beta, cfl_enabled):
    """x0, r0: int32 [nbands, n] CUDA; qm int16 [n].  Returns (x16, r16, xr, prep)
    with prep a uint8 [nbands, 64] tensor of odhip_pvq_refprep records."""
    import torch
    _need(x0, torch.int32, "x0")
    _need(r0, torch.int32, "r0")
    _need(qm, torch.int16, "qm")
    nbands, n = x0.shape
    dev = x0.device
    x16 = torch.empty((nbands, n), dtype=torch.int16, device=dev)
    r16 = torch.empty((nbands, n), dtype=torch.int16, device=dev)
    xr = torch.empty((nbands, n - 1), dtype=torch.int16, device=dev)
    prep = torch.empty((nbands, 64), dtype=torch.uint8, device=dev)
    _check(lib().odhip_pvq_ref_prepare(_p(x0), _p(r0), int(n), ctypes.c_long(nbands), _p(qm),
                                       int(q0), int(beta), int(bool(cfl_enabled)), _p(x16), _p(r16),
                                       _p(xr), _p(prep), _stream()), "odhip_pvq_ref_prepare")
    return x16, r16, xr, prep


def pvq_ref_candidates(prep, theta, n, beta):
    """theta: int32 [nbands] = floor(.5 + OD_THETA_SCALE*acos(corr)) from the host.
    Returns (items uint8 [nbands, 24, 24] of odhip_pvq_refcand, nitems int32 [nbands])."""
    import torch
    _need(theta, torch.int32, "theta")
    nbands = prep.shape[0]
    items = torch.zeros((nbands, MAX_REFCANDS, 24), dtype=torch.uint8, device=prep.device)
    nitems = torch.empty(nbands, dtype=torch.int32, device=prep.device)
    _check(lib().odhip_pvq_ref_candidates(_p(prep), _p(theta), int(n), ctypes.c_long(nbands),
                                          int(beta), _p(items), _p(nitems), _stream()),
           "odhip_pvq_ref_candidates")
    return items, nitems


def pvq_synthesis(y, r16, params, qm_inv):
    """od_pvq_synthesis_partial per band.  y int32 [nbands, n], r16 int16 [nbands, n],
    params int32 [nbands, 5] = (noref, g, theta, m, s), qm_inv int16 [n]."""
    import torch
    _need(y, torch.int32, "y")
    _need(r16, torch.int16, "r16")
    _need(params, torch.int32, "params")
    _need(qm_inv, torch.int16, "qm_inv")
    nbands, n = y.shape
    out = torch.empty_like(y)
    _check(lib().odhip_pvq_synthesis(_p(out), _p(y), _p(r16), int(n), ctypes.c_long(nbands),
                                     _p(params), _p(qm_inv), _stream()), "odhip_pvq_synthesis")
    return out


def pvq_profile(enable):
    """Bracket the band stage's dominant kernel with HIP events (see daala_hip.h)."""
    _check(lib().odhip_pvq_profile(int(bool(enable))), "odhip_pvq_profile")


def pvq_profile_read(max_n=256):
    """Milliseconds of every bracketed launch since the last read."""
    buf = (ctypes.c_float * max_n)()
    n = lib().odhip_pvq_profile_read(buf, max_n)
    if n < 0:
        raise DaalaHipError("odhip_pvq_profile_read failed with code %d" % n)
    return [buf[i] for i in range(n)]


def pvq_band_layout(bs):
    """(nb_bands, offsets[nb_bands + 1], len) for block size 4 << bs."""
    nb = ctypes.c_int()
    offs = (ctypes.c_int * 13)()
    ln = ctypes.c_int()
    _check(lib().odhip_pvq_band_layout(int(bs), ctypes.byref(nb), offs, ctypes.byref(ln)),
           "odhip_pvq_band_layout")
    return nb.value, [offs[i] for i in range(nb.value + 1)], ln.value


def alloc_pvq_cands(nblocks, bs, device, cos_dist=False):
    """Device buffers of odhip_pvq_cands: 'band' is the [B][nb] array of 64-byte
    odhip_pvq_band records (as bytes; unpack_cands decodes it), 'y' the int16
    pulse vectors [2][B][len], 'choice' [B][nb][4]; 'cos_dist' [B][nb][2] only on
    request (parity checks)."""
    import torch
    nb, _, ln = pvq_band_layout(bs)
    out = {
        "band": torch.zeros((nblocks, nb, 64), dtype=torch.uint8, device=device),
        "y": torch.zeros((2, nblocks, ln), dtype=torch.int16, device=device),
        "choice": torch.empty((nblocks, nb, 4), dtype=torch.int32, device=device),
        "cos_dist": None,
    }
    if cos_dist:
        out["cos_dist"] = torch.zeros((nblocks, nb, 2), dtype=torch.float64, device=device)
    return out


def unpack_cands(cands):
    """Host copy of a cands dict as numpy arrays keyed by field name: the record
    fields (cg, gain, k, flags, dist0, yy, dist, moment: [B][nb] or [B][nb][2]) plus y,
    choice and, when present, cos_dist."""
    rec = cands["band"].cpu().numpy().view(BAND_RECORD)[..., 0]
    out = {name: np.ascontiguousarray(rec[name]) for name in
           ("cg", "gain", "k", "flags", "dist0", "yy", "dist", "moment")}
    out["y"] = cands["y"].cpu().numpy()
    out["choice"] = cands["choice"].cpu().numpy()
    if cands.get("cos_dist") is not None:
        out["cos_dist"] = cands["cos_dist"].cpu().numpy()
    return out


class PvqJob:
    """One (plane set, block size) unit of the PVQ band stage; keeps every
    tensor and host array it points to alive."""

    def __init__(self, coef, bs, qm, qm_inv, q_band, beta_band, cands=None, dq=None,
                 rate=None, qg=None, q_band2=None, plane_split=0):
        """q_band2 / plane_split: the planes from plane_split on (the Cr half of a chroma
        plane set) take q_band2 - pvq_qm_q4[pli] is per plane in the reference."""
        import torch
        _need(coef, torch.int32, "coef")
        self.coef, self.bs, self.qm, self.qm_inv = coef, int(bs), qm, qm_inv
        nplanes, h, w = coef.shape
        n = 4 << bs
        self.nblocks = nplanes * (h // n) * (w // n)
        nb = pvq_band_layout(bs)[0]
        assert len(q_band) == nb and len(beta_band) == nb
        self.q_band = (ctypes.c_int32 * 12)(*[int(v) for v in q_band])
        self.beta_band = (ctypes.c_int32 * 12)(*[int(v) for v in beta_band])
        self.q_band2 = (ctypes.c_int32 * 12)(*[int(v) for v in q_band2]) if q_band2 is not None else None
        self.plane_split = int(plane_split) if q_band2 is not None else 0
        self.cands = cands if cands is not None else alloc_pvq_cands(self.nblocks, bs,
                                                                     coef.device)
        self.dq, self.rate, self.qg = dq, rate, qg
        for t, dt, what in ((qm, torch.int16, "qm"), (qm_inv, torch.int16, "qm_inv"),
                            (dq, torch.int32, "dq"), (rate, torch.float64, "rate"),
                            (qg, torch.int32, "qg")):
            if t is not None:
                _need(t, dt, what)

    def struct(self):
        opt = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)  # noqa: E731
        nplanes, h, w = self.coef.shape
        c = _Cands(*[ctypes.c_void_p(self.cands[n].data_ptr() if self.cands.get(n) is not None
                                     else None) for n in _CAND_FIELDS])
        return _Job(_p(self.coef), nplanes, w, h, self.bs, opt(self.qm), opt(self.qm_inv),
                    self.q_band, self.beta_band, c, opt(self.dq), opt(self.rate), opt(self.qg),
                    self.q_band2, self.plane_split)


def _jobs_array(jobs):
    arr = (_Job * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i] = j.struct()
    return arr


def pvq_noref_bands_multi(jobs, pvq_norm_lambda):
    """The adaptation-independent part of pvq_theta (no-reference path) for every
    block and band of every job, in one set of launches."""
    _check(lib().odhip_pvq_noref_bands_multi(_jobs_array(jobs), len(jobs),
                                             ctypes.c_double(pvq_norm_lambda), _stream()),
           "odhip_pvq_noref_bands_multi")


def pvq_select_synth_noref_multi(jobs, pvq_norm_lambda):
    """Choice (`cost <= best_cost`, cost = dist + lambda*rate) + decoder-identical
    dequantisation into each job's dq plane."""
    _check(lib().odhip_pvq_select_synth_noref_multi(_jobs_array(jobs), len(jobs),
                                                    ctypes.c_double(pvq_norm_lambda),
                                                    _stream()),
           "odhip_pvq_select_synth_noref_multi")


def pvq_choose_multi(jobs, pvq_norm_lambda):
    """The `cost <= best_cost` choice alone (fills cands['choice'])."""
    _check(lib().odhip_pvq_choose_multi(_jobs_array(jobs), len(jobs),
                                        ctypes.c_double(pvq_norm_lambda), _stream()),
           "odhip_pvq_choose_multi")


def _resolved(n, what):
    if n < 0:
        raise DaalaHipError("%s failed with code %d" % (what, n))
    return n


def pvq_choose_priced_multi(jobs, pvq_norm_lambda, fused_bands=False):
    """The choice priced with od_pvq_rate's closed form on the device.  fused_bands=False:
    after pvq_noref_bands_multi, a choice kernel reads the records
    (odhip_pvq_choose_priced_multi); True: band stage AND choice in one pass
    (odhip_pvq_noref_bands_priced_multi).  Either way followed by the host-libm resolve;
    returns the number of bands it re-decided."""
    arr = _jobs_array(jobs)
    lam = ctypes.c_double(pvq_norm_lambda)
    if fused_bands:
        _check(lib().odhip_pvq_noref_bands_priced_multi(arr, len(jobs), lam, _stream()),
               "odhip_pvq_noref_bands_priced_multi")
    else:
        _check(lib().odhip_pvq_choose_priced_multi(arr, len(jobs), lam, _stream()),
               "odhip_pvq_choose_priced_multi")
    return _resolved(lib().odhip_pvq_choose_priced_resolve(arr, len(jobs), lam, _stream()),
                     "odhip_pvq_choose_priced_resolve")


def inverse_level_pvq(job, dec, pic_w, pic_h, out=None):
    """Inverse stage fed by the band stage: dequantise-on-load from the chosen
    pulse vectors of `job` (no dequantised plane in HBM)."""
    import torch
    nplanes, h, w = job.coef.shape
    if out is None:
        out = torch.empty((nplanes, h, w), dtype=px_dtype(), device=job.coef.device)
    st = job.struct()
    _check(lib().odhip_inverse_level_pvq(_p(out), w, ctypes.c_long(h * w), ctypes.byref(st),
                                         int(dec), int(pic_w), int(pic_h), _stream()),
           "odhip_inverse_level_pvq")
    return out


def inverse_levels_pvq(jobs, dec, pic_w, pic_h, outs=None):
    """inverse_level_pvq for several levels of one plane set in one set of launches;
    returns the list of reconstructed uint8 [nplanes, h, w] tensors."""
    import torch
    nplanes, h, w = jobs[0].coef.shape
    if outs is None:
        outs = [torch.empty((nplanes, h, w), dtype=px_dtype(), device=jobs[0].coef.device)
                for _ in jobs]
    ptrs = (ctypes.c_void_p * len(jobs))(*[o.data_ptr() for o in outs])
    _check(lib().odhip_inverse_levels_pvq(ptrs, w, ctypes.c_long(h * w), _jobs_array(jobs),
                                          len(jobs), int(dec), int(pic_w), int(pic_h), _stream()),
           "odhip_inverse_levels_pvq")
    return outs


def pvq_noref_bands(coef, bs, qm, q_band, beta_band, pvq_norm_lambda, out=None, cos_dist=False):
    if out is None and cos_dist:
        nplanes, h, w = coef.shape
        n = 4 << bs
        out = alloc_pvq_cands(nplanes * (h // n) * (w // n), bs, coef.device, cos_dist=True)
    job = PvqJob(coef, bs, qm, None, q_band, beta_band, cands=out)
    pvq_noref_bands_multi([job], pvq_norm_lambda)
    return job.cands


def pvq_select_synth_noref(coef, bs, qm_inv, q_band, beta_band, pvq_norm_lambda, cands,
                           rate=None, dq=None, qg=None):
    import torch
    if dq is None:
        dq = torch.empty_like(coef)
    if qg is None:
        qg = torch.empty(tuple(cands["band"].shape[:2]), dtype=torch.int32, device=coef.device)
    job = PvqJob(coef, bs, None, qm_inv, q_band, beta_band, cands=cands, dq=dq, rate=rate, qg=qg)
    pvq_select_synth_noref_multi([job], pvq_norm_lambda)
    return dq, qg


# ---- with-reference band stage on whole planes (odhip_pvq_ref_bands_multi) ---------
REF_SLOTS = 16
REFBAND_RECORD = np.dtype([("xshift", "<i4"), ("rshift", "<i4"), ("g", "<i4"), ("gr", "<i4"),
                           ("cg", "<i4"), ("cgr", "<i4"), ("icgr", "<i4"), ("gain_offset", "<i4"),
                           ("m", "<i2"), ("s", "i1"), ("flags", "u1"), ("theta", "<i4"),
                           ("nitems", "<i4"), ("ntheta", "<i4"), ("corr", "<f8"), ("dist0", "<f8")])
assert REFBAND_RECORD.itemsize == 64
REFITEM_RECORD = np.dtype([("gain", "<i4"), ("theta", "<i4"), ("ts", "<i4"), ("k", "<i4"),
                           ("qcg", "<i4"), ("qtheta", "<i4"), ("flags", "<i4"), ("yslot", "<i4"),
                           ("cos_dist", "<f8"), ("dist", "<f8")])
assert REFITEM_RECORD.itemsize == 48
REFBAND_R_NULL, REFBAND_THETA, REFBAND_NOREF, REFBAND_FLIP, REFBAND_UNCERTAIN = 1, 2, 4, 8, 16
REFITEM_SEARCHED, REFITEM_WITH_REF, REFITEM_K_RANGE = 1, 2, 4
REFITEM_MOMENT_SHIFT = 8


class _RefJob(ctypes.Structure):
    _fields_ = [("d_coef", ctypes.c_void_p), ("d_ref", ctypes.c_void_p), ("nplanes", ctypes.c_int),
                ("w", ctypes.c_int), ("h", ctypes.c_int), ("bs", ctypes.c_int),
                ("is_keyframe", ctypes.c_int), ("pli", ctypes.c_int), ("d_qm", ctypes.c_void_p),
                ("d_qm_inv", ctypes.c_void_p), ("q_band", ctypes.POINTER(ctypes.c_int32)),
                ("beta_band", ctypes.POINTER(ctypes.c_int32)), ("band", ctypes.c_void_p),
                ("items", ctypes.c_void_p), ("y", ctypes.c_void_p), ("r16", ctypes.c_void_p),
                ("x16", ctypes.c_void_p), ("xr", ctypes.c_void_p), ("d_rate", ctypes.c_void_p),
                ("choice", ctypes.c_void_p), ("d_dq", ctypes.c_void_p),
                ("q_band2", ctypes.POINTER(ctypes.c_int32)), ("plane_split", ctypes.c_int),
                ("luma", ctypes.c_void_p)]


class PvqRefJob:
    """One (plane set, reference plane set, block size) unit of the with-reference
    band stage (odhip_pvq_refjob); owns its output and work buffers."""

    def __init__(self, coef, ref, bs, qm, qm_inv, q_band, beta_band, is_keyframe, pli, rate=None,
                 share=None, q_band2=None, plane_split=0):
        """share: another PvqRefJob of the same shape whose output and work buffers this
        one uses too (same planes, another reference buffer: double-buffered references)."""
        import torch
        _need(coef, torch.int32, "coef")
        _need(ref, torch.int32, "ref")
        assert coef.shape == ref.shape
        self.coef, self.ref, self.bs, self.qm, self.qm_inv = coef, ref, int(bs), qm, qm_inv
        self.is_keyframe, self.pli, self.rate = int(is_keyframe), int(pli), rate
        nplanes, h, w = coef.shape
        n = 4 << bs
        self.nblocks = B = nplanes * (h // n) * (w // n)
        self.nb, self.offsets, self.len = pvq_band_layout(bs)
        assert len(q_band) == self.nb and len(beta_band) == self.nb
        self.q_band = (ctypes.c_int32 * 12)(*[int(v) for v in q_band])
        self.beta_band = (ctypes.c_int32 * 12)(*[int(v) for v in beta_band])
        self.q_band2 = (ctypes.c_int32 * 12)(*[int(v) for v in q_band2]) if q_band2 is not None else None
        self.plane_split = int(plane_split) if q_band2 is not None else 0
        dev = coef.device
        if share is not None:
            assert share.coef.shape == coef.shape and share.bs == self.bs
            for f in ("band", "items", "y", "r16", "x16", "xr", "choice", "dq"):
                setattr(self, f, getattr(share, f))
            return
        self.band = torch.zeros((B, self.nb, 64), dtype=torch.uint8, device=dev)
        self.items = torch.zeros((3, self.nb, REF_SLOTS, B, 16), dtype=torch.uint8, device=dev)
        self.y = torch.zeros((REF_SLOTS, B, self.len), dtype=torch.int16, device=dev)
        self.r16 = torch.zeros((B, self.len), dtype=torch.int16, device=dev)
        self.x16 = torch.zeros((B, self.len), dtype=torch.int16, device=dev)
        self.xr = torch.zeros((B, self.len), dtype=torch.int16, device=dev)
        self.choice = torch.zeros((B, self.nb, 16), dtype=torch.int32, device=dev)
        self.dq = torch.zeros_like(coef)
        for t, dt, what in ((qm, torch.int16, "qm"), (qm_inv, torch.int16, "qm_inv"),
                            (rate, torch.float64, "rate")):
            if t is not None:
                _need(t, dt, what)

    def struct(self):
        opt = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)  # noqa: E731
        nplanes, h, w = self.coef.shape
        return _RefJob(_p(self.coef), _p(self.ref), nplanes, w, h, self.bs, self.is_keyframe,
                       self.pli, opt(self.qm), opt(self.qm_inv), self.q_band, self.beta_band,
                       _p(self.band), _p(self.items), _p(self.y), _p(self.r16), _p(self.x16),
                       _p(self.xr), opt(self.rate), _p(self.choice), _p(self.dq), self.q_band2,
                       self.plane_split, None)

    def unpack(self):
        """Host copies: record fields [B][nb], item fields [B][nb][REF_SLOTS], y, choice."""
        rec = self.band.cpu().numpy().view(REFBAND_RECORD)[..., 0]
        # three planes of 16-byte vectors [part][nb][slot][B] -> records [B][nb][slot]
        raw = self.items.cpu().numpy()
        items = np.zeros(raw.shape[1:4], REFITEM_RECORD)
        head = raw[0].view("<i4")
        tail = raw[1].view("<i4")
        res = raw[2].view("<f8")
        for i, f in enumerate(("gain", "theta", "ts", "k")):
            items[f] = head[..., i]
        for i, f in enumerate(("qcg", "qtheta", "flags", "yslot")):
            items[f] = tail[..., i]
        items["cos_dist"] = res[..., 0]
        items["dist"] = res[..., 1]
        items = np.ascontiguousarray(items.transpose(2, 0, 1))
        return {"rec": rec, "items": items, "y": self.y.cpu().numpy(),
                "choice": self.choice.cpu().numpy(), "r16": self.r16.cpu().numpy(),
                "x16": self.x16.cpu().numpy(), "xr": self.xr.cpu().numpy()}


def _refjobs_array(jobs):
    arr = (_RefJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i] = j.struct()
    return arr


def pvq_ref_bands_multi(jobs, pvq_norm_lambda, resolve=True):
    """pvq_theta with a reference up to the rate-dependent choice, for every block
    and band of every job; resolve=True also settles the bands whose theta is
    inside the device-acos margin with the host libm (returns how many were
    re-run)."""
    arr = _refjobs_array(jobs)
    _check(lib().odhip_pvq_ref_bands_multi(arr, len(jobs), ctypes.c_double(pvq_norm_lambda),
                                           _stream()), "odhip_pvq_ref_bands_multi")
    if not resolve:
        return 0
    if resolve == "async":
        _check(lib().odhip_pvq_ref_resolve_begin(_stream()), "odhip_pvq_ref_resolve_begin")
        return 0
    n = lib().odhip_pvq_ref_resolve(arr, len(jobs), ctypes.c_double(pvq_norm_lambda), _stream())
    if n < 0:
        raise DaalaHipError("odhip_pvq_ref_resolve failed with code %d" % n)
    return n


def pvq_ref_resolve_finish(jobs, pvq_norm_lambda):
    """Second half of pvq_ref_bands_multi(..., resolve="async"): waits for the count
    of bands inside the device-acos margin only; 0 (normal) or the number of bands
    re-run with the host's theta - then the caller repeats choice / synthesis."""
    n = lib().odhip_pvq_ref_resolve_finish(_refjobs_array(jobs), len(jobs),
                                           ctypes.c_double(pvq_norm_lambda), _stream())
    if n < 0:
        raise DaalaHipError("odhip_pvq_ref_resolve_finish failed with code %d" % n)
    return n


def pvq_ref_select_synth_multi(jobs, pvq_norm_lambda):
    """The reference's choice among the candidates (cost = dist + lambda*rate, rate
    table optional), skip rules and decoder-identical synthesis into job.dq."""
    _check(lib().odhip_pvq_ref_select_synth_multi(_refjobs_array(jobs), len(jobs),
                                                  ctypes.c_double(pvq_norm_lambda), _stream()),
           "odhip_pvq_ref_select_synth_multi")


# Test hooks live in the library's contexts (odhip_ctx_set_test_hooks); this module keeps the
# values the tests asked for and applies them to the calling thread's current context and to
# every live Pipe (and to Pipes made later).
_hooks = {"margin": 0.0, "perturb": 0, "tol": 1.0}
_pipes = None


def _apply_hooks():
    global _pipes
    import weakref
    if _pipes is None:
        _pipes = weakref.WeakSet()
    L = lib()
    args = (ctypes.c_double(_hooks["margin"]), int(_hooks["perturb"]), ctypes.c_double(_hooks["tol"]))
    _check(L.odhip_ctx_set_test_hooks(None, *args), "odhip_ctx_set_test_hooks")
    for p in list(_pipes):
        if p.h:
            _check(L.odhip_pipe_set_test_hooks(p._p(), *args), "odhip_pipe_set_test_hooks")


def pvq_ref_set_theta_margin(margin, perturb=False):
    """Test hook (per context in the library): the device-acos uncertainty margin (<= 0: the
    default) and a deliberately wrong device theta for the listed bands."""
    _hooks["margin"] = float(margin)
    _hooks["perturb"] = int(bool(perturb))
    _apply_hooks()


def pvq_ref_theta_probe(corr):
    """.5 + OD_THETA_SCALE*acos(corr) as the device evaluates it."""
    import torch
    _need(corr, torch.float64, "corr")
    out = torch.empty_like(corr)
    _check(lib().odhip_pvq_ref_theta_probe(_p(corr), _p(out), ctypes.c_long(corr.numel()),
                                           _stream()), "odhip_pvq_ref_theta_probe")
    return out


def pvq_ref_profile(enable):
    _check(lib().odhip_pvq_ref_profile(int(bool(enable))), "odhip_pvq_ref_profile")


def pvq_ref_profile_read(max_n=256):
    """Milliseconds of each k_refb_search_row<8> launch recorded since the last read."""
    buf = (ctypes.c_float * max_n)()
    n = lib().odhip_pvq_ref_profile_read(buf, max_n)
    if n < 0:
        raise DaalaHipError("odhip_pvq_ref_profile_read failed with code %d" % n)
    return [buf[i] for i in range(n)]


def image_planes_copy_pad(src, plane_w, plane_h, out=None):
    """od_img_plane_copy_pad for a batch of planes: src uint8 [nplanes, pic_h, pic_w]
    -> uint8 [nplanes, plane_h, plane_w] (picture copied, padding low-pass
    extended as the reference's input queue does)."""
    import torch
    _need(src, torch.uint8, "src")
    nplanes, pic_h, pic_w = src.shape
    if out is None:
        out = torch.empty((nplanes, plane_h, plane_w), dtype=torch.uint8, device=src.device)
    _check(lib().odhip_image_planes_copy_pad(_p(out), plane_w, ctypes.c_long(plane_h * plane_w),
                                             plane_w, plane_h, _p(src) if src.numel() else None,
                                             pic_w, ctypes.c_long(pic_h * pic_w), pic_w, pic_h,
                                             nplanes, _stream()), "odhip_image_planes_copy_pad")
    return out


def image_planes_copy_pad16(src, plane_w, plane_h, src_bitdepth=8, out=None):
    """od_img_plane_copy_pad for an encoder with full-precision references: src uint8
    (src_bitdepth 8) or int16 (10 / 12) [nplanes, pic_h, pic_w] -> int16 samples at 12 bits
    [nplanes, plane_h, plane_w]."""
    import torch
    _need(src, torch.uint8 if src_bitdepth == 8 else torch.int16, "src")
    nplanes, pic_h, pic_w = src.shape
    if out is None:
        out = torch.empty((nplanes, plane_h, plane_w), dtype=torch.int16, device=src.device)
    _check(lib().odhip_image_planes_copy_pad16(_p(out), plane_w, ctypes.c_long(plane_h * plane_w),
                                               plane_w, plane_h, _p(src) if src.numel() else None,
                                               int(src_bitdepth), pic_w, ctypes.c_long(pic_h * pic_w),
                                               pic_w, pic_h, nplanes, _stream()),
           "odhip_image_planes_copy_pad16")
    return out


def inverse_levels(coefs, dec, leaf_bs, pic_w, pic_h, outs=None):
    """inverse_level for several partition levels of one plane set in one set of
    launches: coefs[i] (int32 [nplanes, h, w]) at level leaf_bs[i] -> outs[i]."""
    import torch
    nplanes, h, w = coefs[0].shape
    if outs is None:
        outs = [torch.empty((nplanes, h, w), dtype=px_dtype(), device=coefs[0].device) for _ in coefs]
    n = len(coefs)
    for c in coefs:
        _need(c, torch.int32, "coef")
        assert c.shape == coefs[0].shape
    px = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    cf = (ctypes.c_void_p * n)(*[c.data_ptr() for c in coefs])
    lv = (ctypes.c_int * n)(*[int(v) for v in leaf_bs])
    _check(lib().odhip_inverse_levels(px, w, ctypes.c_long(h * w), cf, lv, n, nplanes, w, h, int(dec),
                                      int(pic_w), int(pic_h), _stream()), "odhip_inverse_levels")
    return outs


def cfl_refs_from_luma(luma_jobs, refs=None, copies=2):
    """Chroma-from-luma reference planes from the luma band stage's chosen candidates
    (od_resample_luma_coeffs, non-TF branch): luma_jobs[j] at level bs >= 1 -> refs[j]
    int32 [copies*nplanes, h/2, w/2] for the chroma level bs - 1."""
    import torch
    if refs is None:
        refs = []
        for j in luma_jobs:
            nplanes, h, w = j.coef.shape
            refs.append(torch.empty((copies * nplanes, h // 2, w // 2), dtype=torch.int32,
                                    device=j.coef.device))
    ptrs = (ctypes.c_void_p * len(refs))(*[r.data_ptr() for r in refs])
    _check(lib().odhip_cfl_refs_from_luma(_jobs_array(luma_jobs), len(luma_jobs), ptrs, int(copies),
                                          _stream()), "odhip_cfl_refs_from_luma")
    return refs


def pvq_ref_set_context(ctx):
    """Selects which of the two library contexts the following pvq_ref_* calls of this
    thread use (one call sequence may be in flight per context)."""
    _check(lib().odhip_pvq_ref_set_context(int(ctx)), "odhip_pvq_ref_set_context")
    # the test hooks are per context: a test that set a margin / perturbation / tolerance scale and
    # then switches context would silently run without them
    if any(_hooks[k] != v for k, v in (("margin", 0.0), ("perturb", 0), ("tol", 1.0))):
        _apply_hooks()


def pvq_ref_choose_multi(jobs, pvq_norm_lambda):
    """The with-reference stage's choice alone (choice records incl. the synthesis
    parameters), for inverse_levels_pvq_ref."""
    _check(lib().odhip_pvq_ref_choose_multi(_refjobs_array(jobs), len(jobs),
                                            ctypes.c_double(pvq_norm_lambda), _stream()),
           "odhip_pvq_ref_choose_multi")


def pvq_ref_choose_priced_multi(jobs, pvq_norm_lambda, fused_bands=False):
    """The with-reference stage's choice priced on the device.  fused_bands=False: after
    pvq_ref_bands_multi, every band decided from its candidate records
    (odhip_pvq_ref_choose_priced_multi); True: band stage with the per-lane bands decided
    inside their searches, then the rest (odhip_pvq_ref_bands_priced_multi +
    odhip_pvq_ref_resolve + odhip_pvq_ref_choose_priced_rest_multi).  Followed by the host-libm
    resolve; returns the number of bands it re-decided."""
    arr = _refjobs_array(jobs)
    lam = ctypes.c_double(pvq_norm_lambda)
    L = lib()
    if fused_bands:
        _check(L.odhip_pvq_ref_bands_priced_multi(arr, len(jobs), lam, _stream()), "odhip_pvq_ref_bands_priced_multi")
        if _resolved(L.odhip_pvq_ref_resolve(arr, len(jobs), lam, _stream()), "odhip_pvq_ref_resolve"):
            # bands were re-run with the host's theta: everything is decided from the records
            _check(L.odhip_pvq_ref_choose_priced_multi(arr, len(jobs), lam, _stream()),
                   "odhip_pvq_ref_choose_priced_multi")
        else:
            _check(L.odhip_pvq_ref_choose_priced_rest_multi(arr, len(jobs), lam, _stream()),
                   "odhip_pvq_ref_choose_priced_rest_multi")
    else:
        _check(L.odhip_pvq_ref_choose_priced_multi(arr, len(jobs), lam, _stream()),
               "odhip_pvq_ref_choose_priced_multi")
    return _resolved(L.odhip_pvq_ref_choose_priced_resolve(arr, len(jobs), lam, _stream()),
                     "odhip_pvq_ref_choose_priced_resolve")


def pvq_decode_bands(ref, y, sym, qm, qm_inv, q0, beta, is_keyframe, pli):
    """The decoder's arithmetic for a batch of bands (odhip_pvq_decode_bands): ref, y int32
    [nbands, n] CUDA, sym int32 [nbands, 4] = {gain symbol as read, itheta, noref, 0}, qm /
    qm_inv int16 [n].  Returns (out int32 [nbands, n], info int32 [nbands, 2] = {K, skip})."""
    import torch
    _need(ref, torch.int32, "ref")
    _need(y, torch.int32, "y")
    _need(sym, torch.int32, "sym")
    _need(qm, torch.int16, "qm")
    _need(qm_inv, torch.int16, "qm_inv")
    nbands, n = ref.shape
    assert y.shape == ref.shape and sym.shape == (nbands, 4)
    out = torch.empty_like(ref)
    info = torch.empty((nbands, 2), dtype=torch.int32, device=ref.device)
    _check(lib().odhip_pvq_decode_bands(_p(out), _p(ref), _p(y), int(n), ctypes.c_long(nbands), _p(sym),
                                        _p(qm), _p(qm_inv), int(q0), int(beta), int(is_keyframe), int(pli),
                                        _p(info), _stream()), "odhip_pvq_decode_bands")
    return out, info


def pvq_ref_bands_decided_multi(jobs, pvq_norm_lambda):
    """The with-reference band stage with the priced choice of every band made inside its
    search (odhip_pvq_ref_bands_decided_multi), followed by the two resolves.  Leaves the
    choice records and, in slot 0 of each job's y, the winners' pulse vectors.  Returns
    (bands re-run with the host's theta, bands re-decided with the host libm)."""
    arr = _refjobs_array(jobs)
    lam = ctypes.c_double(pvq_norm_lambda)
    L = lib()
    _check(L.odhip_pvq_ref_bands_decided_multi(arr, len(jobs), lam, _stream()),
           "odhip_pvq_ref_bands_decided_multi")
    n = _resolved(L.odhip_pvq_ref_resolve_finish(arr, len(jobs), lam, _stream()), "odhip_pvq_ref_resolve_finish")
    m = _resolved(L.odhip_pvq_ref_choose_priced_resolve(arr, len(jobs), lam, _stream()),
                  "odhip_pvq_ref_choose_priced_resolve")
    return n, m


def inverse_levels_pvq_ref(jobs, dec, pic_w, pic_h, outs=None):
    """Inverse of several partition levels of one plane set fed by the with-reference
    band stage: dequantise-on-load from the chosen candidates (no dequantised plane)."""
    import torch
    nplanes, h, w = jobs[0].coef.shape
    if outs is None:
        outs = [torch.empty((nplanes, h, w), dtype=px_dtype(), device=jobs[0].coef.device)
                for _ in jobs]
    px = (ctypes.c_void_p * len(jobs))(*[o.data_ptr() for o in outs])
    _check(lib().odhip_inverse_levels_pvq_ref(px, w, ctypes.c_long(h * w), _refjobs_array(jobs),
                                              len(jobs), int(dec), int(pic_w), int(pic_h), _stream()),
           "odhip_inverse_levels_pvq_ref")
    return outs


# ---- contexts ------------------------------------------------------------------------
class Context:
    """odhip_ctx: owner of the library state one call sequence needs (daala_hip.h).
    `with ctx:` makes it the calling thread's current context."""

    def __init__(self, device=0):
        lib().odhip_create.restype = ctypes.c_void_p
        self.handle = lib().odhip_create(int(device))
        if not self.handle:
            raise DaalaHipError("odhip_create(%d) failed" % device)
        self._prev = []

    def __enter__(self):
        lib().odhip_get_current.restype = ctypes.c_void_p
        self._prev.append(lib().odhip_get_current())
        _check(lib().odhip_make_current(ctypes.c_void_p(self.handle)), "odhip_make_current")
        if any(_hooks[k] != v for k, v in (("margin", 0.0), ("perturb", 0), ("tol", 1.0))):
            _apply_hooks()      # per-context test hooks follow the selection
        return self

    def __exit__(self, *exc):
        _check(lib().odhip_make_current(ctypes.c_void_p(self._prev.pop())), "odhip_make_current")

    def set_fpr(self, on=True):
        """Full-precision references: picture planes of this context's calls are int16
        samples at 12 bits (odhip_ctx_set_fpr)."""
        _check(lib().odhip_ctx_set_fpr(ctypes.c_void_p(self.handle), int(bool(on))), "odhip_ctx_set_fpr")
        return self

    def destroy(self):
        if self.handle:
            lib().odhip_destroy(ctypes.c_void_p(self.handle))
            self.handle = None


# ---- odhip_pipe: the frame-batch step as one C call -----------------------------------
PIPE_STAGES = ("image_copy_pad_luma", "forward_pyramid_luma", "pvq_noref_bands", "pvq_choose",
               "cfl_refs_from_luma", "dequant_inverse_luma", "image_copy_pad_chroma",
               "forward_pyramid_chroma", "pvq_ref_bands", "pvq_ref_choose", "dequant_inverse_chroma")
(BUF_PIC, BUF_PX, BUF_LEVEL, BUF_RECON, BUF_BAND, BUF_Y, BUF_CHOICE, BUF_ITEMS, BUF_REF,
 BUF_RATE) = range(10)


class _PipeConfig(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("frames", ctypes.c_int), ("pic_w", ctypes.c_int),
                ("pic_h", ctypes.c_int), ("chroma_cfl", ctypes.c_int), ("serial", ctypes.c_int),
                ("price", ctypes.c_int), ("fpr_bits", ctypes.c_int), ("inter", ctypes.c_int),
                ("reserved", ctypes.c_int),
                ("pvq_norm_lambda", ctypes.c_double), ("quant", ctypes.c_void_p)]


def export_layout_make(nblocks, bs, with_ref):
    """odhip_export_layout_make for sections of nblocks[i] blocks at level bs[i] -> (ctypes layout, dict)."""
    lay = _ExportLayout()
    n = len(bs)
    _check(lib().odhip_export_layout_make(ctypes.byref(lay), n, (ctypes.c_long * n)(*nblocks), (ctypes.c_int * n)(*bs),
                                          (ctypes.c_int * n)(*[int(bool(w)) for w in with_ref])),
           "odhip_export_layout_make")
    return lay, _layout_dict(lay)


def _layout_dict(lay):
    secs = [{k: int(getattr(lay.section[i], k)) for k in ("bs", "ngroups", "cap_words", "blocks_per_group", "record_bytes",
                                                         "nrecords", "records_off", "group_base_off", "stream_off")}
            for i in range(lay.nsections)]
    return {"nsections": int(lay.nsections), "fixed_bytes": int(lay.fixed_bytes),
            "total_bytes": int(lay.total_bytes), "sections": secs}


def decode_export_sections(host, lay):
    """Reference-side decoder of the export format (include/daala_hip.h, export_kernels.hip): `host` = the buffer
    as numpy uint8, `lay` = the layout dict.  Per section (y int32 [B][len], band int32 [B][nb][4] = {coded gain
    index, itheta, max_theta, k}, coded bool [B][nb]).  A host entropy coder walks the same records and words
    sequentially; this vectorised form exists for the tests and the bench.  k of a band that is not coded is 0."""
    hdr = host[:128].view(np.uint32)
    out = []
    rec4 = np.dtype([("qg", "<i2"), ("fn", "<u2")])
    rec8 = np.dtype([("qg", "<i2"), ("itheta", "<i2"), ("max_theta", "<i2"), ("fn", "<u2")])
    for si, sec in enumerate(lay["sections"]):
        nb, offs, ln = pvq_band_layout(sec["bs"])
        G = sec["blocks_per_group"] * nb          # records per group
        B = sec["nrecords"] // nb
        assert int(hdr[16 + si]) == 0, "export stream of section %d overflowed" % si
        rb = sec["record_bytes"]
        rec = host[sec["records_off"]:sec["records_off"] + sec["nrecords"] * rb].view(rec8 if rb == 8 else rec4)
        gbase = host[sec["group_base_off"]:sec["group_base_off"] + sec["ngroups"] * 4].view(np.uint32).astype(np.int64)
        total = int(hdr[si])
        words = host[sec["stream_off"]:sec["stream_off"] + total * 2].view(np.uint16)
        nw = (rec["fn"] & 0x1ff).astype(np.int64)
        assert total == int(nw.sum()), (si, total, int(nw.sum()))
        # first word of every band: its group's base + the words of the group's earlier bands
        pad = (-len(nw)) % G
        nwp = np.concatenate([nw, np.zeros(pad, np.int64)]).reshape(-1, G)
        start = (gbase[:, None] + np.cumsum(nwp, axis=1) - nwp).reshape(-1)[:len(nw)]
        band_of_word = np.repeat(np.arange(len(nw)), nw)
        rank = np.arange(len(band_of_word)) - np.repeat(np.cumsum(nw) - nw, nw)
        w = words[start[band_of_word] + rank].astype(np.int64)
        pos = w & 127
        val = ((w >> 7) ^ 256) - 256          # sign-extend the 9-bit count
        keep = np.ones(len(w), bool)
        idx = np.nonzero(val == -256)[0]
        # escapes: the word after a count of -256 is the count itself.  (A payload word can look like an escape:
        # walk them in order so that a payload is never taken for one.)
        if len(idx):
            is_payload = np.zeros(len(w), bool)
            for i in idx:
                if is_payload[i]:
                    continue
                is_payload[i + 1] = True
                full = int(w[i + 1])
                val[i] = full - 65536 if full >= 32768 else full
            keep = ~is_payload
        y = np.zeros((B, ln), np.int32)
        bw = band_of_word[keep]
        y[bw // nb, np.asarray(offs)[bw % nb] + pos[keep]] = val[keep]
        band = np.zeros((B, nb, 4), np.int32)
        r = rec.reshape(B, nb)
        band[..., 0] = r["qg"]
        band[..., 1] = r["itheta"] if rb == 8 else -1
        band[..., 2] = r["max_theta"] if rb == 8 else 0
        # K of a coded band = the magnitudes of its counts (not exported)
        for i in range(nb):
            band[:, i, 3] = np.abs(y[:, offs[i]:offs[i + 1]]).sum(axis=1)
        coded = ((r["fn"] >> 10) & 3) == 0
        out.append((y, band, coded))
    return out


class _ExportSection(ctypes.Structure):
    _fields_ = [("bs", ctypes.c_int32), ("ngroups", ctypes.c_uint32), ("cap_words", ctypes.c_uint32),
                ("blocks_per_group", ctypes.c_uint32), ("record_bytes", ctypes.c_uint32), ("pad", ctypes.c_uint32),
                ("nrecords", ctypes.c_uint64), ("records_off", ctypes.c_uint64),
                ("group_base_off", ctypes.c_uint64), ("stream_off", ctypes.c_uint64)]


class _ExportLayout(ctypes.Structure):
    _fields_ = [("nsections", ctypes.c_int32), ("pad", ctypes.c_int32), ("fixed_bytes", ctypes.c_uint64),
                ("total_bytes", ctypes.c_uint64), ("section", _ExportSection * 16)]


class Pipe:
    """ctypes mirror of odhip_pipe (include/daala_hip.h): F resident 4:2:0 pictures,
    one C call per step.  Buffers are read / written as numpy arrays."""

    def __init__(self, quant, frames, pic_w, pic_h, chroma_cfl=True, serial=False, device=0,
                 pvq_norm_lambda=0.147, price=False, fpr_bits=0, inter=False):
        """price=True: the choices price every candidate with od_pvq_rate's closed form on
        the device (odhip_pvq_*choose_priced_*), nothing for the host to do in a step.
        fpr_bits = 8 / 10 / 12: full-precision references, pictures of that bit depth (uint8 /
        int16), coded planes and reconstructions int16 at 12 bits."""
        L = lib()
        L.odhip_pipe_create.restype = ctypes.c_void_p
        L.odhip_pipe_theta_reruns.restype = ctypes.c_long
        L.odhip_pipe_price_reruns.restype = ctypes.c_long
        cfg = _PipeConfig(int(device), int(frames), int(pic_w), int(pic_h), int(bool(chroma_cfl)),
                          int(bool(serial)), int(bool(price)), int(fpr_bits), int(bool(inter)), 0,
                          float(pvq_norm_lambda),
                          ctypes.cast(ctypes.byref(quant.c), ctypes.c_void_p))
        self.h = L.odhip_pipe_create(ctypes.byref(cfg))
        if not self.h:
            raise DaalaHipError("odhip_pipe_create failed")
        self.frames, self.pic_w, self.pic_h = int(frames), int(pic_w), int(pic_h)
        self.W, self.H = (pic_w + 63) & ~63, (pic_h + 63) & ~63
        self.chroma_cfl = bool(chroma_cfl)
        self.fpr_bits = int(fpr_bits)
        self.inter = bool(inter)
        # test hooks the tests asked for (module state above) reach this pipe's contexts too
        global _pipes
        if _pipes is None:
            import weakref
            _pipes = weakref.WeakSet()
        _pipes.add(self)
        if _hooks["margin"] > 0 or _hooks["perturb"] or _hooks["tol"] != 1.0:
            _check(L.odhip_pipe_set_test_hooks(self._p(), ctypes.c_double(_hooks["margin"]),
                                               int(_hooks["perturb"]), ctypes.c_double(_hooks["tol"])),
                   "odhip_pipe_set_test_hooks")

    def _p(self):
        return ctypes.c_void_p(self.h)

    def destroy(self):
        if self.h:
            lib().odhip_pipe_destroy(self._p())
            self.h = None

    def set_pictures(self, luma, chroma):
        """luma uint8 [F, pic_h, pic_w], chroma uint8 [2F, pic_h/2, pic_w/2] (all Cb, then
        all Cr): numpy arrays (uploaded) or CUDA tensors (device copy)."""
        dev = hasattr(luma, "data_ptr")
        dt = np.int16 if self.fpr_bits > 8 else np.uint8
        if not dev:
            luma = np.ascontiguousarray(luma, dt)         # kept alive until the call returns
            chroma = np.ascontiguousarray(chroma, dt)
        pl = luma.data_ptr() if dev else luma.ctypes.data
        pc = chroma.data_ptr() if dev else chroma.ctypes.data
        assert tuple(luma.shape) == (self.frames, self.pic_h, self.pic_w)
        assert tuple(chroma.shape) == (2 * self.frames, self.pic_h // 2, self.pic_w // 2)
        _check(lib().odhip_pipe_set_pictures(self._p(), ctypes.c_void_p(pl), ctypes.c_void_p(pc),
                                             int(dev)), "odhip_pipe_set_pictures")

    def set_reference_pictures(self, luma, chroma):
        """Inter mode: the prediction pictures of the batch (numpy, same shapes and depth as
        set_pictures)."""
        dt = np.int16 if self.fpr_bits > 8 else np.uint8
        luma = np.ascontiguousarray(luma, dt)
        chroma = np.ascontiguousarray(chroma, dt)
        assert tuple(luma.shape) == (self.frames, self.pic_h, self.pic_w)
        assert tuple(chroma.shape) == (2 * self.frames, self.pic_h // 2, self.pic_w // 2)
        _check(lib().odhip_pipe_set_reference_pictures(self._p(), ctypes.c_void_p(luma.ctypes.data),
                                                       ctypes.c_void_p(chroma.ctypes.data), 0),
               "odhip_pipe_set_reference_pictures")

    def feed(self, luma, chroma):
        """The pictures of the NEXT step from host memory, copied while the enqueued steps
        compute (odhip_pipe_feed).  luma / chroma: pinned CPU torch tensors (asynchronous) or
        numpy arrays, same shapes as set_pictures; the caller keeps them alive and untouched
        until sync()."""
        dt = np.int16 if self.fpr_bits > 8 else np.uint8
        if hasattr(luma, "data_ptr"):
            assert not luma.is_cuda and luma.is_contiguous() and chroma.is_contiguous()
            pl, pc = luma.data_ptr(), chroma.data_ptr()
        else:
            assert luma.dtype == dt and chroma.dtype == dt and luma.flags["C_CONTIGUOUS"]
            pl, pc = luma.ctypes.data, chroma.ctypes.data
        assert tuple(luma.shape) == (self.frames, self.pic_h, self.pic_w)
        assert tuple(chroma.shape) == (2 * self.frames, self.pic_h // 2, self.pic_w // 2)
        _check(lib().odhip_pipe_feed(self._p(), ctypes.c_void_p(pl), ctypes.c_void_p(pc)), "odhip_pipe_feed")

    def export_bytes(self):
        """Bytes one step copies to the host with set_export (odhip_pipe_export_bytes)."""
        lib().odhip_pipe_export_bytes.restype = ctypes.c_size_t
        return int(lib().odhip_pipe_export_bytes(self._p()))

    def set_export(self, host):
        """host: a pinned CPU uint8 torch tensor of export_bytes() bytes (kept alive by the caller
        until set_export(None)), or None to stop: every following step leaves the record and the pulses of
        every band there, compacted on the device (odhip_pipe_set_export; decode_export reads it back)."""
        if host is None:
            _check(lib().odhip_pipe_set_export(self._p(), ctypes.c_void_p(0)), "odhip_pipe_set_export")
            return
        assert not host.is_cuda and host.is_contiguous() and host.numel() * host.element_size() >= self.export_bytes()
        _check(lib().odhip_pipe_set_export(self._p(), ctypes.c_void_p(host.data_ptr())), "odhip_pipe_set_export")

    def export_layout(self):
        """odhip_pipe_export_layout as a dict: nsections, fixed_bytes, total_bytes, sections[] of
        {bs, ngroups, cap_words, nrecords, records_off, group_base_off, stream_off}."""
        lay = _ExportLayout()
        _check(lib().odhip_pipe_export_layout(self._p(), ctypes.byref(lay)), "odhip_pipe_export_layout")
        return _layout_dict(lay)

    def export_stale(self):
        lib().odhip_pipe_export_stale.restype = ctypes.c_long
        return int(lib().odhip_pipe_export_stale(self._p()))

    def export_shipped_bytes(self, host):
        """Bytes of the export buffer `host` (numpy uint8 view) that crossed the bus for the step it holds: the
        fixed part plus the used prefix of every stream (rounded up to the 16-byte vectors the ship kernel
        moves)."""
        lay = self.export_layout()
        totals = host[:64].view(np.uint32)
        n = lay["fixed_bytes"]
        for i, sec in enumerate(lay["sections"]):
            n += (min(int(totals[i]), sec["cap_words"]) * 2 + 15) // 16 * 16
        return n

    def decode_export(self, host):
        """The export buffer `host` (numpy uint8) decoded back to what tests/_pipeline_check.gpu_decisions reads from
        the dense device buffers: {(set, level): (y int32 [B][len], band int32 [B][nb][4] = {coded gain index,
        itheta, max_theta, k}, coded bool [B][nb])}."""
        secs = decode_export_sections(host, self.export_layout())
        return {((0, i) if i < 5 else (1, i - 5)): v for i, v in enumerate(secs)}

    def step(self):
        _check(lib().odhip_pipe_step(self._p()), "odhip_pipe_step")

    def flush(self):
        _check(lib().odhip_pipe_flush(self._p()), "odhip_pipe_flush")

    def sync(self):
        _check(lib().odhip_pipe_sync(self._p()), "odhip_pipe_sync")

    def k_range(self):
        """Bands above ODHIP_PVQ_MAX_K counted at this pipe's syncs so far."""
        lib().odhip_pipe_k_range.restype = ctypes.c_long
        return int(lib().odhip_pipe_k_range(self._p()))

    def stage(self, name, parity=0):
        _check(lib().odhip_pipe_stage(self._p(), PIPE_STAGES.index(name), int(parity)),
               "odhip_pipe_stage(%s)" % name)

    def buffer(self, what, set_, level=0, parity=-1):
        ptr = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _check(lib().odhip_pipe_buffer(self._p(), int(what), int(set_), int(level), int(parity),
                                       ctypes.byref(ptr), ctypes.byref(n)), "odhip_pipe_buffer")
        return ptr.value, n.value

    def read(self, what, set_, level=0, parity=-1, dtype=np.uint8):
        """parity -1: the buffers of the last step() (0 when the pipe is driven stage by stage)."""
        ptr, n = self.buffer(what, set_, level, parity)
        out = np.empty(n, np.uint8)
        _check(lib().odhip_pipe_read(self._p(), out.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_void_p(ptr), ctypes.c_size_t(n)), "odhip_pipe_read")
        return out.view(dtype)

    def write(self, what, set_, level, data, parity=-1):
        ptr, n = self.buffer(what, set_, level, parity)
        data = np.ascontiguousarray(data)
        assert data.nbytes == n, (data.nbytes, n)
        _check(lib().odhip_pipe_write(self._p(), ctypes.c_void_p(ptr),
                                      data.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n)),
               "odhip_pipe_write")

    def record(self, enable):
        _check(lib().odhip_pipe_record(self._p(), int(bool(enable))), "odhip_pipe_record")

    def timings(self):
        """{stage: (average ms per launch group, groups)} of the recorded steps."""
        ms = (ctypes.c_double * len(PIPE_STAGES))()
        cnt = (ctypes.c_int * len(PIPE_STAGES))()
        _check(lib().odhip_pipe_timings(self._p(), ms, cnt), "odhip_pipe_timings")
        return {PIPE_STAGES[i]: (ms[i], cnt[i]) for i in range(len(PIPE_STAGES)) if cnt[i]}

    def search_timings(self, chroma, max_n=256):
        buf = (ctypes.c_float * max_n)()
        n = lib().odhip_pipe_search_timings(self._p(), int(bool(chroma)), buf, max_n)
        if n < 0:
            raise DaalaHipError("odhip_pipe_search_timings failed with code %d" % n)
        return [buf[i] for i in range(n)]

    def time_pyramid(self, n=10):
        ms = ctypes.c_double()
        _check(lib().odhip_pipe_time_pyramid(self._p(), int(n), ctypes.byref(ms)),
               "odhip_pipe_time_pyramid")
        return ms.value

    def time_stage(self, stage, n=10, parity=-1):
        """Average ms per launch group of one filter + DCT stage (a PIPE_STAGES name) run alone."""
        ms = ctypes.c_double()
        _check(lib().odhip_pipe_time_stage(self._p(), PIPE_STAGES.index(stage), int(parity), int(n),
                                           ctypes.byref(ms)), "odhip_pipe_time_stage")
        return ms.value

    def host_wait_ms(self):
        lib().odhip_pipe_host_wait_ms.restype = ctypes.c_double
        return float(lib().odhip_pipe_host_wait_ms(self._p()))

    def theta_reruns(self):
        return int(lib().odhip_pipe_theta_reruns(self._p()))

    def theta_listed(self):
        """Bands found inside the margin of the device acos so far (theta recomputed by the host's libm)."""
        lib().odhip_pipe_theta_listed.restype = ctypes.c_long
        return int(lib().odhip_pipe_theta_listed(self._p()))

    def price_reruns(self):
        """Priced choices re-decided with the host libm so far (price=True pipes)."""
        return int(lib().odhip_pipe_price_reruns(self._p()))

    def nblocks(self, set_, level):
        n = 4 << level
        dec = 1 if set_ else 0
        planes = self.frames * (2 if set_ else 1)
        return planes * ((self.W >> dec) // n) * ((self.H >> dec) // n)


# ---- od_compute_dist (block-size RDO distortion) ---------------------------------------
def compute_dist(x, y, bs, use_masking=1, flat_qm=0, coded_quantizer=40):
    """od_compute_dist of every (4 << bs)-square block of x (source) against y
    (reconstruction): int32 CUDA tensors [nplanes, h, w].  The device computes everything up
    to the libm call, the host applies pow (odhip_dist_finish).  Returns (dist float64 numpy
    [nplanes, h/n, w/n], parts float64 numpy [nplanes, h/8, w/8, 3])."""
    import torch
    _need(x, torch.int32, "x")
    _need(y, torch.int32, "y")
    assert x.shape == y.shape
    nplanes, h, w = x.shape
    n = 4 << bs
    parts = torch.empty((nplanes, h // 8, w // 8, 3), dtype=torch.float64, device=x.device)
    _check(lib().odhip_dist_parts(_p(parts), _p(x), _p(y), nplanes, w, h, int(bs), int(use_masking),
                                  int(flat_qm), _stream()), "odhip_dist_parts")
    hp = parts.cpu().numpy()
    dist = np.zeros((nplanes, h // n, w // n), np.float64)
    _check(lib().odhip_dist_finish(dist.ctypes.data_as(ctypes.c_void_p),
                                   hp.ctypes.data_as(ctypes.c_void_p), nplanes, w, h, int(bs),
                                   int(use_masking), int(flat_qm), int(coded_quantizer)),
           "odhip_dist_finish")
    return dist, hp


def set_price_tol_scale(scale):
    """Test hook: multiplies the margin inside which a priced choice is left to the host
    libm (odhip_ctx_set_test_hooks on the current context and every live Pipe); 1 restores it."""
    _hooks["tol"] = float(scale)
    _apply_hooks()
