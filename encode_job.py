#!/usr/bin/env python3
"""encode_job.py - the frame-sharded all-intra encode (BASELINE configs[4]) above the shim.

The host of this job is the reference's own encoder: entropy coding, rate pricing on the live
adaptive state, block-size RDO and the packet writer are its sequential C, which this repository
does not re-implement (out of scope) and has only as the build of the reference's
sources under oracle/_ref.  What is bound behind it, through shim/libdaalahipglue.so with explicit
configuration calls (shim/daala_hip_glue.h), is libdaalahip: one batched pyramid per plane, the
PVQ band stage of keyframe luma with batched speed-0 pricing, the deringing level search from
batched passes.

As a module: load_batched_encoder / encode_frames / ... (used by bench.py --encode-frames, by
bench.py's sharded_encode_check and by the tests).  As a program it is ONE ENCODER PROCESS of a
job (bench.py --encode-frames N --procs-per-gpu P starts P of them per rank, all sharing the
rank's GPU; all-intra frames are independent, src/encode.c:303-308,3029,3080):

    python encode_job.py --worker --y4m FILE --frames N --stride S --offset O --device D --out OUT

encodes frames O, O + S, O + 2S, ... (< N) of FILE, each seeded with its GLOBAL frame number
(the display frame number reaches the packet bytes, src/encode.c:3043), prints READY after its
untimed first frame, waits for a line on stdin, encodes, writes OUT (npz: indices, sizes, bytes)
and one JSON line of statistics."""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The host encoder.  libdaalaref_distglue.so = the same build with the shim's hook in front of the file-static
# od_compute_dist (oracle/Makefile DISTGLUE: the level search's distortions from the batched passes); without it
# (or with ODHIP_REFERENCE_LIB pointing at another build) that binding is simply not taken.
_DISTGLUE = os.path.join(ROOT, "oracle", "_ref", "libdaalaref_distglue.so")
REFERENCE_LIB = os.environ.get("ODHIP_REFERENCE_LIB", _DISTGLUE if os.path.exists(_DISTGLUE)
                               else os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))
GLUE_LIB = os.path.join(ROOT, "shim", "libdaalahipglue.so")

_state = {}


class GlueConfig(ctypes.Structure):
    """odhip_glue_config, shim/daala_hip_glue.h."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "device", "bind_filters", "bind_search", "bind_dering", "bind_dct_vtbl", "frame_cache", "band_cache",
        "dering_cache", "pic_w", "pic_h", "check_rates", "check_dering", "gpu_pass_lock", "dist_cache", "check_dist",
        "bind_synthesis")]


class GlueStats(ctypes.Structure):
    """odhip_glue_stats, shim/daala_hip_glue.h."""
    _fields_ = [("calls", ctypes.c_long * 6), ("theta", ctypes.c_long * 4), ("fdct_hits", ctypes.c_long),
                ("fdct_misses", ctypes.c_long), ("band_hits", ctypes.c_long), ("band_misses", ctypes.c_long),
                ("dering_launches", ctypes.c_long), ("dering_served", ctypes.c_long), ("batch_ms", ctypes.c_double),
                ("dering_ms", ctypes.c_double), ("theta_ms", ctypes.c_double), ("dist_served", ctypes.c_long),
                ("dist_left", ctypes.c_long)]


def reference_available():
    return os.path.exists(REFERENCE_LIB) and os.path.exists(GLUE_LIB)


def load_batched_encoder(w, h, device=0, dering_cache=True, check_rates=False, gpu_pass_lock=False, dist_cache=True,
                         check_dist=False, check_dering=False):
    """(reference encoder library, glue library) with the batched GPU stage bound; once per
    process, and only in a process that has not loaded the reference library before."""
    if "r" in _state:
        _state["glue"].odhip_glue_enable_frame_cache(int(w), int(h))
        return _state["r"], _state["glue"]
    with open("/proc/self/maps") as f:
        if os.path.basename(REFERENCE_LIB) in f.read():
            # ctypes binds with RTLD_NOW: a reference library loaded earlier has its calls resolved
            # to its own definitions already and cannot be bound to the shim any more
            raise RuntimeError("%s was loaded before the shim in this process" % REFERENCE_LIB)
    hip = ctypes.CDLL(os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so"), mode=ctypes.RTLD_GLOBAL)
    glue = ctypes.CDLL(GLUE_LIB, mode=ctypes.RTLD_GLOBAL)
    cfg = GlueConfig()
    glue.odhip_glue_default_config(ctypes.byref(cfg))
    # the per-call surfaces stay the reference's own C (one GPU round trip per 4-tap filter call
    # would dominate a 1080p frame); the batched bindings carry the frame
    cfg.device = int(device)
    cfg.bind_filters = cfg.bind_search = cfg.bind_dering = cfg.bind_dct_vtbl = 0
    cfg.frame_cache = cfg.band_cache = 1
    cfg.dering_cache = int(bool(dering_cache))
    cfg.pic_w, cfg.pic_h = int(w), int(h)
    cfg.check_rates = int(bool(check_rates))
    cfg.check_dering = int(bool(check_dering))
    cfg.gpu_pass_lock = int(bool(gpu_pass_lock))
    cfg.dist_cache = int(bool(dist_cache and dering_cache))
    cfg.check_dist = int(bool(check_dist))
    rc = glue.odhip_glue_configure(ctypes.byref(cfg))
    if rc != 0:
        raise RuntimeError("odhip_glue_configure failed with code %d (no CPU fallback exists)" % rc)
    r = ctypes.CDLL(REFERENCE_LIB)
    glue.odhip_glue_set_reference(ctypes.c_void_p(r._handle))
    fd = (ctypes.c_void_p * 5)()
    idt = (ctypes.c_void_p * 5)()
    glue.odhip_glue_cached_dct_vtbl(fd, idt)
    r.ref_set_external_dct_vtbl(fd, None)      # fdct_2d from the batch; idct_2d stays C
    _state.update(r=r, glue=glue, hip=hip)
    return r, glue


def glue_stats(glue):
    st = GlueStats()
    glue.odhip_glue_get_stats(ctypes.byref(st))
    return st


def band_stats(glue):
    """[served from the batch, left to the reference (band has a reference vector), left (other),
    K-pulse searches the batch saved]."""
    st = glue_stats(glue)
    return [st.theta[i] for i in range(4)]


def frame_yuv(index, w, h):
    """Frame `index` of the bench generator, cropped to w x h, planar 4:2:0."""
    import bench
    pl = bench.picture_planes(bench.synth_frame_np(index, 1234))
    return np.concatenate([pl[0][:h, :w].ravel(), pl[1][:h // 2, :w // 2].ravel(),
                           pl[2][:h // 2, :w // 2].ravel()]).astype(np.uint8)


def encode_frames(r, indices, yuv, w, h, quality=20, complexity=7):
    """{global frame index: packet bytes}: yuv[j] (planar 4:2:0 bytes) is frame indices[j] of the
    whole sequence (ref_encode_yuv420_shard, oracle/ref_encoder_driver.c: the reference's public
    encoder API with the display frame number seeded per frame)."""
    if not indices:
        return {}
    frames = np.concatenate([np.ascontiguousarray(f, np.uint8).ravel() for f in yuv])
    idx = (ctypes.c_int * len(indices))(*indices)
    out = np.zeros(max(8 << 20, len(indices) * (w * h)), np.uint8)
    sizes = (ctypes.c_long * len(indices))()
    n = r.ref_encode_yuv420_shard(frames.ctypes.data_as(ctypes.c_void_p), w, h, len(indices), idx,
                                  quality, complexity, out.ctypes.data_as(ctypes.c_void_p),
                                  ctypes.c_long(out.size), sizes)
    assert n == len(indices), n
    local = {}
    pos = 0
    for j, i in enumerate(indices):
        local[i] = bytes(out[pos:pos + sizes[j]])
        pos += sizes[j]
    return local


def encode_owned(r, indices, w, h, quality=20, complexity=7):
    """{global frame index: packet bytes} of the frames in `indices` (bench generator)."""
    return encode_frames(r, indices, [frame_yuv(i, w, h) for i in indices], w, h, quality, complexity)


def digest(packets):
    h = hashlib.sha256()
    for p in packets:
        h.update(len(p).to_bytes(8, "little"))
        h.update(p)
    return h.hexdigest()


def read_y4m_frames(D, path, offset, stride, limit):
    """Frames offset, offset + stride, ... (< limit) of a Y4M file through the library's reader
    (odhip_y4m_open / _read / _skip): ({global index: planar 4:2:0 bytes}, w, h, frames in file)."""
    L = D.lib()
    L.odhip_y4m_open.restype = ctypes.c_void_p
    w, h, fn, fd, err = (ctypes.c_int() for _ in range(5))
    y = L.odhip_y4m_open(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(fn), ctypes.byref(fd),
                         ctypes.byref(err))
    if not y:
        raise SystemExit("cannot read %s as progressive 8-bit 4:2:0 YUV4MPEG2 (code %d)" % (path, err.value))
    w, h = w.value, h.value
    cw, chh = (w + 1) >> 1, (h + 1) >> 1
    out = {}
    i = 0
    while i < limit:
        if i % stride == offset:
            fr = np.empty(w * h + 2 * cw * chh, np.uint8)
            rc = L.odhip_y4m_read(ctypes.c_void_p(y), fr[:w * h].ctypes.data_as(ctypes.c_void_p),
                                  fr[w * h:w * h + cw * chh].ctypes.data_as(ctypes.c_void_p),
                                  fr[w * h + cw * chh:].ctypes.data_as(ctypes.c_void_p))
            if rc == 1:
                out[i] = fr
        else:
            rc = L.odhip_y4m_skip(ctypes.c_void_p(y))
        if rc == 0:
            break
        if rc < 0:
            raise SystemExit("%s: loss of framing at frame %d (code %d)" % (path, i, rc))
        i += 1
    L.odhip_y4m_close(ctypes.c_void_p(y))
    return out, w, h, i


def read_y4m_frames_set(D, path, wanted, limit):
    """As read_y4m_frames, for an arbitrary set of global frame indices (< limit)."""
    L = D.lib()
    L.odhip_y4m_open.restype = ctypes.c_void_p
    w, h, fn, fd, err = (ctypes.c_int() for _ in range(5))
    y = L.odhip_y4m_open(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(fn), ctypes.byref(fd),
                         ctypes.byref(err))
    if not y:
        raise SystemExit("cannot read %s as progressive 8-bit 4:2:0 YUV4MPEG2 (code %d)" % (path, err.value))
    w, h = w.value, h.value
    cw, chh = (w + 1) >> 1, (h + 1) >> 1
    out = {}
    i = 0
    while i < limit:
        if i in wanted:
            fr = np.empty(w * h + 2 * cw * chh, np.uint8)
            rc = L.odhip_y4m_read(ctypes.c_void_p(y), fr[:w * h].ctypes.data_as(ctypes.c_void_p),
                                  fr[w * h:w * h + cw * chh].ctypes.data_as(ctypes.c_void_p),
                                  fr[w * h + cw * chh:].ctypes.data_as(ctypes.c_void_p))
            if rc == 1:
                out[i] = fr
        else:
            rc = L.odhip_y4m_skip(ctypes.c_void_p(y))
        if rc == 0:
            break
        if rc < 0:
            raise SystemExit("%s: loss of framing at frame %d (code %d)" % (path, i, rc))
        i += 1
    L.odhip_y4m_close(ctypes.c_void_p(y))
    return out, w, h, i


def worker(args):
    """One encoder process of the job (see the module docstring)."""
    if args.core >= 0:
        try:
            os.sched_setaffinity(0, {args.core})
        except (AttributeError, OSError):
            pass
    import daala_amd as D
    frames, w, h, total = read_y4m_frames(D, args.y4m, args.offset, args.stride, args.frames)
    owned = sorted(frames)
    # --selfcheck: the shim's own cross-checks against the reference's C definitions (1 = every batched
    # od_pvq_rate, 2 = every served od_dering superblock, 4 = every served od_compute_dist; a difference aborts)
    r, glue = load_batched_encoder(w, h, device=args.device, gpu_pass_lock=args.gpu_lock,
                                   check_rates=bool(args.selfcheck & 1), check_dering=bool(args.selfcheck & 2),
                                   check_dist=bool(args.selfcheck & 4))
    # T encoder contexts in T host threads of this process (the C call releases the GIL; the shim's
    # state is per thread, the HIP context is shared): thread t takes every T-th of the owned frames
    T = max(1, min(args.threads, len(owned)))     # no thread without a frame
    parts = [owned[t::T] for t in range(T)]
    import threading
    go = threading.Event()
    warmed = threading.Barrier(T + 1)
    outs = [{} for _ in parts]
    errors = []

    def run(idx, out):
        # the same thread codes its untimed first frame (allocations, first-use tables, its own frame /
        # band / dering caches in the shim) and, after the start signal, its share of the job
        try:
            if idx:
                encode_frames(r, idx[:1], [frames[idx[0]]], w, h)
            warmed.wait()
            go.wait()
            out.update(encode_frames(r, idx, [frames[i] for i in idx], w, h))
        except Exception as e:      # noqa: BLE001 - reported by the assertion below
            errors.append(repr(e))
            try:
                warmed.abort()
            except Exception:       # noqa: BLE001
                pass

    ths = [threading.Thread(target=run, args=(p_, o)) for p_, o in zip(parts, outs)]
    for th in ths:
        th.start()
    warmed.wait()
    st0 = glue_stats(glue)
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    go.set()
    for th in ths:
        th.join()
    local = {}
    for o in outs:
        local.update(o)
    dt = time.perf_counter() - t0
    assert not errors and sorted(local) == owned, "an encoder thread failed: %s" % errors
    st = glue_stats(glue)
    sizes = np.array([len(local[i]) for i in owned], np.int64)
    blob = np.frombuffer(b"".join(local[i] for i in owned), np.uint8) if owned else np.zeros(0, np.uint8)
    np.savez(args.out, indices=np.array(owned, np.int64), sizes=sizes, bytes=blob)
    print(json.dumps({"frames": len(owned), "seconds": dt, "frames_in_file": total, "w": w, "h": h, "threads": T,
                      "bands_from_batch": st.theta[0] - st0.theta[0],
                      "bands_left_to_reference": st.theta[1] + st.theta[2] - st0.theta[1] - st0.theta[2],
                      "searches_saved": st.theta[3] - st0.theta[3],
                      "batch_ms": st.batch_ms - st0.batch_ms, "dering_ms": st.dering_ms - st0.dering_ms,
                      "theta_ms": st.theta_ms - st0.theta_ms,
                      "dist_served": st.dist_served - st0.dist_served, "dist_left": st.dist_left - st0.dist_left,
                      "dering_served": st.dering_served - st0.dering_served}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--y4m")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--offset", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--core", type=int, default=-1)
    ap.add_argument("--out")
    ap.add_argument("--gpu-lock", type=int, default=0)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--selfcheck", type=int, default=0)
    args = ap.parse_args()
    if not args.worker:
        ap.error("encode_job.py is a module; as a program it only runs as --worker (see bench.py --encode-frames)")
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
