"""Frame-sharded all-intra encode with the batched GPU stage behind the REAL reference
encoder - TEST / CHECK INFRASTRUCTURE (the entropy coder, rate pricing and packet
writer are the reference's own host C, which in this repository exists only as the
test build oracle/_ref).  Used by tests/test_gpu_shard_encode.py and by the
`sharded_encode_check` leg of bench.py --gpus N.

One process per rank; every rank
  - loads libdaalahip, then tests/interpose (binds the frame cache: one batched pyramid
    per plane + the batched PVQ band stage of keyframe luma behind pvq_theta), then the
    reference encoder,
  - encodes the frames it owns (i mod world), each seeded with its GLOBAL frame number
    (ref_encode_yuv420_shard: the display frame number is the one piece of per-frame
    state that reaches the packet bytes, src/encode.c:3043),
and the packets are gathered to rank 0 with daala_amd.shard.gather_packets (RCCL on
GPUs, gloo in the CPU tests)."""
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_state = {}


def load_batched_encoder(w, h, device=0):
    """(reference library, interposer) with the batched GPU stage bound; once per process,
    and only in a process that has not loaded the reference library before."""
    if "r" in _state:
        _state["ipo"].odhip_interpose_enable_cache(w, h)
        return _state["r"], _state["ipo"]
    with open("/proc/self/maps") as f:
        if "libdaalaref.so" in f.read():
            # ctypes binds with RTLD_NOW: a reference library loaded earlier has its calls
            # resolved to its own definitions already and cannot be interposed any more
            raise RuntimeError("libdaalaref.so was loaded before the interposer in this process")
    os.environ["ODHIP_INTERPOSE_PASSTHROUGH"] = "1"    # per-call surfaces stay the reference's
    hip = ctypes.CDLL(os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so"), mode=ctypes.RTLD_GLOBAL)
    assert hip.odhip_init(int(device)) == 0
    ipo = ctypes.CDLL(os.path.join(ROOT, "tests", "interpose", "libinterpose.so"), mode=ctypes.RTLD_GLOBAL)
    r = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))
    ipo.odhip_interpose_set_reference(ctypes.c_void_p(r._handle))
    ipo.odhip_interpose_enable_cache(w, h)
    ipo.odhip_interpose_enable_bands()
    fd = (ctypes.c_void_p * 5)()
    idt = (ctypes.c_void_p * 5)()
    hip.odhip_install_cached_dct_vtbl(fd, idt)
    r.ref_set_external_dct_vtbl(fd, None)      # fdct_2d from the batch; idct_2d stays C
    _state.update(r=r, ipo=ipo, hip=hip)
    return r, ipo


def frame_yuv(index, w, h):
    """Frame `index` of the bench generator, cropped to w x h, planar 4:2:0."""
    import bench
    pl = bench.picture_planes(bench.synth_frame_np(index, 1234))
    return np.concatenate([pl[0][:h, :w].ravel(), pl[1][:h // 2, :w // 2].ravel(),
                           pl[2][:h // 2, :w // 2].ravel()]).astype(np.uint8)


def encode_owned(r, indices, w, h, quality=20, complexity=7):
    """{global frame index: packet bytes} of the frames in `indices` (bench generator)."""
    return encode_frames(r, indices, [frame_yuv(i, w, h) for i in indices], w, h, quality, complexity)


def encode_frames(r, indices, yuv, w, h, quality=20, complexity=7):
    """{global frame index: packet bytes}: yuv[j] (planar 4:2:0 bytes) is frame indices[j] of
    the whole sequence."""
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


def band_stats(ipo):
    arr = (ctypes.c_long * 4).in_dll(ipo, "odhip_interposed_theta")
    return [arr[i] for i in range(4)]


def digest(packets):
    h = hashlib.sha256()
    for p in packets:
        h.update(len(p).to_bytes(8, "little"))
        h.update(p)
    return h.hexdigest()


def sequential_digest(nframes, w, h, quality=20, complexity=7):
    """The plain C encoder on frames 0..nframes-1, in a child process (this process has
    the batched stage bound)."""
    import json
    import subprocess
    e = dict(os.environ)
    e.pop("ODHIP_INTERPOSE_PASSTHROUGH", None)
    e.update(NFRAMES=str(nframes), CONTENT="bench", QUALITY=str(quality), COMPLEXITY=str(complexity),
             PACKET_DIGEST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interpose", "run_interposed.py"), "0",
                        str(w), str(h)], capture_output=True, text=True, timeout=1800, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])["digest"]
