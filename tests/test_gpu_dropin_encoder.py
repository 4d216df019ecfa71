"""GPU drop-in test: the UNMODIFIED reference encoder (oracle/_ref, built from
the reference's own sources) with libdaalahip's od_bin_{f,i}dctNxN_hip bound into
its od_state_opt_vtbl.fdct_2d/idct_2d slots produces byte-identical packets.

Skipped when oracle/_ref is not present on the box (it is a prebuilt file that
travels with the snapshot; /root/reference itself is never read here)."""
import ctypes
import os

import numpy as np
import pytest

from _libs import P, ref, synth_frame

pytestmark = pytest.mark.gpu


def _encode(r, frames, w, h, nframes, quality=20, complexity=7, count=0):
    out = np.zeros(4 << 20, np.uint8)
    sizes = (ctypes.c_long * 64)()
    n = r.ref_encode_yuv420(P(frames), w, h, nframes, quality, complexity, count, P(out),
                            ctypes.c_long(out.size), sizes)
    assert n == nframes, n
    total = sum(sizes[i] for i in range(n))
    return bytes(out[:total]), [sizes[i] for i in range(n)]


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_reference_encoder_with_hip_transforms_is_byte_identical():
    import torch
    import daala_amd
    assert torch.cuda.is_available()
    daala_amd.init(0)
    L = daala_amd.lib()
    r = ref()
    w = h = 64
    nframes = 2
    frames = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=7, phase=5 * f)])
                             for f in range(nframes)]).astype(np.uint8)
    r.ref_set_external_dct_vtbl(None, None)
    want, sizes_c = _encode(r, frames, w, h, nframes)
    fd = (ctypes.c_void_p * 5)()
    idt = (ctypes.c_void_p * 5)()
    L.odhip_install_dct_vtbl(fd, idt)
    assert all(fd[i] and idt[i] for i in range(5))
    r.ref_set_external_dct_vtbl(fd, idt)
    try:
        got, sizes_h = _encode(r, frames, w, h, nframes, count=0)
    finally:
        r.ref_set_external_dct_vtbl(None, None)
    assert sizes_h == sizes_c
    assert got == want, "packets differ between the C and the HIP transform tables"
    assert len(want) > 200


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_reference_encoder_with_interposed_filters_and_pvq_search():
    """Load-time override (INTEGRATION.md sections 2-3): the reference's own symbols
    od_prefilter_split / od_postfilter_split / od_apply_{pre,post}filter_frame_sbs
    and pvq_search_rdo_double are bound to libdaalahip inside the UNMODIFIED
    reference encoder; packets stay byte-identical and the call counters show the
    GPU path really ran."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "interpose", "libinterpose.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so,
                        os.path.join(here, "interpose", "interpose.c"),
                        "-L" + os.path.join(here, "..", "daala_amd", "lib"), "-ldaalahip",
                        "-Wl,-rpath,$ORIGIN/../../daala_amd/lib", "-ldl"], check=True)
    def run(mode, *size, env=None):
        e = dict(os.environ)
        e.update(env or {})
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"),
                            str(mode)] + [str(v) for v in size], capture_output=True, text=True,
                           timeout=900, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    plain = run(0)
    bound = run(1)
    assert plain["calls"] == [0] * 6
    assert all(c > 0 for c in bound["calls"]), bound["calls"]
    assert bound["sizes"] == plain["sizes"]
    assert bound["packets"] == plain["packets"], "packets differ with the HIP surfaces bound"
    # Frame cache: a 180x116 picture (padded to 192x128 by the encoder, so the
    # picture-edge gating of the split filters matters).  Every fdct_2d call of
    # both RDO passes is served from ONE batched GPU pyramid per plane; with
    # ODHIP_CACHE_CHECK=1 every hit is also verified against the per-call path.
    plain2 = run(0, 180, 116, env={"NFRAMES": "1"})
    cached = run(2, 180, 116, env={"NFRAMES": "1", "ODHIP_CACHE_CHECK": "1"})
    assert cached["sizes"] == plain2["sizes"]
    assert cached["packets"] == plain2["packets"], "packets differ with the frame cache"
    hits, misses = cached["cache"]
    assert hits > 1000 and misses == 0, cached["cache"]


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_batched_band_stage_behind_the_real_encoder_is_byte_identical():
    """Frame integration with host pricing: the unmodified
    reference encoder, -z 7 (od_pvq_rate prices with the LIVE adaptive entropy coder),
    with one batched GPU pass per keyframe - pyramid of every plane + the PVQ band
    stage of every luma block of every level - serving its fdct_2d calls and every
    pvq_theta call whose reference vector is null; the host prices, chooses and
    synthesises.  Packets must be byte-identical to plain C, for several picture
    sizes / qualities, with every served band checked against the band the encoder
    presents (ODHIP_CACHE_CHECK=1)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))

    def run(mode, w, h, **env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"),
                            str(mode), str(w), str(h)], capture_output=True, text=True, timeout=1500, env=e)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    for (w, h, nframes, quality, complexity) in ((64, 64, 2, 20, 7), (180, 116, 1, 20, 7),
                                                 (320, 192, 2, 40, 7), (256, 128, 1, 5, 2)):
        env = dict(NFRAMES=nframes, QUALITY=quality, COMPLEXITY=complexity,
                   ODHIP_INTERPOSE_PASSTHROUGH=1)
        plain = run(0, w, h, **env)
        # ODHIP_RATE_CHECK: at the default complexity the served bands are priced by the
        # library's batched host routine (odhip_pvq_rate_batch16); every price is compared
        # with the reference's own od_pvq_rate on the live context inside the encoder
        batch = run(3, w, h, ODHIP_CACHE_CHECK=1, ODHIP_RATE_CHECK=1, **env)
        assert batch["sizes"] == plain["sizes"], (w, h)
        assert batch["packets"] == plain["packets"], (w, h, quality, complexity)
        served, with_ref, other, searches = batch["theta"]
        assert served > 100 and searches > 0, batch["theta"]
        assert other == 0, batch["theta"]


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_batched_band_stage_1080p_frame_byte_identical():
    """The same at BASELINE configs[1]'s size: one 1920x1080 keyframe of the bench
    generator, -v 20 -z 7."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in (0, 3):
        e = dict(os.environ)
        e.update(NFRAMES="1", CONTENT="bench", ODHIP_INTERPOSE_PASSTHROUGH="1")
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"),
                            str(mode), "1920", "1080"], capture_output=True, text=True, timeout=1500, env=e)
        assert p.returncode == 0, p.stderr[-3000:]
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res[3]["sizes"] == res[0]["sizes"]
    assert res[3]["packets"] == res[0]["packets"]
    served, with_ref, other, searches = res[3]["theta"]
    assert served > 50000 and other == 0, res[3]["theta"]
    print("1080p keyframe: %d bands from the batch (%d searches), %d bands with a neighbour "
          "prediction left to the reference; encode %.2f s (of which the batched GPU pass incl. PCIe "
          "both ways %.1f ms) vs %.2f s plain C"
          % (served, searches, with_ref, res[3]["encode_seconds"], res[3]["gpu_batch_ms"],
             res[0]["encode_seconds"]))


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_dering_level_search_from_batched_passes_is_byte_identical():
    """SURVEY.md 8(f) rank 1 at the frame level: every od_dering call of the encoder's
    deringing level search (five luma levels per superblock, then the three planes with
    the level chosen, src/encode.c:2697-2832) served by odhip_dering_cache - one launch per
    (plane, threshold) pair filters every superblock of the plane.  Packets must be
    byte-identical; with ODHIP_DERING_CHECK every served superblock and direction array is
    also compared with the reference's own od_dering."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))

    def run(mode, w, h, **env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"),
                            str(mode), str(w), str(h)], capture_output=True, text=True, timeout=1500, env=e)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    for (w, h, nframes, quality) in ((320, 192, 2, 20), (180, 116, 2, 40), (256, 128, 1, 5)):
        env = dict(NFRAMES=nframes, QUALITY=quality, COMPLEXITY=7, ODHIP_INTERPOSE_PASSTHROUGH=1)
        plain = run(0, w, h, **env)
        cached = run(1, w, h, ODHIP_INTERPOSE_DERING_CACHE=1, ODHIP_DERING_CHECK=1, **env)
        assert cached["sizes"] == plain["sizes"], (w, h)
        assert cached["packets"] == plain["packets"], (w, h, quality)
        launches, served = cached["dering"]
        assert served > 0 and launches > 0, cached["dering"]
        # one launch per (plane, threshold) pair per frame, at most 5 + 2 * 5
        assert launches <= 15 * nframes, cached["dering"]
        assert served == cached["calls"][5], (cached["dering"], cached["calls"])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle",
                                                    "_ref", "libdaalaref_4x4.so")),
                    reason="oracle/_ref/libdaalaref_4x4.so not present")
def test_configs0_reference_built_for_4x4_blocks_only():
    """BASELINE configs[0]: 64x64 4:2:0, two frames, the reference BUILT with block sizes limited
    to 4x4 (src/internal.h:100-101; oracle/_ref/libdaalaref_4x4.so).  With the HIP surfaces bound -
    the transforms through the vtbl slots, the filter drivers and the PVQ search by symbol - its
    packets are byte-identical to its own plain-C packets, and differ from the default build's
    (the limit really changes what is coded)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))

    def run(mode, **env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"), str(mode)],
                           capture_output=True, text=True, timeout=900, env=e)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    default = run(0)
    plain = run(0, REF_LIB="libdaalaref_4x4.so")
    bound = run(1, REF_LIB="libdaalaref_4x4.so", ODHIP_INTERPOSE_VTBL=1)
    assert plain["sizes"] != default["sizes"]
    assert bound["sizes"] == plain["sizes"]
    assert bound["packets"] == plain["packets"], "4x4-only build: packets differ with the HIP surfaces bound"
    # every superblock is split all the way down (the split filters of all four levels run),
    # only 4x4 blocks are coded
    assert all(c > 0 for c in bound["calls"]), bound["calls"]


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                    "oracle", "_ref", "libdaalaref_disthip.so")),
                    reason="oracle/_ref/libdaalaref_disthip.so not present")
def test_od_compute_dist_bound_behind_the_block_size_rdo_is_byte_identical():
    """SURVEY.md 8(f) rank 2 bound where the reference calls it: a build of the reference whose
    file-static od_compute_dist (src/encode.c:1170; call sites :1418-1421, :1797-1798: every
    split / no-split decision of the block-size RDO) forwards to the library's per-call surface
    od_compute_dist_hip - device parts + host pow.  Its doubles feed `dist + lambda*rate`
    comparisons, so the packets are byte-identical to plain C only if every one of them is the
    reference's bit for bit."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for lib in ("libdaalaref.so", "libdaalaref_disthip.so"):
        e = dict(os.environ)
        e.update(NFRAMES="2", QUALITY="20", COMPLEXITY="7", REF_LIB=lib)
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"), "0", "192", "128"],
                           capture_output=True, text=True, timeout=1500, env=e)
        assert p.returncode == 0, p.stderr[-3000:]
        res[lib] = json.loads(p.stdout.strip().splitlines()[-1])
    a, b = res["libdaalaref.so"], res["libdaalaref_disthip.so"]
    assert b["dist_hip_calls"] and b["dist_hip_calls"] > 500 and a["dist_hip_calls"] is None
    assert b["sizes"] == a["sizes"] and b["packets"] == a["packets"]
    print("od_compute_dist bound: %d calls through od_compute_dist_hip, %.1f us per call incl. PCIe both ways "
          "(encode %.2f s vs %.2f s plain C)"
          % (b["dist_hip_calls"], 1e6 * (b["encode_seconds"] - a["encode_seconds"]) / b["dist_hip_calls"],
             b["encode_seconds"], a["encode_seconds"]))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref",
                                                    "libdaalaref_distglue.so")),
                    reason="oracle/_ref/libdaalaref_distglue.so not built")
def test_level_search_distortions_from_the_batched_passes():
    """The deringing level search's od_compute_dist calls (src/encode.c:2776-2801) served from the
    same batched passes as its od_dering calls (odhip_dering_cache_set_source / _dist behind the
    shim's odhip_glue_compute_dist, bound by the DISTGLUE lines of oracle/Makefile): packets stay
    byte-identical to the plain C encoder's, five of the six calls per searched superblock are served,
    and with ODHIP_DIST_CHECK every served value was compared bit for bit with the C function's."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))

    def run(mode, w, h, **env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        p = subprocess.run([sys.executable, os.path.join(here, "interpose", "run_interposed.py"), str(mode), str(w),
                            str(h)], capture_output=True, text=True, timeout=1500, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    for (w, h, nframes) in ((320, 192, 2), (1920, 1080, 1)):
        env = dict(NFRAMES=nframes, CONTENT="bench", ODHIP_INTERPOSE_PASSTHROUGH=1, ODHIP_CACHE_FDCT_ONLY=1,
                   REF_LIB="libdaalaref_distglue.so")
        plain = run(0, w, h, **env)
        bound = run(3, w, h, ODHIP_INTERPOSE_DERING_CACHE=1, ODHIP_INTERPOSE_DIST_CACHE=1, ODHIP_DIST_CHECK=1, **env)
        assert bound["sizes"] == plain["sizes"] and bound["packets"] == plain["packets"]
        dc = bound["dist_cache"]
        assert dc["served"] > 0 and dc["reference_side"][0] == dc["served"], dc
        # five served calls per searched superblock (the sixth compares the unfiltered reconstruction)
        assert dc["served"] % 5 == 0, dc
        nsb = ((w + 63) // 64) * ((h + 63) // 64) * nframes
        assert dc["served"] >= 5 * nsb // 2, (dc, nsb)
