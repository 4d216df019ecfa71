"""Decoder mirror (second half): inside the REAL reference decoder,
for every plane of every frame of a clip, odhip_inverse_partition - idct_2d of every leaf
block and od_postfilter_split of every split node at the frame's own (arbitrary) partition,
superblock-edge post-filter, pixel conversion - from the decoder's dequantised coefficient
plane and block-size map alone must give the pixels the decoder's own reconstruction gives
(src/decode.c:482-660, :988-996).  Entropy decoding, od_pvq_decode and the predictions stay
the reference's sequential host code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))


def _run(w, h, nframes, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interpose", "run_decode_check.py"),
                        str(w), str(h), str(nframes)], capture_output=True, text=True, timeout=1500, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
@pytest.mark.parametrize("size", [(64, 64, 2, 20), (180, 116, 2, 20), (320, 192, 2, 60), (256, 128, 1, 5)])
def test_gpu_reconstruction_equals_reference_decoder(size):
    w, h, nframes, quality = size
    plain = _run(w, h, nframes, QUALITY=quality, DECODE_CHECK=0)
    res = _run(w, h, nframes, QUALITY=quality)
    planes, pixels, bad = res["check"]
    assert planes == 3 * nframes, res["check"]
    W, H = (w + 63) & ~63, (h + 63) & ~63
    assert pixels == nframes * (W * H + 2 * (W // 2) * (H // 2))
    assert bad == 0, res["check"]
    # the hooks did not disturb the decoder, and the clip really decodes to something picture-like
    assert res["decoded"] == plain["decoded"]
    assert res["mean_abs_error_vs_source"] < 25


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_gpu_reconstruction_1080p_frame():
    res = _run(1920, 1080, 1, CONTENT="bench")
    planes, pixels, bad = res["check"]
    assert planes == 3 and bad == 0, res["check"]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_decoder_dering_from_batched_passes():
    """The decoder applies the transmitted deringing level superblock by superblock
    (src/decode.c: od_dering per plane); with odhip_dering_cache bound (encoder and decoder)
    every call is served from one launch per (plane, threshold), each served superblock is
    compared with the reference's od_dering, and the decoded clip is unchanged."""
    for (w, h, nframes, quality) in ((320, 192, 2, 20), (180, 116, 2, 60)):
        plain = _run(w, h, nframes, QUALITY=quality, DECODE_CHECK=0)
        res = _run(w, h, nframes, QUALITY=quality, DECODE_CHECK=0, DERING_CACHE=1)
        assert res["decoded"] == plain["decoded"], (w, h)
        launches, served = res["dering"]
        assert served > 0 and 0 < launches < served, res["dering"]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_decoder_synthesis_on_the_gpu():
    """od_pvq_decode's data-parallel half: every od_pvq_synthesis_partial call of the real decoder's
    pvq_decode_partition (src/pvq_decoder.c:77-89) - and of the encoder's pvq_theta, whose reconstruction
    the decoder must reproduce (src/pvq_encoder.c:631) - served by od_pvq_synthesis_partial_hip; the bitstream
    decodes to the same pictures, so every band's synthesis was bit-equal on both sides."""
    for (w, h, nframes, quality) in ((64, 64, 2, 20), (180, 116, 2, 45)):
        plain = _run(w, h, nframes, QUALITY=quality, DECODE_CHECK=0)
        res = _run(w, h, nframes, QUALITY=quality, DECODE_CHECK=0, SYNTHESIS=1)
        assert res["decoded"] == plain["decoded"], (w, h)
        assert res["synthesis_calls"] > 100, res
        # with the GPU reconstruction check on top: decoded coefficients from the GPU synthesis feed
        # odhip_inverse_partition and the pixels are the decoder's own
    res = _run(64, 64, 2, QUALITY=20, SYNTHESIS=1)
    planes, pixels, bad = res["check"]
    assert planes == 6 and bad == 0 and res["synthesis_calls"] > 100, res
