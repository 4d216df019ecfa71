"""BASELINE configs[0] / SURVEY 8(b) last row: the reference's
examples/encoder_example.c, UNMODIFIED, built against the reference library
(oracle/_ref, compiled from the reference's own sources) and the Ogg framing
stand-in of tests/oggshim.

CPU (plumbing, no GPU): a synthetic 64x64 Y4M goes in, a well-formed Ogg
stream comes out (page CRCs, sequence numbers, flags checked) and its data
packets equal the packets of the headless public-API driver
(oracle/ref_encoder_driver.c) on the same frames.

GPU: the same binary with tests/interpose/libinterpose.so in LD_PRELOAD - the
load-time form of INTEGRATION.md sections 1-3: libdaalahip's transforms in
od_state.opt_vtbl, its lapping-filter drivers, PVQ search and deringing bound to
the reference's own symbol names - writes a byte-identical file.

Skipped when oracle/_ref is absent (prebuilt files that travel with the
snapshot; /root/reference is never read here)."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from _libs import P, ref, synth_frame

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(ROOT, "oracle", "_ref", "encoder_example")
needs_exe = pytest.mark.skipif(not os.path.exists(EXE) or ref() is None,
                               reason="oracle/_ref/encoder_example not present")


def write_y4m(path, w, h, nframes, seed=7):
    frames = []
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (w, h))
        for fr in range(nframes):
            f.write(b"FRAME\n")
            planes = [p.astype(np.uint8) for p in synth_frame(w, h, seed=seed, phase=5 * fr)]
            for p in planes:
                f.write(p.tobytes())
            frames.append(np.concatenate([p.ravel() for p in planes]))
    return np.concatenate(frames)


def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        t.append(r)
    return t


_CRC = _crc_table()


def ogg_crc(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFFFF) ^ _CRC[((c >> 24) & 0xFF) ^ b]
    return c


def ogg_packets(data):
    """RFC 3533 demultiplexer for a single logical stream: checks every page,
    returns (packets, granulepos of each page)."""
    pos = 0
    packets = []
    cur = bytearray()
    pageno = 0
    granules = []
    serial = None
    last_flags = 0
    while pos < len(data):
        assert data[pos:pos + 4] == b"OggS", pos
        version, flags = data[pos + 4], data[pos + 5]
        granule, ser, seq, crc, nseg = struct.unpack_from("<qIIIB", data, pos + 6)
        assert version == 0
        lacing = data[pos + 27:pos + 27 + nseg]
        body_len = sum(lacing)
        page = bytearray(data[pos:pos + 27 + nseg + body_len])
        page[22:26] = b"\0\0\0\0"
        assert ogg_crc(page) == crc, "page %d CRC" % pageno
        assert seq == pageno
        assert serial is None or ser == serial
        serial = ser
        assert bool(flags & 2) == (pageno == 0)
        assert bool(flags & 1) == (len(cur) > 0), "continued-packet flag"
        body = data[pos + 27 + nseg:pos + 27 + nseg + body_len]
        off = 0
        for lv in lacing:
            cur += body[off:off + lv]
            off += lv
            if lv < 255:
                packets.append(bytes(cur))
                cur = bytearray()
        granules.append(granule)
        last_flags = flags
        pageno += 1
        pos += 27 + nseg + body_len
    assert not cur
    assert last_flags & 4, "last page must carry end-of-stream"
    return packets, granules


def run_example(tmp_path, name, w, h, nframes, env=None):
    y4m = str(tmp_path / "in.y4m")
    frames = write_y4m(y4m, w, h, nframes)
    out = str(tmp_path / name)
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([EXE, "-v", "20", "-k", "1", "-z", "7", "-o", out, y4m], capture_output=True,
                       text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    with open(out, "rb") as f:
        return f.read(), frames, p.stderr


@needs_exe
@pytest.mark.parametrize("w,h,nframes", [(64, 64, 2), (176, 120, 3)])
def test_encoder_example_unmodified_cpu_plumbing(tmp_path, w, h, nframes):
    ogv, frames, _ = run_example(tmp_path, "c.ogv", w, h, nframes)
    packets, granules = ogg_packets(ogv)
    # three header packets (info, comment, setup: src/infoenc.c) then one packet per frame
    assert len(packets) == 3 + nframes
    assert packets[0][:6] == b"\x80daala"
    assert granules[0] == 0
    assert len(granules) >= 2
    r = ref()
    r.ref_set_external_dct_vtbl(None, None)
    out = np.zeros(1 << 20, np.uint8)
    sizes = (ctypes.c_long * 16)()
    n = r.ref_encode_yuv420(P(frames), w, h, nframes, 20, 7, 0, P(out), ctypes.c_long(out.size), sizes)
    assert n == nframes
    off = 0
    for i in range(n):
        assert packets[3 + i] == bytes(out[off:off + sizes[i]]), "data packet %d" % i
        off += sizes[i]


@needs_exe
@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 64), (176, 120)])
def test_encoder_example_unmodified_with_libdaalahip_is_byte_identical(tmp_path, size):
    import torch
    assert torch.cuda.is_available()
    w, h = size
    nframes = 2
    want, _, _ = run_example(tmp_path, "c.ogv", w, h, nframes)
    ipo = os.path.join(HERE, "interpose", "libinterpose.so")
    hip = os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so")
    assert os.path.exists(ipo) and os.path.exists(hip)
    got, _, err = run_example(tmp_path, "h.ogv", w, h, nframes,
                              env={"LD_PRELOAD": ipo + ":" + hip, "ODHIP_INTERPOSE_VTBL": "1",
                                   "ODHIP_INTERPOSE_REPORT": "1"})
    line = [l for l in err.splitlines() if l.startswith("odhip_interposed_calls")]
    assert line, err[-1000:]
    calls = [int(v) for v in line[-1].split()[1:]]
    assert all(c > 0 for c in calls), calls
    # every packet and granule position of the container; the raw files differ in the Ogg
    # stream serial number only when the two runs straddle a second (examples/
    # encoder_example.c:927-928: srand(time(NULL)); serial = rand())
    assert ogg_packets(got) == ogg_packets(want), \
        "the .ogv written through libdaalahip differs from the plain C one"
    assert len(ogg_packets(got)[0]) == 3 + nframes
