"""YUV4MPEG2 input of the frame-batch path (odhip_y4m_*, host code): the files the
reference's encoder_example reads (examples/encoder_example.c:89-160, :449-508).  The
same Y4M that tests/test_encoder_example.py feeds to the unmodified encoder_example must
come back plane for plane; odd sizes, frame parameters, a missing C tag; and the formats
the batched path does not take are refused, not guessed."""
import ctypes

import numpy as np

import daala_amd
from test_encoder_example import write_y4m


def _open(path):
    L = daala_amd.lib()
    L.odhip_y4m_open.restype = ctypes.c_void_p
    w, h, fn, fd, err = (ctypes.c_int() for _ in range(5))
    y = L.odhip_y4m_open(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(fn),
                         ctypes.byref(fd), ctypes.byref(err))
    return L, y, w.value, h.value, (fn.value, fd.value), err.value


def _read_all(L, y, w, h):
    cw, ch = (w + 1) >> 1, (h + 1) >> 1
    out = []
    while True:
        planes = [np.zeros((h, w), np.uint8), np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)]
        rc = L.odhip_y4m_read(ctypes.c_void_p(y), *[p.ctypes.data_as(ctypes.c_void_p) for p in planes])
        if rc == 0:
            return out
        assert rc == 1, rc
        out.append(np.concatenate([p.ravel() for p in planes]))


def test_reads_what_encoder_example_reads(tmp_path):
    for (w, h, n) in ((64, 64, 2), (176, 120, 3)):
        path = tmp_path / ("in_%d.y4m" % w)
        want = write_y4m(str(path), w, h, n)
        L, y, gw, gh, fps, err = _open(path)
        assert y and err == 0 and (gw, gh) == (w, h) and fps == (30, 1)
        got = _read_all(L, y, w, h)
        L.odhip_y4m_close(ctypes.c_void_p(y))
        assert len(got) == n and np.array_equal(np.concatenate(got), want)


def test_odd_size_frame_parameters_and_default_chroma(tmp_path):
    w, h = 35, 21
    rng = np.random.RandomState(4)
    frames = [rng.randint(0, 256, size=w * h + 2 * 18 * 11).astype(np.uint8) for _ in range(2)]
    path = tmp_path / "odd.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W35 H21 F25:1 Ip A1:1 Xcomment\n")      # no C tag: 4:2:0
        for fr in frames:
            f.write(b"FRAME Ip\n")                                   # frame parameters
            f.write(fr.tobytes())
    L, y, gw, gh, fps, err = _open(path)
    assert y and (gw, gh, fps) == (w, h, (25, 1))
    got = _read_all(L, y, w, h)
    L.odhip_y4m_close(ctypes.c_void_p(y))
    assert len(got) == 2 and all(np.array_equal(a, b) for a, b in zip(got, frames))


def test_refuses_what_the_batched_path_cannot_take(tmp_path):
    for tags, code in ((b"W16 H16 F30:1 Ip C444", -23), (b"W16 H16 F30:1 Ip C420p10", -23),
                       (b"W16 H16 F30:1 It C420jpeg", -23), (b"W16 H16 F30:1 Ip Cmono", -23)):
        path = tmp_path / "x.y4m"
        with open(path, "wb") as f:
            f.write(b"YUV4MPEG2 " + tags + b"\n")
        _, y, _, _, _, err = _open(path)
        assert not y and err == code, (tags, err)
    path = tmp_path / "junk.y4m"
    with open(path, "wb") as f:
        f.write(b"RIFF....\n")
    _, y, _, _, _, err = _open(path)
    assert not y and err == -10
    # loss of framing / short read
    path = tmp_path / "short.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W16 H16 F30:1 Ip C420\nFRAME\n" + bytes(100))
    L, y, w, h, _, _ = _open(path)
    planes = [np.zeros(256, np.uint8), np.zeros(64, np.uint8), np.zeros(64, np.uint8)]
    assert L.odhip_y4m_read(ctypes.c_void_p(y), *[p.ctypes.data_as(ctypes.c_void_p) for p in planes]) == -1
    L.odhip_y4m_close(ctypes.c_void_p(y))


def test_skip_reads_only_the_frames_a_rank_owns(tmp_path):
    """odhip_y4m_skip: rank r of a world of 3 reads frames r, r + 3, ... of a 7-frame file
    (with frame parameters on some FRAME lines) and steps over the rest."""
    w, h, n = 48, 32, 7
    rng = np.random.RandomState(9)
    fsz = w * h + 2 * (w // 2) * (h // 2)
    frames = [rng.randint(0, 256, size=fsz).astype(np.uint8) for _ in range(n)]
    path = tmp_path / "shard.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W48 H32 F30:1 Ip C420jpeg\n")
        for i, fr in enumerate(frames):
            f.write(b"FRAME Ip\n" if i % 2 else b"FRAME\n")
            f.write(fr.tobytes())
    for rank in range(3):
        L, y, gw, gh, fps, err = _open(path)
        assert y and (gw, gh) == (w, h)
        planes = [np.zeros(w * h, np.uint8), np.zeros(w * h // 4, np.uint8), np.zeros(w * h // 4, np.uint8)]
        got = []
        for i in range(n + 1):
            if i % 3 == rank:
                rc = L.odhip_y4m_read(ctypes.c_void_p(y), *[p.ctypes.data_as(ctypes.c_void_p) for p in planes])
                if rc == 1:
                    got.append((i, np.concatenate(planes)))
            else:
                rc = L.odhip_y4m_skip(ctypes.c_void_p(y))
            assert rc == (1 if i < n else 0), (rank, i, rc)
        L.odhip_y4m_close(ctypes.c_void_p(y))
        assert [i for i, _ in got] == list(range(rank, n, 3))
        assert all(np.array_equal(a, frames[i]) for i, a in got)


def test_skip_reports_a_truncated_frame_like_read_does(tmp_path):
    """A rank that SKIPS a truncated last frame must see the same error as the rank that reads it
    (a seek past the end of a file succeeds, so skip used to count the frame and the
    ranks of a sharded encode disagreed on the total)."""
    w = h = 16
    nbytes = w * h + 2 * 8 * 8
    path = tmp_path / "trunc.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W16 H16 F30:1 Ip C420\n")
        f.write(b"FRAME\n" + bytes(nbytes))           # a whole frame
        f.write(b"FRAME\n" + bytes(nbytes - 7))       # a truncated one
    L, y, _, _, _, _ = _open(path)
    assert L.odhip_y4m_skip(ctypes.c_void_p(y)) == 1
    assert L.odhip_y4m_skip(ctypes.c_void_p(y)) < 0
    L.odhip_y4m_close(ctypes.c_void_p(y))
    L, y, _, _, _, _ = _open(path)
    planes = [np.zeros((h, w), np.uint8), np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8)]
    ptrs = [p.ctypes.data_as(ctypes.c_void_p) for p in planes]
    assert L.odhip_y4m_read(ctypes.c_void_p(y), *ptrs) == 1
    assert L.odhip_y4m_read(ctypes.c_void_p(y), *ptrs) < 0
    L.odhip_y4m_close(ctypes.c_void_p(y))
    # whole frames: skip, skip, end of stream
    path = tmp_path / "whole.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W16 H16 F30:1 Ip C420\n")
        for _ in range(2):
            f.write(b"FRAME\n" + bytes(nbytes))
    L, y, _, _, _, _ = _open(path)
    assert [L.odhip_y4m_skip(ctypes.c_void_p(y)) for _ in range(3)] == [1, 1, 0]
    L.odhip_y4m_close(ctypes.c_void_p(y))
