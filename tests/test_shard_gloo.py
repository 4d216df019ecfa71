"""CPU, world_size 2, gloo: the N > 1 path - frame ownership and the packet
gather - against the sequential result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _packet(i):
    # deterministic variable-length pseudo-packet for frame i (incl. an empty one)
    n = 0 if i == 5 else 17 + (i * 7919) % 301
    return bytes(((i * 131 + j * 17) & 0xFF) for j in range(n))


def _worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daala_amd.shard import frames_of_rank, gather_packets
    mine = frames_of_rank(nframes, rank, world)
    local = {i: _packet(i) for i in mine}
    out = gather_packets(local, nframes)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("nframes", [1, 7, 12])
def test_gather_matches_sequential(nframes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [_packet(i) for i in range(nframes)]


def _drop_worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daala_amd.shard import frames_of_rank, gather_packets
    # rank 1 loses frame 3: nobody owns it
    local = {i: _packet(i) for i in frames_of_rank(nframes, rank, world) if i != 3}
    try:
        gather_packets(local, nframes)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_dropped_frame_is_an_error_on_every_rank():
    """A frame no rank owns must not come back as an empty packet (frame 5 of the
    other tests IS a legitimately empty packet and does come back)."""
    world, nframes = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_drop_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert set(got) == {0, 1}
    for msg in got.values():
        assert "owned by no rank" in msg and "3" in msg


def test_ownership_partitions_frames():
    from daala_amd.shard import frames_of_rank
    for world in (1, 2, 4, 8):
        for n in (0, 1, 9, 300):
            got = sorted(i for r in range(world) for i in frames_of_rank(n, r, world))
            assert got == list(range(n))


def _encode_worker(rank, world, port, nframes, q):
    """BASELINE configs[4] in miniature, on CPU: every rank runs the REAL reference
    encoder (oracle/_ref) on the frames it owns, seeded with their global frame
    numbers, and the packets are gathered to rank 0."""
    import ctypes
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import P, ref, synth_frame
    from daala_amd.shard import frames_of_rank, gather_packets
    r = ref()
    w = h = 64
    mine = frames_of_rank(nframes, rank, world)
    local = {}
    if mine:
        fr = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=7, phase=5 * i)])
                             for i in mine]).astype(np.uint8)
        idx = (ctypes.c_int * len(mine))(*mine)
        out = np.zeros(1 << 20, np.uint8)
        sizes = (ctypes.c_long * 64)()
        k = r.ref_encode_yuv420_shard(P(fr), w, h, len(mine), idx, 20, 7, P(out),
                                      ctypes.c_long(out.size), sizes)
        assert k == len(mine)
        pos = 0
        for j, i in enumerate(mine):
            local[i] = bytes(out[pos:pos + sizes[j]])
            pos += sizes[j]
    got = gather_packets(local, nframes)
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_reference_encode_equals_sequential_byte_for_byte():
    """All-intra frames are independent except for the display frame number coded
    in the frame header (src/encode.c:3043); a shard seeded with the global frame
    numbers (ref_encode_yuv420_shard) yields the sequential encoder's packets."""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import P, ref, synth_frame
    r = ref()
    if r is None:
        pytest.skip("oracle/_ref not present")
    nframes, world, w, h = 5, 2, 64, 64
    frames = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=7, phase=5 * i)])
                             for i in range(nframes)]).astype(np.uint8)
    out = np.zeros(1 << 20, np.uint8)
    sizes = (ctypes.c_long * 64)()
    assert r.ref_encode_yuv420(P(frames), w, h, nframes, 20, 7, 0, P(out), ctypes.c_long(out.size),
                               sizes) == nframes
    want, pos = [], 0
    for i in range(nframes):
        want.append(bytes(out[pos:pos + sizes[i]]))
        pos += sizes[i]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_encode_worker, args=(rk, world, port, nframes, q)) for rk in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == want
    assert len(set(want)) == nframes
