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


def test_ownership_partitions_frames():
    from daala_amd.shard import frames_of_rank
    for world in (1, 2, 4, 8):
        for n in (0, 1, 9, 300):
            got = sorted(i for r in range(world) for i in frames_of_rank(n, r, world))
            assert got == list(range(n))
