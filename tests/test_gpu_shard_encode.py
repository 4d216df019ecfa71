"""BASELINE configs[4] in miniature ON THE GPU: two processes (both on cuda:0, gloo
control plane - the box has one GPU), each running the real reference encoder with the
batched GPU stage (pyramids + keyframe-luma PVQ band stage) on the frames it owns,
packets gathered to rank 0 with daala_amd.shard.gather_packets and compared with the
plain C encoder run sequentially, byte for byte."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
# NOT _libs.ref(): the spawned workers import this module, and a reference library loaded
# before the interposer can no longer be interposed (see _shard_encode.load_batched_encoder)
HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))


def _worker(rank, world, port, nframes, w, h, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _shard_encode as S
    from daala_amd.shard import frames_of_rank, gather_packets
    r, ipo = S.load_batched_encoder(w, h, device=0)
    local = S.encode_owned(r, frames_of_rank(nframes, rank, world), w, h)
    got = gather_packets(local, nframes)
    q.put((rank, S.digest(got) if rank == 0 else None, S.band_stats(ipo)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not present")
def test_two_rank_gpu_encode_equals_sequential_c_encoder():
    import _shard_encode as S
    nframes, world, w, h = 5, 2, 320, 192
    want = S.sequential_digest(nframes, w, h)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, nframes, w, h, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = dict((rk, (dg, st)) for rk, dg, st in (q.get(timeout=600) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][0] == want
    for rk in range(world):
        served, with_ref, other, searches = res[rk][1]
        assert served > 1000 and other == 0, res[rk][1]
