"""Frame sharding of an all-intra encode over the GPUs of one node.

All-intra frames are independent in the reference (keyframe_rate = 1 makes
every frame OD_I_FRAME, src/encode.c:303-308; the range coder and adaptation
context are reset per frame, :3029,:3080), so the only exchange step of a
sharded encode is collecting the coded packets: there is NO collective inside a
frame.  One process per GPU; `torch.distributed` backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.

    owner(frame i)      = i mod world_size
    gather_packets(...) = all_gather of packet sizes, then one padded all_gather
                          of packet bytes (RCCL has no gatherv; the payload is
                          ~0.23 MB per 1080p frame, negligible against one xGMI
                          link), rank 0 returns the packets in frame order.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def frames_of_rank(nframes: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: frame i is encoded by rank i mod world."""
    return list(range(rank, nframes, world))


def gather_packets(local: Dict[int, bytes], nframes: int, device=None) -> Optional[List[bytes]]:
    """Collect {frame index: packet bytes} from every rank.  Returns the list of
    packets in frame order on rank 0 (None elsewhere).  Every frame index must be
    owned by exactly one rank: a frame nobody owns (a dropped frame) or two owners
    raise on every rank; an owned frame may be an empty packet."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend() == "nccl" else torch.device("cpu")
    # 1. sizes: a dense [nframes] vector per rank (0 where the rank does not own)
    sizes = torch.zeros(nframes, dtype=torch.int64, device=device)
    for i, b in local.items():
        if not 0 <= i < nframes:
            raise ValueError("frame index %d out of range" % i)
        sizes[i] = len(b) + 1     # 0 = not owned; an owned empty packet is 1
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    table = torch.stack(all_sizes)  # [world, nframes]
    owners = (table > 0).sum(0)
    if nframes and int(owners.max()) > 1:
        raise ValueError("a frame is owned by more than one rank")
    if nframes and int(owners.min()) < 1:
        missing = [int(i) for i in (owners == 0).nonzero().flatten()[:8]]
        raise ValueError("frames owned by no rank (dropped): %s" % missing)
    table = (table - 1).clamp(min=0)
    # 2. bytes: each rank concatenates its packets in frame order, padded to the max
    per_rank = table.sum(1)
    cap = int(per_rank.max())
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=device)
    pos = 0
    for i in sorted(local):
        b = local[i]
        if b:
            buf[pos:pos + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
            pos += len(b)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != 0:
        return None
    out: List[bytes] = [b""] * nframes
    table_c = table.cpu()
    for r in range(world):
        data = bufs[r].cpu().numpy().tobytes()
        pos = 0
        for i in range(nframes):
            n = int(table_c[r, i])
            if n:
                out[i] = data[pos:pos + n]
                pos += n
    return out
