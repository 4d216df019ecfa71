"""bench.py's N > 1 flow end to end on a one-GPU box: two ranks launched the way the driver
launches them (torch.distributed.run), both on cuda:0 with the control plane over gloo (the
test hooks ODHIP_BENCH_ONE_GPU / ODHIP_BENCH_BACKEND) - sharded pictures, barrier + max over
ranks timing, rank 0's single JSON line with the whole-job value, and the sharded encode check
(one 1080p frame per rank through the real encoder, packets gathered and compared with the
sequential plain-C encoder)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ)
    env.update(ODHIP_BENCH_ONE_GPU="1", ODHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "4"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["frames_per_gpu_per_step"] == 4
    # whole-job value: both ranks' blocks over the max time
    assert abs(d["value"] - 2 * 4 * 3 * d["config"]["blocks_per_frame"] / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]
    assert d["pipelined_equals_serial"] is True
    chk = d.get("sharded_encode_check")
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so")):
        assert chk and chk["ran"], chk
        assert chk["packets_equal_sequential_c_encoder"] is True
        assert chk["frames"] == 2
