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


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so")),
                    reason="oracle/_ref (the reference encoder: the host half) not present")
def test_configs4_encode_mode_two_ranks_one_gpu(tmp_path):
    """BASELINE configs[4] as a command, in miniature: `bench.py --gpus 2 --encode-frames 5` -
    a Y4M file in through odhip_y4m_*, frame i -> rank i mod 2, every rank the reference
    encoder with the batched GPU stage behind it, packets gathered to rank 0, the first three
    compared with the plain C encoder run sequentially; one JSON line with frames/s."""
    env = dict(os.environ)
    env.update(ODHIP_BENCH_ONE_GPU="1", ODHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29633", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--encode-frames", "12", "--encode-check", "12"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["frames"] == 12 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["rank0"]["frames"] == 6 and d["rank0"]["bands_from_batch"] > 100000
    # no --threads-per-proc: the encoder threads of a rank are sized from what the host grants -
    # the CPU quota over the two ranks that share this box - and the line says whether that starves a GPU
    sys.path.insert(0, ROOT)
    import bench
    quota = bench.host_cpu_quota()
    assert d["host_cpu_quota"] == quota
    assert d["encoder_threads_per_process"] == max(1, min(32, quota // 2))
    assert d["encoder_threads_source"].startswith("host_cpu_quota")
    assert d["host_starved"] == (d["encoders_per_gpu"] < 16)
    if d["host_starved"]:
        assert "below the 16 per GPU" in p.stderr
    assert d["prefix_check"]["frames"] == 12         # every frame of the job, not a prefix
    assert d["prefix_check"]["packets_equal_sequential_c_encoder"] is True
    assert d["prefix_check"]["covers"] == "every frame of the job"


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so")),
                    reason="oracle/_ref (the reference encoder: the host half) not present")
def test_configs4_encode_mode_rccl_world_size_one_two_encoders_per_gpu():
    """The RCCL branch itself: one rank launched by torch.distributed.run with backend "nccl"
    (ODHIP_BENCH_FORCE_DIST=1 makes a process group of world size 1), two encoder processes of two
    encoder threads each sharing the GPU (--procs-per-gpu 2 --threads-per-proc 2), packets gathered with device tensors over RCCL
    (daala_amd.shard.gather_packets) and every frame compared with the sequential C encoder."""
    env = dict(os.environ)
    env.update(ODHIP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ODHIP_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29635", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--encode-frames", "6", "--procs-per-gpu", "2", "--threads-per-proc", "2",
           "--encode-check", "6"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["frames"] == 6 and d["encoder_processes_per_gpu"] == 2
    assert d["encoder_threads_per_process"] == 2 and d["encoders_per_gpu"] == 4
    assert d["rank0"]["frames"] == 6 and len(d["rank0"]["encoder_seconds_per_process"]) == 2
    assert d["prefix_check"]["frames"] == 6
    assert d["prefix_check"]["packets_equal_sequential_c_encoder"] is True


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref_distglue.so")),
                    reason="oracle/_ref (the reference encoder: the host half) not present")
def test_configs4_four_encoder_threads_with_the_shims_cross_checks():
    """Four encoder THREADS of one process sharing one HIP context, every batched price, every served
    od_dering superblock and every served od_compute_dist cross-checked inside the encoders against the
    reference's own C definitions (a difference aborts the encoder), and the check on the steady-state
    frames of every thread - the LAST frame each thread coded - not on the warm-up prefix."""
    env = dict(os.environ)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("ODHIP_BENCH_BACKEND", "ODHIP_BENCH_FORCE_DIST", "ODHIP_BENCH_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--encode-frames", "12", "--threads-per-proc", "4",
           "--encode-selfcheck", "7"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["encoder_threads_per_process"] == 4 and d["encoder_selfchecks"] == 7
    chk = d["prefix_check"]
    assert chk["frame_indices"] == [8, 9, 10, 11], chk      # thread t's last frame: 8 + t
    assert chk["packets_equal_sequential_c_encoder"] is True, chk
    r0 = d["rank0"]
    assert r0["bands_from_batch"] > 12 * 100000 and r0["od_compute_dist_served_per_frame"] > 2000
    assert r0["dering_served_per_frame"] > 2000
