"""End-to-end frames/s of the UNMODIFIED reference encoder with the batched GPU stages bound
(tests/interpose mode 3: pyramids + keyframe-luma PVQ band stage per frame, host pricing;
then also the deringing level search from batched passes, odhip_dering_cache), against plain
C, with N encoder processes sharing one GPU (all-intra frames are independent: the
frame-sharded driver runs several encoder contexts per GPU).  1080p, -v 20 -z 7."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tests", "interpose", "run_interposed.py")


def run(mode, nproc, nframes, dering=False):
    e = dict(os.environ)
    e.update(NFRAMES=str(nframes), CONTENT="bench", ODHIP_INTERPOSE_PASSTHROUGH="1", ODHIP_CACHE_FDCT_ONLY="1")
    if dering:
        e["ODHIP_INTERPOSE_DERING_CACHE"] = "1"
    t0 = time.perf_counter()
    procs = [subprocess.Popen(["taskset", "-c", str(i), sys.executable, RUN, str(mode), "1920", "1080"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
             for i in range(nproc)]
    outs = [p.communicate() for p in procs]
    wall = time.perf_counter() - t0
    res = []
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-800:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    enc = [r["encode_seconds"] for r in res]
    return {"processes": nproc, "frames_per_process": nframes, "wall_s": round(wall, 2),
            "encode_s_mean": round(sum(enc) / len(enc), 3),
            "fps_encode_only": round(nproc * nframes / max(enc), 3),
            "gpu_batch_ms_per_frame": round(sum(r.get("gpu_batch_ms", 0) for r in res) / (nproc * nframes), 1),
            "dering_launches_served": res[0].get("dering"),
            "packets": res[0]["packets"][:16]}


if __name__ == "__main__":
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    run(0, 1, 1)      # untimed: pages the libraries and the interpreter in on a fresh box
    for nproc in (1, 8):
        for mode, dering, name in ((0, False, "plain C"), (3, False, "batched band stage"),
                                   (3, True, "batched band stage + dering search")):
            print(name, json.dumps(run(mode, nproc, nframes, dering)), flush=True)
