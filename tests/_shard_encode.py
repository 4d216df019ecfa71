"""Frame-sharded all-intra encode with the batched GPU stage behind the REAL reference
encoder: the job itself lives in encode_job.py at the repository root (the shim,
shim/libdaalahipglue.so, configured by explicit calls); this module re-exports it for the
tests and adds the CHECK side: the plain C encoder run sequentially in a child process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from encode_job import (band_stats, digest, encode_frames, encode_owned, frame_yuv,  # noqa: E402,F401
                        glue_stats, load_batched_encoder)


def sequential_digest(nframes, w, h, quality=20, complexity=7):
    """The plain C encoder on frames 0..nframes-1, in a child process (this process has
    the batched stage bound)."""
    import json
    import subprocess
    e = dict(os.environ)
    e.pop("ODHIP_INTERPOSE_PASSTHROUGH", None)
    e.update(NFRAMES=str(nframes), CONTENT="bench", QUALITY=str(quality), COMPLEXITY=str(complexity),
             PACKET_DIGEST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interpose", "run_interposed.py"), "0",
                        str(w), str(h)], capture_output=True, text=True, timeout=1800, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])["digest"]
