"""Host side of the frame-sharded encode job (encode_job.py, bench.py --encode-frames): which frames an
encoder takes from a Y4M file (odhip_y4m_skip / _read through the library, CPU only) and the CPU quota
bench.py reports."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_every_encoder_reads_exactly_its_frames(tmp_path):
    import daala_amd as D
    import encode_job as S
    w, h, n = 48, 32, 7
    rng = np.random.RandomState(5)
    frames = [rng.randint(0, 256, size=w * h * 3 // 2).astype(np.uint8) for _ in range(n)]
    path = tmp_path / "j.y4m"
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W48 H32 F30:1 Ip A1:1 C420jpeg\n")
        for fr in frames:
            f.write(b"FRAME\n" + fr.tobytes())
    seen = {}
    for stride, limit in ((3, n), (2, 5), (1, n), (4, 100)):
        seen.clear()
        for offset in range(stride):
            got, gw, gh, total = S.read_y4m_frames(D, str(path), offset, stride, limit)
            assert (gw, gh) == (w, h) and total == min(limit, n)
            assert sorted(got) == list(range(offset, min(limit, n), stride))
            for i, fr in got.items():
                assert np.array_equal(fr, frames[i])
                assert i not in seen
                seen[i] = True
        assert sorted(seen) == list(range(min(limit, n)))       # every frame owned by exactly one encoder


def test_host_cpu_quota_is_positive_and_bounded():
    import bench
    q = bench.host_cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)


def test_glue_config_mirror_matches_the_header():
    """encode_job.GlueConfig is filled by field name and read by the shim by offset: its fields must be the
    header's, in the header's order (a field inserted in the middle of odhip_glue_config once shifted every
    batched switch of the encode job)."""
    import re
    import encode_job as S
    src = open(os.path.join(ROOT, "shim", "daala_hip_glue.h")).read()
    body = src[src.index("typedef struct odhip_glue_config {"):src.index("} odhip_glue_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"\bint\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in S.GlueConfig._fields_]
