"""CPU: the export format of include/daala_hip.h (export_kernels.hip) - the layout arithmetic of
odhip_export_layout_make (host code of the library, no GPU needed) and the reference-side reader
daala_amd.decode_export_sections on a buffer WRITTEN HERE in numpy from the header's description: records, groups
placed in arbitrary order, escapes.  The GPU tests decode what the kernels wrote; this pins the reader and the
documented format to each other independently of the kernels."""
import numpy as np

import daala_amd as D


def _write_section(buf, hdr, si, sec, y, qg, itheta, max_theta, noref, skip, rng):
    nb, offs, ln = D.pvq_band_layout(sec["bs"])
    B = y.shape[0]
    rb = sec["record_bytes"]
    rec4 = np.dtype([("qg", "<i2"), ("fn", "<u2")])
    rec8 = np.dtype([("qg", "<i2"), ("itheta", "<i2"), ("max_theta", "<i2"), ("fn", "<u2")])
    rec = np.zeros(B * nb, rec8 if rb == 8 else rec4)
    bpg = sec["blocks_per_group"]
    ngroups = sec["ngroups"]
    assert ngroups == (B + bpg - 1) // bpg
    group_words = []
    for g in range(ngroups):
        words = []
        for blk in range(g * bpg, min(B, (g + 1) * bpg)):
            for i in range(nb):
                n0 = len(words)
                if skip[blk, i] == 0:
                    for j in range(offs[i + 1] - offs[i]):
                        v = int(y[blk, offs[i] + j])
                        if v == 0:
                            continue
                        if -255 <= v <= 255:
                            words.append(((v << 7) | j) & 0xffff)
                        else:
                            words.append(((-256 << 7) | j) & 0xffff)
                            words.append(v & 0xffff)
                r = rec[blk * nb + i]
                r["qg"] = qg[blk, i]
                if rb == 8:
                    r["itheta"] = itheta[blk, i]
                    r["max_theta"] = max_theta[blk, i]
                r["fn"] = (len(words) - n0) | int(noref[blk, i]) << 9 | int(skip[blk, i]) << 10
        group_words.append(words)
    # groups land in the stream in ANY order
    order = rng.permutation(ngroups)
    base = np.zeros(ngroups, np.uint32)
    stream = []
    for g in order:
        base[g] = len(stream)
        stream += group_words[g]
    assert len(stream) <= sec["cap_words"]
    buf[sec["records_off"]:sec["records_off"] + rec.nbytes] = rec.view(np.uint8)
    buf[sec["group_base_off"]:sec["group_base_off"] + 4 * ngroups] = base.view(np.uint8)
    s = np.asarray(stream, np.uint16)
    buf[sec["stream_off"]:sec["stream_off"] + s.nbytes] = s.view(np.uint8)
    hdr[si] = len(stream)


def test_layout_and_reader_agree_with_the_documented_format():
    rng = np.random.RandomState(5)
    bs_list = [0, 1, 3, 2, 4]
    with_ref = [0, 1, 1, 0, 0]
    nblocks = [300, 77, 9, 40, 5]
    _, lay = D.export_layout_make(nblocks, bs_list, with_ref)
    assert lay["nsections"] == 5 and lay["fixed_bytes"] % 16 == 0 and lay["total_bytes"] % 16 == 0
    pos = 128
    for sec, bs, wr, B in zip(lay["sections"], bs_list, with_ref, nblocks):
        nb, offs, ln = D.pvq_band_layout(bs)
        assert sec["nrecords"] == B * nb and sec["record_bytes"] == (8 if wr else 4)
        assert sec["blocks_per_group"] == 2048 // ln and sec["cap_words"] == B * ln
        assert sec["records_off"] % 16 == 0 and sec["group_base_off"] % 16 == 0 and sec["stream_off"] % 16 == 0
        assert sec["records_off"] >= pos
        pos = sec["group_base_off"] + 4 * sec["ngroups"]
    assert lay["sections"][0]["stream_off"] >= lay["fixed_bytes"] - 15
    buf = np.zeros(lay["total_bytes"], np.uint8)
    hdr = buf[:64].view(np.uint32)
    want = []
    for si, (sec, bs, wr, B) in enumerate(zip(lay["sections"], bs_list, with_ref, nblocks)):
        nb, offs, ln = D.pvq_band_layout(bs)
        y = np.zeros((B, ln), np.int32)
        m = rng.rand(B, ln) < 0.1
        y[m] = rng.randint(-9, 10, size=int(m.sum()))
        big = rng.rand(B, ln) < 0.01
        y[big] = rng.choice([-256, 256, 999, -32768, 32767], size=int(big.sum()))
        y[:, 0] = 0                                        # the DC slot holds no pulse
        qg = rng.randint(-5, 400, size=(B, nb))
        itheta = rng.randint(-1, 50, size=(B, nb)) if wr else np.full((B, nb), -1)
        max_theta = rng.randint(0, 60, size=(B, nb)) if wr else np.zeros((B, nb), int)
        noref = rng.randint(0, 2, size=(B, nb)) if wr else np.ones((B, nb), int)
        skip = rng.choice([0, 0, 0, 1, 2], size=(B, nb)) if wr else np.zeros((B, nb), int)
        for i in range(nb):
            y[skip[:, i] != 0, offs[i]:offs[i + 1]] = 0
        _write_section(buf, hdr, si, sec, y, qg, itheta, max_theta, noref, skip, rng)
        band = np.zeros((B, nb, 4), np.int32)
        band[..., 0] = qg
        band[..., 1] = itheta
        band[..., 2] = max_theta
        for i in range(nb):
            band[:, i, 3] = np.abs(y[:, offs[i]:offs[i + 1]]).sum(axis=1)
        want.append((y, band, skip == 0))
    got = D.decode_export_sections(buf, lay)
    for si in range(5):
        for name, g, w in zip(("pulses", "record", "coded"), got[si], want[si]):
            assert np.array_equal(g, w), (si, name)
