"""The frame-batch step (odhip_pipe) as bench.py runs it, verified:

(i)  the default two-stream, software-pipelined step at 1080p - luma chain and chroma
     chain in their own contexts on their own streams, the luma chain of step i + 1
     overlapping the chroma chain of step i - leaves exactly the records, pulse vectors,
     choices and reconstructed pixels of the same steps run serially on one stream
     (round 1's shared edge-strip scratch made this fail silently);
(ii) a WHOLE 1080p frame (every block of every level of all three planes) through the
     GPU stages with the host pricing every candidate in between equals the compiled
     reference's pvq_theta path pixel for pixel, for both synthetic content types;
(iii) the same whole frames through price = 1 steps - od_pvq_rate's closed form evaluated
     on the device inside the choice kernels, one C call per step, the configuration
     bench.py times - equal the compiled reference too, also with the decision margin
     forced wide so that the host-libm path re-decides bands."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _libs import ref  # noqa: E402

pytestmark = pytest.mark.gpu

PIC_W, PIC_H = 1920, 1080


def _bench():
    import bench
    return bench


def _pictures(gen, nframes, seed):
    b = _bench()
    fr = [b.picture_planes(gen(i, seed)) for i in range(nframes)]
    return (np.stack([f[0] for f in fr]), np.concatenate([np.stack([f[1] for f in fr]),
                                                          np.stack([f[2] for f in fr])]))


def _dump(D, pipe, cfl):
    out = {}
    for set_ in (0, 1):
        for bs in range(5 - set_):
            out[("recon", set_, bs)] = pipe.read(D.BUF_RECON, set_, bs)
            out[("choice", set_, bs)] = pipe.read(D.BUF_CHOICE, set_, bs)
            out[("band", set_, bs)] = pipe.read(D.BUF_BAND, set_, bs)
            out[("y", set_, bs)] = pipe.read(D.BUF_Y, set_, bs)
            if set_ and cfl:
                out[("items", set_, bs)] = pipe.read(D.BUF_ITEMS, set_, bs)
    return out


@pytest.mark.parametrize("cfl,price", [(True, False), (False, False), (True, True), (False, True)])
def test_pipelined_step_equals_serial_step_1080p(cfl, price):
    import daala_amd as D
    D.init(0)
    b = _bench()
    qt = D.QuantTables.load()
    F = 3
    pipes = [D.Pipe(qt, F, PIC_W, PIC_H, chroma_cfl=cfl, serial=s, price=price) for s in (False, True)]
    try:
        for rnd, gen in enumerate((b.synth_frame_np, b.natural_like_frame_np)):
            luma, chroma = _pictures(gen, F, 77 + rnd)
            dumps = []
            for pipe in pipes:
                pipe.set_pictures(luma, chroma)
                for _ in range(4):
                    pipe.step()
                pipe.flush()
                pipe.sync()
                dumps.append(_dump(D, pipe, cfl))
            for key in dumps[0]:
                assert np.array_equal(dumps[0][key], dumps[1][key]), (rnd, key)
            # the steps really reconstruct something picture-like at the finest level
            rec = dumps[0][("recon", 0, 0)].reshape(F, pipes[0].H, pipes[0].W)[:, :PIC_H]
            assert np.abs(rec.astype(np.int32) - luma).mean() < 12
    finally:
        for pipe in pipes:
            pipe.destroy()


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
@pytest.mark.parametrize("content", ["checker", "natural"])
def test_whole_frame_equals_compiled_reference_1080p(content):
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.load()
    pics = b.picture_planes(b.CONTENT[content](3, 4321))
    cpu, blocks, _ = C.cpu_frame(qt, pics, PIC_W, PIC_H, chroma_cfl=True)
    assert blocks == b.blocks_per_frame()
    gpu = C.gpu_priced_frame(D, qt, pics, PIC_W, PIC_H, chroma_cfl=True)
    assert C.compare_frame(gpu, cpu) == []


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
@pytest.mark.parametrize("content", ["checker", "natural"])
def test_device_priced_steps_equal_compiled_reference_1080p(content):
    """(iii): two 1080p pictures per step, pipelined, priced on the device."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.load()
    F = 2
    fr = [b.picture_planes(b.CONTENT[content](3 + i, 4321)) for i in range(F)]
    pics = [np.stack([f[p] for f in fr]) for p in range(3)]
    gpu, reruns, dec = C.gpu_device_priced(D, qt, pics, PIC_W, PIC_H, chroma_cfl=True, frames=F, steps=3,
                                           decisions=True)
    for i in range(F):
        want = []
        cpu, _, _ = C.cpu_frame(qt, fr[i], PIC_W, PIC_H, chroma_cfl=True, decisions=want)
        assert C.compare_frame(gpu, cpu, frame=i, frames=F) == [], (content, i)
        # north_star's "coefficients and PVQ pulse vectors": the coded gain index, itheta,
        # max_theta, K and the pulse vector of EVERY band of every block of every level of the
        # whole frame (Y, Cb, Cr) against the reference's pvq_theta (src/pvq_encoder.c:333-641)
        assert C.compare_decisions(dec, want, frame=i, frames=F) == [], (content, i)
    print("%s: %d priced decisions left to the host libm in 3 steps of %d frames" % (content, reruns, F))


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
def test_device_priced_host_libm_path_forced():
    """With the margin scaled up every contested decision is listed and re-decided from
    rates the HOST libm computes: same pixels, and the path really ran."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.for_quality(40)
    full = b.natural_like_frame_np(5, 99)
    pw, ph = 312, 180
    pics = [full[0][:ph, :pw], full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]]
    try:
        for cfl in (False, True):
            cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=cfl)
            for scale in (1., 1e7, 1e12):
                D.set_price_tol_scale(scale)
                for serial in (False, True):
                    gpu, reruns = C.gpu_device_priced(D, qt, pics, pw, ph, chroma_cfl=cfl, serial=serial)
                    assert C.compare_frame(gpu, cpu) == [], (cfl, scale, serial)
                    if scale > 1e10:
                        assert reruns > 100, reruns
    finally:
        D.set_price_tol_scale(1.)


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
def test_theta_margin_rerun_inside_priced_steps():
    """The other deferred host path of a step: bands whose theta the device cannot be trusted
    with (margin forced wide, device theta deliberately off by one) are re-run with the
    host's acos one step late, and - the band stage having made the priced choice of the
    per-lane bands itself - every band is then decided again from the candidate records.
    Same pixels as the compiled reference, pipelined and serial."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.for_quality(40)
    full = b.natural_like_frame_np(6, 5)
    pw, ph = 312, 180
    pics = [full[0][:ph, :pw], full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]]
    cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=True)
    D.pvq_ref_set_theta_margin(0.25, True)
    try:
        for serial in (False, True):
            luma = np.ascontiguousarray(pics[0])[None]
            chroma = np.stack([pics[1], pics[2]])
            pipe = D.Pipe(qt, 1, pw, ph, chroma_cfl=True, serial=serial, price=True)
            try:
                pipe.set_pictures(luma, chroma)
                for _ in range(3):
                    pipe.step()
                pipe.flush()
                pipe.sync()
                assert pipe.theta_reruns() > 100, pipe.theta_reruns()
                gpu = [[pipe.read(D.BUF_RECON, 0, bs).reshape(1, pipe.H, pipe.W) for bs in range(5)],
                       [pipe.read(D.BUF_RECON, 1, bs).reshape(2, pipe.H // 2, pipe.W // 2) for bs in range(4)]]
            finally:
                pipe.destroy()
            assert C.compare_frame(gpu, cpu) == [], serial
    finally:
        D.pvq_ref_set_theta_margin(0, False)


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
def test_whole_frame_noref_chroma_and_ragged_size():
    """Chroma through the no-reference stage, and a picture whose size is not a
    multiple of the superblock (padding + gated split filters), whole frame."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.for_quality(40)
    full = b.synth_frame_np(5, 99)
    pw, ph = 312, 180
    pics = [full[0][:ph, :pw], full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]]
    for cfl in (False, True):
        cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=cfl)
        gpu = C.gpu_priced_frame(D, qt, pics, pw, ph, chroma_cfl=cfl)
        assert C.compare_frame(gpu, cpu) == [], cfl


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
@pytest.mark.parametrize("size", [(312, 180, 40), (1920, 1080, 20)])
def test_inter_frame_step_equals_compiled_reference(size):
    """odhip_pipe_config.inter: an INTER frame - luma and chroma through pvq_theta with
    is_keyframe = 0 against the pyramid of a prediction picture (here: the same scene half a
    phase later, a plausible motion-compensated prediction), the choice priced on the device,
    positions PVQ never codes taken from the prediction.  Every reconstructed pixel of every
    level of Y, Cb, Cr against the reference's own C functions run the same way."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    pw, ph, quality = size
    qt = D.QuantTables.for_quality(quality)
    cur = b.natural_like_frame_np(8, 3)
    prev = b.natural_like_frame_np(8, 3)
    rng = np.random.RandomState(4)
    pics, pred = [], []
    for pl, pp, (w_, h_) in zip(cur, prev, ((pw, ph), (pw // 2, ph // 2), (pw // 2, ph // 2))):
        pics.append(np.ascontiguousarray(pl[:h_, :w_]))
        # the prediction: the picture shifted by one sample with a little noise
        q = np.roll(pp[:h_, :w_].astype(np.int32), 1, axis=1) + rng.randint(-6, 7, size=(h_, w_))
        pred.append(np.clip(q, 0, 255).astype(np.uint8))
    cpu, blocks, _ = C.cpu_frame(qt, pics, pw, ph, inter_pred=pred)
    gpu, _ = C.gpu_device_priced(D, qt, pics, pw, ph, inter_pred=pred, steps=3)
    assert C.compare_frame(gpu, cpu) == [], size
    # the prediction matters: coding the same picture as a keyframe gives other pixels
    key, _, _ = C.cpu_frame(qt, pics, pw, ph)
    assert not np.array_equal(key[0][2], cpu[0][2])


def test_fed_pictures_equal_resident_pictures():
    """odhip_pipe_feed: a different set of pictures for every step, copied from pinned host
    memory into the back buffers while the previous steps compute.  After each of six steps
    the pipe must hold exactly what a pipe with those pictures resident (set_pictures) holds -
    reconstructions, choice records, band records (pulse-vector slots of pruned candidates keep
    whatever an earlier step left there, so those buffers depend on the history) - i.e. no copy
    ever lands in a buffer a padding kernel is still reading and no step starts before its
    pictures arrived."""
    import torch
    import daala_amd as D
    D.init(0)
    b = _bench()
    qt = D.QuantTables.load()
    F = 2
    pw, ph = 640, 360
    gens = (b.synth_frame_np, b.natural_like_frame_np)

    def pictures(k):
        fr = [b.picture_planes(gens[(k + i) % 2](10 * k + i, 7 + k)) for i in range(F)]
        luma = np.stack([f[0][:ph, :pw] for f in fr])
        chroma = np.concatenate([np.stack([f[1][:ph // 2, :pw // 2] for f in fr]),
                                 np.stack([f[2][:ph // 2, :pw // 2] for f in fr])])
        return np.ascontiguousarray(luma), np.ascontiguousarray(chroma)

    sets = [pictures(k) for k in range(6)]
    pinned = [(torch.from_numpy(l).pin_memory(), torch.from_numpy(c).pin_memory()) for l, c in sets]
    fed = D.Pipe(qt, F, pw, ph, chroma_cfl=True, price=True)
    resident = D.Pipe(qt, F, pw, ph, chroma_cfl=True, price=True, serial=True)
    try:
        # all six steps in flight, then the last state; and step by step
        for l, c in pinned:
            fed.feed(l, c)
            fed.step()
        fed.flush()
        fed.sync()
        resident.set_pictures(*sets[5])
        resident.step()
        resident.flush()
        resident.sync()
        want = _dump(D, resident, True)
        got = _dump(D, fed, True)
        for key in want:
            if key[0] in ("recon", "choice", "band"):
                assert np.array_equal(got[key], want[key]), ("in flight", key)
        assert np.array_equal(fed.read(D.BUF_PIC, 0), sets[5][0].ravel())
        for k in (1, 4, 2):
            fed.feed(*pinned[k])
            fed.step()
            fed.flush()
            fed.sync()
            resident.set_pictures(*sets[k])
            resident.step()
            resident.flush()
            resident.sync()
            want = _dump(D, resident, True)
            got = _dump(D, fed, True)
            for key in want:
                if key[0] in ("recon", "choice", "band"):
                    assert np.array_equal(got[key], want[key]), (k, key)
    finally:
        fed.destroy()
        resident.destroy()


def test_two_contexts_keep_their_own_scratch():
    """Two inverse calls in flight on two streams with one context each reproduce the
    serial results (the API-level form of (i))."""
    import torch
    import daala_amd as D
    D.init(0)
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(5)
    coefs = [torch.from_numpy((rng.randint(-300, 300, size=(4, 1088, 1920)) * 16).astype(np.int32)).to(dev),
             torch.from_numpy((rng.randint(-300, 300, size=(8, 544, 960)) * 16).astype(np.int32)).to(dev)]
    want = [D.inverse_level(coefs[0], 0, 2, 1920, 1080), D.inverse_level(coefs[1], 1, 1, 1920, 1080)]
    torch.cuda.synchronize()
    ctxs = [D.Context(0), D.Context(0)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [torch.empty_like(w) for w in want]
    for _ in range(6):
        for i in (0, 1):
            with ctxs[i], torch.cuda.stream(streams[i]):
                D.inverse_level(coefs[i], i, 2 - i, 1920, 1080, out=outs[i])
    torch.cuda.synchronize()
    for i in (0, 1):
        assert torch.equal(outs[i], want[i])
    for c in ctxs:
        c.destroy()


@pytest.mark.skipif(ref() is None, reason="oracle/_ref (the compiled reference) not present")
@pytest.mark.parametrize("quality,masking,hvs", [(1, 1, 1), (5, 1, 1), (40, 1, 1), (100, 1, 1), (20, 0, 1), (20, 1, 0),
                                                 (511, 1, 1)])
def test_device_priced_step_across_quantisers(quality, masking, hvs):
    """The priced step (every band decided inside its search, the no-reference bands in place) against the
    compiled reference's pvq_theta at other operating points than -v 20: fine and coarse quantisers (K from 1 to
    hundreds of pulses), activity masking off (beta = 1 everywhere), the flat quantisation matrices - pixels of
    every level and gain / theta / K / pulses of every band, a 640x360 picture of each content type."""
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.for_quality(quality, use_masking=masking, hvs_qm=hvs)
    pw, ph = 640, 360
    for gen in (b.synth_frame_np, b.natural_like_frame_np):
        full = b.picture_planes(gen(2, 1717))
        pics = [full[0][:ph, :pw], full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]]
        gpu, _, dec = C.gpu_device_priced(D, qt, pics, pw, ph, chroma_cfl=True, decisions=True)
        want = []
        cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=True, decisions=want)
        assert C.compare_frame(gpu, cpu) == [], (quality, masking, hvs, gen.__name__)
        assert C.compare_decisions(dec, want) == [], (quality, masking, hvs, gen.__name__)


def test_exported_decisions_are_the_device_buffers():
    """odhip_pipe_set_export (the output side of the PCIe-inclusive rate): fed steps whose decisions - the
    record and the pulses of every band, compacted on the device (export_kernels.hip: 4- / 8-byte records + 16-bit
    (position, count) words) - leave for pinned host memory on a third stream.  After the sync that follows
    step i the host buffer DECODES (daala_amd.decode_export_sections, the reference-side reader of the format)
    to exactly what the dense device buffers of step i hold (gain index, theta, its range, K, skip and every
    pulse of every band of every level, read back through odhip_pipe_read), for odd and even steps (the luma
    outputs are double-buffered by step parity, the chroma ones shared); far fewer bytes cross the bus than
    the dense vectors hold; the reconstruction is the one of the same steps without export; a pipe mode that
    cannot export says so."""
    import torch
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.load()
    F, pw, ph = 2, 640, 360
    full = [b.picture_planes(b.synth_frame_np(i, 99)) for i in range(F + 1)]
    def pics(k0):
        lum = np.stack([full[k0 + i][0][:ph, :pw] for i in range(F - 0)][:F])
        chr_ = np.concatenate([np.stack([full[k0 + i][1][:ph // 2, :pw // 2] for i in range(F)]),
                               np.stack([full[k0 + i][2][:ph // 2, :pw // 2] for i in range(F)])])
        return np.ascontiguousarray(lum), np.ascontiguousarray(chr_)
    sets = [pics(0), pics(1)]
    pipe = D.Pipe(qt, F, pw, ph, chroma_cfl=True, price=True)
    plain = D.Pipe(qt, F, pw, ph, chroma_cfl=True, price=True)
    nbytes = pipe.export_bytes()
    lay = pipe.export_layout()
    assert nbytes == lay["total_bytes"] > lay["fixed_bytes"] > 0 and lay["nsections"] == 9
    host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    pinned = [(torch.from_numpy(l).pin_memory(), torch.from_numpy(c).pin_memory()) for l, c in sets]
    pipe.set_export(host)
    for step in range(5):
        l, c = pinned[step & 1]
        pipe.feed(l, c)
        pipe.step()
        if step < 2:
            continue          # from the third step on every wait of the export path has a predecessor
        pipe.flush()
        pipe.sync()
        got = pipe.decode_export(host.numpy())
        want = C.gpu_decisions(D, pipe)
        assert set(got) == set(want)
        dense = 0
        for key in sorted(want):
            yw, bw, cw = want[key]
            yg, bg, cg = got[key]
            assert np.array_equal(cg, cw), (step, key, "coded")
            assert np.array_equal(bg[..., :3], bw[..., :3]), (step, key, "gain index / theta / max_theta")
            # K is not exported: a coded band's K is the sum of its counts' magnitudes
            assert np.array_equal(bg[..., 3][cw], bw[..., 3][cw]), (step, key, "K of the coded bands")
            assert np.array_equal(yg, yw), (step, key, "pulses")
            dense += yw.size * 2
        shipped = pipe.export_shipped_bytes(host.numpy())
        assert shipped < dense / 2, (shipped, dense)
        assert pipe.export_stale() == 0
        plain.set_pictures(*sets[step & 1])
        plain.step()
        plain.flush()
        plain.sync()
        for set_ in (0, 1):
            for bs in range(5 - set_):
                assert np.array_equal(pipe.read(D.BUF_RECON, set_, bs), plain.read(D.BUF_RECON, set_, bs))
    pipe.set_export(None)
    pipe.destroy()
    plain.destroy()
    noref = D.Pipe(qt, F, pw, ph, chroma_cfl=False, price=True)
    assert noref.export_bytes() == 0
    with pytest.raises(Exception):
        noref.set_export(host)
    noref.destroy()


def test_export_words_escape_and_group_order():
    """The export format on hand-made buffers (odhip_export_pack / odhip_export_ship called directly): pulse counts
    beyond +-255 take the two-word escape, a theta winner's last position is not exported, null and skipped bands
    carry no words, groups of 256 bands decode whatever order their workgroups finished in, and the ship kernel
    moves exactly the used prefix of each stream."""
    import ctypes
    import torch
    import daala_amd as D
    D.init(0)
    L = D.lib()
    rng = np.random.RandomState(77)
    bs_list = [0, 2, 1]
    with_ref = [0, 0, 1]
    nblocks = [1500, 300, 700]
    lay_c, lay = D.export_layout_make(nblocks, bs_list, with_ref)
    dev = torch.zeros(lay["total_bytes"], dtype=torch.uint8, device="cuda")
    host = torch.zeros(lay["total_bytes"], dtype=torch.uint8).pin_memory()
    host.fill_(0xAB)
    assert L.odhip_export_begin(ctypes.c_void_p(dev.data_ptr()), ctypes.byref(lay_c), None) == 0
    want = []
    keep = []
    for si, (bs, wr, B) in enumerate(zip(bs_list, with_ref, nblocks)):
        nb, offs, ln = D.pvq_band_layout(bs)
        slots = 3 if wr else 2
        y = np.zeros((slots, B, ln), np.int16)
        mask = rng.rand(slots, B, ln) < 0.08
        y[mask] = rng.randint(-6, 7, size=int(mask.sum()))
        big = rng.rand(slots, B, ln) < 0.004
        y[big] = rng.choice([-256, 256, -300, 1000, -32768, 32767, 255, -255], size=int(big.sum()))
        yw = np.zeros((B, ln), np.int32)
        band = np.zeros((B, nb, 4), np.int32)
        coded = np.ones((B, nb), bool)
        if wr:
            ch = np.zeros((B, nb, 16), np.int32)
            ch[..., 2] = rng.randint(0, 2, size=(B, nb))                      # noref
            ch[..., 3] = rng.randint(0, 40, size=(B, nb))                     # itheta
            ch[..., 4] = ch[..., 3] + rng.randint(1, 9, size=(B, nb))         # max_theta
            ch[..., 5] = rng.randint(0, 500, size=(B, nb))                    # k (not exported)
            ch[..., 6] = rng.choice([0, 0, 0, 1, 2], size=(B, nb))            # skip
            ch[..., 7] = rng.randint(-3, 300, size=(B, nb))                   # coded gain index
            ch[..., 9] = rng.choice([-1, 0, 1, 2], size=(B, nb))              # yslot
            for i in range(nb):
                a, b_ = offs[i], offs[i + 1]
                on = (ch[:, i, 6] == 0) & (ch[:, i, 9] >= 0)
                idx = np.nonzero(on)[0]
                v = y[ch[idx, i, 9], idx, a:b_].astype(np.int32)
                v[ch[idx, i, 2] == 0, -1] = 0
                yw[idx, a:b_] = v
            band[..., 0] = ch[..., 7]
            band[..., 1] = ch[..., 3]
            band[..., 2] = ch[..., 4]
            for i in range(nb):
                band[:, i, 3] = np.abs(yw[:, offs[i]:offs[i + 1]]).sum(axis=1)
            coded = ch[..., 6] == 0
        else:
            ch = np.zeros((B, nb, 4), np.int32)
            ch[..., 0] = rng.randint(0, 2, size=(B, nb))
            ch[..., 1] = rng.choice([0, 1, 2, 7, 300], size=(B, nb))
            for i in range(nb):
                a, b_ = offs[i], offs[i + 1]
                idx = np.nonzero(ch[:, i, 1] != 0)[0]
                yw[idx, a:b_] = y[ch[idx, i, 0], idx, a:b_]
                band[:, i, 3] = np.abs(yw[:, a:b_]).sum(axis=1)
            band[..., 0] = ch[..., 1]
            band[..., 1] = -1
        d_ch = torch.from_numpy(ch).cuda()
        d_y = torch.from_numpy(y).cuda()
        keep += [d_ch, d_y]
        assert L.odhip_export_pack(ctypes.c_void_p(dev.data_ptr()), ctypes.byref(lay_c), si, ctypes.c_void_p(d_ch.data_ptr()),
                                   ctypes.c_void_p(d_y.data_ptr()), ctypes.c_long(B), bs, wr, None) == 0
        want.append((yw, band, coded))
    assert L.odhip_export_ship(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(dev.data_ptr()), ctypes.byref(lay_c), None) == 0
    torch.cuda.synchronize()
    h = host.numpy()
    got = D.decode_export_sections(h, lay)
    for si in range(3):
        for name, g_, w_ in zip(("pulses", "record", "coded"), got[si], want[si]):
            assert np.array_equal(g_, w_), (si, name)
    # only the used prefix of a stream crossed: the byte after its last 16-byte vector still holds the fill pattern
    totals = h[:64].view(np.uint32)
    for si, sec in enumerate(lay["sections"]):
        used = (int(totals[si]) * 2 + 15) // 16 * 16
        assert 0 < used < sec["cap_words"] * 2
        assert (h[sec["stream_off"] + used:sec["stream_off"] + sec["cap_words"] * 2] == 0xAB).all()
    assert int(h[64:128].view(np.uint32).sum()) == 0          # no overflow flag


def test_export_follows_a_late_resolve():
    """a chroma band that the host-libm resolve re-decides one step late changes choices and pulses
    AFTER they were packed for the host.  With the margins forced wide (theta margin + a deliberately wrong device
    theta; price margin scaled up) hundreds of bands are re-decided per step: `step, flush, sync, read` must decode
    to the FINAL device buffers (the flush re-exports the step), and steps issued back to back count their superseded
    exports (odhip_pipe_export_stale)."""
    import torch
    import daala_amd as D
    import _pipeline_check as C
    D.init(0)
    b = _bench()
    qt = D.QuantTables.for_quality(40)
    full = b.natural_like_frame_np(6, 5)
    pw, ph = 312, 180
    luma = np.ascontiguousarray(full[0][:ph, :pw])[None]
    chroma = np.stack([full[1][:ph // 2, :pw // 2], full[2][:ph // 2, :pw // 2]])
    D.pvq_ref_set_theta_margin(0.25, True)
    D.set_price_tol_scale(1e7)
    try:
        pipe = D.Pipe(qt, 1, pw, ph, chroma_cfl=True, price=True)
        try:
            pipe.set_pictures(luma, chroma)
            host = torch.zeros(pipe.export_bytes(), dtype=torch.uint8).pin_memory()
            pipe.set_export(host)
            for step in range(3):
                before = pipe.theta_reruns() + pipe.price_reruns()
                pipe.step()
                pipe.flush()
                pipe.sync()
                assert pipe.theta_reruns() + pipe.price_reruns() > before + 50      # the late paths really ran
                got = pipe.decode_export(host.numpy())
                want = C.gpu_decisions(D, pipe)
                for key in sorted(want):
                    yw, bw, cw = want[key]
                    yg, bg, cg = got[key]
                    assert np.array_equal(cg, cw), (step, key)
                    assert np.array_equal(bg[..., :3], bw[..., :3]), (step, key)
                    assert np.array_equal(bg[..., 3][cw], bw[..., 3][cw]), (step, key)
                    assert np.array_equal(yg, yw), (step, key, "pulses after the resolve")
            assert pipe.export_stale() == 0
            # back to back: the resolve of step i runs inside step i + 1, after the host could have read the buffer
            for _ in range(3):
                pipe.step()
            pipe.flush()
            pipe.sync()
            assert pipe.export_stale() >= 2, pipe.export_stale()
            pipe.set_export(None)
        finally:
            pipe.destroy()
    finally:
        D.pvq_ref_set_theta_margin(0, False)
        D.set_price_tol_scale(1.)


def test_pulse_count_above_int16_is_reported_not_silently_different():
    """The reference's pulse count K is an int; the pulse vectors here are int16 (ODHIP_PVQ_MAX_K = 32767).  Found by
    tests/soak/parity_soak.py: at a coded quantiser of 8 (below encoder_example's range), flat matrices, no masking, a
    saturated edge inside a 64x64 block asks for K > 35 000 in band 0; such a candidate is not searched here, so the
    band would differ from the reference's.  It must not differ SILENTLY: the device counts every such band and
    odhip_pipe_sync returns ODHIP_ERANGE (PulseRangeError)."""
    import daala_amd as D
    D.init(0)
    D.pvq_k_range_take()                                   # (clear whatever an earlier test left)
    pw, ph = 128, 128
    luma = np.zeros((1, ph, pw), np.uint8)
    luma[:, :, 32:64] = 255                                # a saturated step inside every 64x64 block
    luma[:, :, 96:128] = 255
    chroma = np.full((2, ph // 2, pw // 2), 128, np.uint8)
    for (base, quantizer, expect) in ((16, 8, True), (320, 243, False)):
        qt = D.QuantTables(base, quantizer, 0, 0)
        pipe = D.Pipe(qt, 1, pw, ph, chroma_cfl=True, price=True)
        try:
            pipe.set_pictures(luma, chroma)
            pipe.step()
            pipe.flush()
            if expect:
                with pytest.raises(D.PulseRangeError):
                    pipe.sync()
                assert pipe.k_range() > 0
                pipe.sync()                                # reported once: the counters were taken
            else:
                pipe.sync()
                assert pipe.k_range() == 0
        finally:
            pipe.destroy()
    assert D.pvq_k_range_take() == (0, 0)
