"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle.

Tolerances (BASELINE.json north_star): logits within 1e-4 of the CPU reference path, argmax labels
identical; MFCC within 1e-4 (float32 kernel vs float64 psf arithmetic); integer/bit-level pieces
(PCM scaling + pre-emphasis, labels, run-to-run determinism) exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import synth_clips
from oracle import dscnn as o_dscnn
from oracle import psf_mfcc as o_mfcc

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def native():
    from kws import _native

    return _native


def make_ctx(native):
    """Context enqueuing on torch's current stream, so tensor fills/copies and kernels are ordered."""
    c = native.Context(0)
    c.use_torch_stream()
    return c


@pytest.fixture(scope="module")
def ctx(native):
    c = make_ctx(native)
    yield c
    c.close()


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def state_from_blob(blob):
    st, off = {}, 0
    for k, shp in o_dscnn.state_shapes(12).items():
        n = int(np.prod(shp))
        st[k] = torch.from_numpy(blob[off:off + n].reshape(shp).copy())
        off += n
    return st


def he_model(e2e_golden):
    """kws.libs.models.DepthwiseSeparableConv holding the golden signal-preserving weights."""
    from kws.libs.models import DepthwiseSeparableConv

    model = DepthwiseSeparableConv(num_classes=12)
    model.load_state_dict(state_from_blob(e2e_golden["he.blob"]))
    return model


def peak_ratio(clip):
    """log(largest bin power) - min log(mel energy) per frame, from the oracle's float64 spectrum: the quantity the float32 kernel's
    precision flag measures (include/kws_hip.h, kws_set_frontend_refine).  An all-zero frame gives -inf (never flagged)."""
    sig = o_mfcc.fix_length(o_mfcc.pcm16_to_float(clip), 16000)
    feat, _ = o_mfcc.fbank(sig)
    ps = o_mfcc.powspec(o_mfcc.framesig(o_mfcc.preemphasis(sig, 0.97), 400, 160), 512)
    with np.errstate(divide="ignore"):
        return np.log(ps.max(axis=1)) - np.log(feat).min(axis=1)


def gpu_mfcc(ctx, dev, clips):
    wav = torch.from_numpy(np.ascontiguousarray(clips)).to(dev)
    nf, nc = ctx.frontend_shape()
    out = torch.empty((len(clips), 1, nf, nc), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(wav, out)
    ctx.sync()
    return out.cpu().numpy()


def gpu_infer(ctx, dev, clips):
    wav = torch.from_numpy(np.ascontiguousarray(clips)).to(dev)
    logits = torch.empty((len(clips), 12), dtype=torch.float32, device=dev)
    labels = torch.empty((len(clips),), dtype=torch.int32, device=dev)
    ctx.infer_i16(wav, logits, labels)
    ctx.sync()
    return logits.cpu().numpy(), labels.cpu().numpy()


# ------------------------------------------------------------------------------------------- MFCC
def test_mfcc_golden_clips(ctx, dev, sigproc_golden):
    clips = sigproc_golden["clips"]
    got = gpu_mfcc(ctx, dev, clips)
    want = np.stack([o_mfcc.extract_features_pcm16(c) for c in clips])
    assert got.shape == (len(clips), 1, 99, 10) and got.dtype == np.float32
    err = np.abs(got[:, 0] - want).max(axis=(1, 2))
    for name, e in zip(sigproc_golden["names"], err):
        assert e <= TOL, f"{name}: MFCC max abs err {e}"
    # all-zero clip: c0 = log(eps) to float32 rounding, every other coefficient exactly 0
    assert np.all(np.abs(got[0, 0, :, 0] - (-36.04365338911715)) < 1e-5)
    assert np.all(got[0, 0, :, 1:] == 0.0)


def test_mfcc_diverse_clips_incl_level_steps(native, ctx, dev, e2e_golden):
    """The 48 diverse golden clips: tones, chirps, noise at four levels and gated bursts over a quiet floor -- EVERY frame
    within 1e-4 of the float64 oracle.  The bursts put a loud and a quiet frame into one packed transform (without the
    per-frame level equalisation the quiet frame's cepstra were off by up to 2e-2), and a clean tone over a quiet floor
    spreads its mel bands over 80 dB inside one frame, where a float32 transform (rounding noise ~138 dB below the strongest
    component) misses by up to 6e-4: such frames are flagged by the float32 kernel (weakest mel band more than 10.2 log units of
    power below the largest spectral bin) and recomputed in float64 by the refinement launch.  No
    per-frame allowance is left in this gate."""
    clips, names = e2e_golden["clips"], e2e_golden["names"]
    before = ctx.frontend_stats()
    got = gpu_mfcc(ctx, dev, clips)[:, 0]
    total, refined, last = ctx.frontend_stats()
    want = np.stack([o_mfcc.extract_features_pcm16(c) for c in clips])
    err = np.abs(got - want).max(axis=2)                      # [clip, frame]
    over = err > TOL
    assert not over.any(), [(str(names[i]), int(f), float(err[i, f])) for i, f in zip(*np.nonzero(over))][:5]
    # what the refinement did: counted per call, a few percent of these frames, none of them on white noise at full scale
    assert total - before[0] == clips.shape[0] * 99 and refined - before[1] == last
    assert 0.03 * err.size <= last <= 0.25 * err.size, last
    # ... and nothing else: with the refinement off the other rows keep their bits, and the float32 kernel alone misses
    ctx.set_frontend_refine(0.0)
    try:
        plain = gpu_mfcc(ctx, dev, clips)[:, 0]
    finally:
        ctx.set_frontend_refine(native.FE_REFINE_SPAN_DEFAULT)
    changed = np.any(plain != got, axis=2)
    assert changed.sum() <= last
    ratio = np.stack([peak_ratio(c) for c in clips])
    assert last == pytest.approx((ratio > native.FE_REFINE_SPAN_DEFAULT).sum(), abs=4)  # the flag is the quantity the header names
    assert not changed[ratio < native.FE_REFINE_SPAN_DEFAULT - 0.01].any()      # below the threshold: bit-identical to the float32 kernel
    assert changed[ratio > native.FE_REFINE_SPAN_DEFAULT + 0.01].mean() > 0.98  # above it: recomputed (a float64 row may round to the same floats)
    assert np.abs(plain - want).max() > 2 * TOL                                # the float32 kernel alone does not meet the gate here
    # a frame's result must not depend on its partner in the packed pair: frame 2k of a clip whose odd frames are loud
    loud = synth_clips(1, 12)[0].astype(np.int32)
    quiet = np.clip(np.round(np.random.default_rng(13).standard_normal(16000) * 3), -32768, 32767).astype(np.int32)
    t = np.arange(16000)
    env = ((t // 160) % 2 == 1)                                    # alternate 10 ms loud / 10 ms quiet
    mixed = np.where(env, loud, quiet).astype(np.int16)
    got2 = gpu_mfcc(ctx, dev, mixed[None])[0, 0]
    want2 = o_mfcc.extract_features_pcm16(mixed)
    assert np.abs(got2 - want2).max() <= TOL


def test_mfcc_audit_clips_every_frame_within_tol(native, ctx, dev):
    """Eight clips of the 3.56 M-frame audit (tools/fe_precision_audit.py, profiles/r03_precision_audit.txt) on which round 3's
    FIRST flag -- the span max - min of a frame's log-mel values -- failed: four with a frame that missed 1e-4 at a span threshold
    of 12.0, four more that missed it (1.2e-4 .. 1.5e-4) at 11.5.  The flag the kernel has now measures the weakest band against
    the frame's largest spectral BIN (what the float32 transform's noise is relative to): every frame of the eight is within
    the tolerance, every named frame is over the threshold by a margin, and with the refinement off they miss again."""
    named, clips = [], []
    for name in ("audit_hard_clips.npz", "audit_exception_clips.npz"):
        d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
        for c, fr in zip(d["clips"], d["frames"]):
            named += [(len(clips), int(f)) for f in np.atleast_1d(fr) if f >= 0]
            clips.append(c)
    clips = np.stack(clips)
    want = np.stack([o_mfcc.extract_features_pcm16(c) for c in clips])
    got = gpu_mfcc(ctx, dev, clips)[:, 0]
    err = np.abs(got - want).max(axis=2)
    assert err.max() <= TOL, (float(err.max()), np.unravel_index(err.argmax(), err.shape))
    ratio = np.stack([peak_ratio(c) for c in clips])
    assert all(ratio[i, f] > native.FE_REFINE_SPAN_DEFAULT + 0.5 for i, f in named), [float(ratio[i, f]) for i, f in named]
    ctx.set_frontend_refine(0.0)
    try:
        raw = gpu_mfcc(ctx, dev, clips)[:, 0]
    finally:
        ctx.set_frontend_refine(native.FE_REFINE_SPAN_DEFAULT)
    err_raw = np.abs(raw - want).max(axis=2)
    assert all(err_raw[i, f] > TOL for i, f in named), [(i, f, float(err_raw[i, f])) for i, f in named]


@pytest.mark.parametrize("kind,seed", [("uniform", 0), ("gauss", 1)])
def test_mfcc_random_batch(ctx, dev, kind, seed):
    clips = synth_clips(96, seed, kind)
    got = gpu_mfcc(ctx, dev, clips)[:, 0]
    want = np.stack([o_mfcc.extract_features_pcm16(c) for c in clips])
    assert np.abs(got - want).max() <= TOL


def test_mfcc_f32_entry_equals_i16_entry(ctx, dev):
    clips = synth_clips(5, 3)
    a = gpu_mfcc(ctx, dev, clips)
    x = torch.from_numpy(o_mfcc.pcm16_to_float(clips)).to(dev)
    out = torch.empty((5, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_f32(x, out)
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), a)


def test_mfcc_unaligned_pointer_takes_scalar_path(ctx, dev):
    clips = synth_clips(3, 4)
    ref = gpu_mfcc(ctx, dev, clips)
    buf = torch.zeros(3 * 16000 + 1, dtype=torch.int16, device=dev)
    view = buf[1:].view(3, 16000)
    view.copy_(torch.from_numpy(clips))
    out = torch.empty((3, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(view, out)
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_mfcc_other_geometry(native, dev):
    """Half-second clips, 13 cepstra, 40 filters, 8 kHz: exercises the table builder and ragged tail."""
    c = make_ctx(native)
    try:
        spec = o_mfcc.FrontendSpec(sample_rate=8000, n_samples=4000, winlen=0.032, winstep=0.012, nfft=512, nfilt=40, numcep=13)
        c.set_frontend(sample_rate=8000, n_samples=4000, frame_len=spec.frame_len, frame_step=spec.frame_step,
                       nfft=512, nfilt=40, numcep=13)
        assert c.frontend_shape() == (spec.num_frames, 13)
        clips = np.random.default_rng(9).integers(-20000, 20000, size=(7, 4000), dtype=np.int16)
        got = gpu_mfcc(c, dev, clips)[:, 0]
        want = np.stack([o_mfcc.mfcc(o_mfcc.pcm16_to_float(x), spec) for x in clips])
        assert np.abs(got - want).max() <= TOL
    finally:
        c.close()


@pytest.mark.parametrize("frame_len", [320, 384, 385, 400, 447, 448, 449, 512])
def test_mfcc_frame_length_boundaries(native, dev, frame_len):
    """The float32 kernel has two instantiations: frame lengths in (384, 448] (the reference's 400: the eighth 64-sample
    block of a frame is empty, the seventh is cut by a lane bound) and every other length up to 512.  Both sides of each
    boundary, odd frame counts (the last pair has no partner) and a batch that is not a multiple of anything."""
    c = make_ctx(native)
    try:
        n = 16000 // 2 + 37
        spec = o_mfcc.FrontendSpec(sample_rate=16000, n_samples=n, winlen=frame_len / 16000.0, winstep=0.01, nfft=512)
        assert spec.frame_len == frame_len
        c.set_frontend(sample_rate=16000, n_samples=n, frame_len=frame_len, frame_step=160, nfft=512)
        clips = np.random.default_rng(frame_len).integers(-20000, 20000, size=(5, n), dtype=np.int16)
        clips[3, : n // 2] = 0  # leading silence: all-zero frames beside live ones
        got = gpu_mfcc(c, dev, clips)[:, 0]
        want = np.stack([o_mfcc.mfcc(o_mfcc.pcm16_to_float(x), spec) for x in clips])
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= TOL
    finally:
        c.close()


def test_frontend_rejects_unsupported(native):
    from kws.common.errors import AudioProcessingError

    c = native.Context(0)
    try:
        with pytest.raises(AudioProcessingError):
            c.set_frontend(nfft=8192)                 # beyond the float64 kernel's LDS
        with pytest.raises(AudioProcessingError):
            c.set_frontend(nfft=3000)                 # not a power of two and > 2048
        with pytest.raises(AudioProcessingError):
            c.set_frontend(nfilt=65)
        assert c.frontend_shape() == (99, 10)  # a refused configuration leaves the previous one intact
        assert c.frontend_math() == native.FE_F32
    finally:
        c.close()


PRECISE_TOL = 1e-5  # float64 front end: the float32 rounding of cepstra of magnitude <= 64 is 3.8e-6


def test_mfcc_float64_frontend_matches_everywhere(native, dev, e2e_golden, sigproc_golden):
    """KWS_FE_F64 (everything after framing in float64, as psf): every golden clip -- the clean tones and gated bursts on
    which the float32 kernel needs its dynamic-range allowance included -- within 1e-5 of the oracle, no allowance; the
    fused path with it agrees with the reference model's golden logits."""
    c = make_ctx(native)
    try:
        c.set_frontend_math(native.FE_F64)
        assert c.frontend_math() == native.FE_F64
        clips = np.concatenate([e2e_golden["clips"], synth_clips(16, 0, "uniform"), synth_clips(16, 1, "gauss")])
        got = gpu_mfcc(c, dev, clips)[:, 0]
        want = np.stack([o_mfcc.extract_features_pcm16(x) for x in clips])
        err = np.abs(got - want).max(axis=(1, 2))
        assert err.max() <= PRECISE_TOL, (int(err.argmax()), float(err.max()))
        assert np.all(np.abs(got[0, :, 1:]) < 1e-10) and np.all(np.abs(got[0, :, 0] + 36.04365338911715) < 1e-5)   # all-zero clip
        c.load_dscnn(e2e_golden["he.blob"], 12)
        logits, labels = gpu_infer(c, dev, e2e_golden["clips"])
        assert np.abs(logits - e2e_golden["he.logits"][8:]).max() <= 2e-5
        assert np.array_equal(labels, e2e_golden["he.label"][8:])
        x = torch.from_numpy(o_mfcc.pcm16_to_float(clips[:5])).to(dev)           # float32-signal entry point
        out = torch.empty((5, 1, 99, 10), dtype=torch.float32, device=dev)
        c.mfcc_f32(x, out)
        c.sync()
        assert np.array_equal(out.cpu().numpy()[:, 0], got[:5])
    finally:
        c.close()


@pytest.mark.parametrize("winlen,nfft", [(0.04, 640), (0.064, 1024), (0.025, 512), (0.1, 1600)])
def test_mfcc_other_transform_lengths(dev, winlen, nfft):
    """extract_features with winlen * samplerate > 512: the reference then uses nfft = int(winlen * samplerate)
    (audio_processor.py:268), e.g. 640 -- not a power of two (direct DFT), 1024 (FFT), 1600.  float64 kernel vs oracle."""
    from kws.libs.audio_processor import AudioConfig, AudioProcessor

    ap = AudioProcessor(None, AudioConfig(), precise=(nfft == 512))
    clip = synth_clips(1, 40, "gauss")[0] if nfft != 1600 else synth_clips(1, 41, "uniform")[0]
    clip[2000:2400] = 0
    feat = ap.extract_features(o_mfcc.pcm16_to_float(clip), winlen=winlen)
    spec = o_mfcc.FrontendSpec(winlen=winlen, nfft=max(512, int(winlen * 16000)))
    assert spec.nfft == nfft
    want = o_mfcc.mfcc(o_mfcc.pcm16_to_float(clip), spec)
    assert feat.shape == want.shape == (spec.num_frames, 10)
    assert np.abs(feat - want).max() <= PRECISE_TOL


# ------------------------------------------------------------------------------------------- sigproc operators
def test_sigproc_operators_against_reference_golden(ctx, dev, sigproc_golden):
    from kws.libs.speech_features import sigproc

    g = sigproc_golden
    keep = g["keep_frames"]
    for clip, head, ps_ref in zip(g["clips"], g["preemph_head_f32"], g["powspec_f64"]):
        x = torch.from_numpy(o_mfcc.pcm16_to_float(clip)).to(dev)
        y = sigproc.preemphasis(x, 0.97)
        assert np.array_equal(y[:64].cpu().numpy(), head)  # float32 pre-emphasis is bit-exact
        frames = sigproc.framesig(y, 400, 160, winfunc=lambda n, device=None: torch.ones(n, device=device))
        assert tuple(frames.shape) == (99, 400)
        assert torch.equal(frames[1], y[160:560]) and float(frames[98, 320:].abs().max()) == 0.0
        ps = sigproc.powspec(frames, 512).cpu().numpy().astype(np.float64)
        scale = max(ps_ref.max(), 1e-30)
        assert np.abs(ps[keep] - ps_ref).max() <= 2e-6 * scale
        mag = sigproc.magspec(frames, 512).cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(mag[keep] ** 2 / 512, ps_ref, atol=4e-6 * scale)


@pytest.mark.parametrize("NFFT", [64, 256, 400, 640, 1024, 2048, 4096])
def test_sigproc_any_nfft(dev, NFFT):
    """magspec / powspec take any NFFT (sigproc.py:55-90): frames shorter than NFFT are zero-padded, longer ones truncated;
    float64 transform behind a float32 boundary -> relative error of a float32 rounding."""
    from kws.libs.speech_features import sigproc

    rng = np.random.default_rng(NFFT)
    frames = rng.standard_normal((7, 400))
    fr = torch.from_numpy(frames).to(dev)
    want = np.abs(np.fft.rfft(frames.astype(np.float32).astype(np.float64), NFFT))
    mag = sigproc.magspec(fr, NFFT)
    assert mag.dtype == torch.float64 and tuple(mag.shape) == (7, NFFT // 2 + 1)
    assert np.abs(mag.cpu().numpy() - want).max() <= 2e-7 * want.max()
    ps = sigproc.powspec(fr, NFFT).cpu().numpy()
    assert np.abs(ps - want ** 2 / NFFT).max() <= 3e-7 * (want ** 2 / NFFT).max()


def test_sigproc_reference_unit_test_recipe(dev, sigproc_golden):
    """The reference's own test (tests/kws/libs/speech_features/test_sigproc.py:8-21) with the rectangular
    window, at float32 precision: rand(16000) -> framesig(400,160) -> magspec(512)."""
    from kws.libs.speech_features import sigproc

    np.random.seed(0)
    sig = torch.from_numpy(np.random.rand(16000)).to(dev)  # float64 in, float64 out, float32 inside
    frames = sigproc.framesig(sig, 400, 160, winfunc=lambda n, device=None: torch.ones(n, device=device))
    mag = sigproc.magspec(frames, 512)
    assert mag.dtype == torch.float64
    ref = sigproc_golden["reftest_magspec_f64"]
    np.testing.assert_allclose(mag.cpu().numpy()[sigproc_golden["keep_frames"]], ref, rtol=1e-5, atol=2e-4)
    hann = sigproc.framesig(sig.float(), 400, 160)  # default window is Hann, as in the reference signature
    want = sig.float()[:400] * torch.hann_window(400, device=dev)
    assert torch.allclose(hann[0], want, atol=1e-7)


# ------------------------------------------------------------------------------------------- DS-CNN
def split_act(act):
    sizes = [("conv1", 141), ("dsconv1", 141), ("dsconv2", 245), ("dsconv3", 357)]
    out, off = {}, 0
    for name, p in sizes:
        out[name] = act[:, off:off + 64 * p].reshape(-1, 64, p)
        off += 64 * p
    out["pool"] = act[:, off:off + 64]
    off += 64
    out["dsconv4"] = act[:, off:off + 64 * 477].reshape(-1, 64, 477)
    return out


LAYER_RTOL = 2e-5  # per-layer gate: max abs error <= LAYER_RTOL * max |reference activation| of that layer (no floor)


def golden_case(dscnn_golden, e2e_golden, tag):
    """(blob, x, reference logits, reference labels) of a golden tag; 'he' = signal-preserving weights on 8 random
    maps + the oracle MFCC of 48 diverse clips (e2e_golden.npz), 'n01' / 'default' = the round-1 tags."""
    if tag == "he":
        g = e2e_golden
        x = np.concatenate([g["x_rand"], o_mfcc.collate_pcm16(g["clips"])])
        return g["he.blob"], x, g["he.logits"], g["he.label"]
    g = dscnn_golden
    return g[f"{tag}.blob"], g["x"], g[f"{tag}.logits"], g[f"{tag}.label"]


@pytest.mark.parametrize("tag", ["he", "n01", "default"])
@pytest.mark.parametrize("use_mfma", [0, 1, 4, 5])  # VALU cross-check, f32 MFMA, split-bf16 MFMA, f16-pair MFMA
def test_dscnn_layers_and_logits(native, ctx, dev, dscnn_golden, e2e_golden, tag, use_mfma):
    blob, x_np, ref_logits, ref_label = golden_case(dscnn_golden, e2e_golden, tag)
    ctx.load_dscnn(blob, 12)
    x = torch.from_numpy(x_np).to(dev)
    B = x.shape[0]
    logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
    labels = torch.empty((B,), dtype=torch.int32, device=dev)
    act = torch.zeros((B, native.ACT_FLOATS_PER_CLIP), dtype=torch.float32, device=dev)
    ctx.forward_debug_f32(x, logits, labels, act, use_mfma=use_mfma)
    ctx.sync()
    _, layers = o_dscnn.forward(state_from_blob(blob), torch.from_numpy(x_np), return_layers=True)
    got = split_act(act.cpu().numpy())
    want = {"conv1": layers["conv1"].numpy().reshape(B, 64, -1), "pool": layers["pool"].numpy()}
    for i in range(1, 5):
        want[f"dsconv{i}"] = layers[f"dsconv{i}"].numpy()[:, :, 1:-1, 1:-1].reshape(B, 64, -1)
    for name in ["conv1", "dsconv1", "dsconv2", "dsconv3", "dsconv4", "pool"]:
        scale = float(np.abs(want[name]).max())
        err = float(np.abs(got[name] - want[name]).max())
        assert err <= LAYER_RTOL * scale, f"{tag} mfma={use_mfma} layer {name}: max abs err {err:.3e} (scale {scale:.3e})"
    # against the reference module's own logits (golden), and its labels
    lg = logits.cpu().numpy()
    err = float(np.abs(lg - ref_logits).max())
    assert err <= min(TOL, LAYER_RTOL * max(float(np.abs(ref_logits).max()), 1e-30) * 4), f"{tag}: logits err {err:.3e}"
    if tag == "he":
        # the fixture depends on its input: many classes, logits that move from clip to clip -- and the kernel follows
        assert len(set(ref_label[8:].tolist())) >= 6 and float(ref_logits[8:].std(axis=0).mean()) >= 0.1
        assert np.array_equal(labels.cpu().numpy(), ref_label)
    elif tag == "n01":
        assert np.array_equal(labels.cpu().numpy(), ref_label)


@pytest.mark.parametrize("use_mfma", [0, 1, 4, 5])
def test_argmax_ties_first_maximum_wins(native, ctx, dev, e2e_golden, use_mfma):
    """torch.max(outputs, 1) returns the first maximum (kws/libs/training.py:371).  tie_all: twelve identical class
    rows -> every logit ties -> label 0 for every input; tie_pair: row 11 is a copy of row lo -> wherever lo wins the two
    tie exactly and lo must be reported.  Expected labels come from the reference module itself (make_golden.py)."""
    g = e2e_golden
    x = torch.from_numpy(np.concatenate([g["x_rand"], o_mfcc.collate_pcm16(g["clips"])])).to(dev)
    B = x.shape[0]
    lo, hi = (int(v) for v in g["tie_pair.lo_hi"])
    for tag in ("tie_all", "tie_pair"):
        ctx.load_dscnn(g[f"{tag}.blob"], 12)
        logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
        labels = torch.empty((B,), dtype=torch.int32, device=dev)
        if use_mfma in (4, 5):
            ctx.set_pointwise_math(use_mfma)
            ctx.forward_f32(x, logits, labels)  # the product instantiations
            ctx.set_pointwise_math(native.PW_DEFAULT)
        else:
            act = torch.zeros((B, native.ACT_FLOATS_PER_CLIP), dtype=torch.float32, device=dev)
            ctx.forward_debug_f32(x, logits, labels, act, use_mfma=use_mfma)
        ctx.sync()
        lg, lb = logits.cpu().numpy(), labels.cpu().numpy()
        assert np.abs(lg - g[f"{tag}.logits"]).max() <= TOL
        if tag == "tie_all":
            assert np.all(lg == lg[:, :1]) and np.all(lb == 0)
        else:
            assert np.array_equal(lg[:, lo], lg[:, hi])          # identical rows -> bit-identical logits
            assert np.array_equal(lb, g["tie_pair.label"])
            assert (lb == lo).sum() >= 3 and not (lb == hi).any()


def test_standalone_block_forward(dev, dsblock_golden):
    """DepthwiseSeparableConvBlock.forward on its own (kws_dsblock_forward_f32) against the imported reference module's
    outputs: four shapes incl. 5x5 / stride 2 / padding 2, padding 0, and channel counts off the kernel's tile sizes;
    relative gate 2e-5 of the output's scale; the relu(bias) ring bit-exact."""
    from kws.common.errors import ModelError
    from kws.libs.models import DepthwiseSeparableConvBlock

    g = dsblock_golden
    for i, (ci, co, k, st, pd, B, H, W) in enumerate(g["cases"].tolist()):
        blk = DepthwiseSeparableConvBlock(ci, co, kernel_size=k, stride=st, padding=pd)
        blk.load_state_dict({n: torch.from_numpy(g[f"c{i}.{n}"]) for n in ("depthwise.weight", "depthwise.bias", "pointwise.weight",
                                                                           "pointwise.bias")})
        y = blk(torch.from_numpy(g[f"c{i}.x"]).to(dev)).cpu().numpy()
        want = g[f"c{i}.y"]
        assert y.shape == want.shape
        err = float(np.abs(y - want).max())
        assert err <= LAYER_RTOL * float(np.abs(want).max()), (i, err)
        if pd:
            assert np.array_equal(y[:, :, 0, :], want[:, :, 0, :]) and np.array_equal(y[:, :, :, -1], want[:, :, :, -1])
    with pytest.raises(ModelError):
        blk(torch.zeros(1, 3, 4, 4))                       # CPU tensor: no fallback
    with pytest.raises(ModelError):
        blk(torch.zeros(1, 4, 4, 4, device=dev))           # wrong channel count


@pytest.mark.parametrize("tag", ["t149", "f13", "small", "c3"])
def test_dscnn_on_any_feature_map(native, dev, anymap_golden, tag):
    """DepthwiseSeparableConv.forward accepts any [B,C,T,F] (models.py:160-183): 149 x 10 (clip_duration_ms = 1500), 99 x 13
    (num_cepstral_coeffs = 13), a 20 x 8 map and a 3-channel 50 x 12 one take the composed path -- conv1, the four blocks with
    their relu(bias) rings, pool + fc -- and must give the imported reference model's stage outputs (2e-5 of the stage's
    largest value), logits (1e-4) and labels."""
    g = anymap_golden
    x = torch.from_numpy(g[f"{tag}.x"]).to(dev)
    B, cin, T, F = x.shape
    c = make_ctx(native)
    try:
        c.load_dscnn(g[f"{tag}.blob"], 12, cin)
        H1, W1 = (T - 6) // 2 + 1, (F - 6) // 2 + 1
        sizes = [64 * (H1 + 2 * k) * (W1 + 2 * k) for k in range(5)]
        layers = torch.zeros((B * sum(sizes),), dtype=torch.float32, device=dev)
        logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
        labels = torch.empty((B,), dtype=torch.int32, device=dev)
        c.forward_map_f32(x, logits, labels, layers=layers)
        c.sync()
        off = 0
        for k, n in enumerate(sizes):
            got = layers[off:off + B * n].reshape(B, 64, H1 + 2 * k, W1 + 2 * k)[:2].cpu().numpy()
            want = g[f"{tag}.layer{k}"]
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), f"stage {k}"
            if k:  # the ring is relu(bias) exactly, corners included
                assert np.array_equal(got[:, :, 0, :], want[:, :, 0, :]) and np.array_equal(got[:, :, :, -1], want[:, :, :, -1])
            off += B * n
        assert np.abs(logits.cpu().numpy() - g[f"{tag}.logits"]).max() <= TOL
        assert np.array_equal(labels.cpu().numpy(), g[f"{tag}.label"])
        # the product entry (no stage dump) gives the same bits
        l2 = torch.empty_like(logits)
        c.forward_map_f32(x, l2, None)
        c.sync()
        assert torch.equal(l2, logits)
    finally:
        c.close()


def test_any_map_end_to_end_and_python_surface(native, dev, anymap_golden, e2e_golden):
    """wav -> label when AudioConfig changes the map: 1.5 s clips (149 frames) and 13 cepstra, through kws_infer_i16 (which
    follows kws_frontend_shape) and through the drop-in model class; and the composed path on the reference geometry
    agrees with the fused kernel."""
    from kws.libs.models import DepthwiseSeparableConv

    g = anymap_golden
    for tag, clips, kw in (("t149", g["long_clips"], {"n_samples": 24000}), ("f13", g["clips13"], {"numcep": 13})):
        c = make_ctx(native)
        try:
            c.set_frontend(**kw)
            assert c.frontend_shape() == tuple(g[f"{tag}.x"].shape[2:])
            c.load_dscnn(g[f"{tag}.blob"], 12)
            wav = torch.from_numpy(np.ascontiguousarray(clips)).to(dev)
            logits = torch.empty((len(clips), 12), dtype=torch.float32, device=dev)
            labels = torch.empty((len(clips),), dtype=torch.int32, device=dev)
            c.infer_i16(wav, logits, labels)
            c.sync()
            assert np.abs(logits.cpu().numpy() - g[f"{tag}.logits"]).max() <= TOL, tag
            assert np.array_equal(labels.cpu().numpy(), g[f"{tag}.label"]), tag
        finally:
            c.close()
    model = DepthwiseSeparableConv(num_classes=12)
    st, off = {}, 0
    for k, shp in o_dscnn.state_shapes(12).items():
        n = int(np.prod(shp))
        st[k] = torch.from_numpy(g["t149.blob"][off:off + n].reshape(shp).copy())
        off += n
    model.load_state_dict(st)
    out, lab = model(torch.from_numpy(g["t149.x"]).to(dev), return_labels=True)
    assert np.abs(out.cpu().numpy() - g["t149.logits"]).max() <= TOL and np.array_equal(lab.cpu().numpy(), g["t149.label"])
    # 99 x 10 through the composed path (debug entry) vs the fused kernel: two implementations of one network
    c = make_ctx(native)
    try:
        c.load_dscnn(e2e_golden["he.blob"], 12)
        x = torch.from_numpy(o_mfcc.collate_pcm16(e2e_golden["clips"][:16])).to(dev)
        fused = torch.empty((16, 12), dtype=torch.float32, device=dev)
        comp = torch.empty_like(fused)
        layers = torch.zeros((16 * 64 * (141 + 245 + 357 + 477 + 605),), dtype=torch.float32, device=dev)
        c.forward_f32(x, fused, None)
        c.forward_map_f32(x, comp, None, layers=layers)
        c.sync()
        assert np.abs(fused.cpu().numpy() - comp.cpu().numpy()).max() <= 2e-5 * float(fused.abs().max())
        assert np.abs(comp.cpu().numpy() - e2e_golden["he.logits"][8:24]).max() <= TOL
    finally:
        c.close()


def test_multichannel_model(dev):
    """DepthwiseSeparableConv(input_channels=3) (models.py:125,135): conv1 over three channels in the general kernel, then
    the fused kernel from block 1; logits and labels against the imported reference module's (golden)."""
    from kws.common.errors import ModelError
    from kws.libs.models import DepthwiseSeparableConv

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multichannel_golden.npz"))
    m = DepthwiseSeparableConv(num_classes=12, input_channels=3)
    state, off = {}, 0
    for k, v in m.state_dict().items():
        n = v.numel()
        state[k] = torch.from_numpy(g["blob"][off:off + n].reshape(tuple(v.shape)).copy())
        off += n
    assert off == g["blob"].size
    m.load_state_dict(state)
    logits, labels = m.forward(torch.from_numpy(g["x"]).to(dev), return_labels=True)
    err = float(np.abs(logits.cpu().numpy() - g["logits"]).max())
    assert err <= min(TOL, 4 * LAYER_RTOL * float(np.abs(g["logits"]).max())), err
    assert np.array_equal(labels.cpu().numpy(), g["label"])
    assert float(g["logits"].std(axis=0).mean()) >= 0.1
    # the same on the exact bf16 triple (the default above is the f16-pair arithmetic: the pre-convolved entry measures conv1's
    # maximum while staging it)
    from kws import _native
    ctx = m._context(0)
    ctx.set_pointwise_math(_native.PW_SPLIT_BF16)
    logits3, labels3 = m.forward(torch.from_numpy(g["x"]).to(dev), return_labels=True)
    ctx.set_pointwise_math(_native.PW_DEFAULT)
    assert float(np.abs(logits3.cpu().numpy() - g["logits"]).max()) <= min(TOL, 4 * LAYER_RTOL * float(np.abs(g["logits"]).max()))
    assert np.array_equal(labels3.cpu().numpy(), g["label"])
    assert float((logits3 - logits).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["logits"]).max()))
    with pytest.raises(ModelError):
        m.forward(torch.zeros(2, 1, 99, 10, device=dev))            # wrong channel count
    with pytest.raises(ModelError):
        m.infer_pcm16(torch.zeros(2, 16000, dtype=torch.int16, device=dev))   # an MFCC map has one channel


def test_forward_entry_matches_debug_entry(ctx, dev, dscnn_golden):
    g = dscnn_golden
    ctx.load_dscnn(g["n01.blob"], 12)
    x = torch.from_numpy(g["x"]).to(dev)
    a = torch.empty((x.shape[0], 12), dtype=torch.float32, device=dev)
    la = torch.empty((x.shape[0],), dtype=torch.int32, device=dev)
    ctx.forward_f32(x, a, la)
    b = torch.empty_like(a)
    ctx.forward_f32(x, b, None)  # label pointer may be NULL
    ctx.sync()
    assert torch.equal(a, b)
    assert np.array_equal(la.cpu().numpy(), g["n01.label"])


def test_dscnn_f16_pair_arithmetic_holds_over_range(native, ctx, dev, e2e_golden):
    """The default DS-CNN arithmetic (KWS_PW_PAIR_F16: f16 pairs, per-clip power-of-two scales derived two layers ahead from
    measured maxima and bounds that hold for any input) against a float64 evaluation of the model, over inputs and weights
    that stress f16's range: feature maps scaled by 1e-3 .. 1e3, all zero, one huge value among tiny ones, half empty,
    constant; convolution weights x 6 and x 0.1, biases x 200 and zero.  Nothing may overflow: finite logits everywhere,
    error against float64 within 4 x what torch's own f32 forward shows on the same clip (floor 2e-6 of the logit scale), and
    within 3e-6 of the scale (or 8 x the f32 forward's error) of the exact three-way bf16 arithmetic on the same context."""
    torch.manual_seed(41)
    x = torch.randn(24, 1, 99, 10) * 6.0
    x[0] = 0.0
    x[1] *= 1e-3
    x[2] *= 1e3
    x[3] *= 40.0
    x[4, :, 50:] = 0.0
    x[5] = torch.randn(1, 99, 10) * 0.01
    x[5, 0, 40, 3] = 3000.0
    x[6] = 7.5
    x[7] = -x[7].abs()
    x[8, :, :, 0] = -36.04365338911715      # the log-energy floor of digital silence in cepstrum 0
    base = state_from_blob(e2e_golden["he.blob"])
    # (the last two: every stage's values far below 1 -- the units are capped at 2^120 there -- and far above)
    # (-5.0: every pointwise row alternates +5, -5 plus the trained part -- sum|w| is ~10^3 x |sum w|, the bounds are loose by that)
    for w_gain, b_gain in ((1.0, 1.0), (6.0, 1.0), (0.1, 1.0), (1.0, 200.0), (1.0, 0.0), (1e-9, 0.0), (300.0, 1e6), (-5.0, 1.0)):
        state = {k: v.clone() for k, v in base.items()}
        cancel = w_gain < 0
        for k in state:
            if k.endswith("weight") and not k.startswith("fc") and not cancel:
                state[k] = state[k] * (w_gain if "pointwise" in k or k.startswith("conv1") else 1.0)
            if k.endswith("bias") and not k.startswith("fc"):
                state[k] = (state[k] + 0.01) * b_gain
        if cancel:
            for i in range(1, 5):
                alt = torch.tensor([1.0, -1.0]).repeat(32).reshape(1, 64, 1, 1) * abs(w_gain)
                state[f"dsconv{i}.pointwise.weight"] = alt.expand(64, 64, 1, 1).clone() + state[f"dsconv{i}.pointwise.weight"]
        blob = np.concatenate([state[k].reshape(-1).numpy() for k in o_dscnn.state_shapes(12)]).astype(np.float32)
        ctx.load_dscnn(blob, 12)
        ref64 = o_dscnn.forward({k: v.double() for k, v in state.items()}, x.double())
        ref32 = o_dscnn.forward(state, x)
        xd = x.to(dev)
        out = {}
        for math in (native.PW_PAIR_F16, native.PW_SPLIT_BF16):
            ctx.set_pointwise_math(math)
            logits = torch.empty((x.shape[0], 12), dtype=torch.float32, device=dev)
            ctx.forward_f32(xd, logits, None)
            ctx.sync()
            out[math] = logits.cpu().double()
        ctx.set_pointwise_math(native.PW_DEFAULT)
        pair, triple = out[native.PW_PAIR_F16], out[native.PW_SPLIT_BF16]
        assert torch.isfinite(pair).all() and torch.isfinite(triple).all(), (w_gain, b_gain)
        for i in range(x.shape[0]):
            scale = max(1.0, float(ref64[i].abs().max()))
            e_pair = float((pair[i] - ref64[i]).abs().max())
            e_f32 = float((ref32[i].double() - ref64[i]).abs().max())
            assert e_pair <= max(4.0 * e_f32, 2e-6 * scale), (w_gain, b_gain, i, e_pair, e_f32, scale)
            # (two f32-grade results differ by at most the sum of their errors: where cancellation amplifies every arithmetic's
            # rounding, the mutual gate follows the f32 forward's own error like the gate above)
            assert float((pair[i] - triple[i]).abs().max()) <= max(3e-6 * scale, 8.0 * e_f32), (w_gain, b_gain, i)


def test_pointwise_math_settings_agree(native, ctx, dev, dscnn_golden):
    """KWS_PW_SPLIT_BF16 (default) and KWS_PW_F32 are two arithmetic routes to the same f32 result: both within
    TOL of the reference logits, within 1e-5 of each other, identical labels; unknown settings are refused."""
    from kws.common.errors import ModelError

    g = dscnn_golden
    ctx.load_dscnn(g["n01.blob"], 12)
    x = torch.from_numpy(g["x"]).to(dev)
    out = {}
    for math in (native.PW_SPLIT_BF16, native.PW_F32, native.PW_PAIR_F16):
        ctx.set_pointwise_math(math)
        logits = torch.empty((x.shape[0], 12), dtype=torch.float32, device=dev)
        labels = torch.empty((x.shape[0],), dtype=torch.int32, device=dev)
        ctx.forward_f32(x, logits, labels)
        ctx.sync()
        out[math] = (logits.cpu().numpy(), labels.cpu().numpy())
        assert np.abs(out[math][0] - g["n01.logits"]).max() <= TOL
        assert np.array_equal(out[math][1], g["n01.label"])
    scale = max(1.0, float(np.abs(g["n01.logits"]).max()))
    assert np.abs(out[native.PW_SPLIT_BF16][0] - out[native.PW_F32][0]).max() <= 1e-5 * scale
    assert np.abs(out[native.PW_SPLIT_BF16][0] - out[native.PW_PAIR_F16][0]).max() <= 1e-5 * scale
    with pytest.raises(ModelError):
        ctx.set_pointwise_math(2)
    ctx.set_pointwise_math(native.PW_DEFAULT)


# ------------------------------------------------------------------------------------------- fused wav -> label
def assert_labels_match(labels, want_logits, err):
    """argmax must be identical wherever the reference's top-2 margin exceeds 10x the measured logit error (a closer
    call may legitimately flip); returns the fraction of clips that were compared."""
    want_logits = torch.as_tensor(want_logits)
    top2 = torch.topk(want_logits, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 10.0 * max(err, 1e-7)).numpy()
    assert np.array_equal(np.asarray(labels)[clear], o_dscnn.predict(want_logits).numpy()[clear])
    return float(clear.mean())


def test_infer_golden_diverse_clips(ctx, dev, e2e_golden):
    """wav -> label on the 48 diverse golden clips with signal-preserving weights: the expected logits and labels are
    the imported reference model's (on the oracle's MFCC); they span >= 6 classes and move from clip to clip, so a
    kernel that ignores its input cannot pass."""
    g = e2e_golden
    ctx.load_dscnn(g["he.blob"], 12)
    logits, labels = gpu_infer(ctx, dev, g["clips"])
    want, want_label = g["he.logits"][8:], g["he.label"][8:]
    assert len(set(want_label.tolist())) >= 6 and float(want.std(axis=0).mean()) >= 0.1
    err = float(np.abs(logits - want).max())
    assert err <= TOL, f"logits max abs err {err:.3e}"
    assert np.array_equal(labels, want_label)           # min top-2 margin of the fixture is 4.4e-3 >> TOL
    # silence vs full-scale noise must differ by far more than the tolerance (round 1's weights: 6e-4)
    assert np.abs(logits[0] - logits[4]).max() > 0.5


@pytest.mark.parametrize("tag", ["he", "he5", "raw"])
def test_infer_stress_clips_and_weight_tags(native, dev, e2e_golden, stress_golden, tag):
    """Inputs and weights chosen to break the logit bound: 16 speech-like clips (harmonics over a floor 60-90 dB down, pauses
    of exact zeros and of +-1 LSB dither, -6 .. -50 dBFS) next to the 48 diverse clips, under the 'he' weights, the same
    with a 5x classifier gain (logits up to 40) and weights whose conv1 is NOT pre-divided by the MFCC maps' RMS (logits up
    to 200).  Expected values: the imported reference model on the oracle's MFCC (make_golden.py stress)."""
    clips = np.concatenate([e2e_golden["clips"], stress_golden["speech_clips"]])
    want, want_label = stress_golden[f"{tag}.logits"], stress_golden[f"{tag}.label"]
    c = make_ctx(native)
    try:
        c.load_dscnn(stress_golden[f"{tag}.blob"], 12)
        logits, labels = gpu_infer(c, dev, clips)
    finally:
        c.close()
    err = np.abs(logits - want).max(axis=1)
    scale = float(np.abs(want).max())
    # 1e-4 absolute (north_star) while the logits are O(10); float32 itself resolves 6e-8 of the largest activation, and two
    # correct float32 implementations of this network differ by ~2e-6 of the largest logit (the generator's own
    # oracle-vs-reference gate), so beyond |logit| ~ 50 the bound is that float32 floor
    gate = max(TOL, 2e-6 * scale)
    names = [str(n) for n in e2e_golden["names"]] + [str(n) for n in stress_golden["speech_names"]]
    worst = int(err.argmax())
    assert err.max() <= gate, f"{tag}: clip {names[worst]} logits off by {err.max():.2e} (gate {gate:.1e}, |logit| max {scale:.1f})"
    assert np.array_equal(labels, want_label), [names[i] for i in np.nonzero(labels != want_label)[0]]
    assert len(set(want_label.tolist())) >= 6


@pytest.mark.parametrize("kind,seed", [("uniform", 0), ("gauss", 1)])
def test_infer_matches_oracle(ctx, dev, e2e_golden, kind, seed):
    blob = e2e_golden["he.blob"]
    state = state_from_blob(blob)
    ctx.load_dscnn(blob, 12)
    clips = synth_clips(160, seed, kind)
    logits, labels = gpu_infer(ctx, dev, clips)
    want = o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips)))
    err = float(np.abs(logits - want.numpy()).max())
    assert err <= TOL, f"logits max abs err {err:.3e}"
    assert assert_labels_match(labels, want, err) > 0.95
    if kind == "gauss":  # every 16th clip is silence: its logits differ from the noise clips' by far more than TOL
        assert np.abs(logits[0] - logits[1]).max() > 0.5


def test_infer_requires_model(native, dev):
    from kws.common.errors import ModelError

    c = make_ctx(native)
    try:
        wav = torch.zeros((2, 16000), dtype=torch.int16, device=dev)
        logits = torch.empty((2, 12), dtype=torch.float32, device=dev)
        with pytest.raises(ModelError):
            c.infer_i16(wav, logits, None)
        with pytest.raises(ModelError):
            c.load_dscnn(np.zeros(100, np.float32), 12)  # wrong blob size
    finally:
        c.close()


def test_full_batch_properties(ctx, dev, e2e_golden):
    """BASELINE workload size (B = 4096): properties that need no oracle at that size.  Signal-preserving weights;
    the batch mixes full-scale noise, quieter noise at many levels, silence and the diverse golden clips, so the labels
    spread over the classes."""
    blob = e2e_golden["he.blob"]
    state = state_from_blob(blob)
    ctx.load_dscnn(blob, 12)
    B = 4096
    clips = synth_clips(B, 0, "uniform")
    gain = np.random.default_rng(3).uniform(0.0, 1.0, B) ** 4          # levels from full scale down to a few LSBs
    clips[1::2] = np.round(clips[1::2] * gain[1::2, None]).astype(np.int16)
    clips[5] = 0
    clips[777] = 0
    clips[1000:1048] = e2e_golden["clips"]
    logits, labels = gpu_infer(ctx, dev, clips)
    assert np.isfinite(logits).all() and labels.min() >= 0 and labels.max() < 12
    # argmax consistency (first maximum wins) on the device's own logits
    assert np.array_equal(labels, np.argmax(logits, axis=1).astype(np.int32))
    # determinism: bit-identical on a second run
    logits2, labels2 = gpu_infer(ctx, dev, clips)
    assert np.array_equal(logits, logits2) and np.array_equal(labels, labels2)
    # clips are independent: a sub-batch and a permutation give bit-identical rows
    sub = [0, 5, 4095, 1234, 777, 31]
    ls, _ = gpu_infer(ctx, dev, clips[sub])
    assert np.array_equal(ls, logits[sub])
    perm = np.random.default_rng(1).permutation(B)
    lp, _ = gpu_infer(ctx, dev, clips[perm])
    assert np.array_equal(lp, logits[perm])
    # BASELINE configs[3]'s shard size (8192 over 8 GPUs = 1024 clips) and a single clip: the same bits as inside the batch
    for lo, n in ((0, 1024), (3072, 1024), (4095, 1), (0, 1)):
        lpart, ypart = gpu_infer(ctx, dev, clips[lo:lo + n])
        assert np.array_equal(lpart, logits[lo:lo + n]) and np.array_equal(ypart, labels[lo:lo + n])
    # identical inputs -> identical outputs; a sample of the batch against the oracle
    assert np.array_equal(logits[5], logits[777])
    pick = np.arange(0, B, 64)
    want = o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips[pick])))
    err = float(np.abs(logits[pick] - want.numpy()).max())
    assert err <= TOL, err
    assert assert_labels_match(labels[pick], want, err) > 0.9
    assert np.array_equal(labels[1000:1048], e2e_golden["he.label"][8:])
    assert len(np.unique(labels)) >= 6 and float(logits.std(axis=0).mean()) >= 0.1


def test_batch_of_one_and_ragged_sizes(ctx, dev, e2e_golden):
    ctx.load_dscnn(e2e_golden["he.blob"], 12)
    clips = synth_clips(67, 5)
    full, _ = gpu_infer(ctx, dev, clips)
    for n in (1, 2, 63, 67):
        part, _ = gpu_infer(ctx, dev, clips[:n])
        assert np.array_equal(part, full[:n])


# ------------------------------------------------------------------------------------------- host mirror of the reference API
def test_python_surface_end_to_end(dev, tmp_path, e2e_golden):
    import wave

    from kws.inference import KeywordSpotter
    from kws.libs.audio_processor import AudioConfig, AudioProcessor
    from kws.libs.models import DepthwiseSeparableConv

    clips = synth_clips(4, 6, "gauss")
    clips[0] = synth_clips(1, 7)[0]
    clips[2] = 0                                   # silence next to noise: the logits must tell them apart
    # AudioProcessor.extract_features: float signal in [-1,1] -> float64 [99,10]
    ap = AudioProcessor(None, AudioConfig())
    feat = ap.extract_features(o_mfcc.pcm16_to_float(clips[1]))
    assert feat.shape == (99, 10) and feat.dtype == np.float64
    assert np.abs(feat - o_mfcc.extract_features_pcm16(clips[1])).max() <= TOL
    batch = ap.extract_features_batch(torch.from_numpy(clips).to(dev))
    assert tuple(batch.shape) == (4, 1, 99, 10) and batch.dtype == torch.float32
    # model with the reference's state_dict layout
    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    logits = model(batch)
    want = o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips)))
    assert float((logits.cpu() - want).abs().max()) <= TOL
    assert float(want.std(dim=0).mean()) >= 0.1  # the four clips give different logits
    # save / load round trip, then wav files -> words
    path = tmp_path / "model.pth"
    model.save(str(path))
    spotter = KeywordSpotter()
    spotter.load_weights(str(path))
    files = []
    for i, c in enumerate(clips):
        f = tmp_path / f"clip{i}.wav"
        with wave.open(str(f), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(c[: 16000 - 100 * i].tobytes())  # shorter files are zero-padded to one second
        files.append(str(f))
    got = spotter.infer_files(files)
    padded = np.stack([np.concatenate([c[: 16000 - 100 * i], np.zeros(100 * i, np.int16)]) for i, c in enumerate(clips)])
    want2 = o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(padded)))
    assert [g[0] for g in got] == o_dscnn.predict(want2).tolist()
    assert all(word == spotter.words[idx] for idx, word in got)


def test_own_stream_and_profiling_counters(native, dev, e2e_golden):
    """A context on its own (non-blocking) stream: explicit syncs order it against torch; the per-kernel
    event timers count one launch per kernel per call."""
    c = native.Context(0)
    try:
        c.load_dscnn(e2e_golden["he.blob"], 12)
        clips = synth_clips(32, 8)
        wav = torch.from_numpy(clips).to(dev)
        logits = torch.empty((32, 12), dtype=torch.float32, device=dev)
        labels = torch.empty((32,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.prof_enable(True)
        c.prof_reset()
        for _ in range(3):
            c.infer_i16(wav, logits, labels)
        ms_c, n_c = c.prof_read(native.KWS_K_DSCNN)
        ms_m, n_m = c.prof_read(native.KWS_K_MFCC)
        assert (n_c, n_m) == (3, 3) and ms_c > 0 and ms_m > 0
        c.prof_enable(False)
        c.sync()
        ref = make_ctx(native)
        try:
            ref.load_dscnn(e2e_golden["he.blob"], 12)
            l2, _ = gpu_infer(ref, dev, clips)
        finally:
            ref.close()
        assert np.array_equal(logits.cpu().numpy(), l2)
    finally:
        c.close()


# ------------------------------------------------------------------------------------------- augmentation
def test_augment_matches_numpy_bit_exact(ctx, dev):
    rng = np.random.default_rng(21)
    B, n = 9, 16000
    clips = synth_clips(B, 9, "gauss")
    shift = rng.integers(-1600, 1600, B).astype(np.int32)
    shift[0], shift[1] = 0, 1599
    bg = (rng.standard_normal(50000) * 0.1).astype(np.float32)
    off = rng.integers(0, 50000 - n, B).astype(np.int32)
    vol = rng.uniform(0, 1, B).astype(np.float32)
    vol[2] = 0.0
    sil = np.zeros(B, np.uint8)
    sil[3] = 1
    out = torch.empty((B, n), dtype=torch.float32, device=dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    ctx.augment_i16(t(clips), out, shift=t(shift), bg=t(bg), bg_off=t(off), bg_vol=t(vol), silence=t(sil))
    ctx.sync()
    want = np.empty((B, n), np.float32)
    for b in range(B):
        a = np.zeros(n, np.float32)
        if not sil[b]:
            x = o_mfcc.pcm16_to_float(clips[b])
            s = int(shift[b])
            if s >= 0:
                a[s:] = x[: n - s]
            else:
                a[: n + s] = x[-s:]
        want[b] = a + bg[off[b]: off[b] + n] * vol[b]          # float32 array * float32 scalar, as NumPy does
    assert np.array_equal(out.cpu().numpy(), want)
    # no background pool, no shift: plain PCM scaling
    ctx.augment_i16(t(clips), out)
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), o_mfcc.pcm16_to_float(clips))


# ------------------------------------------------------------------------------------------- streaming
@pytest.mark.parametrize("use_graph", [False, True])
def test_streaming_frames_and_labels(native, dev, e2e_golden, use_graph):
    from kws.inference import StreamingSpotter
    from kws.libs.models import DepthwiseSeparableConv

    S, hops = 5, 112                     # odd stream count: the last wavefront carries a single stream
    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(31)
    pcm = rng.integers(-20000, 20000, size=(S, hops * 160), dtype=np.int16)
    pcm[1, : 40 * 160] = 0              # a stream that starts with digital silence
    pcm[2] = (pcm[2] * np.linspace(0.001, 1.0, hops * 160)).astype(np.int16)   # a stream that fades in
    pcm[3] = np.round(9000 * np.sin(2 * np.pi * 700 * np.arange(hops * 160) / 16000.0)).astype(np.int16)  # a tone
    sp = StreamingSpotter(S, model, use_graph=use_graph)
    try:
        checks = {2, 3, 50, 99, 100, 101, hops - 1}
        for t in range(hops):
            labels, logits = sp.push(pcm[:, t * 160:(t + 1) * 160])
            if t not in checks:
                continue
            feats, pushed = sp.features()
            assert pushed == t + 1
            newest = t - 2                  # newest complete frame of the continuous signal
            want = np.zeros((S, 99, 10), np.float32)
            for s in range(S):
                sig = o_mfcc.pcm16_to_float(pcm[s, : (t + 1) * 160])
                spec = o_mfcc.FrontendSpec(n_samples=len(sig))
                allf = o_mfcc.mfcc(sig, spec)    # frames of the continuous signal (the tail frames are zero-padded: unused)
                for i in range(99):
                    f = newest - 98 + i
                    if 0 <= f <= newest:
                        want[s, i] = allf[f]
            assert np.abs(feats - want).max() <= TOL, f"hop {t}"
            ref = o_dscnn.forward(state, torch.from_numpy(want)[:, None])
            err = float(np.abs(logits - ref.numpy()).max())
            assert err <= TOL, f"hop {t}: {err:.3e}"
            assert_labels_match(labels, ref, err)
            if t >= 99:  # full windows: silence-then-noise, a fade-in, a tone and plain noise give different logits
                assert float(ref.std(dim=0).mean()) >= 0.1, "the streams' logits must differ"
    finally:
        sp.close()


def test_streaming_high_dynamic_range_streams_meet_the_gate(native, dev, e2e_golden):
    """Streams on which float32 alone misses 1e-4 -- a clean tone over digital silence, a gated burst over a quiet floor,
    a speech-like signal with pauses of exact zeros -- next to white noise: every frame of every window within 1e-4 of the
    float64 oracle, on the one-launch push (flagged frames are redone in float64 by the wavefront that computed them,
    inside the DS-CNN launch) and on the features-only push (frame kernel, two streams per transform), logits within 1e-4."""
    import speechlike
    from kws.inference import StreamingSpotter

    names = [str(n) for n in e2e_golden["names"]]
    clips = e2e_golden["clips"]
    speech, _ = speechlike.speechlike_set(4, 400)
    gated = np.round(8000 * np.sin(2 * np.pi * 3000 * np.arange(16000) / 16000.0)).astype(np.int16)
    gated[:4000] = 0
    gated[9000:12000] = 0
    rows = [clips[names.index("sine_7000Hz_a8000")], clips[names.index("burst_44")], speech[0], speech[3], gated,
            synth_clips(1, 5)[0], clips[names.index("sine_200Hz_a8000")]]
    hops = 106
    pcm = np.zeros((len(rows), hops * 160), np.int16)
    for i, r in enumerate(rows):
        pcm[i, :16000] = r
    S = len(rows)
    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    fo = native.Context(dev.index)     # features-only pushes: the two-launch route's frame kernel
    fo.stream_open(S)
    sp = StreamingSpotter(S, model)
    ring = torch.empty((S, 99, 10), dtype=torch.float32, device=dev)
    hop_dev = torch.empty((S, 160), dtype=torch.int16, device=dev)
    try:
        allf = [o_mfcc.mfcc(o_mfcc.pcm16_to_float(pcm[s]), o_mfcc.FrontendSpec(n_samples=pcm.shape[1])) for s in range(S)]
        for t in range(hops):
            labels, logits = sp.push(pcm[:, t * 160:(t + 1) * 160])
            hop_dev.copy_(torch.from_numpy(pcm[:, t * 160:(t + 1) * 160]))
            torch.cuda.synchronize()
            fo.stream_push_i16(hop_dev)
            if t not in (2, 30, 60, 99, 100, hops - 1):
                continue
            newest = t - 2
            want = np.zeros((S, 99, 10), np.float32)
            for s in range(S):
                for i in range(99):
                    f = newest - 98 + i
                    if 0 <= f <= newest:
                        want[s, i] = allf[s][f]
            feats, pushed = sp.features()
            assert pushed == t + 1
            e = np.abs(feats - want).max(axis=(1, 2))
            assert e.max() <= TOL, f"one-launch push, hop {t}: per-stream errors {e}"
            fo.stream_copy_features(ring)
            fo.sync()
            two = np.roll(ring.cpu().numpy(), -((t + 1 - 2) % 99), axis=1)
            e2 = np.abs(two - want).max(axis=(1, 2))
            assert e2.max() <= TOL, f"frame kernel, hop {t}: per-stream errors {e2}"
            ref = o_dscnn.forward(state, torch.from_numpy(want)[:, None])
            err = float(np.abs(logits - ref.numpy()).max())
            assert err <= TOL, f"hop {t}: {err:.3e}"
            assert_labels_match(labels, ref, err)
        # the refinement ran (and was counted) on both routes; the noise stream alone would have listed next to nothing
        for c in (sp._ctx, fo):
            _, refined, _ = c.frontend_stats()
            assert refined >= 50, refined
        # and it is what makes the gate hold: the same streams with the refinement off miss it
        off = native.Context(dev.index)
        try:
            off.set_frontend_refine(0.0)
            off.stream_open(S)
            for t in range(100):
                hop_dev.copy_(torch.from_numpy(pcm[:, t * 160:(t + 1) * 160]))
                torch.cuda.synchronize()
                off.stream_push_i16(hop_dev)
            off.stream_copy_features(ring)
            off.sync()
            plain = np.roll(ring.cpu().numpy(), -((100 - 2) % 99), axis=1)
            w = np.zeros((S, 99, 10), np.float32)
            for s in range(S):
                for i in range(99):
                    f = 97 - 98 + i
                    if 0 <= f <= 97:
                        w[s, i] = allf[s][f]
            assert np.abs(plain - w).max() > 2 * TOL
            assert off.frontend_stats()[1] == 0
        finally:
            off.close()
    finally:
        sp.close()
        fo.close()


@pytest.mark.parametrize("route", ["one_launch", "two_launch"])
def test_streaming_with_two_hops_per_frame(native, dev, e2e_golden, route):
    """A 20 ms window (320 samples) over 10 ms hops still gives 99 frames per second, but a frame now spans TWO hops, not
    three: the newest frame after h pushes is h - 2 and the window starts at row (h - 1) mod 99.  Every place that derives
    the window's first row (fused push, two-launch push, endpointer, host mirror) takes it from ceil(frame_len / step)."""
    from kws.inference import StreamingSpotter
    from kws.libs.audio_processor import AudioConfig

    S, hops = 3, 108
    cfg = AudioConfig(frame_length=0.02)
    spec = lambda n: o_mfcc.FrontendSpec(winlen=0.02, n_samples=n)
    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    pcm = np.random.default_rng(91).integers(-20000, 20000, size=(S, hops * 160), dtype=np.int16)
    pcm[1] //= 40
    sp = StreamingSpotter(S, model, config=cfg, vad_log_energy=-3.0)
    try:
        if route == "two_launch":
            sp._ctx.set_pointwise_math(native.PW_F32)
        for t in range(hops):
            labels, logits = sp.push(pcm[:, t * 160:(t + 1) * 160])
            if t not in (1, 2, 50, 98, 99, 100, hops - 1):
                continue
            feats, pushed = sp.features()
            assert pushed == t + 1
            newest = t - 1
            want = np.zeros((S, 99, 10), np.float32)
            for s in range(S):
                allf = o_mfcc.mfcc(o_mfcc.pcm16_to_float(pcm[s, : (t + 1) * 160]), spec((t + 1) * 160))
                for i in range(99):
                    f = newest - 98 + i
                    if 0 <= f <= newest:
                        want[s, i] = allf[f]
            assert np.abs(feats - want).max() <= TOL, f"hop {t}"
            ref = o_dscnn.forward(state, torch.from_numpy(want)[:, None])
            err = float(np.abs(logits - ref.numpy()).max())
            assert err <= (TOL if route == "one_launch" else 5e-5 * max(1.0, float(ref.abs().max()))), f"hop {t}: {err:.3e}"
            # the endpointer reads the newest frame's log energy: loud streams voiced, the quiet one not
            want_voiced = want[:, -1, 0] > -3.0
            assert sp.vad_state is not None
            if t >= 60:
                assert np.array_equal((sp.vad_state & 1).astype(bool), want_voiced), (t, sp.vad_state, want[:, -1, 0])
    finally:
        sp.close()


def test_streaming_one_launch_push_agrees_with_the_two_launch_path(native, dev, e2e_golden):
    """A push that asks for logits from the product DS-CNN is one launch (each stream's new frame is computed in the prologue
    of its DS-CNN workgroup); a features-only push, and a push under the f32-MFMA pointwise variant, take the separate frame
    kernel (two streams per packed transform) + the DS-CNN kernel.  Same rings, same hop counts, same logits up to the
    arithmetic of the variants -- at an odd stream count, across the ring wrap (hop > 99)."""
    S, hops = 7, 130
    blob = e2e_golden["he.blob"]
    pcm = np.random.default_rng(77).integers(-20000, 20000, size=(hops, S, 160), dtype=np.int16)
    pcm[:, 2] //= 50
    ctxs = {name: native.Context(dev.index) for name in ("fused", "features_only", "f32_two_launch")}
    try:
        for c in ctxs.values():
            c.use_torch_stream()   # the three pushes of a hop and the copy of the next hop stay in program order
            c.load_dscnn(blob, 12)
            c.stream_open(S)
        ctxs["f32_two_launch"].set_pointwise_math(native.PW_F32)
        hop = torch.empty((S, 160), dtype=torch.int16, device=dev)
        logits = {k: torch.empty((S, 12), dtype=torch.float32, device=dev) for k in ctxs}
        labels = {k: torch.empty((S,), dtype=torch.int32, device=dev) for k in ctxs}
        rings = {k: torch.empty((S, 99, 10), dtype=torch.float32, device=dev) for k in ctxs}
        for t in range(hops):
            hop.copy_(torch.from_numpy(pcm[t]))
            ctxs["fused"].stream_push_i16(hop, logits["fused"], labels["fused"])
            ctxs["features_only"].stream_push_i16(hop)
            ctxs["f32_two_launch"].stream_push_i16(hop, logits["f32_two_launch"], labels["f32_two_launch"])
            if t in (1, 2, 60, 98, 99, 100, hops - 1):
                for k, c in ctxs.items():
                    c.stream_copy_features(rings[k])
                    c.sync()
                    assert c.stream_state()[1] == t + 1, k
                a, b, d = (rings[k].cpu().numpy() for k in ("fused", "features_only", "f32_two_launch"))
                assert np.array_equal(b, d)                         # the same kernel on the same samples
                assert np.abs(a - b).max() <= TOL, f"hop {t}"        # one frame vs two frames per packed transform (float32 roundings)
                la, ld = logits["fused"].cpu().numpy(), logits["f32_two_launch"].cpu().numpy()
                assert np.abs(la - ld).max() <= 5e-5 * max(1.0, np.abs(ld).max()), f"hop {t}"
    finally:
        for c in ctxs.values():
            c.close()


def test_streaming_graph_survives_a_weight_reload(dev, e2e_golden):
    """A captured push holds the weight pointers by value: after kws_load_dscnn (StreamingSpotter.load_model) the next
    push must classify with the NEW weights (the graph is re-captured), eager and replayed alike; the feature ring is
    untouched by the reload."""
    from kws.inference import StreamingSpotter

    S, hops = 4, 104
    first = he_model(e2e_golden)
    second = he_model(e2e_golden)
    with torch.no_grad():
        second.fc.weight.copy_(torch.flip(second.fc.weight, dims=[0]))      # a visibly different classifier
        second.fc.bias.copy_(torch.flip(second.fc.bias, dims=[0]))
        second.dsconv2.pointwise.weight.mul_(0.5)
    state2 = {k: v.clone() for k, v in second.state_dict().items()}
    pcm = np.random.default_rng(41).integers(-20000, 20000, size=(S, hops * 160), dtype=np.int16)
    pcm[1] //= 50
    sp = StreamingSpotter(S, first, use_graph=True)
    try:
        for t in range(hops - 2):
            sp.push(pcm[:, t * 160:(t + 1) * 160])
        _, before = sp.push(pcm[:, (hops - 2) * 160:(hops - 1) * 160])
        sp.load_model(second)
        _, after = sp.push(pcm[:, (hops - 1) * 160:])
        feats, pushed = sp.features()
        assert pushed == hops
        want = o_dscnn.forward(state2, torch.from_numpy(feats)[:, None]).numpy()
        assert np.abs(after - want).max() <= TOL
        assert np.abs(after - before).max() > 0.1          # the old weights would have given `before`-like logits
    finally:
        sp.close()


def test_trainer_checkpoint_loads(dev, tmp_path, e2e_golden):
    """KeywordSpottingModel.load accepts the trainer's checkpoint dict ({"model_state_dict": ...},
    kws/libs/training.py:199-216) as well as a bare state_dict (train.py:77); an edit through .data needs
    sync_weights()."""
    from kws.libs.models import DepthwiseSeparableConv

    g = e2e_golden
    src = he_model(g)
    ckpt = tmp_path / "best_model.pth"
    torch.save({"model_state_dict": src.state_dict(), "epoch": 3, "metrics": {"val_acc": 0.5}}, str(ckpt))
    m = DepthwiseSeparableConv(12)
    m.load(str(ckpt), device=torch.device("cpu"))
    x = torch.from_numpy(np.concatenate([g["x_rand"], o_mfcc.collate_pcm16(g["clips"][:8])])).to(dev)
    logits, labels = m.forward(x, return_labels=True)
    assert np.abs(logits.cpu().numpy() - g["he.logits"][:16]).max() <= TOL
    assert np.array_equal(labels.cpu().numpy(), g["he.label"][:16])
    m.fc.bias.data.add_(1.0)                       # behind torch's back: _version unchanged
    m.sync_weights()
    assert np.abs(m.forward(x).cpu().numpy() - (g["he.logits"][:16] + 1.0)).max() <= TOL


def test_host_ingest_pipeline(native, dev, e2e_golden):
    """kws_infer_host_i16 (pack threads -> pinned staging -> H2D || MFCC + DS-CNN || D2H, chunk by chunk) returns exactly
    what the device-resident fused call returns: ragged batch sizes around the chunk size, every pipeline shape, numpy
    (pageable), CPU-tensor and pinned inputs; KeywordSpotter.infer_batches yields batch by batch, in order."""
    from kws.common.errors import ModelError
    from kws.inference import KeywordSpotter

    model = he_model(e2e_golden)
    big = synth_clips(2500, 31, "uniform")
    big[100:148] = e2e_golden["clips"]
    want_logits, want_labels = model.infer_pcm16(torch.from_numpy(big).to(dev))
    want_logits, want_labels = want_logits.cpu().numpy(), want_labels.cpu().numpy()
    # not only HIP path against HIP path: the 48 golden clips inside the batch against the imported reference model's values
    assert np.abs(want_logits[100:148] - e2e_golden["he.logits"][8:]).max() <= TOL
    assert np.array_equal(want_labels[100:148], e2e_golden["he.label"][8:])
    c = native.Context(0)
    try:
        c.load_dscnn(e2e_golden["he.blob"], 12)
        for chunk, slots, threads in ((0, 0, 0), (100, 2, 1), (512, 4, -1), (3000, 3, 3), (1, 16, 2)):
            c.ingest_config(chunk, slots, threads)
            n = 2500 if chunk != 1 else 40
            logits, labels = c.infer_host_i16(big[:n])
            assert np.array_equal(logits, want_logits[:n]) and np.array_equal(labels, want_labels[:n]), (chunk, slots, threads)
            if n > 148:
                assert np.abs(logits[100:148] - e2e_golden["he.logits"][8:]).max() <= TOL   # the ingest route itself vs the reference
        c.ingest_config(0, 0, 0)
        for n in (1, 1023, 1024, 1025, 2049):
            logits, labels = c.infer_host_i16(big[:n])
            assert np.array_equal(logits, want_logits[:n]) and np.array_equal(labels, want_labels[:n]), n
        pinned = torch.from_numpy(big).pin_memory()
        logits, labels = c.infer_host_i16(pinned)                   # the DMA reads the caller's pinned buffer
        assert np.array_equal(logits, want_logits) and np.array_equal(labels, want_labels)
        logits, labels = c.infer_host_i16(torch.from_numpy(big))    # pageable CPU tensor
        assert np.array_equal(logits, want_logits)
        with pytest.raises(ModelError):
            c.ingest_config(0, 1, 0)                                # fewer than two slots cannot overlap anything
        import ctypes
        d_wav = torch.from_numpy(big[:8]).to(dev)                   # a DEVICE pointer where a host pointer belongs: refused, not dereferenced
        out_l, out_y = np.empty((8, 12), np.float32), np.empty((8,), np.int32)
        rc = native.lib().kws_infer_host_i16(c._h, ctypes.c_void_p(d_wav.data_ptr()), 8, ctypes.c_void_p(out_l.ctypes.data),
                                             ctypes.c_void_p(out_y.ctypes.data))
        assert rc == native.KWS_EINVAL
    finally:
        c.close()
    sp = KeywordSpotter(model)
    batches = [synth_clips(b, seed, "uniform") for seed, b in enumerate([64, 7, 1300, 1, 64])]
    mixed = [b if i % 2 == 0 else torch.from_numpy(b).pin_memory() for i, b in enumerate(batches)]
    got = list(sp.infer_batches(mixed))
    assert len(got) == len(batches)
    for clips, (labels, logits) in zip(batches, got):
        wl, wy = model.infer_pcm16(torch.from_numpy(clips).to(dev))
        assert np.array_equal(labels, wy.cpu().numpy()) and np.array_equal(logits, wl.cpu().numpy())
    labels, logits = sp.infer_pcm16(big[:10, :15000])               # short clips are zero-padded to one second
    padded = np.concatenate([big[:10, :15000], np.zeros((10, 1000), np.int16)], axis=1)
    wl, _ = model.infer_pcm16(torch.from_numpy(padded).to(dev))
    assert np.array_equal(logits, wl.cpu().numpy())


def test_infer_batches_keeps_the_pipeline_full_across_small_batches(native, dev, e2e_golden):
    """KeywordSpotter.infer_batches over batches of the reference's own size (1028, train.py:110) and smaller ones: batch k+1
    is submitted before batch k is waited for, every batch is cut into as many chunks as the ring has slots -- and every
    result equals the device-resident call's, bit for bit."""
    from kws.inference import KeywordSpotter

    model = he_model(e2e_golden)
    spotter = KeywordSpotter(model)
    sizes = [1028, 256, 1028, 5, 130, 3000]
    big = np.concatenate([synth_clips(s, 60 + i) for i, s in enumerate(sizes)])
    big[7:7 + 48] = e2e_golden["clips"]
    want_logits, want_labels = gpu_infer(model._context(dev.index or 0), dev, big)
    assert np.abs(want_logits[7:7 + 48] - e2e_golden["he.logits"][8:]).max() <= TOL       # anchored on the reference model's logits
    off = 0
    batches = []
    for s_ in sizes:
        batches.append(big[off:off + s_])
        off += s_
    got = list(spotter.infer_batches(iter(batches)))
    assert [len(lab) for lab, _ in got] == sizes
    assert np.array_equal(np.concatenate([lab for lab, _ in got]), want_labels)
    assert np.array_equal(np.concatenate([lg for _, lg in got]), want_logits)
    # a consumer that stops early leaves nothing in flight
    it = spotter.infer_batches(iter(batches))
    next(it)
    it.close()
    lab, lg = spotter.infer_pcm16(batches[1])
    assert np.array_equal(lg, want_logits[1028:1028 + 256])


def test_mfcc_batch_beyond_grid_limit(ctx, dev):
    """B > 65535 clips: the front end splits the launch (grid.y limit); first, last and boundary clips match the
    small-batch result bit for bit."""
    B = 65535 + 70
    base = torch.from_numpy(synth_clips(256, 11, "uniform")).to(dev)
    wav = base.repeat((B + 255) // 256, 1)[:B].contiguous()
    out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(wav, out)
    ref = torch.empty((256, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(base, ref)
    ctx.sync()
    for idx in (0, 255, 65534, 65535, 65536, B - 1):
        assert torch.equal(out[idx], ref[idx % 256]), idx


# ------------------------------------------------------------------------------ model zoo / posteriors (section 8 f-4)
def test_batchnorm_variant_folds_into_the_fused_kernel(dev, e2e_golden):
    """DepthwiseSeparableConvBN (build-defined: inference BatchNorm after every convolution) equals its own
    unfolded torch-CPU definition (oracle.dscnn.forward_bn) within the logit tolerance, identical argmax."""
    from kws.libs.models import DepthwiseSeparableConvBN

    torch.manual_seed(5)
    m = DepthwiseSeparableConvBN(12)
    m.plain.load_state_dict(state_from_blob(e2e_golden["he.blob"]))   # signal-preserving weights, MFCC-scaled inputs
    with torch.no_grad():
        for bn in [m.bn_conv1, *m.bn_dw, *m.bn_pw]:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0.0, 0.2)
            bn.running_mean.normal_(0.0, 0.3)
            bn.running_var.uniform_(0.3, 2.0)
    x = torch.from_numpy(o_mfcc.collate_pcm16(e2e_golden["clips"][:33]))
    state = {k: v.detach().clone() for k, v in m.plain.state_dict().items()}
    names = ["conv1"] + [f"dw{i}" for i in range(1, 5)] + [f"pw{i}" for i in range(1, 5)]
    mods = [m.bn_conv1] + list(m.bn_dw) + list(m.bn_pw)
    bn = {n: (b.weight.detach(), b.bias.detach(), b.running_mean.clone(), b.running_var.clone()) for n, b in zip(names, mods)}
    want = o_dscnn.forward_bn(state, bn, x)
    logits, labels = m.forward(x.to(dev), return_labels=True)
    err = (logits.cpu() - want).abs().max().item()
    assert err <= TOL, err
    assert float(want.std(dim=0).mean()) >= 0.1 and len(set(want.argmax(dim=1).tolist())) >= 3
    assert_labels_match(labels.cpu().numpy(), want, err)
    # the folded model is cached until a statistic changes
    assert m.fold() is m.fold()
    with torch.no_grad():
        m.bn_pw[3].bias.add_(1.0)
    assert m.fold() is not None and m._folded_key is not None
    logits2 = m.forward(x.to(dev))
    assert (logits2.cpu() - logits.cpu()).abs().max().item() > 1e-3


def test_softmax_and_streaming_posterior_smoothing(native, dev, e2e_golden):
    from kws.inference import StreamingSpotter
    from kws.libs.models import DepthwiseSeparableConv

    # softmax entry against numpy
    c = native.Context(0)
    c.use_torch_stream()
    z = torch.randn(257, 12, device=dev) * 5
    p = torch.empty_like(z)
    c.softmax_f32(z, p)
    c.sync()
    zn = z.cpu().numpy().astype(np.float64)
    e = np.exp(zn - zn.max(axis=1, keepdims=True))
    assert np.abs(p.cpu().numpy() - e / e.sum(axis=1, keepdims=True)).max() <= 1e-6
    c.close()

    # streaming: smoothed posteriors = mean of the softmax of the last W hops' logits (fewer at the start)
    model = he_model(e2e_golden)
    S, W, hops = 5, 4, 11
    raw = StreamingSpotter(S, model)
    smo = StreamingSpotter(S, model, smooth_window=W)
    rng = np.random.default_rng(2)
    hist = []
    for h in range(hops):
        hop = rng.integers(-20000, 20000, size=(S, raw.hop), dtype=np.int16)
        _, logits = raw.push(hop)
        labels, post = smo.push(hop)
        zn = logits.astype(np.float64)
        e = np.exp(zn - zn.max(axis=1, keepdims=True))
        hist.append(e / e.sum(axis=1, keepdims=True))
        want = np.mean(hist[-W:], axis=0)
        assert np.abs(post - want).max() <= 2e-6, h
        assert np.array_equal(labels, want.argmax(axis=1))
    raw.close()
    smo.close()


def test_cnn_trad_fpool3_matches_its_cpu_definition(dev):
    """cnn-trad-fpool3 (build-defined) on the GPU vs oracle.cnn_trad on the CPU: logits within
    1e-4 relative to their scale, identical argmax where the margin allows, ragged batch (not a multiple of 8)."""
    from oracle import cnn_trad as o_ct
    from kws.libs.models import CnnTradFpool3

    state = o_ct.random_state(seed=4)
    m = CnnTradFpool3(12)
    m.load_state_dict(state)
    torch.manual_seed(8)
    x = torch.randn(21, 1, 99, 10) * 3.0
    want = o_ct.forward(state, x)
    logits, labels = m.forward(x.to(dev), return_labels=True)
    scale = max(1.0, float(want.abs().max()))
    err = float((logits.cpu() - want).abs().max())
    assert err <= TOL * scale, (err, scale)
    top2 = torch.topk(want, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 2 * TOL * scale).numpy()
    assert np.array_equal(labels.cpu().numpy()[clear], want.argmax(dim=1).numpy()[clear])
    assert clear.mean() > 0.8


def test_cnn_trad_f16_pair_arithmetic_holds_over_range(dev):
    """cnn-trad-fpool3's default arithmetic (f16 pairs, three MFMAs per k-block, KWS_CT_F16_PAIR) against a float64
    evaluation of the model's CPU definition, over inputs and weights that stress f16's range: features scaled by 1e-3 .. 1e3,
    an all-zero clip, a clip with one huge value, half-empty clips; convolution weights x 8 and x 0.05, large biases.  The
    per-clip power-of-two scales come from rigorous bounds, so nothing may overflow: finite logits everywhere, error against
    float64 within 4x of what torch's own f32 forward shows on the same clip (floor 2e-6 of the logit scale), and within
    3e-6 of the scale of the exact three-way bf16 arithmetic (KWS_CT_BF16_TRIPLE) run on the same context."""
    from kws import _native
    from kws.libs.models import CnnTradFpool3
    from oracle import cnn_trad as o_ct

    torch.manual_seed(31)
    x = torch.randn(24, 1, 99, 10) * 8.0
    x[0] = 0.0
    x[1] *= 1e-3
    x[2] *= 1e3
    x[3] *= 60.0
    x[4, :, 50:] = 0.0
    x[5] = torch.randn(1, 99, 10) * 0.01
    x[5, 0, 40, 3] = 5000.0                      # one huge feature among tiny ones
    x[6] = x[6].abs()
    x[7] = -x[7].abs()
    for seed, gain, bias_gain in ((11, 1.0, 1.0), (12, 8.0, 1.0), (13, 0.05, 1.0), (14, 1.0, 300.0)):
        state = o_ct.random_state(seed=seed)
        for k in state:
            if k.startswith("conv") and k.endswith("weight"):
                state[k] = state[k] * gain
            if k.startswith("conv") and k.endswith("bias"):
                state[k] = state[k] * bias_gain
        m = CnnTradFpool3(12)
        m.load_state_dict(state)
        ref64 = o_ct.forward({k: v.double() for k, v in state.items()}, x.double())
        ref32 = o_ct.forward(state, x)
        ctx = m._context(0)
        ctx.set_cnn_trad_math(_native.KWS_CT_F16_PAIR)
        pair = m.forward(x.to(dev)).cpu().double()
        ctx.set_cnn_trad_math(_native.KWS_CT_BF16_TRIPLE)
        triple = m.forward(x.to(dev)).cpu().double()
        ctx.set_cnn_trad_math(_native.KWS_CT_F16_PAIR)
        assert torch.isfinite(pair).all() and torch.isfinite(triple).all(), (seed, gain)
        for i in range(x.shape[0]):
            scale = max(1.0, float(ref64[i].abs().max()))
            e_pair = float((pair[i] - ref64[i]).abs().max())
            e_f32 = float((ref32[i].double() - ref64[i]).abs().max())
            assert e_pair <= max(4.0 * e_f32, 2e-6 * scale), (seed, gain, bias_gain, i, e_pair, e_f32, scale)
            assert float((pair[i] - triple[i]).abs().max()) <= 3e-6 * scale, (seed, gain, i)


def test_cnn_trad_fpool3_fused_wav_to_label(dev):
    """BASELINE configs[2] (MFCC + cnn-trad-fpool3 fused): int16 PCM through kws_infer_cnn_trad_i16 vs the oracle's
    MFCC followed by the model's CPU definition; the two-call path (kws_mfcc_i16, kws_forward_cnn_trad_f32) gives
    the same bits; a context without the model refuses the call."""
    from kws.common.errors import ModelError
    from kws.libs.models import CnnTradFpool3
    from oracle import cnn_trad as o_ct

    state = o_ct.random_state(seed=5)
    m = CnnTradFpool3(12)
    m.load_state_dict(state)
    clips = synth_clips(40, 3, "gauss")
    want = o_ct.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips)))
    wav = torch.from_numpy(clips).to(dev)
    logits, labels = m.infer_pcm16(wav)
    scale = max(1.0, float(want.abs().max()))
    err = float((logits.cpu() - want).abs().max())
    assert err <= TOL * scale, (err, scale)
    top2 = torch.topk(want, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 2 * TOL * scale).numpy()
    assert np.array_equal(labels.cpu().numpy()[clear], want.argmax(dim=1).numpy()[clear])
    assert clear.mean() > 0.8
    ctx = m._context(0)
    feat = torch.empty((40, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(wav, feat)
    two_call = m.forward(feat)
    torch.cuda.synchronize()
    assert torch.equal(two_call, logits)
    with pytest.raises(ModelError):
        m.infer_pcm16(wav.cpu())


def test_streaming_energy_endpointer(dev):
    """kws_stream_vad_f32 against oracle.endpointer fed with the oracle's own log energies: three streams (silence,
    a loud burst in silence, continuous noise) over 260 hops; open / close events at the same hops."""
    from kws.inference import StreamingSpotter
    from oracle.endpointer import EnergyEndpointer

    hops, step, S = 260, 160, 3
    rng = np.random.default_rng(11)
    pcm = np.zeros((S, hops * step), np.int16)
    pcm[1, 60 * step:140 * step] = rng.integers(-20000, 20000, 80 * step)
    pcm[2] = rng.integers(-20000, 20000, hops * step)
    thr = -10.0  # silence sits at log(eps) = -36, the noise at about -1.8 (float PCM in [-1, 1))
    sp = StreamingSpotter(S, vad_log_energy=thr)
    try:
        refs = [EnergyEndpointer(thr) for _ in range(S)]
        # log energies of all frames from the whole signals: pre-emphasis runs over the continuous stream, so a frame
        # that starts right after the burst still sees its last sample
        c0_all = [o_mfcc.mfcc(o_mfcc.pcm16_to_float(pcm[s]), o_mfcc.FrontendSpec(n_samples=pcm.shape[1]))[:, 0] for s in range(S)]
        got_events = [[] for _ in range(S)]
        want_events = [[] for _ in range(S)]
        for t in range(hops):
            sp.push(pcm[:, t * step:(t + 1) * step])
            for s in range(S):
                st = int(sp.vad_state[s])
                if st >> 1:
                    got_events[s].append((t, st >> 1))
                if t < 2:  # no complete frame yet: the device reports 0 and keeps its history untouched
                    assert st == 0
                    continue
                c0 = c0_all[s][t - 2]
                trig, ev = refs[s].update(float(c0))
                assert abs(c0 - thr) > 3.0  # the decision is never close to the threshold
                assert (st & 1) == int(trig), (s, t)
                if ev:
                    want_events[s].append((t, ev))
        assert got_events == want_events
        assert want_events[0] == [] and len(want_events[1]) == 2 and len(want_events[2]) == 1
        assert want_events[1][0][1] == 1 and want_events[1][1][1] == 2
    finally:
        sp.close()


# ------------------------------------------------------------------------------ full-size runs of the other BASELINE configs
def test_cnn_trad_fused_full_batch_properties(dev):
    """BASELINE configs[2] read literally at its full size (B = 4096, MFCC + cnn-trad-fpool3 fused): finite, deterministic,
    permutation-invariant, a sub-batch gives the same bits, every 64th clip against the model's CPU definition."""
    from kws.libs.models import CnnTradFpool3
    from oracle import cnn_trad as o_ct

    state = o_ct.random_state(seed=6)
    m = CnnTradFpool3(12)
    m.load_state_dict(state)
    B = 4096
    clips = synth_clips(B, 21, "uniform")
    gain = np.random.default_rng(22).uniform(0.0, 1.0, B) ** 4
    clips[1::2] = np.round(clips[1::2] * gain[1::2, None]).astype(np.int16)
    clips[9] = 0
    wav = torch.from_numpy(clips).to(dev)
    logits, labels = m.infer_pcm16(wav)
    lg, lb = logits.cpu().numpy(), labels.cpu().numpy()
    assert np.isfinite(lg).all() and lb.min() >= 0 and lb.max() < 12
    assert np.array_equal(lb, np.argmax(lg, axis=1).astype(np.int32))
    l2, y2 = m.infer_pcm16(wav)
    assert torch.equal(l2, logits) and torch.equal(y2, labels)
    perm = torch.from_numpy(np.random.default_rng(23).permutation(B)).to(dev)
    lp, _ = m.infer_pcm16(wav[perm].contiguous())
    assert torch.equal(lp, logits[perm])
    ls, _ = m.infer_pcm16(wav[1000:1021].contiguous())          # 21 clips: not a multiple of the dense kernel's 16
    assert torch.equal(ls, logits[1000:1021])
    pick = np.arange(0, B, 64)
    want = o_ct.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips[pick])))
    scale = max(1.0, float(want.abs().max()))
    err = float(np.abs(lg[pick] - want.numpy()).max())
    assert err <= TOL * scale, (err, scale)
    top2 = torch.topk(want, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 10 * max(err, 1e-7)).numpy()
    assert np.array_equal(lb[pick][clear], want.argmax(dim=1).numpy()[clear]) and clear.mean() > 0.8
    assert float(want.std(dim=0).mean()) > 1e-3 * scale         # the logits move with the input


@pytest.mark.parametrize("use_graph", [False, True])
def test_streaming_full_size_64_and_257_streams(dev, e2e_golden, use_graph):
    """BASELINE configs[4] at its size (64 streams) and at an odd count (257: the last wavefront carries one stream),
    130 hops: four streams against the oracle at full windows, and independence -- a stream's bits do not depend on how
    many other streams are open (the two streams of a wavefront share one packed transform, so a stream is compared with
    the run in which it has the same partner, or none)."""
    from kws.inference import StreamingSpotter

    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    hops = 130
    rng = np.random.default_rng(51)
    pcm = rng.integers(-20000, 20000, size=(257, hops * 160), dtype=np.int16)
    level = rng.uniform(0.0, 1.0, 257) ** 3
    pcm = np.round(pcm * level[:, None]).astype(np.int16)       # every stream at its own level
    pcm[5] = 0
    pcm[7] = np.round(6000 * np.sin(2 * np.pi * 1234 * np.arange(hops * 160) / 16000.0)).astype(np.int16)
    out = {}
    for S in (257, 64, 1):
        src = pcm[:S] if S > 1 else pcm[256:257]
        sp = StreamingSpotter(S, model, use_graph=use_graph)
        sp._ctx.stream_cluster(1)   # one launch shape for all three runs: bit-for-bit independence is a statement about one shape
        try:
            for t in range(hops):
                labels, logits = sp.push(src[:, t * 160:(t + 1) * 160])
            feats, pushed = sp.features()
            out[S] = (labels.copy(), logits.copy(), feats.copy())
            assert pushed == hops
        finally:
            sp.close()
    y257, l257, f257 = out[257]
    assert np.array_equal(l257[:64], out[64][1]) and np.array_equal(y257[:64], out[64][0]) and np.array_equal(f257[:64], out[64][2])
    assert np.array_equal(l257[256:], out[1][1]) and np.array_equal(f257[256:], out[1][2])     # the unpaired last stream
    newest = hops - 3
    for s in (0, 5, 7, 63, 200, 256):
        sig = o_mfcc.pcm16_to_float(pcm[s])
        allf = o_mfcc.mfcc(sig, o_mfcc.FrontendSpec(n_samples=len(sig)))
        want = allf[newest - 98:newest + 1].astype(np.float32)
        assert np.abs(f257[s] - want).max() <= TOL, s
        ref = o_dscnn.forward(state, torch.from_numpy(want)[None, None])
        err = float(np.abs(l257[s] - ref.numpy()[0]).max())
        assert err <= TOL, (s, err)
        assert_labels_match(y257[s:s + 1], ref, err)
    assert len(np.unique(y257)) >= 3 and float(l257.std(axis=0).mean()) >= 0.1


@pytest.mark.parametrize("use_graph", [False, True])
def test_streaming_time_tile_clusters_agree(native, dev, e2e_golden, use_graph):
    """The one-launch push at 64 streams as 4, 2 and 1 workgroups per stream (time tiles with recomputed halos; the tiles'
    pooled partial sums meet in global memory and the last workgroup to arrive runs fc + argmax).  Same feature rings bit for
    bit; logits within 2e-5 of their scale (only the order in which the pooled sums are added differs: per wavefront, then
    per tile) and within 1e-4 of the oracle for every shape; labels identical; a shape is deterministic run to run; the
    automatic choice is 4 tiles at 64 streams.  Across the ring wrap (hop > 99), with a silent and a fading stream."""
    from kws.inference import StreamingSpotter

    S, hops = 64, 118
    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(61)
    pcm = rng.integers(-20000, 20000, size=(S, hops * 160), dtype=np.int16)
    pcm = np.round(pcm * (rng.uniform(0.0, 1.0, S) ** 3)[:, None]).astype(np.int16)
    pcm[3] = 0
    pcm[9] = (pcm[9] * np.linspace(0.0, 1.0, hops * 160)).astype(np.int16)
    pcm[11] = np.round(7000 * np.sin(2 * np.pi * 900 * np.arange(hops * 160) / 16000.0)).astype(np.int16)
    runs = {}
    for shape in (0, 4, 2, 1, 4):           # 0 = automatic; the second "4" checks determinism
        sp = StreamingSpotter(S, model, use_graph=use_graph)
        try:
            sp._ctx.stream_cluster(shape)
            checks = []
            for t in range(hops):
                labels, logits = sp.push(pcm[:, t * 160:(t + 1) * 160])
                if t in (2, 40, 98, 99, 100, hops - 1):
                    checks.append((labels.copy(), logits.copy()))
            feats, pushed = sp.features()
            assert pushed == hops
            runs.setdefault(shape, []).append((checks, feats))
        finally:
            sp.close()
    base_checks, base_feats = runs[1][0]
    newest = hops - 3
    want = np.zeros((S, 99, 10), np.float32)
    for s_ in range(S):
        allf = o_mfcc.mfcc(o_mfcc.pcm16_to_float(pcm[s_]), o_mfcc.FrontendSpec(n_samples=pcm.shape[1]))
        want[s_] = allf[newest - 98:newest + 1]
    ref = o_dscnn.forward(state, torch.from_numpy(want)[:, None])
    for shape, lst in runs.items():
        for checks, feats in lst:
            assert np.array_equal(feats, base_feats), shape
            for (lab, lg), (lab1, lg1) in zip(checks, base_checks):
                assert np.abs(lg - lg1).max() <= 2e-5 * max(1.0, float(np.abs(lg1).max())), shape
                assert np.array_equal(lab, lab1), shape
            err = float(np.abs(checks[-1][1] - ref.numpy()).max())
            assert err <= TOL, (shape, err)
            assert_labels_match(checks[-1][0], ref, err)
    assert np.abs(base_feats - want).max() <= TOL
    a, b = runs[4]
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(a[0], b[0]))                      # deterministic
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(runs[0][0][0], a[0]))             # automatic == 4 tiles at 64 streams


def test_streaming_host_results_are_the_device_results(native, dev, e2e_golden):
    """Zero-copy delivery (kws_stream_host_results / kws_stream_wait_host): the one-launch push writes logits and labels to
    pinned host memory and raises a flag there.  What the host reads after the wait is bit for bit what the device arrays
    hold, at every push, for 1, 64 (4 tiles per stream) and 257 streams; a spotter without it returns the same; the
    wait refuses when the newest push delivered nothing (features only) and before the delivery is enabled."""
    from kws.inference import StreamingSpotter
    from kws.common.errors import ModelError

    model = he_model(e2e_golden)
    rng = np.random.default_rng(77)
    for S, hops in ((1, 30), (64, 110), (257, 12)):
        pcm = rng.integers(-12000, 12000, size=(S, hops * 160), dtype=np.int16)
        a = StreamingSpotter(S, model)                      # host results (the default)
        b = StreamingSpotter(S, model, host_results=False)  # stream synchronise + device-to-host copies
        try:
            assert a._host and not b._host
            for t in range(hops):
                hop_t = pcm[:, t * 160:(t + 1) * 160]
                # host samples take kws_stream_push_host_i16 (the kernel reads the hop from pinned host memory), a device
                # tensor takes kws_stream_push_i16 + kws_stream_wait_host: alternate, the stream state is the same
                la, ga = a.push(hop_t if t % 2 == 0 else torch.from_numpy(np.ascontiguousarray(hop_t)).to(dev))
                lb, gb = b.push(hop_t)
                assert np.array_equal(ga, gb) and np.array_equal(la, lb), (S, t)
                if t % 2:
                    a._ctx.sync()
                    assert np.array_equal(ga, a._logits.cpu().numpy()) and np.array_equal(la, a._labels.cpu().numpy()), (S, t)
            fa, na = a.features()
            fb, nb = b.features()
            assert na == nb == hops and np.array_equal(fa, fb)
            # a features-only push delivers nothing: the wait says so instead of returning the previous hop's results
            a._ctx.stream_push_i16(a._hop_buf, None, None)
            with pytest.raises(ModelError, match="did not deliver"):
                a._ctx.stream_wait_host(S)
            la, ga = a.push(torch.from_numpy(np.ascontiguousarray(pcm[:, :160])).to(dev))  # and the next full push delivers again
            a._ctx.sync()
            assert np.array_equal(ga, a._logits.cpu().numpy())
            with pytest.raises(ModelError, match="kws_stream_host_results"):
                b._ctx.stream_wait_host(S)
            a._ctx.stream_host_results(False)               # switched off: pushes keep working through the device arrays
            a._host = False
            la2, ga2 = a.push(pcm[:, 160:320])
            assert np.isfinite(ga2).all()
        finally:
            a.close()
            b.close()


def test_infer_files_of_any_wav_encoding(dev, tmp_path, e2e_golden):
    """inference(wav) on files that are not 16-bit mono PCM: 24-bit and float32 stereo are decoded to float32 mono as
    librosa.load does (load_audio) and take the float32 path (kws_infer_f32); the labels are the oracle's for the same
    float signals.  A 16-bit mono file next to them still gets the label of the int16 path."""
    import struct

    from kws.inference import KeywordSpotter
    from kws.libs.audio_processor import load_audio

    model = he_model(e2e_golden)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    sp = KeywordSpotter(model)
    clips = e2e_golden["clips"]

    def write(path, payload, tag, ch, bits):
        block = ch * bits // 8
        fmt = struct.pack("<HHIIHH", tag, ch, 16000, 16000 * block, block, bits)
        body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", len(body)) + body)

    a, b, c = clips[15].astype(np.int32), clips[30].astype(np.int32), clips[40]
    v24 = (a << 8) + 77                                                      # 24-bit samples that are not 16-bit values
    write(tmp_path / "x24.wav", b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in v24), 1, 1, 24)
    st = np.stack([a / 32768.0, b / 32768.0 * 0.5], axis=1).astype("<f4")    # float32 stereo
    write(tmp_path / "xf32s.wav", st.tobytes(), 3, 2, 32)
    write(tmp_path / "x16.wav", c.tobytes(), 1, 1, 16)
    files = [str(tmp_path / n) for n in ("x24.wav", "xf32s.wav", "x16.wav")]
    got = sp.infer_files(files)
    sigs = [load_audio(f) for f in files]
    feats = np.stack([o_mfcc.mfcc(s_, o_mfcc.DEFAULT_SPEC).astype(np.float32) for s_ in sigs])[:, None]
    want = o_dscnn.forward(state, torch.from_numpy(feats))
    assert [g[0] for g in got] == o_dscnn.predict(want).tolist()
    assert got[2][0] == int(e2e_golden["he.label"][8 + 40])                  # the 16-bit file: same label as the int16 path
    labels, logits = sp.infer_f32(np.stack(sigs))
    assert np.abs(logits - want.numpy()).max() <= TOL
