#!/usr/bin/env python3
"""Generate the golden fixtures by importing the reference's two importable
hot-path modules.  Runs ONLY in the build container (needs /root/reference):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes (data only -- inputs and expected outputs, no reference source):
  tests/golden/sigproc_golden.npz   PCM16 clips -> reference preemphasis /
                                    framesig(rect) / powspec outputs
  tests/golden/dscnn_golden.npz     seeded state_dicts + inputs -> reference
                                    DepthwiseSeparableConv logits, per-layer probes
  tests/golden/dsblock_golden.npz   DepthwiseSeparableConvBlock on its own: four shapes -> reference outputs
  tests/golden/multichannel_golden.npz  DepthwiseSeparableConv(input_channels=3) on ten 3-channel maps -> reference logits
  tests/golden/stress_golden.npz    16 speech-like clips (tests/golden/speechlike.py: glottal pulse trains through formant
                                    resonators, pauses of exact zeros / +-1 LSB dither, -6 .. -50 dBFS) and three weight
                                    tags -- the 'he' weights, the same with a 5x classifier gain, and weights whose conv1
                                    is NOT pre-divided by the MFCC maps' RMS (activations and logits ~15x larger) ->
                                    reference logits / labels for these 16 + the 48 diverse clips of e2e_golden.npz
  tests/golden/anymap_golden.npz    DepthwiseSeparableConv on feature maps other than 99 x 10: 149 x 10 (clip_duration_ms=1500),
                                    99 x 13 (num_cepstral_coeffs=13), a 20 x 8 map and a 3-channel 50 x 12 map -> reference
                                    logits, labels and every stage's output for two probe inputs
  tests/golden/e2e_golden.npz       48 diverse PCM16 clips + 8 random maps, signal-preserving ("he") weights ->
                                    reference logits / labels / per-layer probes whose VALUES DEPEND ON THE
                                    AUDIO (labels span >= 6 classes, logit std across clips >= 0.1 -- asserted
                                    here), plus two tie cases for the first-maximum rule of torch.max

Reference modules used (imported, never copied):
  kws/libs/speech_features/sigproc.py:14-103   (framesig, magspec, powspec, preemphasis)
  kws/libs/models.py:122-183                   (DepthwiseSeparableConv)
The psf-only tail (mel/log/DCT/lifter/energy) has no importable reference
here (python_speech_features absent) -> no golden vector for it: parity
unpinned, covered by analytic known answers in tests/test_oracle.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from kws.libs import models as ref_models  # noqa: E402  (reference)
from kws.libs.speech_features import sigproc as ref_sigproc  # noqa: E402  (reference)

from oracle import dscnn as o_dscnn  # noqa: E402
from oracle import psf_mfcc as o_mfcc  # noqa: E402

N = 16000
FRAME_LEN, FRAME_STEP, NFFT = 400, 160, 512
KEEP_FRAMES = np.array([0, 1, 2, 48, 49, 50, 96, 97, 98])
PROBE = [0, 4]  # samples whose activations are probed in detail (one random input, one MFCC input)


def make_clips() -> "tuple[np.ndarray, list]":
    rs = np.random.RandomState(20251004)
    t = np.arange(N)
    clips, names = [], []
    names.append("zeros"); clips.append(np.zeros(N, np.int16))
    imp = np.zeros(N, np.int16); imp[1000] = 32767
    names.append("impulse@1000"); clips.append(imp)
    names.append("square_fullscale_p64"); clips.append(np.where((t // 32) % 2 == 0, 32767, -32768).astype(np.int16))
    names.append("sine_1kHz_a12000"); clips.append(np.round(12000 * np.sin(2 * np.pi * 1000 * t / 16000.0)).astype(np.int16))
    names.append("uniform_fullrange"); clips.append(rs.randint(-32768, 32768, size=N).astype(np.int16))
    names.append("gauss_sigma3000"); clips.append(np.clip(np.round(rs.standard_normal(N) * 3000), -32768, 32767).astype(np.int16))
    quiet = np.zeros(N, np.int16); quiet[8000:] = rs.randint(-3, 4, size=N - 8000)
    names.append("half_silent_lsb_noise"); clips.append(quiet)
    names.append("dc_minus_full"); clips.append(np.full(N, -32768, np.int16))
    return np.stack(clips), names


def ones_window(n, device=None):
    return torch.ones(n, dtype=torch.float64, device=device)


def sigproc_golden():
    clips, names = make_clips()
    pre32, pspec, energy = [], [], []
    for clip in clips:
        x32 = torch.from_numpy(o_mfcc.pcm16_to_float(clip))           # float32, as librosa hands over
        y32 = ref_sigproc.preemphasis(x32, 0.97)                      # reference, float32
        frames = ref_sigproc.framesig(y32.double(), FRAME_LEN, FRAME_STEP, winfunc=ones_window)  # float64 like psf
        ps = ref_sigproc.powspec(frames, NFFT).numpy()                # reference, float64 [99,257]
        assert ps.shape == (99, 257) and ps.dtype == np.float64
        # the oracle must agree with the reference before its output is trusted anywhere
        o_y = o_mfcc.preemphasis(x32.numpy(), 0.97)
        assert o_y.dtype == np.float32 and np.array_equal(o_y, y32.numpy())
        o_ps = o_mfcc.powspec(o_mfcc.framesig(o_y, FRAME_LEN, FRAME_STEP), NFFT)
        np.testing.assert_allclose(o_ps, ps, rtol=1e-9, atol=1e-18 + 1e-12 * ps.max())
        pre32.append(y32.numpy()[:64].copy())
        pspec.append(ps[KEEP_FRAMES])
        energy.append(ps.sum(1))

    # the recipe of the reference's own unit test (tests/kws/libs/speech_features/test_sigproc.py:8-21),
    # with the rectangular window psf uses (the test as written feeds Hann on one side)
    np.random.seed(0)
    sig = np.random.rand(N)
    fr = ref_sigproc.framesig(torch.tensor(sig), FRAME_LEN, FRAME_STEP, winfunc=ones_window)
    mag = ref_sigproc.magspec(fr, NFFT).numpy()
    o_mag = o_mfcc.magspec(o_mfcc.framesig(sig, FRAME_LEN, FRAME_STEP), NFFT)
    np.testing.assert_allclose(o_mag, mag, rtol=1e-5, atol=1e-8)      # the reference test's own tolerance

    np.savez_compressed(
        os.path.join(HERE, "sigproc_golden.npz"),
        clips=clips, names=np.array(names), keep_frames=KEEP_FRAMES,
        preemph_head_f32=np.stack(pre32), powspec_f64=np.stack(pspec), frame_energy_f64=np.stack(energy),
        reftest_magspec_f64=mag[KEEP_FRAMES], reftest_magspec_colsum_f64=mag.sum(0),
    )
    print("sigproc_golden.npz:", len(names), "clips", names)


def probe_layers(layers: dict) -> dict:
    """Compact, layout-sensitive summary of every activation tensor."""
    out = {}
    for name, t in layers.items():
        a = t.detach().numpy()
        if a.ndim == 4:
            out[name + ".shape"] = np.array(a.shape)
            out[name + ".chan_mean"] = a.mean(axis=(2, 3))                       # [B,64]
            out[name + ".corner"] = a[PROBE, :, :3, :3].copy()                   # ring + first interior values
            out[name + ".row_mid"] = a[PROBE, :, a.shape[2] // 2, :].copy()      # one full row
            out[name + ".col1"] = a[PROBE, :, :, 1].copy()                       # one full column
        else:
            out[name] = a.copy()
    return out


def dscnn_golden():
    torch.manual_seed(1234)
    rs = np.random.RandomState(7)
    x_rand = torch.from_numpy(rs.standard_normal((3, 1, 99, 10)).astype(np.float32))
    clips, _ = make_clips()
    x_mfcc = torch.from_numpy(o_mfcc.collate_pcm16(clips[[0, 3, 4, 5]]))         # realistic feature ranges incl. silence
    x = torch.cat([x_rand, x_mfcc], 0)

    save = {"x": x.numpy()}
    # (a) every parameter N(0, 0.1), biases included  (b) the reference's own default init, seed 0
    torch.manual_seed(0)
    ref_default = ref_models.DepthwiseSeparableConv(num_classes=12).eval()
    states = {
        "n01": o_dscnn.random_state(seed=1, std=0.1),
        "default": {k: v.detach().clone() for k, v in ref_default.state_dict().items()},
    }
    assert list(ref_default.state_dict().keys()) == o_dscnn.STATE_KEYS
    for tag, st in states.items():
        ref = ref_models.DepthwiseSeparableConv(num_classes=12).eval()
        ref.load_state_dict(st)
        with torch.no_grad():
            # reference forward with per-layer captures via the module's own submodules
            h = torch.relu(ref.conv1(x)); layers = {"conv1": h}
            for i, blk in enumerate([ref.dsconv1, ref.dsconv2, ref.dsconv3, ref.dsconv4], 1):
                d = blk.depthwise(h); layers[f"dsconv{i}.depthwise"] = d
                h = blk(h); layers[f"dsconv{i}"] = h
            logits = ref(x)
            o_logits, o_layers = o_dscnn.forward(st, x, return_layers=True)
        np.testing.assert_allclose(o_logits.numpy(), logits.numpy(), rtol=0, atol=1e-6)
        for name, t in layers.items():
            np.testing.assert_allclose(o_layers[name].numpy(), t.numpy(), rtol=0, atol=2e-6 * max(1.0, float(t.abs().max())))
        save[f"{tag}.blob"] = o_dscnn.flatten_state(st)
        save[f"{tag}.logits"] = logits.numpy()
        save[f"{tag}.label"] = torch.max(logits, 1)[1].numpy()
        for k, v in probe_layers(layers).items():
            save[f"{tag}.{k}"] = v
        print(tag, "logits[0] =", np.round(logits[0].numpy(), 4))
    np.savez_compressed(os.path.join(HERE, "dscnn_golden.npz"), **save)
    print("dscnn_golden.npz written")


def diverse_clips() -> "tuple[np.ndarray, list]":
    """48 int16 clips whose spectra, levels and envelopes differ: the 8 sigproc clips, sines, chirps, uniform and
    Gaussian noise at several levels, gated noise/tone bursts over a noise floor."""
    base, base_names = make_clips()
    rs = np.random.RandomState(5)
    t = np.arange(N)
    clips, names = list(base), list(base_names)
    for f in (200, 440, 1000, 2500, 5000, 7000):
        names.append(f"sine_{f}Hz_a8000"); clips.append(np.round(8000 * np.sin(2 * np.pi * f * t / 16000.0)).astype(np.int16))
    for f0, f1 in ((100, 4000), (6000, 300), (50, 7900)):
        ph = 2 * np.pi * (f0 * t / 16000.0 + (f1 - f0) * t * t / (2.0 * 16000 * 16000))
        names.append(f"chirp_{f0}_{f1}"); clips.append(np.round(15000 * np.sin(ph)).astype(np.int16))
    for amp in (32767, 3000, 300, 30):
        names.append(f"uniform_a{amp}"); clips.append(rs.randint(-amp, amp + 1, size=N).astype(np.int16))
    for sigma in (9000, 1000, 100, 10):
        names.append(f"gauss_s{sigma}")
        clips.append(np.clip(np.round(rs.standard_normal(N) * sigma), -32768, 32767).astype(np.int16))
    while len(clips) < 48:
        env = np.zeros(N)
        a = rs.randint(0, 12000)
        env[a:a + rs.randint(1000, 4000)] = 1
        f = rs.uniform(100, 6000)
        tone = np.sin(2 * np.pi * f * t / 16000.0) * rs.choice([0, 1000, 6000])
        x = (rs.standard_normal(N) * rs.choice([30, 300, 3000, 9000]) + tone) * env + rs.standard_normal(N) * rs.choice([0, 2, 20])
        names.append(f"burst_{len(clips)}"); clips.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    return np.stack(clips), names


def he_state(seed: int, in_scale: float = 15.0, fc_std: float = 0.5, bias_std: float = 0.1):
    """Signal-preserving weights: conv weights N(0, 2/fan_in) (fan_in = in_channels/groups * kh * kw), conv1 also
    divided by `in_scale` (an input normalisation folded into the first layer: the MFCC maps have RMS ~ 15), biases
    N(0, 0.1), fc N(0, 0.5).  Activations stay O(1..30) through the net, so the logits depend on the audio."""
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in o_dscnn.state_shapes(12).items():
        if k.endswith("bias"):
            w = rs.standard_normal(shp) * bias_std
        elif k.startswith("fc"):
            w = rs.standard_normal(shp) * fc_std
        else:
            w = rs.standard_normal(shp) * np.sqrt(2.0 / int(np.prod(shp[1:])))
            if k.startswith("conv1"):
                w = w / in_scale
        out[k] = torch.from_numpy(w.astype(np.float32))
    return out


def ref_forward_with_layers(st, x):
    ref = ref_models.DepthwiseSeparableConv(num_classes=12).eval()
    ref.load_state_dict(st)
    with torch.no_grad():
        h = torch.relu(ref.conv1(x)); layers = {"conv1": h}
        for i, blk in enumerate([ref.dsconv1, ref.dsconv2, ref.dsconv3, ref.dsconv4], 1):
            d = blk.depthwise(h); layers[f"dsconv{i}.depthwise"] = d
            h = blk(h); layers[f"dsconv{i}"] = h
        logits = ref(x)
        labels = torch.max(logits, 1)[1]                                         # kws/libs/training.py:371
    return logits, labels, layers


def e2e_golden():
    clips, names = diverse_clips()
    rs = np.random.RandomState(11)
    x_rand = torch.from_numpy((3.0 * rs.standard_normal((8, 1, 99, 10))).astype(np.float32))
    x_mfcc = torch.from_numpy(o_mfcc.collate_pcm16(clips))     # oracle front end (psf tail unpinned), float32 as the loader casts
    x = torch.cat([x_rand, x_mfcc], 0)

    st = he_state(seed=2)
    logits0, _, _ = ref_forward_with_layers(st, x_mfcc)
    # a trained classifier's classes are balanced over its data: centre the logits over the clip set (float32)
    st["fc.bias"] = (st["fc.bias"] - logits0.mean(0)).float()
    logits, labels, layers = ref_forward_with_layers(st, x)
    o_logits, o_layers = o_dscnn.forward(st, x, return_layers=True)
    np.testing.assert_allclose(o_logits.numpy(), logits.numpy(), rtol=0, atol=2e-6 * float(logits.abs().max()))
    for name, t in layers.items():
        np.testing.assert_allclose(o_layers[name].numpy(), t.numpy(), rtol=0, atol=2e-6 * float(t.abs().max()))
    assert torch.equal(o_dscnn.predict(o_logits), labels)
    clip_logits, clip_labels = logits[8:], labels[8:]
    n_classes = len(set(clip_labels.tolist()))
    spread = float(clip_logits.std(0).mean())
    top2 = torch.topk(logits, 2, dim=1).values
    print(f"he: {n_classes} classes over {len(clips)} clips, logit std across clips {spread:.3f}, "
          f"|logit| max {float(logits.abs().max()):.2f}, min top-2 margin {float((top2[:, 0] - top2[:, 1]).min()):.2e}")
    assert n_classes >= 6 and spread >= 0.1, "the fixture must depend on the audio"

    save = {"clips": clips, "names": np.array(names), "x_rand": x_rand.numpy(),
            "he.blob": o_dscnn.flatten_state(st), "he.logits": logits.numpy(), "he.label": labels.numpy()}
    for k, v in probe_layers(layers).items():
        save[f"he.{k}"] = v
    save["he.dsconv4.full"] = layers["dsconv4"][PROBE].numpy()                   # block 4 output, ring included
    save["he.pool"] = torch.nn.functional.adaptive_avg_pool2d(layers["dsconv4"], (1, 1)).reshape(len(x), -1).numpy()

    # ties (torch.max returns the FIRST maximum): (a) every class row identical -> every logit ties -> label 0;
    # (b) row hi := row lo for the most frequent label lo -> wherever lo wins, hi ties with it and lo must be reported
    tie_a = {k: v.clone() for k, v in st.items()}
    tie_a["fc.weight"] = st["fc.weight"][3:4].repeat(12, 1).contiguous()
    tie_a["fc.bias"] = st["fc.bias"][3:4].repeat(12).contiguous()
    la, ya, _ = ref_forward_with_layers(tie_a, x)
    assert bool((la == la[:, :1]).all()) and bool((ya == 0).all())
    lo = int(torch.bincount(clip_labels, minlength=12)[:11].argmax())
    hi = 11
    tie_b = {k: v.clone() for k, v in st.items()}
    tie_b["fc.weight"][hi] = st["fc.weight"][lo]
    tie_b["fc.bias"][hi] = st["fc.bias"][lo]
    lb, yb, _ = ref_forward_with_layers(tie_b, x)
    assert bool((lb[:, lo] == lb[:, hi]).all()) and int((yb == lo).sum()) >= 3 and not bool((yb == hi).any())
    save.update({"tie_all.blob": o_dscnn.flatten_state(tie_a), "tie_all.logits": la.numpy(), "tie_all.label": ya.numpy(),
                 "tie_pair.blob": o_dscnn.flatten_state(tie_b), "tie_pair.logits": lb.numpy(), "tie_pair.label": yb.numpy(),
                 "tie_pair.lo_hi": np.array([lo, hi])})
    np.savez_compressed(os.path.join(HERE, "e2e_golden.npz"), **save)
    print("e2e_golden.npz written; labels:", labels.tolist())


def stress_golden():
    """Inputs and weights that can break the 1e-4 logit bound: speech-like clips -- a few strong
    harmonics over a floor 60-90 dB down inside one frame, onsets next to digital silence, levels down to -50 dBFS -- and
    weights with larger gains than the 'he' tag.  Logits and labels from the imported reference model on the oracle's MFCC."""
    sys.path.insert(0, HERE)
    import speechlike

    sp_clips, sp_names = speechlike.speechlike_set(16, 400)
    div_clips, _ = diverse_clips()
    clips = np.concatenate([div_clips, sp_clips])
    # how hard these clips are for a float32 front end: frames whose mel bands span more than 50 dB (11.5 in log power)
    over50 = []
    for c in sp_clips:
        feat, _ = o_mfcc.fbank(o_mfcc.fix_length(o_mfcc.pcm16_to_float(c), N))
        lm = np.log(feat)
        over50.append(((lm.max(1) - lm.min(1)) > 11.5).mean())
    print(f"speech-like clips: {100 * np.mean(over50):.1f} % of frames span more than 50 dB (per clip: "
          + " ".join(f"{100 * v:.0f}" for v in over50) + ")")
    x = torch.from_numpy(o_mfcc.collate_pcm16(clips))
    g = np.load(os.path.join(HERE, "e2e_golden.npz"))
    he = {}
    off = 0
    for k, shp in o_dscnn.state_shapes(12).items():
        n = int(np.prod(shp))
        he[k] = torch.from_numpy(g["he.blob"][off:off + n].reshape(shp).copy())
        off += n
    he5 = {k: v.clone() for k, v in he.items()}
    he5["fc.weight"] = (he["fc.weight"] * 5.0).float()
    he5["fc.bias"] = (he["fc.bias"] * 5.0).float()
    raw = he_state(seed=3, in_scale=1.0)                      # conv1 NOT pre-divided by the MFCC maps' RMS
    l0, _, _ = ref_forward_with_layers(raw, x)
    raw["fc.bias"] = (raw["fc.bias"] - l0.mean(0)).float()      # classes balanced over the clip set, as in the 'he' tag
    save = {"speech_clips": sp_clips, "speech_names": np.array(sp_names), "speech_frac_frames_over_50dB": np.array(over50)}
    for tag, st in (("he", he), ("he5", he5), ("raw", raw)):
        logits, labels, _ = ref_forward_with_layers(st, x)
        o_logits = o_dscnn.forward(st, x)
        np.testing.assert_allclose(o_logits.numpy(), logits.numpy(), rtol=0, atol=2e-6 * float(logits.abs().max()))
        assert torch.equal(o_dscnn.predict(o_logits), labels)
        top2 = torch.topk(logits, 2, dim=1).values
        print(f"{tag}: {len(set(labels.tolist()))} classes over {len(clips)} clips, logit std across clips {float(logits.std(0).mean()):.2f}, "
              f"|logit| max {float(logits.abs().max()):.1f}, min top-2 margin {float((top2[:, 0] - top2[:, 1]).min()):.2e}, "
              f"oracle vs reference {float((o_logits - logits).abs().max()):.1e}")
        save[f"{tag}.blob"] = o_dscnn.flatten_state(st)
        save[f"{tag}.logits"] = logits.numpy()
        save[f"{tag}.label"] = labels.numpy()
    if "he" in save:
        np.testing.assert_array_equal(save["he.logits"][:48], g["he.logits"][8:])  # the same model on the same 48 clips
    np.savez_compressed(os.path.join(HERE, "stress_golden.npz"), **save)
    print("stress_golden.npz written")


def anymap_golden():
    """DepthwiseSeparableConv.forward takes any [B,C,T,F] (kws/libs/models.py:160-183; adaptive pooling) and AudioConfig's
    clip_duration_ms / num_cepstral_coeffs change T and F (kws/libs/audio_processor.py:37-46): the imported reference model on
    such maps -- MFCC maps of 1.5 s clips (149 x 10) and with 13 cepstra (99 x 13), a small random map and a 3-channel one."""
    div, _ = diverse_clips()
    rs = np.random.RandomState(77)
    pick = [3, 4, 9, 15, 18, 23, 30, 40]
    long_clips = np.concatenate([div[pick], div[pick][:, :8000]], axis=1)          # int16 [8, 24000]: 1.5 s
    spec_long = o_mfcc.FrontendSpec(n_samples=24000)
    spec_13 = o_mfcc.FrontendSpec(numcep=13)
    assert spec_long.num_frames == 149
    cases = {
        "t149": torch.from_numpy(np.stack([o_mfcc.extract_features_pcm16(c, spec_long).astype(np.float32) for c in long_clips])[:, None]),
        "f13": torch.from_numpy(np.stack([o_mfcc.extract_features_pcm16(c, spec_13).astype(np.float32) for c in div[pick]])[:, None]),
        "small": torch.from_numpy((3.0 * rs.standard_normal((5, 1, 20, 8))).astype(np.float32)),
        "c3": torch.from_numpy((3.0 * rs.standard_normal((4, 3, 50, 12))).astype(np.float32)),
    }
    save = {"long_clips": long_clips, "clips13": div[pick]}
    for tag, x in cases.items():
        cin = x.shape[1]
        ref = ref_models.DepthwiseSeparableConv(num_classes=12, input_channels=cin).eval()
        st = {}
        srs = np.random.RandomState(900 + cin)
        for k, v in ref.state_dict().items():
            shp = tuple(v.shape)
            if k.endswith("bias"):
                w = srs.standard_normal(shp) * 0.1
            elif k.startswith("fc"):
                w = srs.standard_normal(shp) * 0.5
            else:
                w = srs.standard_normal(shp) * np.sqrt(2.0 / int(np.prod(shp[1:])))
                if k.startswith("conv1") and tag in ("t149", "f13"):
                    w = w / 15.0
            st[k] = torch.from_numpy(w.astype(np.float32))
        ref.load_state_dict(st)
        with torch.no_grad():
            st["fc.bias"] = (st["fc.bias"] - ref(x).mean(0)).float()
            ref.load_state_dict(st)
            h = torch.relu(ref.conv1(x)); layers = [h]
            for blk in (ref.dsconv1, ref.dsconv2, ref.dsconv3, ref.dsconv4):
                h = blk(h); layers.append(h)
            logits = ref(x)
            labels = torch.max(logits, 1)[1]
        if cin == 1:
            o_logits = o_dscnn.forward(st, x)
            np.testing.assert_allclose(o_logits.numpy(), logits.numpy(), rtol=0, atol=2e-6 * float(logits.abs().max()))
        save[f"{tag}.x"] = x.numpy()
        save[f"{tag}.blob"] = np.concatenate([st[k].reshape(-1).numpy() for k in ref.state_dict().keys()])
        save[f"{tag}.logits"] = logits.numpy()
        save[f"{tag}.label"] = labels.numpy()
        for i, t in enumerate(layers):
            save[f"{tag}.layer{i}"] = t[:2].numpy()                       # stage outputs (rings included) of the first two inputs
        print(f"anymap {tag}: x {tuple(x.shape)} -> stages {[tuple(t.shape[2:]) for t in layers]}, labels {labels.tolist()}, "
              f"logit std {float(logits.std(0).mean()):.2f}")
    np.savez_compressed(os.path.join(HERE, "anymap_golden.npz"), **save)
    print("anymap_golden.npz written")


def dsblock_golden():
    """DepthwiseSeparableConvBlock (kws/libs/models.py:75-119) on its own, imported and run: three shapes incl. a
    non-default kernel size / stride / padding and channel counts that are not multiples of the kernel's tiles."""
    cases = [  # (c_in, c_out, k, stride, padding, B, H, W)
        (64, 64, 3, 1, 1, 3, 9, 5),
        (8, 24, 3, 1, 1, 2, 6, 7),
        (20, 70, 5, 2, 2, 2, 11, 9),
        (3, 5, 3, 1, 0, 1, 4, 4),
    ]
    save = {"cases": np.array(cases)}
    for i, (ci, co, k, st, pd, B, H, W) in enumerate(cases):
        torch.manual_seed(100 + i)
        blk = ref_models.DepthwiseSeparableConvBlock(ci, co, kernel_size=k, stride=st, padding=pd).eval()
        with torch.no_grad():
            for prm in blk.parameters():
                prm.copy_(torch.randn_like(prm) * (0.5 if prm.dim() > 1 else 0.3))
            x = torch.randn(B, ci, H, W) * 2.0
            y = blk(x)
        o_y = o_dscnn.block_forward({n: t.detach() for n, t in blk.state_dict().items()}, x, k, st, pd)
        np.testing.assert_allclose(o_y.numpy(), y.numpy(), rtol=0, atol=2e-6 * float(y.abs().max()))
        save[f"c{i}.x"] = x.numpy()
        save[f"c{i}.y"] = y.numpy()
        for n, t in blk.state_dict().items():
            save[f"c{i}.{n}"] = t.detach().numpy()
        print(f"dsblock case {i}: in {tuple(x.shape)} -> out {tuple(y.shape)}")
    np.savez_compressed(os.path.join(HERE, "dsblock_golden.npz"), **save)


def multichannel_golden():
    """DepthwiseSeparableConv(num_classes=12, input_channels=3) (kws/libs/models.py:125,135), imported and run on
    3-channel maps; weights scaled like the 'he' tag so the logits follow the input."""
    rs = np.random.RandomState(21)
    ref = ref_models.DepthwiseSeparableConv(num_classes=12, input_channels=3).eval()
    st = {}
    for k, v in ref.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("bias"):
            w = rs.standard_normal(shp) * 0.1
        elif k.startswith("fc"):
            w = rs.standard_normal(shp) * 0.5
        else:
            w = rs.standard_normal(shp) * np.sqrt(2.0 / int(np.prod(shp[1:])))
        st[k] = torch.from_numpy(w.astype(np.float32))
    ref.load_state_dict(st)
    x = rs.standard_normal((10, 3, 99, 10)) * np.array([0.2, 0.0, 1, 2, 4, 8, 3, 3, 3, 3]).reshape(-1, 1, 1, 1)
    x[6, 0] = 0; x[7, 1] = 0; x[8, 2] = 0; x[9] += 5.0           # a silent channel each, and an offset
    x = torch.from_numpy(x.astype(np.float32))
    with torch.no_grad():
        st["fc.bias"] = (st["fc.bias"] - ref(x).mean(0)).float()    # classes balanced over the inputs, as in the 'he' tag
        ref.load_state_dict(st)
        logits = ref(x)
        conv1 = torch.relu(ref.conv1(x))
    assert len(set(torch.max(logits, 1)[1].tolist())) >= 3
    blob = np.concatenate([st[k].reshape(-1).numpy() for k in ref.state_dict().keys()])
    assert blob.size == 6400 * 3 + 19264 + 65 * 12
    np.savez_compressed(os.path.join(HERE, "multichannel_golden.npz"), x=x.numpy(), blob=blob, logits=logits.numpy(),
                        label=torch.max(logits, 1)[1].numpy(), conv1=conv1[:2].numpy())
    print("multichannel: logits std across inputs", float(logits.std(0).mean()), "labels", torch.max(logits, 1)[1].tolist())


if __name__ == "__main__":
    only = set(sys.argv[1:])  # e.g. `make_golden.py stress` regenerates one file; no argument = all
    for name, fn in (("sigproc", sigproc_golden), ("dscnn", dscnn_golden), ("e2e", e2e_golden), ("stress", stress_golden), ("anymap", anymap_golden),
                     ("dsblock", dsblock_golden), ("multichannel", multichannel_golden)):
        if not only or name in only:
            fn()
