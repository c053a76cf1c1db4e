#!/usr/bin/env python3
"""CPU study behind the float32 front end's precision flag: which per-frame quantity, computable
from what the float32 kernel already holds, predicts that the float32 cepstra miss the float64 reference by more than
1e-4?  Simulates the float32 arithmetic (scipy pocketfft in single precision on single frames, float32 mel / log / DCT)
on the golden clips + speech-like clips and prints, per candidate criterion and threshold, the worst error among the
frames it would NOT flag and the fraction of frames it flags.  Test/diagnostic infrastructure: imports oracle/."""
import os
import sys

import numpy as np
import scipy.fft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import psf_mfcc as o  # noqa: E402
import speechlike  # noqa: E402


def f32_frontend(pcm):
    """float32 model of the kernel: returns (cepstra [99,10], log-mel [99,26], log-energy [99])."""
    x = o.pcm16_to_float(pcm)
    y = o.preemphasis(x, 0.97).astype(np.float32)
    fr = o.framesig(y, 400, 160).astype(np.float32)
    pad = np.zeros((fr.shape[0], 512), np.float32)
    pad[:, :400] = fr
    spec = scipy.fft.rfft(pad, axis=1)
    assert spec.dtype == np.complex64
    p = ((spec.real ** 2 + spec.imag ** 2) * np.float32(1.0 / 512)).astype(np.float32)
    p[np.all(fr == 0, axis=1)] = 0
    fb = o.get_filterbanks().astype(np.float32)
    mel = (p @ fb.T).astype(np.float32)
    en = p.sum(1, dtype=np.float32)
    mel = np.where(mel == 0, np.float32(o.EPS), mel)
    en = np.where(en == 0, np.float32(o.EPS), en)
    lm = np.log(mel).astype(np.float32)
    d = (o.dct2_ortho_matrix(26, 10) * o.lifter_vector(10)[:, None]).astype(np.float32)
    c = ((lm - lm[:, :1]) @ d.T).astype(np.float32)
    c[:, 0] = np.log(en)
    return c.astype(np.float64), lm.astype(np.float64), np.log(en).astype(np.float64)


def gpu_frontend(clips):
    """The product kernel's cepstra (GPU box only)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
    from kws import _native
    ctx = _native.Context(0)
    dev = torch.device("cuda", 0)
    out = torch.empty((len(clips), 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(torch.from_numpy(np.ascontiguousarray(clips)).to(dev), out)
    ctx.sync()
    return out.cpu().numpy()[:, 0].astype(np.float64)


def main():
    use_gpu = "--gpu" in sys.argv
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_golden.npz"))
    sp, sp_names = speechlike.speechlike_set(32, 400)
    rng = np.random.default_rng(0)
    noise = rng.integers(-32768, 32768, size=(256 if use_gpu else 16, 16000), dtype=np.int16)
    gauss = np.clip(np.round(np.random.default_rng(1).standard_normal((64, 16000)) * 3000.0), -32768, 32767).astype(np.int16)
    sets = {"golden48": g["clips"], "speech32": sp, "uniform": noise, "gauss64": gauss}
    rows = []
    edges = o.mel_bin_edges(26, 512, 16000).astype(int)
    logw = np.log((edges[2:] - edges[:-2]) / 2.0)   # sum of a triangle's weights ~ half its base
    for tag, clips in sets.items():
        gpu = gpu_frontend(clips) if use_gpu else None
        for ci, c in enumerate(clips):
            sim, _, _ = f32_frontend(c)
            got = gpu[ci] if use_gpu else sim
            want = o.extract_features_pcm16(c)
            feat, en = o.fbank(o.fix_length(o.pcm16_to_float(c), 16000))
            lm, le = np.log(feat), np.log(en)
            err = np.abs(got - want).max(axis=1)
            span = lm.max(1) - lm.min(1)
            span_e = le - lm.min(1)
            span_w = (le[:, None] - lm + logw[None, :]).max(1)
            rows.append(np.stack([err, span, span_e, span_w, np.full(99, list(sets).index(tag))], 1))
    a = np.concatenate(rows)
    err, span, span_e, span_w, which = a.T
    print(("GPU kernel" if use_gpu else "CPU float32 model") + f": {len(err)} frames; err > 1e-4 on {np.mean(err > 1e-4) * 100:.2f} %, worst {err.max():.2e}")
    for i, t in enumerate(sets):
        e = err[which == i]
        print(f"  {t}: err > 1e-4 on {np.mean(e > 1e-4) * 100:.2f} % of frames, > 5e-5 on {np.mean(e > 5e-5) * 100:.2f} %, worst {e.max():.2e}")
    for name, crit in (("max-min log-mel", span), ("log E - min log-mel", span_e), ("max_j log E - log(mel_j / width_j)", span_w)):
        print(f"criterion: {name}")
        for thr in (10.0, 11.0, 11.5, 12.0, 12.5, 13.0, 13.5, 14.0, 14.5, 15.0, 16.0, 17.0, 18.0):
            keep = crit <= thr
            per_set = " ".join(f"{t}:{np.mean(~keep[which == i]) * 100:5.1f}%" for i, t in enumerate(sets))
            print(f"  thr {thr:5.1f}: flagged {np.mean(~keep) * 100:5.2f} % ({per_set})   worst unflagged err {err[keep].max():.2e}")


if __name__ == "__main__":
    main()
