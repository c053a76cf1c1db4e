#!/usr/bin/env python3
"""Diagnostics (GPU box): per golden clip, the MFCC error of the kernel against the float64 oracle (worst frame and
coefficient) and the logit error it leaves at the end of the fused path (signal-preserving golden weights)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
from kws import _native
from oracle import psf_mfcc as o
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_golden.npz"))
clips, names = g["clips"], g["names"]
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
ctx.load_dscnn(g["he.blob"], 12)
B = len(clips)
wav = torch.from_numpy(clips).to(dev)
feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
ctx.mfcc_i16(wav, feat); ctx.infer_i16(wav, logits, None); ctx.sync()
got = feat.cpu().numpy()[:, 0].astype(np.float64)
want = np.stack([o.extract_features_pcm16(c) for c in clips])
lerr = np.abs(logits.cpu().numpy() - g["he.logits"][8:]).max(axis=1)
for i in range(B):
    e = np.abs(got[i] - want[i])
    f, k = np.unravel_index(e.argmax(), e.shape)
    print(f"{i:2d} {names[i]:24s} mfcc max err {e.max():.2e} at frame {f:2d} coef {k}  rms {np.sqrt((e**2).mean()):.2e}   logit err {lerr[i]:.2e}")
print("overall mfcc", np.abs(got - want).max(), "logits", lerr.max())
