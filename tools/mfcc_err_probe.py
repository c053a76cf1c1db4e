#!/usr/bin/env python3
"""Diagnostics (GPU box): where the float32 MFCC differs most from the float64 oracle (per coefficient)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native
from oracle import psf_mfcc as o
B = 512
clips = bench.synth_clips(B, 0)
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
ctx.mfcc_i16(torch.from_numpy(clips).to(dev), out); ctx.sync()
got = out.cpu().numpy()[:, 0].astype(np.float64)
want = o.collate_pcm16(clips)[:, 0].astype(np.float64) if o.collate_pcm16(clips[:1]).ndim == 4 else None
want = np.stack([o.extract_features_pcm16(c) for c in clips])
err = np.abs(got - want)
print("max err per coefficient:", np.array2string(err.max(axis=(0, 1)), precision=2))
print("rms err per coefficient:", np.array2string(np.sqrt((err ** 2).mean(axis=(0, 1))), precision=2))
print("value rms per coefficient:", np.array2string(np.sqrt((want ** 2).mean(axis=(0, 1))), precision=3))
b, f, k = np.unravel_index(err.argmax(), err.shape)
print("worst at clip", b, "frame", f, "coef", k, "got", got[b, f, k], "want", want[b, f, k])
