#!/usr/bin/env python3
"""Diagnostics (GPU box): fused wav -> label against the oracle over several seeds / input kinds / weight scales."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
from kws import _native
from oracle import dscnn as o_dscnn, psf_mfcc as o_mfcc
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
worst = 0.0
for seed in range(6):
    rng = np.random.default_rng(100 + seed)
    B = 257 + 64 * seed
    kind = seed % 3
    if kind == 0:
        clips = rng.integers(-32768, 32768, size=(B, 16000), dtype=np.int16)
    elif kind == 1:
        clips = np.clip(rng.normal(0, 3000, size=(B, 16000)), -32768, 32767).astype(np.int16)
    else:  # sparse: silence with bursts and single impulses
        clips = np.zeros((B, 16000), np.int16)
        for b in range(B):
            a = int(rng.integers(0, 15000)); n = int(rng.integers(1, 900))
            clips[b, a:a + n] = rng.integers(-20000, 20000, n)
    state = o_dscnn.random_state(seed=seed, std=0.05 + 0.03 * seed)
    ctx.load_dscnn(o_dscnn.flatten_state(state), 12)
    wav = torch.from_numpy(clips).to(dev)
    logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
    labels = torch.empty((B,), dtype=torch.int32, device=dev)
    ctx.infer_i16(wav, logits, labels); ctx.sync()
    want = o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips)))
    err = float((logits.cpu() - want).abs().max())
    top2 = torch.topk(want, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 2e-4).numpy()
    same = bool(np.array_equal(labels.cpu().numpy()[clear], want.argmax(1).numpy()[clear]))
    worst = max(worst, err)
    print(f"seed {seed} kind {kind} B {B}: max |logit err| {err:.3e}, scale {float(want.abs().max()):.2f}, argmax equal on {int(clear.sum())}/{B} clear clips: {same}", flush=True)
    assert err <= 1e-4 and same
print("worst", worst)
