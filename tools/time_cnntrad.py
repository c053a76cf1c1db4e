#!/usr/bin/env python3
"""Diagnostics (GPU box): time kws_forward_cnn_trad_f32 (build-defined cnn-trad-fpool3) for KWS_HIP_LIB."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = _native.Context(0); ctx.use_torch_stream()
ctx.load_cnn_trad(bench.synth_cnn_trad_weights(4), 12)
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
ctx.mfcc_i16(wav, feat)
logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
labels = torch.empty((B,), dtype=torch.int32, device=dev)
for _ in range(3): ctx.forward_cnn_trad_f32(feat, logits, labels)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5): ctx.forward_cnn_trad_f32(feat, logits, labels)
    t1.record(); torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1) / 5)
flop = 2 * (99 * 10 * 64 * 160 + 297 * 64 * 2560 + 19008 * 32 + 32 * 128 + 128 * 12)
print(f"{os.environ.get('KWS_HIP_LIB', 'default'):40s} cnn-trad-fpool3 B={B}: {best:.3f} ms  = {B / best * 1e3 / 1e6:.2f} M clips/s, "
      f"{flop * B / best / 1e9:.1f} TFLOP/s algorithmic  checksum {float(logits.double().sum()):.5f}")
