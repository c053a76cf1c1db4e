#!/usr/bin/env python3
"""Diagnostics (GPU box): time kws_forward_f32 (DS-CNN alone) for the library selected by KWS_HIP_LIB.

    python tools/time_dscnn.py [clips = 4096] [pointwise math: 4 = bf16 triple, 5 = f16 pair, 1 = f32 MFMA; default = the context's]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = _native.Context(0); ctx.use_torch_stream()
ctx.load_dscnn(bench.bench_weights()[0], 12)
if len(sys.argv) > 2:
    ctx.set_pointwise_math(int(sys.argv[2]))
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
ctx.mfcc_i16(wav, feat)
logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
labels = torch.empty((B,), dtype=torch.int32, device=dev)
for _ in range(100): ctx.forward_f32(feat, logits, labels)  # clocks settle
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(40): ctx.forward_f32(feat, logits, labels)
    t1.record(); torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1) / 40)
print(f"{os.environ.get('KWS_HIP_LIB', 'default'):40s} math {sys.argv[2] if len(sys.argv) > 2 else 'default'}  dscnn best-of-5 {best:.4f} ms  checksum {float(logits.double().sum()):.6f}")
