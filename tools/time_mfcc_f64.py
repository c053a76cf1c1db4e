#!/usr/bin/env python3
"""Diagnostics (GPU box): time kws_mfcc_i16 with the float64 front end for the library selected by KWS_HIP_LIB."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native
B = 4096
dev = torch.device("cuda", 0)
ctx = _native.Context(0); ctx.use_torch_stream(); ctx.set_frontend_math(_native.FE_F64)
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
for _ in range(20): ctx.mfcc_i16(wav, out)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20): ctx.mfcc_i16(wav, out)
t1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('KWS_HIP_LIB', 'default'):60s} mfcc f64 {t0.elapsed_time(t1) / 20:.4f} ms  checksum {float(out.double().sum()):.6f}")
