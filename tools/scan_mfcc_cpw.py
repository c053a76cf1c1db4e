#!/usr/bin/env python3
"""Diagnostics (GPU box): the float32 MFCC kernel's time at one batch size for the chunks-per-wavefront choice of the launcher
(default) or the one forced with KWS_X_MFCC_CPW (read once per process: one process per setting).

    [KWS_X_MFCC_CPW=n] python tools/scan_mfcc_cpw.py <clips>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native
B = int(sys.argv[1])
dev = torch.device("cuda", 0)
ctx = _native.Context(0); ctx.use_torch_stream()
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
for _ in range(200): ctx.mfcc_i16(wav, out)
torch.cuda.synchronize()
ctx.prof_enable(1); ctx.prof_reset()
for _ in range(100): ctx.mfcc_i16(wav, out)
ctx.sync()
ms, n = ctx.prof_read(_native.KWS_K_MFCC)
print(f"B={B:5d} cpw={os.environ.get('KWS_X_MFCC_CPW', 'auto'):>4s}  mfcc kernel {ms / n * 1e3:8.2f} us")
