#!/usr/bin/env python3
"""Diagnostics (GPU box): what the HIP events bench.py records around every kernel launch cost per step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
ctx.load_dscnn(bench.bench_weights()[0], 12); ctx.reserve(B)
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
lg = torch.empty((B, 12), dtype=torch.float32, device=dev); lb = torch.empty((B,), dtype=torch.int32, device=dev)
for _ in range(100): ctx.infer_i16(wav, lg, lb)
ctx.sync()
for rep in range(3):
    for on in (False, True):
        ctx.prof_enable(on); ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(200): ctx.infer_i16(wav, lg, lb)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 200 * 1e3
        extra = ""
        if on:
            k, kn = ctx.prof_read(_native.KWS_K_DSCNN); m, mn = ctx.prof_read(_native.KWS_K_MFCC)
            extra = f"  kernels {k / kn:.4f} + {m / mn:.4f} = {k / kn + m / mn:.4f} ms"
        print(f"B={B} events {'on ' if on else 'off'}: {dt:.4f} ms per step{extra}")
ctx.prof_enable(False); ctx.close()
