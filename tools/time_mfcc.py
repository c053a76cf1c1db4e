#!/usr/bin/env python3
"""Diagnostics (GPU box): time kws_mfcc_i16 alone for the library selected by KWS_HIP_LIB -- the call as a whole with the
selective float64 refinement off and on (alternating rounds, so clock drift hits both alike), the per-kernel times of the
float32 kernel and of the refinement launch (HIP events on the context's stream), and how many frames were refined.
  python tools/time_mfcc.py [uniform|golden|speech]      workload: bench noise (default), the 48 golden clips tiled to
                                                          4096, or 32 speech-like clips tiled to 4096"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench
from kws import _native
B = 4096
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
if kind == "golden":
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_golden.npz"))["clips"]
    clips = np.ascontiguousarray(np.tile(g, (B // len(g) + 1, 1))[:B])
elif kind == "speech":
    import speechlike
    g, _ = speechlike.speechlike_set(32, 400)
    clips = np.ascontiguousarray(np.tile(g, (B // len(g), 1))[:B])
else:
    clips = bench.synth_clips(B, 0)
dev = torch.device("cuda", 0)
ctx = _native.Context(0); ctx.use_torch_stream()
wav = torch.from_numpy(clips).to(dev)
out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
for _ in range(150): ctx.mfcc_i16(wav, out)  # clocks settle after ~30 launches (bench.py --spinup)
torch.cuda.synchronize()
res = {0.0: [], 1e9: [], _native.FE_REFINE_SPAN_DEFAULT: []}  # off / flags computed but nothing listed (an empty refinement launch) / on
for rnd in range(3):
    for span in res:
        ctx.set_frontend_refine(span)
        for _ in range(10): ctx.mfcc_i16(wav, out)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(100): ctx.mfcc_i16(wav, out)
        t1.record(); torch.cuda.synchronize()
        res[span].append(t0.elapsed_time(t1) / 100)
kern = {}
for span in (0.0, 1e9, _native.FE_REFINE_SPAN_DEFAULT):  # kernel-only times (HIP events on the stream, every 4th launch)
    ctx.set_frontend_refine(span)
    ctx.prof_reset(); ctx.prof_enable(4)
    for _ in range(200): ctx.mfcc_i16(wav, out)
    kern[span] = (ctx.prof_read(_native.KWS_K_MFCC), ctx.prof_read(_native.KWS_K_MFCC_REFINE))
    ctx.prof_enable(0)
(ms_f32, n_f32), (ms_ref, n_ref) = kern[_native.FE_REFINE_SPAN_DEFAULT]
total, refined, last = ctx.frontend_stats()
name = os.environ.get('KWS_HIP_LIB', 'default')
print(f"{os.path.basename(name):24s} {kind}: mfcc call {min(res[0.0]):.4f} ms without refinement, {min(res[1e9]):.4f} with an empty list, {min(res[_native.FE_REFINE_SPAN_DEFAULT]):.4f} ms with; "
      f"float32 kernel {kern[0.0][0][0] / max(kern[0.0][0][1], 1):.4f} ms without the flag, {kern[1e9][0][0] / max(kern[1e9][0][1], 1):.4f} with, empty refinement launch {kern[1e9][1][0] / max(kern[1e9][1][1], 1):.4f} ms; "
      f"float32 kernel {ms_f32 / max(n_f32, 1):.4f} ms, refinement launch {ms_ref / max(n_ref, 1):.4f} ms, "
      f"{last} of {B * 99} frames refined per call ({100.0 * last / (B * 99):.3f} %)  checksum {float(out.double().sum()):.6f}")
