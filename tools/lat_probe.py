#!/usr/bin/env python3
"""Diagnostics (GPU box): host-visible latency of single calls at B = 64 (launch + kernel + sync), the streaming hop,
and blocking sync vs spinning on an event query (no difference: the overhead is on the launch side)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'keyword-spotting_amd'))
import bench
from kws import _native
dev = torch.device('cuda',0)
ctx = _native.Context(0)
ctx.load_dscnn(bench.synth_weights(), 12)
S=64
feat = torch.randn((S,1,99,10), device=dev)
logits = torch.empty((S,12), dtype=torch.float32, device=dev); labels = torch.empty((S,), dtype=torch.int32, device=dev)
def p50(f, n=300):
    lat=[]
    for i in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); f(); ctx.sync(); lat.append((time.perf_counter()-t0)*1e6)
    return float(np.percentile(lat[20:],50))
print('forward_f32 B=64 + sync  p50 us', p50(lambda: ctx.forward_f32(feat, logits, labels)))
print('sync only p50 us', p50(lambda: None))
wav = torch.zeros((S,16000), dtype=torch.int16, device=dev)
print('mfcc_i16 B=64 + sync p50 us', p50(lambda: ctx.mfcc_i16(wav, feat)))
sm = torch.empty((S,12), dtype=torch.float32, device=dev)
print('softmax + sync p50 us', p50(lambda: ctx.softmax_f32(logits, sm)))
ctx.stream_open(S)
hop = torch.zeros((S,160), dtype=torch.int16, device=dev)
print('stream push + sync p50 us', p50(lambda: ctx.stream_push_i16(hop, logits, labels, use_graph=False)))
# kernel-only durations via events
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(50): ctx.forward_f32(feat, logits, labels)
ms,n = ctx.prof_read(_native.KWS_K_DSCNN); print('dscnn kernel B=64 event us', ms/n*1e3)
# blocking sync vs spinning on an event query (same stream as torch's current stream)
ctx.use_torch_stream()
def p50_spin(f, n=300):
    lat=[]
    for i in range(n):
        torch.cuda.synchronize(); ev = torch.cuda.Event()
        t0=time.perf_counter(); f(); ev.record()
        while not ev.query(): pass
        lat.append((time.perf_counter()-t0)*1e6)
    return float(np.percentile(lat[20:],50))
print('stream push, spin on event query p50 us', p50_spin(lambda: ctx.stream_push_i16(hop, logits, labels, use_graph=False)))
print('stream push + blocking sync (torch stream) p50 us', p50(lambda: ctx.stream_push_i16(hop, logits, labels, use_graph=False)))
print('forward B=64, spin p50 us', p50_spin(lambda: ctx.forward_f32(feat, logits, labels)))
