#!/usr/bin/env python3
"""Diagnostics (GPU box): PCIe-inclusive rate of the host-ingest path.

Host int16 batches -> pinned staging -> H2D on a copy stream, overlapped with MFCC + DS-CNN of the previous
batch (KeywordSpotter.infer_batches).  Prints one JSON line; this rate is never bench.py's `value`.
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws.inference import KeywordSpotter
from kws.libs.models import DepthwiseSeparableConv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = DepthwiseSeparableConv(12)
sp = KeywordSpotter(model)
clips = [bench.synth_clips(B, seed) for seed in range(4)]
list(sp.infer_batches(clips[:2], max_batch=B))  # warm-up (context, weights, pinned buffers)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for labels, logits in sp.infer_batches((clips[i % 4] for i in range(N)), max_batch=B):
    n += labels.shape[0]
dt = time.perf_counter() - t0
# the same with batches the caller already holds in pinned memory (no host-side pack)
pinned = [torch.from_numpy(c).pin_memory() for c in clips]
t2 = time.perf_counter()
n2 = 0
for labels, logits in sp.infer_batches((pinned[i % 4] for i in range(N)), max_batch=B):
    n2 += labels.shape[0]
dt_pin = time.perf_counter() - t2
# device-resident reference for the same batches
wav = torch.from_numpy(clips[0]).cuda()
for _ in range(3): model.infer_pcm16(wav)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(N): model.infer_pcm16(wav)
torch.cuda.synchronize(); dt_dev = time.perf_counter() - t1
print(json.dumps({"batch": B, "batches": N, "host_ingest_clips_per_s": n / dt, "host_ingest_GBps": n * 32000 / dt / 1e9,
                  "pinned_input_clips_per_s": n2 / dt_pin, "pinned_input_GBps": n2 * 32000 / dt_pin / 1e9,
                  "device_resident_clips_per_s": B * N / dt_dev,
                  "note": "host numpy int16 -> pinned pack (memcpy) -> H2D on a copy stream overlapped with compute; includes the D2H of logits+labels"}))
