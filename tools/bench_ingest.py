#!/usr/bin/env python3
"""Diagnostics (GPU box): PCIe-inclusive rate of the host-ingest path, against the raw link.

    python tools/bench_ingest.py [clips per call = 16384] [calls = 12]

Host int16 batches -> kws_infer_host_i16 (pack threads -> pinned staging -> H2D on a copy stream || MFCC + DS-CNN ||
D2H of logits + labels).  Also measured in the same process: the raw pinned H2D rate of this box's link (1 and 2 copy
streams) and the single-thread / pooled host memcpy rate, so the pipeline's rate can be read against its two ceilings.
Prints one JSON line; this rate is never bench.py's `value`.
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
blob, _ = bench.bench_weights()
out = {"clips_per_call": B, "calls": N, "PCIe_Gen5_x16_spec_GBps": 63.0}

# ---- raw link: pinned -> device, 128 MiB transfers
nbytes = 128 << 20
h = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
d = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for k in (1, 2):
    for rep in range(2):  # first repetition warms up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8):
            with torch.cuda.stream(streams[i % k]):
                d[i % 2].copy_(h[i % 2], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out[f"raw_h2d_GBps_{k}_stream"] = 8 * nbytes / dt / 1e9
# ---- host memcpy (pageable -> pinned), one thread
src = np.random.default_rng(0).integers(0, 255, nbytes, dtype=np.uint8)
dst = h[0].numpy()
for rep in range(2):
    t0 = time.perf_counter(); dst[...] = src; dt = time.perf_counter() - t0
out["host_memcpy_GBps_1_thread"] = nbytes / dt / 1e9
del h, d, src, dst

clips = bench.synth_clips(B, 0)
def run(ctx, src, calls):
    logits = np.empty((B, 12), np.float32); labels = np.empty((B,), np.int32)
    ctx.infer_host_i16(src, logits, labels)  # warm-up: staging rings, threads, clocks
    t0 = time.perf_counter()
    for _ in range(calls):
        ctx.infer_host_i16(src, logits, labels)
    dt = time.perf_counter() - t0
    return B * calls / dt, labels.copy()

res = {}
for name, (chunk, slots, threads) in {"default (1024 x 3 slots, 16 threads)": (0, 0, 0), "8 pack threads": (0, 0, 8), "1 pack thread": (0, 0, 1), "4 pack threads": (0, 0, 4),
                                       "16 pack threads": (0, 0, 16), "chunk 512": (512, 0, 0), "chunk 2048": (2048, 0, 0),
                                       "chunk 4096 x 2 slots": (4096, 2, 0), "6 slots": (0, 6, 0)}.items():
    ctx = _native.Context(0)
    ctx.load_dscnn(blob, 12)
    ctx.ingest_config(chunk, slots, threads)
    rate, lab = run(ctx, clips, N if name.startswith("default") else max(3, N // 3))
    res[name] = {"clips_per_s": rate, "GBps": rate * 32000 / 1e9}
    ctx.close()
out["pageable_input"] = res
pinned = torch.from_numpy(clips).pin_memory()
ctx = _native.Context(0)
ctx.load_dscnn(blob, 12)
rate, lab_p = run(ctx, pinned, N)
out["pinned_input"] = {"clips_per_s": rate, "GBps": rate * 32000 / 1e9}
# device-resident reference for the same batch, and a check that the three routes agree bit for bit
wav = torch.from_numpy(clips).to(dev)
lg = torch.empty((B, 12), dtype=torch.float32, device=dev); lb = torch.empty((B,), dtype=torch.int32, device=dev)
for _ in range(3): ctx.infer_i16(wav, lg, lb)
ctx.sync(); t1 = time.perf_counter()
for _ in range(N): ctx.infer_i16(wav, lg, lb)
ctx.sync(); dt_dev = time.perf_counter() - t1
out["device_resident_clips_per_s"] = B * N / dt_dev
out["labels_identical_across_routes"] = bool(np.array_equal(lab, lb.cpu().numpy()) and np.array_equal(lab_p, lab))
# ---- many SMALL batches (the reference's own batch size is 1028, train.py:110): one synchronous kws_infer_host_i16 call per
# batch (the pipeline fills and drains inside every call) against submit / wait with the next batch submitted before the
# previous one is waited for (what KeywordSpotter.infer_batches does)
small = {}
for bs in (256, 1028, 4096):
    nb = max(8, min(64, 65536 // bs))
    batches = [clips[(i * bs) % (B - bs):(i * bs) % (B - bs) + bs] for i in range(nb)]
    c2 = _native.Context(0)
    c2.load_dscnn(blob, 12)
    c2.infer_host_i16(batches[0])
    t0 = time.perf_counter()
    for b_ in batches:
        c2.infer_host_i16(b_)
    dt_sync = time.perf_counter() - t0
    t0 = time.perf_counter()
    pend = None
    for b_ in batches:
        nxt = c2.infer_host_submit_i16(b_)
        if pend is not None:
            c2.infer_host_wait(pend[2])
        pend = nxt
    c2.infer_host_wait(0)
    dt_pipe = time.perf_counter() - t0
    small[f"B={bs}"] = {"batches": nb, "one_call_per_batch_clips_per_s": bs * nb / dt_sync, "submit_wait_pipelined_clips_per_s": bs * nb / dt_pipe}
    c2.close()
out["small_batches"] = small
best = max(v["clips_per_s"] for v in res.values())
out["best_pageable_clips_per_s"] = best
out["best_pageable_frac_of_raw_link"] = best * 32000 / 1e9 / max(out["raw_h2d_GBps_1_stream"], out["raw_h2d_GBps_2_stream"])
out["note"] = "host numpy int16 (pageable) -> pack threads -> pinned staging -> H2D || MFCC + DS-CNN || D2H (kws_infer_host_i16); results in host memory on return"
print(json.dumps(out))
