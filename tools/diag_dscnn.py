#!/usr/bin/env python3
"""Diagnostics (GPU box): per-phase shader-clock breakdown of kws_dscnn_fwd_kernel from in-kernel stamps.

    python tools/diag_dscnn.py [batch]

Prints the median cycles each phase takes on one CU (thread 0 of the clip's workgroup), the in-kernel
clock (s_memtime / s_memrealtime), and what fraction of a clip's lifetime each phase is.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench  # noqa: E402
from kws import _native  # noqa: E402

NAMES = ["stage features", "conv1 units", "conv1 barrier", "block1 units", "block1 barrier", "block2 units",
         "block2 barrier", "block3 units", "block3 barrier", "block4 units", "block4 barrier", "pool+fc"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.use_torch_stream()
    ctx.load_dscnn(bench.synth_weights(), 12)
    wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
    feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(wav, feat)
    logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
    stamps = torch.zeros((B, 16), dtype=torch.int64, device=dev)
    label = {0: "VALU cross-check", 1: "product", 2: "ablation: matrix core only", 3: "ablation: stencil only", 4: "split-bf16 MFMA", 5: "f16-pair MFMA", 6: "ablation of 4: split + matrix core, no stencil"}
    for mode in modes:
        for _ in range(3):
            ctx.forward_stamps_f32(feat, logits, stamps, mode)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            ctx.forward_stamps_f32(feat, logits, stamps, mode)
        t1.record(); torch.cuda.synchronize()
        st = stamps.cpu().numpy().astype(np.int64)
        d = np.diff(st[:, :13], axis=1)
        total = st[:, 12] - st[:, 0]
        rt = (st[:, 15] - st[:, 14]).astype(np.float64)  # 100 MHz ticks
        clock_ghz = np.median(total / np.maximum(rt, 1)) * 0.1
        print(f"mode {mode} ({label[mode]})  B={B}  kernel {t0.elapsed_time(t1) / 5:.3f} ms  median clip lifetime "
              f"{np.median(total):.0f} cycles  clock ~{clock_ghz:.2f} GHz = {np.median(total) / clock_ghz / 1e3:.1f} us per clip per CU")
        for i, n in enumerate(NAMES):
            print(f"  {n:16s} {np.median(d[:, i]):9.0f} cyc  {100 * np.median(d[:, i]) / np.median(total):5.1f} %")
    ctx.close()


if __name__ == "__main__":
    main()
