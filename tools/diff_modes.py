#!/usr/bin/env python3
"""Diagnostics (GPU box): compare the per-layer activations of two DS-CNN kernel variants.

    python tools/diff_modes.py [mode_a] [mode_b]
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


def main():
    ma = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.use_torch_stream()
    ctx.load_dscnn(bench.synth_weights(), 12)
    B = 4
    wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
    feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
    ctx.mfcc_i16(wav, feat)
    outs = []
    for mode in (ma, mb):
        logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
        labels = torch.empty((B,), dtype=torch.int32, device=dev)
        act = torch.zeros((B, _native.ACT_FLOATS_PER_CLIP), dtype=torch.float32, device=dev)
        ctx.forward_debug_f32(feat, logits, labels, act, use_mfma=mode)
        torch.cuda.synchronize()
        outs.append((act.cpu().numpy(), logits.cpu().numpy()))
    sizes = [("conv1", 141, 3), ("dsconv1", 141, 3), ("dsconv2", 245, 5), ("dsconv3", 357, 7)]
    off = 0
    for name, p, wdt in sizes:
        a = outs[0][0][:, off:off + 64 * p].reshape(B, 64, p)
        b = outs[1][0][:, off:off + 64 * p].reshape(B, 64, p)
        off += 64 * p
        d = np.abs(a - b)
        print(f"{name}: max abs diff {d.max():.3e}  (scale {np.abs(a).max():.3f})")
        if d.max() > 1e-3 * max(1.0, np.abs(a).max()):
            bad = d[0] > 1e-3
            print("  clip 0: bad channels:", np.where(bad.any(axis=1))[0][:64])
            print("  clip 0: bad positions:", np.where(bad.any(axis=0))[0][:200])
            c = np.where(bad.any(axis=1))[0][0]
            print(f"  channel {c} a:", np.round(a[0, c, :12], 4))
            print(f"  channel {c} b:", np.round(b[0, c, :12], 4))
            break
    print("logits max abs diff", np.abs(outs[0][1] - outs[1][1]).max())
    ctx.close()


if __name__ == "__main__":
    main()
