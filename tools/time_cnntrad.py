#!/usr/bin/env python3
"""Diagnostics (GPU box): cnn-trad-fpool3's two kernels under both arithmetics, same context, same call.

    python tools/time_cnntrad.py [clips = 4096] [steps = 40]

Per arithmetic (KWS_CT_F16_PAIR: three f16 MFMAs per f32 k-block; KWS_CT_BF16_TRIPLE: six bf16 ones): ms per step of
kws_forward_cnn_trad_f32 on resident features, the two kernels' own durations (HIP events), and the largest logit
difference between the two.  Prints one JSON line."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
ctx.load_cnn_trad(bench.synth_cnn_trad_weights(), 12)
feat = (torch.randn((B, 1, 99, 10), generator=torch.Generator().manual_seed(3)) * 6.0).to(dev)
logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
labels = torch.empty((B,), dtype=torch.int32, device=dev)
out = {"clips": B, "steps": steps}
keep = {}
for rep in range(2):  # two rounds: the second one's numbers are reported (clocks settled)
    for tag, math in (("bf16_triple", _native.KWS_CT_BF16_TRIPLE), ("f16_pair", _native.KWS_CT_F16_PAIR)):
        ctx.set_cnn_trad_math(math)
        for _ in range(10):
            ctx.forward_cnn_trad_f32(feat, logits, labels)
        ctx.sync()
        ctx.prof_enable(1); ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.forward_cnn_trad_f32(feat, logits, labels)
        ctx.sync()
        dt = time.perf_counter() - t0
        c_ms, c_n = ctx.prof_read(_native.KWS_K_CNNTRAD_CONV)
        d_ms, d_n = ctx.prof_read(_native.KWS_K_CNNTRAD_DENSE)
        ctx.prof_enable(0)
        out[tag] = {"ms_per_step": dt / steps * 1e3, "conv_ms": c_ms / max(c_n, 1), "dense_ms": d_ms / max(d_n, 1)}
        keep[tag] = logits.cpu().numpy().copy()
sc = max(1.0, float(np.abs(keep["bf16_triple"]).max()))
out["max_abs_logit_diff_over_scale"] = float(np.abs(keep["f16_pair"] - keep["bf16_triple"]).max() / sc)
out["logit_scale"] = sc
out["labels_equal"] = bool(np.array_equal(keep["f16_pair"].argmax(1), keep["bf16_triple"].argmax(1)))
ctx.close()
print(json.dumps(out))
