"""CPU model of the DS-CNN's f16-pair arithmetic (csrc/kws_dscnn.hip, KWS_PW_PAIR_F16: MODE 5) against a float64 evaluation of
the reference architecture (oracle/dscnn.py).  No GPU: NumPy float16 / float32 stand in for the matrix core's operand and
accumulate types.  What the kernel does, restated step by step:

* every GEMM operand v, scaled by a power of two, is hi = f16(v) and lo = f16(v - hi) (the residual is NOT rescaled: it may
  go subnormal, which the matrix core honours); hi*hi + hi*lo + lo*hi accumulate in f32;
* the activations are kept in per-clip power-of-two units: stage n's accumulators, stored plane, ring value and pointwise
  bias are in units 2^sg[n]; the next depthwise runs on the scaled plane with its bias in those units;
* the exponents are decided two layers ahead: ky[n] (block n's depthwise output, true units, times 2^ky[n] < 2^15) comes from
  the MEASURED maximum of stage n - 2's stored output and the weight-derived bounds |dw out| <= dw_abs |in| + dw_bmax,
  |pw out| <= pw_abs |dw out| + pw_bmax; conv1's from the clip's largest |feature|.

Claims checked: no operand overflows f16 whatever the input level, weight gain or bias size; the bound chain really bounds
(every depthwise output is below its 2^15 / scale); the logits' error against float64 is of the size of torch's own f32
forward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dscnn as o_dscnn


def _exp_for(bound: float) -> int:
    """k with bound * 2^k < 2^15 (the kernel's pow2_exp_for: from the float's exponent field, clamped to +-100)."""
    b = np.float32(bound)
    if not b > 0:
        return 100
    return int(np.clip(15 - int(np.frexp(b)[1]), -100, 100))


def _cap_units(ky: int, k_w: int, bz: float):
    """(ky, sg): the stage's units 2^sg = 2^(ky + k_w), lowered (through ky) until the stage's stored output -- bias included,
    bounded by bz in true units -- stays below 2^100 in them and sg <= 120 (the kernel's cap_units)."""
    sg = ky + k_w
    b = np.float32(bz)
    eb = int(np.frexp(b)[1]) if b > 0 else -126
    limit = min(100 - eb, 120)
    if sg > limit:
        ky -= sg - limit
        sg = limit
    return ky, sg


def _pair(a: np.ndarray):
    """hi, lo of already scaled values; asserts the range claim."""
    x = a.astype(np.float32)
    with np.errstate(over="ignore"):
        hi = x.astype(np.float16)
    assert np.isfinite(hi).all(), f"an operand overflowed f16 (max |x| = {np.abs(x).max():.3g})"
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _gemm(a, b):
    (ah, al), (bh, bl) = a, b
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def _row_abs(w: np.ndarray) -> float:
    return float(np.abs(w.astype(np.float64)).sum(axis=1).max() * 1.0000002)


def _forward_pair(state, x):
    st = {k: v.numpy() for k, v in state.items()}
    w1 = st["conv1.weight"].reshape(64, 100)
    k_c1 = _exp_for(float(np.abs(w1).max()))
    w1p = _pair(w1 * np.float32(2.0 ** k_c1))
    c1_abs, c1_bmax = _row_abs(w1), float(np.abs(st["conv1.bias"]).max())
    blk = []
    for i in range(1, 5):
        dw, pw = st[f"dsconv{i}.depthwise.weight"].reshape(64, 9), st[f"dsconv{i}.pointwise.weight"].reshape(64, 64)
        k_pw = _exp_for(float(np.abs(pw).max()))
        blk.append(dict(dw=st[f"dsconv{i}.depthwise.weight"], dwb=st[f"dsconv{i}.depthwise.bias"], pwp=_pair(pw * np.float32(2.0 ** k_pw)),
                        pwb=st[f"dsconv{i}.pointwise.bias"], k_pw=k_pw, dw_abs=_row_abs(dw), dw_bmax=float(np.abs(st[f"dsconv{i}.depthwise.bias"]).max()),
                        pw_abs=_row_abs(pw), pw_bmax=float(np.abs(st[f"dsconv{i}.pointwise.bias"]).max())))
    out = []
    for n in range(x.shape[0]):
        xi = x[n:n + 1]
        mx = float(xi.abs().max())
        kx = _exp_for(mx)
        sg = [0, 0, 0, 0, 0]
        ky = [0, 0, 0, 0, 0]
        kx, sg[0] = _cap_units(kx, k_c1, (c1_abs * mx + c1_bmax) * 1.001)
        # conv1: im2col of the features times 2^kx, accumulators and stored plane in units 2^sg[0]
        cols = F.unfold(F.pad(xi, (2, 2, 2, 2)), (10, 10), stride=2).numpy()[0]                       # [100, 141]
        acc = _gemm(w1p, _pair(cols * np.float32(2.0 ** kx)))
        z = np.maximum(acc + st["conv1.bias"][:, None] * np.float32(2.0 ** sg[0]), 0).reshape(1, 64, 47, 3)
        # scales of block 1: from the a-priori bound on conv1's output (two layers ahead of the features)
        bz = (c1_abs * mx + c1_bmax) * 1.001
        by = (blk[0]["dw_abs"] * bz + blk[0]["dw_bmax"]) * 1.001
        ky[1], sg[1] = _cap_units(_exp_for(by), blk[0]["k_pw"], (blk[0]["pw_abs"] * by + blk[0]["pw_bmax"]) * 1.001)
        measured = float(z.max()) * 2.0 ** -sg[0]                                                      # conv1's largest output, true units
        for i, b in enumerate(blk, start=1):
            if i < 4:  # decided at the start of block i for block i + 1: measured max of stage i - 1 (and its ring) -> two bounds
                m_in = measured if i == 1 else max(measured, blk[i - 2]["pw_bmax"])
                bz = (b["pw_abs"] * ((b["dw_abs"] * m_in + b["dw_bmax"]) * 1.001) + b["pw_bmax"]) * 1.001
                by = (blk[i]["dw_abs"] * bz + blk[i]["dw_bmax"]) * 1.001
                ky[i + 1], sg[i + 1] = _cap_units(_exp_for(by), blk[i]["k_pw"], (blk[i]["pw_abs"] * by + blk[i]["pw_bmax"]) * 1.001)
            # depthwise on the scaled plane (its ring = relu(previous pointwise bias) in the same units), bias in the plane's units
            zin = torch.from_numpy(z)
            if i > 1:
                ring = np.maximum(blk[i - 2]["pwb"], 0) * np.float32(2.0 ** sg[i - 1])
                zp = torch.from_numpy(np.broadcast_to(ring[None, :, None, None], (1, 64, z.shape[2] + 2, z.shape[3] + 2)).copy())
                zp[:, :, 1:-1, 1:-1] = zin
                zin = zp
            y = F.conv2d(zin, torch.from_numpy(b["dw"]), torch.from_numpy(b["dwb"] * np.float32(2.0 ** sg[i - 1])), padding=1, groups=64).numpy()
            e = np.float32(2.0 ** (ky[i] - sg[i - 1]))
            assert float(np.abs(y * e).max()) < 2.0 ** 15 * 1.002, "the bound chain failed to bound a depthwise output"
            h, w_ = y.shape[2], y.shape[3]
            acc = _gemm(b["pwp"], _pair(y.reshape(64, h * w_) * e))
            z = np.maximum(acc + b["pwb"][:, None] * np.float32(2.0 ** sg[i]), 0).reshape(1, 64, h, w_)
            assert np.isfinite(z).all(), "a stored plane left the float range in its units"
            measured = float(z.max()) * 2.0 ** -sg[i]
        # pool over the interior and the ring of block 4's output, true units
        interior = z.reshape(64, -1).sum(axis=1) * np.float32(2.0 ** -sg[4])
        ring_n = (z.shape[2] + 2) * (z.shape[3] + 2) - z.shape[2] * z.shape[3]
        pooled = (interior + ring_n * np.maximum(blk[3]["pwb"], 0)) / ((z.shape[2] + 2) * (z.shape[3] + 2))
        out.append(st["fc.weight"] @ pooled.astype(np.float32) + st["fc.bias"])
    return np.stack(out)


@pytest.mark.parametrize("seed,w_gain,b_gain", [(1, 1.0, 1.0), (2, 5.0, 1.0), (3, 0.2, 1.0), (4, 1.0, 100.0), (5, 1.0, 0.0),
                                                  (6, 1e-9, 0.0), (7, 300.0, 1e7),  # vanishing / bias-dominated stages
                                                  (8, -40.0, 1.0)])                 # cancelling pointwise rows: bounds loose by 10^3 per layer
def test_f16_pair_model_is_f32_grade_and_never_overflows(seed, w_gain, b_gain):
    state = o_dscnn.random_state(seed, std=0.1)
    cancel = w_gain < 0
    w_gain = abs(w_gain)
    for k in list(state):
        if k.endswith("weight") and not k.startswith("fc"):
            fan_in = int(np.prod(state[k].shape[1:]))
            state[k] = state[k] * float((2.0 / fan_in) ** 0.5 / 0.1) * (w_gain if ("pointwise" in k or k.startswith("conv1")) else 1.0)
        if k.endswith("bias") and not k.startswith("fc"):
            state[k] = state[k] * b_gain
    if cancel:  # every pointwise row alternates +g, -g (+ a small random part): sum|w| is ~10^3 x |sum w|
        for i in range(1, 5):
            w = state[f"dsconv{i}.pointwise.weight"]
            alt = torch.tensor([1.0, -1.0]).repeat(32).reshape(1, 64, 1, 1) * w_gain * 0.125
            state[f"dsconv{i}.pointwise.weight"] = alt.expand(64, 64, 1, 1).clone() + w / w_gain
    g = torch.Generator().manual_seed(200 + seed)
    x = torch.randn(6, 1, 99, 10, generator=g) * 6.0
    x[0] = 0.0
    x[1] *= 1e-3
    x[2] *= 300.0
    x[3, :, 50:] = 0.0
    x[4] = torch.randn(1, 99, 10, generator=g) * 0.01
    x[4, 0, 40, 3] = 3000.0
    if cancel:
        x[5] = 4.0  # a constant map: the alternating rows cancel almost exactly
    ref64 = o_dscnn.forward({k: v.double() for k, v in state.items()}, x.double()).numpy()
    ref32 = o_dscnn.forward(state, x).numpy()
    got = _forward_pair(state, x)
    assert np.isfinite(got).all()
    for i in range(x.shape[0]):
        scale = max(1.0, float(np.abs(ref64[i]).max()))
        e_pair = float(np.abs(got[i] - ref64[i]).max())
        e_f32 = float(np.abs(ref32[i] - ref64[i]).max())
        assert e_pair <= max(4.0 * e_f32, 2e-6 * scale), (seed, w_gain, b_gain, i, e_pair, e_f32, scale)
