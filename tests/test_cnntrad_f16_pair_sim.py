"""CPU model of cnn-trad-fpool3's f16-pair arithmetic (csrc/kws_cnntrad.hip, KWS_CT_F16_PAIR), against a float64 evaluation of
the model's CPU definition (oracle/cnn_trad.py).  No GPU: NumPy float16 / float32 stand in for the matrix core's operand and
accumulate types.  What the kernel does, restated: every GEMM operand v is scaled by a power of two s into f16's range and
written as hi = f16(v s), lo' = f16((v s - hi) 2^11); hi*hi accumulates in one f32 sum, hi*lo' + lo'*hi in a second one that
is scaled by 2^-11 at the end; weights get s per layer from max |w|, activations per clip from bounds known before the
values exist (max |feature|, then sum|w| * bound + max|b| layer by layer).  Claims checked here: no operand overflows f16
whatever the input level or weight gain; the logits' error against float64 is of the size of torch's own f32 forward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cnn_trad as o_ct


def _exp_for(bound: float) -> int:
    """k with bound * 2^k < 2^15 (the kernel's pow2_exp_for: from the float's exponent field, clamped to +-100)."""
    if not bound > 0.0:
        return 100
    e = int(np.frexp(np.float32(bound))[1])          # bound = f * 2^e, f in [0.5, 1)
    return int(np.clip(15 - e, -100, 100))


def _pair(a: np.ndarray, k: int):
    x = (a.astype(np.float32) * np.float32(2.0 ** k)).astype(np.float32)
    hi = x.astype(np.float16)
    assert np.isfinite(hi).all(), "an operand overflowed f16"
    lo = ((x - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _gemm(a_pair, b_pair):
    (ah, al), (bh, bl) = a_pair, b_pair
    main = ah @ bh
    cross = ah @ bl + al @ bh
    return (main + cross * np.float32(2.0 ** -11)).astype(np.float32)


def _forward_pair(state, x):
    w1 = state["conv1.weight"].numpy().reshape(64, -1)
    w2 = state["conv2.weight"].numpy().reshape(64, -1)
    wl = state["lin.weight"].numpy()
    b1, b2, bl = state["conv1.bias"].numpy(), state["conv2.bias"].numpy(), state["lin.bias"].numpy()
    kw1, kw2, kwl = (_exp_for(float(np.abs(w).max())) for w in (w1, w2, wl))
    w1p, w2p, wlp = _pair(w1, kw1), _pair(w2, kw2), _pair(wl, kwl)
    w1_abs, w2_abs = float(np.abs(w1).sum(1).max()), float(np.abs(w2).sum(1).max())
    out = []
    for i in range(x.shape[0]):
        xi = x[i:i + 1]
        m0 = float(xi.abs().max())
        k0 = _exp_for(m0)
        cols = F.unfold(F.pad(xi, (3, 4, 9, 10)), (20, 8)).numpy()[0]
        y1 = _gemm(w1p, _pair(cols, k0)) * np.float32(2.0 ** -(k0 + kw1)) + b1[:, None]
        yp = F.max_pool2d(torch.from_numpy(np.maximum(y1, 0).reshape(1, 64, 99, 10)), (1, 3), (1, 3))
        bound1 = (w1_abs * m0 + float(np.abs(b1).max())) * 1.001
        assert float(yp.max()) <= bound1
        k1 = _exp_for(bound1)
        cols2 = F.unfold(F.pad(yp, (1, 2, 4, 5)), (10, 4)).numpy()[0]
        y2 = np.maximum(_gemm(w2p, _pair(cols2, k1)) * np.float32(2.0 ** -(k1 + kw2)) + b2[:, None], 0)
        bound2 = (w2_abs * bound1 + float(np.abs(b2).max())) * 1.001
        assert float(y2.max()) <= bound2
        k2 = _exp_for(bound2)
        h = _gemm(wlp, _pair(y2.reshape(-1, 1), k2))[:, 0] * np.float32(2.0 ** -(k2 + kwl)) + bl
        d = np.maximum(state["dnn.weight"].numpy() @ h + state["dnn.bias"].numpy(), 0)
        out.append(state["fc.weight"].numpy() @ d + state["fc.bias"].numpy())
    return np.stack(out)


@pytest.mark.parametrize("seed,gain", [(1, 1.0), (2, 8.0), (3, 0.05)])
def test_f16_pair_model_is_f32_grade_and_never_overflows(seed, gain):
    state = o_ct.random_state(seed)
    for k in state:
        if k.startswith("conv") and k.endswith("weight"):
            state[k] = state[k] * gain
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(4, 1, 99, 10, generator=g) * 20.0
    x[1] *= 1e-3
    x[2] *= 50.0
    x[3, :, :50] = 0.0
    ref64 = o_ct.forward({k: v.double() for k, v in state.items()}, x.double()).numpy()
    ref32 = o_ct.forward(state, x).numpy()
    got = _forward_pair(state, x)
    for i in range(x.shape[0]):
        scale = max(1.0, float(np.abs(ref64[i]).max()))
        e_pair = float(np.abs(got[i] - ref64[i]).max())
        e_f32 = float(np.abs(ref32[i] - ref64[i]).max())
        assert e_pair <= max(4.0 * e_f32, 2e-6 * scale), (seed, gain, i, e_pair, e_f32, scale)
