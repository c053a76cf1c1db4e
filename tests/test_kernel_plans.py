"""CPU tests of the host logic and of the algorithms the HIP kernels implement.

The kernels cannot run here (no GPU), so their *plans* are replayed in NumPy with the same index
arithmetic as the device code and checked against the oracle:
  * the 8 x 8 x 8 wavefront FFT (register radix-8 passes + two LDS exchanges) and the split of two
    packed real frames                                    (csrc/kws_mfcc.hip: fft512, split_power)
  * the sparse mel decomposition, through the C ABI's host helpers
  * the ring-slot / zero-slot depthwise addressing        (csrc/kws_dscnn.hip: block_phase)
  * the im2col addressing of conv1 on the zero-padded MFCC map (conv1_phase)
"""
import numpy as np
import pytest
import torch

from oracle import dscnn as o_dscnn
from oracle import psf_mfcc as o

XROW = 72
R = np.float64(0.70710678118654752440)


def dft8(v):
    """Same butterfly network as the device dft8 (axis 0 = register index)."""
    b0, b4 = v[0] + v[4], v[0] - v[4]
    b1, b5 = v[1] + v[5], v[1] - v[5]
    b2, b6 = v[2] + v[6], v[2] - v[6]
    b3, b7 = v[3] + v[7], v[3] - v[7]
    mi = lambda a: a.imag - 1j * a.real  # * -i
    b5 = (b5.real + b5.imag) * R + 1j * (b5.imag - b5.real) * R
    b6 = mi(b6)
    b7 = (b7.imag - b7.real) * R - 1j * (b7.real + b7.imag) * R
    d0, d1, d2, d3 = b0 + b2, b0 - b2, b1 + b3, mi(b1 - b3)
    e0, e1, e2, e3 = b4 + b6, b4 - b6, b5 + b7, mi(b5 - b7)
    out = np.empty_like(v)
    out[0], out[4], out[2], out[6] = d0 + d2, d0 - d2, d1 + d3, d1 - d3
    out[1], out[5], out[3], out[7] = e0 + e2, e0 - e2, e1 + e3, e1 - e3
    return out


def fft512_plan(z):
    tw = np.exp(-2j * np.pi * np.arange(512) / 512)
    lane = np.arange(64)
    k1, q = lane >> 3, lane & 7
    v = np.stack([z[64 * n1 + lane] for n1 in range(8)])
    v = dft8(v)
    for i in range(8):
        v[i] = v[i] * tw[(lane * i) & 511]
    xbuf = np.zeros(8 * XROW, complex)
    for i in range(8):
        xbuf[i * XROW + lane] = v[i]
    v = np.stack([xbuf[k1 * XROW + 8 * a + q] for a in range(8)])
    v = dft8(v)
    for i in range(8):
        v[i] = v[i] * tw[(8 * q * i) & 511]
    for c in range(8):
        xbuf[k1 * XROW + 8 * c + q] = v[c]
    v = np.stack([xbuf[k1 * XROW + 8 * q + b] for b in range(8)])
    v = dft8(v)
    Z = np.zeros(512, complex)
    for d in range(8):
        Z[k1 + 8 * q + 64 * d] = v[d]
    return Z


def test_dft8_network():
    x = np.random.default_rng(0).standard_normal((8, 5)) + 1j * np.random.default_rng(1).standard_normal((8, 5))
    np.testing.assert_allclose(dft8(x), np.fft.fft(x, axis=0), atol=1e-12)


def test_wavefront_fft_plan_and_real_pair_split():
    rng = np.random.default_rng(2)
    a, b = rng.standard_normal(400), rng.standard_normal(400)
    z = np.zeros(512, complex)
    z[:400] = a + 1j * b
    Z = fft512_plan(z)
    np.testing.assert_allclose(Z, np.fft.fft(z), atol=1e-10)
    # split_power: bins k and 512-k of the packed transform give both real spectra
    k = np.arange(257)
    zk, wk = Z[k], Z[(512 - k) & 511]
    pa = ((zk.real + wk.real) ** 2 + (zk.imag - wk.imag) ** 2) / (4 * 512)
    pb = ((zk.imag + wk.imag) ** 2 + (zk.real - wk.real) ** 2) / (4 * 512)
    ref = o.powspec(np.stack([a, b]), 512)
    np.testing.assert_allclose(pa, ref[0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pb, ref[1], rtol=1e-10, atol=1e-12)


# ------------------------------------------------------------------ host tables through the C ABI
native = pytest.importorskip("kws._native")


@pytest.mark.parametrize("nfilt,sr", [(26, 16000), (40, 16000), (20, 8000), (13, 22050), (26, 44100)])
def test_host_mel_tables_match_oracle(nfilt, sr):
    edges = native.host_mel_edges(nfilt, 512, sr)
    assert edges.tolist() == o.mel_bin_edges(nfilt, 512, sr).astype(int).tolist()
    fb = native.host_mel_dense(nfilt, 512, sr)
    ref = o.get_filterbanks(nfilt, 512, sr)
    np.testing.assert_allclose(fb, ref, atol=1e-7)
    assert np.array_equal(fb != 0, ref != 0)  # identical sparsity: 459 non-zeros for the default


@pytest.mark.parametrize("nfilt,sr", [(26, 16000), (40, 16000), (20, 8000), (13, 22050), (26, 44100), (54, 16000)])
def test_host_mel_lane_layout(nfilt, sr):
    """Segments sit on adjacent lanes, in order, without overlap, inside the wavefront, and none of them straddles a
    16-lane DPP row (the kernel's segmented sums shift with row_shl); a filterbank that cannot be laid out so is refused."""
    first, count, used, row_safe = native.host_mel_layout(nfilt, 512, sr)
    edges = native.host_mel_edges(nfilt, 512, sr)
    assert count.tolist() == [-(-int(edges[s + 1] - edges[s]) // 8) for s in range(nfilt + 1)]
    end = 0
    for f, c in zip(first, count):
        assert f >= end  # in order, no overlap (idle lanes may pad)
        end = f + c
        if c:
            assert f // 16 == (f + c - 1) // 16
    assert row_safe and end <= used <= 64
    with pytest.raises(Exception):
        native.host_mel_layout(64, 512, 16000)  # 65 segments do not fit 64 lanes


@pytest.mark.parametrize("nfilt,numcep,L", [(26, 10, 22), (26, 13, 22), (40, 12, 0)])
def test_host_dct_lifter_matches_oracle(nfilt, numcep, L):
    want = o.dct2_ortho_matrix(nfilt, numcep) * o.lifter_vector(numcep, L)[:, None]
    np.testing.assert_allclose(native.host_dct_lifter(nfilt, numcep, L), want, atol=2e-7 * np.abs(want).max())


# ------------------------------------------------------------------ DS-CNN addressing plans
def block_geom(n):
    H, W = 45 + 2 * n, 1 + 2 * n
    ring = n > 1
    HI, WI = (H - 2, W - 2) if ring else (H, W)
    return H, W, ring, HI, WI


def depthwise_plan(n, z_in_interior, ring_val, w_dw, b_dw):
    """z_in_interior [64, HI*WI]; ring_val [64]; returns depthwise output [64, H*W] via slot addressing."""
    H, W, ring, HI, WI = block_geom(n)
    pin = HI * WI
    plane = np.zeros((64, pin + 2))
    plane[:, :pin] = z_in_interior
    plane[:, pin] = ring_val if ring else 0.0
    plane[:, pin + 1] = 0.0
    out = np.zeros((64, H * W))
    o_ = 1 if ring else 0
    for pos in range(H * W):
        h, x = divmod(pos, W)
        acc = b_dw.copy()
        for dh in (-1, 0, 1):
            for dx in (-1, 0, 1):
                hh, xx = h + dh - o_, x + dx - o_
                inside = 0 <= hh < HI and 0 <= xx < WI
                in_map = 0 <= h + dh < H and 0 <= x + dx < W
                a = hh * WI + xx if inside else (pin if (ring and in_map) else pin + 1)
                acc = acc + w_dw[:, dh + 1, dx + 1] * plane[:, a]
        out[:, pos] = acc
    return out


def test_depthwise_slot_addressing_matches_oracle():
    st = o_dscnn.random_state(seed=11)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((1, 1, 99, 10)).astype(np.float32))
    _, layers = o_dscnn.forward(st, x.double(), return_layers=True)
    prev = layers["conv1"][0].numpy().reshape(64, -1)  # block 1 input: conv1 output, no ring
    for n in range(1, 5):
        H, W, ring, HI, WI = block_geom(n)
        ring_val = np.maximum(st[f"dsconv{n - 1}.pointwise.bias"].double().numpy(), 0) if ring else np.zeros(64)
        got = depthwise_plan(n, prev, ring_val, st[f"dsconv{n}.depthwise.weight"].double().numpy()[:, 0],
                             st[f"dsconv{n}.depthwise.bias"].double().numpy())
        want = layers[f"dsconv{n}.depthwise"][0].numpy().reshape(64, -1)
        assert want.shape == (64, H * W)
        np.testing.assert_allclose(got, want, atol=1e-12)
        # next block's stored interior = this block's pointwise output without its ring
        full = layers[f"dsconv{n}"][0].numpy()
        assert full.shape == (64, H + 2, W + 2)
        prev = full[:, 1:-1, 1:-1].reshape(64, -1)


def test_conv1_im2col_addressing_matches_oracle():
    st = o_dscnn.random_state(seed=12)
    x = np.random.default_rng(4).standard_normal((99, 10))
    pad = np.zeros((103, 14))
    pad[2:101, 2:12] = x
    w = st["conv1.weight"].double().numpy()[:, 0]  # [64,10,10]
    b = st["conv1.bias"].double().numpy()
    out = np.zeros((64, 141))
    for pos in range(141):
        oh, ow = divmod(pos, 3)
        base = (2 * oh) * 14 + 2 * ow
        acc = b.copy()
        for s in range(50):
            for half in (0, 1):
                k = 2 * s + half
                off = ((2 * s) // 10) * 14 + (2 * s) % 10 + half
                acc = acc + w[:, k // 10, k % 10] * pad.reshape(-1)[base + off]
        out[:, pos] = np.maximum(acc, 0)
    _, layers = o_dscnn.forward(st, torch.from_numpy(x)[None, None], return_layers=True)
    np.testing.assert_allclose(out, layers["conv1"][0].numpy().reshape(64, -1), atol=1e-12)


def test_pool_ring_term():
    """mean over 55x11 = (sum of 53x9 interior + 128 * relu(bias)) / 605."""
    st = o_dscnn.random_state(seed=13)
    x = torch.randn(1, 1, 99, 10, dtype=torch.float64)
    _, layers = o_dscnn.forward(st, x, return_layers=True)
    z = layers["dsconv4"][0].numpy()
    interior = z[:, 1:-1, 1:-1].reshape(64, -1).sum(1)
    ring = np.maximum(st["dsconv4.pointwise.bias"].double().numpy(), 0)
    np.testing.assert_allclose((interior + 128 * ring) / 605, layers["pool"][0].numpy(), atol=1e-12)
