"""Clip generators of the front-end precision audit (test infrastructure; used by tools/fe_precision_audit.py and
tests/test_gpu_parity.py): int16 [n, 16000] sets that stress the float32 transform's dynamic range in different ways."""
import numpy as np

import speechlike

T = np.arange(16000) / 16000.0


def clip16(x):
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def make_sets(n: int, seed: int):
    """{name: int16 [n, 16000]} -- uniform / Gaussian noise at random levels, gated bursts over silence or +-1 LSB dither, tones
    and chirps with fades and gates, mixtures, speech-like clips with random parameters.  Deterministic in (n, seed)."""
    rng = np.random.default_rng(seed)

    def gen_uniform():
        return clip16(rng.uniform(-1, 1, 16000) * 10 ** rng.uniform(1.0, 4.5))

    def gen_gauss():
        return clip16(rng.standard_normal(16000) * 10 ** rng.uniform(0.5, 4.0))

    def gen_bursts():
        x = np.zeros(16000)
        for _ in range(int(rng.integers(1, 6))):
            a, m = int(rng.integers(0, 15000)), int(rng.integers(1, 3000))
            x[a:a + m] = rng.uniform(-1, 1, len(x[a:a + m])) * 10 ** rng.uniform(1.5, 4.4)
        if rng.random() < 0.5:
            x += rng.integers(-1, 2, 16000)  # +-1 LSB dither under the silence
        return clip16(x)

    def gen_tone():
        f0, f1 = rng.uniform(60, 7800, 2)
        ph = 2 * np.pi * (f0 * T + (f1 - f0) * T * T / 2 * (rng.random() < 0.5))
        x = np.sin(ph) * 10 ** rng.uniform(2.0, 4.4)
        if rng.random() < 0.5:
            x *= np.linspace(rng.uniform(0, 1), rng.uniform(0, 1), 16000) ** 2   # a fade
        if rng.random() < 0.5:
            x[: int(rng.integers(0, 12000))] = 0
        return clip16(x + rng.standard_normal(16000) * 10 ** rng.uniform(-1, 1.5))

    def gen_mix():
        a, b = gen_tone().astype(np.float64), gen_bursts().astype(np.float64)
        return clip16(a * rng.uniform(0, 1) + b)

    sp, _ = speechlike.speechlike_set(n, 1000 + seed)
    gens = {"uniform": gen_uniform, "gauss": gen_gauss, "bursts": gen_bursts, "tones": gen_tone, "mix": gen_mix}
    sets = {k: np.stack([g() for _ in range(n)]) for k, g in gens.items()}   # (generator by generator: the order of the draws matters)
    sets["speechlike"] = sp
    return sets
