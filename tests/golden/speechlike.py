"""Speech-like synthetic PCM16 clips (no dataset is reachable from the build container or the GPU box).

A clip is a sequence of voiced / unvoiced segments separated by pauses:
  * voiced: a glottal pulse train (f0 80-260 Hz with vibrato and a falling contour, Rosenberg-like pulse shape) through
    three two-pole resonators (formants F1-F3 drawn per segment, bandwidths 60-200 Hz) and a +6 dB/oct lip radiation;
  * unvoiced: white noise through one wide resonator at 2.5-6 kHz (fricative);
  * pauses of 100-300 ms that are EXACT zeros (digital silence, as in edited recordings) or +-1 LSB dither;
  * overall level from -6 to -50 dBFS, attack / release ramps of 5-20 ms.
These are the inputs on which a float32 front end is weakest: a few strong harmonics over a floor 60-90 dB below them
inside one frame, onsets next to digital silence, and very low levels where the int16 quantisation is the floor.
Deterministic for a given seed (NumPy RandomState), so the fixture generator and the GPU-side probes see the same clips.
"""
import numpy as np

SR = 16000
N = 16000


def _resonator(x: np.ndarray, freq: float, bw: float) -> np.ndarray:
    """Two-pole resonator y[n] = x[n] + 2 r cos(w) y[n-1] - r^2 y[n-2], unity gain at DC removed by (1 - r)."""
    r = np.exp(-np.pi * bw / SR)
    a1, a2 = 2.0 * r * np.cos(2.0 * np.pi * freq / SR), -r * r
    y = np.zeros_like(x)
    y1 = y2 = 0.0
    g = 1.0 - r
    for n in range(len(x)):
        v = g * x[n] + a1 * y1 + a2 * y2
        y[n] = v
        y2, y1 = y1, v
    return y


def _glottal_train(n: int, f0_start: float, f0_end: float, rs: np.random.RandomState) -> np.ndarray:
    t = np.arange(n) / SR
    f0 = np.linspace(f0_start, f0_end, n) * (1.0 + 0.02 * np.sin(2 * np.pi * 5.0 * t + rs.uniform(0, 6.28)))
    phase = np.cumsum(f0) / SR
    frac = phase - np.floor(phase)
    # Rosenberg-like pulse: rising half-cosine over 40 % of the period, falling quarter-cosine over 16 %, closed after
    open_, close_ = 0.40, 0.16
    g = np.where(frac < open_, 0.5 * (1.0 - np.cos(np.pi * frac / open_)),
                 np.where(frac < open_ + close_, np.cos(0.5 * np.pi * (frac - open_) / close_), 0.0))
    return np.diff(g, prepend=g[0])  # lip radiation: first difference


def speechlike_clip(seed: int, level_dbfs: float, pause: str = "zeros") -> np.ndarray:
    """One int16 clip of N samples.  pause: 'zeros' (exact digital silence) or 'dither' (+-1 LSB noise)."""
    rs = np.random.RandomState(seed)
    x = np.zeros(N)
    pos = int(rs.uniform(0.02, 0.15) * SR)
    while pos < N - 800:
        seg = int(rs.uniform(0.08, 0.35) * SR)
        seg = min(seg, N - pos)
        if rs.uniform() < 0.75:  # voiced
            src = _glottal_train(seg, rs.uniform(90, 260), rs.uniform(80, 200), rs)
            y = src
            for lo, hi in ((250, 900), (900, 2500), (2400, 3600)):
                y = _resonator(y, rs.uniform(lo, hi), rs.uniform(60, 200)) * 8.0
        else:                    # unvoiced (fricative)
            y = _resonator(rs.standard_normal(seg), rs.uniform(2500, 6000), rs.uniform(800, 2000))
        y = y / (np.abs(y).max() + 1e-12)
        ramp = int(rs.uniform(0.005, 0.02) * SR)
        env = np.ones(seg)
        env[:ramp] = np.linspace(0, 1, ramp)
        env[-ramp:] = np.linspace(1, 0, ramp)
        x[pos:pos + seg] += y * env * rs.uniform(0.3, 1.0)
        pos += seg + int(rs.uniform(0.10, 0.30) * SR)  # pause of 100-300 ms
    x = x / (np.abs(x).max() + 1e-12) * (32767.0 * 10.0 ** (level_dbfs / 20.0))
    pcm = np.round(x)
    if pause == "dither":
        quiet = pcm == 0
        pcm = np.where(quiet, rs.randint(-1, 2, size=N), pcm)
    return np.clip(pcm, -32768, 32767).astype(np.int16)


def speechlike_set(n: int = 16, seed: int = 400):
    """n clips: levels spread over -6 .. -50 dBFS, alternating exact-zero and dithered pauses."""
    levels = np.linspace(-6.0, -50.0, n)
    clips, names = [], []
    for i in range(n):
        pause = "zeros" if i % 2 == 0 else "dither"
        clips.append(speechlike_clip(seed + i, float(levels[i]), pause))
        names.append(f"speech_{i:02d}_{levels[i]:+.0f}dBFS_{pause}")
    return np.stack(clips), names
