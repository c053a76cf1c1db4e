"""CPU oracle: the MFCC front end of the reference, restated in NumPy/SciPy.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) -- never imported by the
product path.

What is restated, and from where
--------------------------------
The reference computes features with one call,
``psf.mfcc(signal, samplerate, numcep, nfft, winlen, winstep, nfilt)``
(``kws/libs/audio_processor.py:270-278``), on a float32 signal in [-1, 1]
that ``librosa.load`` decoded from 16-bit PCM (``:145``) and
``librosa.util.fix_length`` padded/trimmed to 16000 samples (``:148``).
``python-speech-features==0.6`` is a third-party dependency that is not in
``/root/reference`` (pin: ``requirements.txt:11``, ``uv.lock:804-807``); its
published algorithm (``base.py`` mfcc/fbank/get_filterbanks/lifter and
``sigproc.py`` preemphasis/framesig/powspec) is restated here stage by stage.
The first four stages also exist in the reference itself as a torch
restatement (``kws/libs/speech_features/sigproc.py:14-103``) and are pinned
against it by ``tests/golden/sigproc_golden.npz``.

Numeric types follow what NumPy does in the reference pipeline:
  * PCM16 -> float32 by x / 32768            (librosa / soundfile decode)
  * pre-emphasis in float32                  (float32 array * Python float)
  * everything after framing in float64      (concatenate with float64 zeros)
  * the data loader casts the result to float32
    (``kws/libs/data_loader.py:103``).
"""
from __future__ import annotations

import decimal
import math
from dataclasses import dataclass

import numpy as np
from scipy.fftpack import dct as _scipy_dct

EPS = float(np.finfo(float).eps)  # 2.220446049250313e-16 (psf floors zeros to this)


def _round_half_up(x: float) -> int:
    # psf.sigproc.round_half_up: decimal quantize with ROUND_HALF_UP.
    return int(decimal.Decimal(x).quantize(decimal.Decimal("1"), rounding=decimal.ROUND_HALF_UP))


@dataclass(frozen=True)
class FrontendSpec:
    """Front-end parameters; defaults = ``AudioConfig`` (kws/libs/audio_processor.py:37-46)
    plus the psf.mfcc defaults the reference leaves untouched."""

    sample_rate: int = 16000
    n_samples: int = 16000          # AudioConfig.desired_samples
    winlen: float = 0.025           # frame_length
    winstep: float = 0.01           # frame_step
    nfft: int = 512                 # max(fft_size, int(winlen*sr)), audio_processor.py:268
    nfilt: int = 26
    numcep: int = 10
    preemph: float = 0.97           # psf.mfcc default
    ceplifter: int = 22             # psf.mfcc default
    lowfreq: float = 0.0
    highfreq: float | None = None   # -> sample_rate / 2
    append_energy: bool = True

    @property
    def frame_len(self) -> int:
        return _round_half_up(self.winlen * self.sample_rate)

    @property
    def frame_step(self) -> int:
        return _round_half_up(self.winstep * self.sample_rate)

    @property
    def num_frames(self) -> int:
        # sigproc.framesig, kws/libs/speech_features/sigproc.py:31-35
        if self.n_samples <= self.frame_len:
            return 1
        return 1 + int(math.ceil((1.0 * self.n_samples - self.frame_len) / self.frame_step))

    @property
    def n_bins(self) -> int:
        return self.nfft // 2 + 1


DEFAULT_SPEC = FrontendSpec()


# --------------------------------------------------------------------------- a1
def pcm16_to_float(pcm: np.ndarray) -> np.ndarray:
    """int16 PCM -> float32 in [-1, 1): what ``librosa.load`` hands the reference
    for a 16 kHz mono 16-bit WAV (kws/libs/audio_processor.py:145)."""
    pcm = np.asarray(pcm)
    if pcm.dtype != np.int16:
        raise TypeError("pcm16_to_float expects int16")
    return pcm.astype(np.float32) / np.float32(32768.0)


def fix_length(x: np.ndarray, size: int) -> np.ndarray:
    """``librosa.util.fix_length`` along the last axis: trim or zero-pad at the end
    (kws/libs/audio_processor.py:148)."""
    n = x.shape[-1]
    if n > size:
        return x[..., :size]
    if n < size:
        pad = [(0, 0)] * (x.ndim - 1) + [(0, size - n)]
        return np.pad(x, pad, mode="constant")
    return x


# --------------------------------------------------------------------------- a2
def preemphasis(signal: np.ndarray, coeff: float = 0.97) -> np.ndarray:
    """y[0] = x[0]; y[n] = x[n] - coeff*x[n-1]   (sigproc.py:93-103).

    Stays in the dtype of ``signal`` (float32 in the real pipeline) because the
    coefficient is a Python float."""
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


# --------------------------------------------------------------------------- a3
def framesig(sig: np.ndarray, frame_len: int, frame_step: int, window: np.ndarray | None = None) -> np.ndarray:
    """Overlapping frames, zero-padded tail, times a window (sigproc.py:14-52).

    ``window=None`` is the rectangular window ``psf.mfcc`` uses by default
    (the reference's torch default is Hann, which the real pipeline never uses).
    The zero tail is float64, which promotes the frames to float64."""
    slen = len(sig)
    if slen <= frame_len:
        num_frames = 1
    else:
        num_frames = 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((num_frames - 1) * frame_step + frame_len)
    padded = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + (np.arange(num_frames) * frame_step)[:, None]
    frames = padded[idx]
    win = np.ones((frame_len,)) if window is None else np.asarray(window, dtype=np.float64)
    return frames * win


# --------------------------------------------------------------------------- a4
def magspec(frames: np.ndarray, nfft: int) -> np.ndarray:
    """|rfft(frame, n=nfft)| (sigproc.py:55-77); frames longer than nfft are truncated."""
    return np.absolute(np.fft.rfft(frames, nfft))


def powspec(frames: np.ndarray, nfft: int) -> np.ndarray:
    """1/nfft * |rfft|^2 (sigproc.py:80-90)."""
    return 1.0 / nfft * np.square(magspec(frames, nfft))


# --------------------------------------------------------------------------- a5
def hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def mel_bin_edges(nfilt: int, nfft: int, sample_rate: int, lowfreq: float = 0.0, highfreq: float | None = None) -> np.ndarray:
    """nfilt+2 FFT-bin indices: points equally spaced in mel, floor((nfft+1)*hz/sr)."""
    highfreq = highfreq or sample_rate / 2
    assert highfreq <= sample_rate / 2, "highfreq is greater than samplerate/2"
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    return np.floor((nfft + 1) * mel2hz(melpoints) / sample_rate)


def get_filterbanks(nfilt: int = 26, nfft: int = 512, sample_rate: int = 16000, lowfreq: float = 0.0, highfreq: float | None = None) -> np.ndarray:
    """Triangular mel filterbank, float64 [nfilt, nfft//2+1] (psf base.get_filterbanks)."""
    b = mel_bin_edges(nfilt, nfft, sample_rate, lowfreq, highfreq)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(b[j]), int(b[j + 1])):
            fb[j, i] = (i - b[j]) / (b[j + 1] - b[j])
        for i in range(int(b[j + 1]), int(b[j + 2])):
            fb[j, i] = (b[j + 2] - i) / (b[j + 2] - b[j + 1])
    return fb


def fbank(signal: np.ndarray, spec: FrontendSpec = DEFAULT_SPEC):
    """(mel energies [frames, nfilt], frame energy [frames]); zeros floored to eps."""
    sig = preemphasis(signal, spec.preemph)
    frames = framesig(sig, spec.frame_len, spec.frame_step)
    pspec = powspec(frames, spec.nfft)
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, EPS, energy)
    fb = get_filterbanks(spec.nfilt, spec.nfft, spec.sample_rate, spec.lowfreq, spec.highfreq)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, EPS, feat)
    return feat, energy


# --------------------------------------------------------------------------- a6
def lifter_vector(ncoeff: int, L: int = 22) -> np.ndarray:
    if L > 0:
        return 1.0 + (L / 2.0) * np.sin(np.pi * np.arange(ncoeff) / L)
    return np.ones(ncoeff)


def dct2_ortho_matrix(n_in: int, n_out: int) -> np.ndarray:
    """Explicit DCT-II (norm='ortho') matrix [n_out, n_in]; equals
    ``scipy.fftpack.dct(type=2, norm='ortho')`` row by row."""
    n = np.arange(n_in)
    k = np.arange(n_out)[:, None]
    m = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * math.sqrt(2.0 / n_in)
    m[0, :] = math.sqrt(1.0 / n_in)
    return m


def mfcc(signal: np.ndarray, spec: FrontendSpec = DEFAULT_SPEC) -> np.ndarray:
    """psf.mfcc for one clip: float64 [num_frames, numcep]."""
    feat, energy = fbank(signal, spec)
    feat = np.log(feat)
    feat = _scipy_dct(feat, type=2, axis=1, norm="ortho")[:, : spec.numcep]
    feat = lifter_vector(feat.shape[1], spec.ceplifter) * feat
    if spec.append_energy:
        feat[:, 0] = np.log(energy)
    return feat


# --------------------------------------------------------------------------- a7/a8
def extract_features_pcm16(pcm: np.ndarray, spec: FrontendSpec = DEFAULT_SPEC) -> np.ndarray:
    """One int16 clip -> float64 [frames, numcep], the value
    ``AudioProcessor.extract_features`` returns for that clip
    (kws/libs/audio_processor.py:235-278) when no augmentation is applied."""
    x = fix_length(pcm16_to_float(pcm), spec.n_samples)
    return mfcc(x, spec)


def collate_pcm16(pcm_batch: np.ndarray, spec: FrontendSpec = DEFAULT_SPEC) -> np.ndarray:
    """int16 [B, n] -> float32 [B, 1, frames, numcep]: per-clip features, float32
    cast and channel axis of ``SpeechCommandsDataLoader.__getitem__``
    (kws/libs/data_loader.py:103-104) stacked by torch's default collate.
    Per-clip Python loop on purpose: that is the reference's structure."""
    out = np.empty((len(pcm_batch), 1, spec.num_frames, spec.numcep), dtype=np.float32)
    for i, clip in enumerate(pcm_batch):
        out[i, 0] = extract_features_pcm16(clip, spec).astype(np.float32)
    return out
