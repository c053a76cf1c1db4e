"""Feature extraction for keyword spotting on MI355X.

Drop-in for the reference's ``kws/libs/audio_processor.py``: same ``AudioConfig`` fields and defaults
(``:20-72``), same ``AudioProcessor(dataset_path, config)`` with ``transform(filepath, label)``
(``:130-170``) and ``extract_features(signal, ...)`` (``:235-278``).  The MFCC arithmetic the reference
delegates to ``python_speech_features.mfcc`` runs in the batched HIP kernel behind ``kws_mfcc_i16`` /
``kws_mfcc_f32`` (include/kws_hip.h).  WAV decoding and the random augmentations stay on the host.

Additions for batched use: ``extract_features_batch`` (device tensor in, ``float32[B,1,T,F]`` out --
the collated batch of the reference's data loader), ``load_pcm16`` (16-bit files as int16, the fast path's input) and
``load_audio`` (any PCM / float WAV as float32 mono, what ``librosa.load`` returns).
"""
from __future__ import annotations

import random
import wave
from dataclasses import asdict, dataclass
from decimal import ROUND_HALF_UP, Decimal
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from kws.common.errors import AudioProcessingError, handle_error

BACKGROUND_NOISE_DIR = "_background_noise_"
SILENCE_INDEX = 0


@dataclass
class AudioConfig:
    """Audio / feature parameters (defaults = reference ``AudioConfig``)."""

    # background noise mixing
    background_volume: float = 0.1
    background_frequency: float = 0.8
    use_background_noise: bool = True
    # clip geometry
    sample_rate: int = 16000
    clip_duration_ms: int = 1000
    time_shift_ms: float = 100
    # MFCC
    frame_length: float = 0.025
    frame_step: float = 0.01
    num_cepstral_coeffs: int = 10
    num_mel_filters: int = 26
    fft_size: int = 512

    @property
    def time_shift(self) -> int:
        return int((self.time_shift_ms * self.sample_rate) / 1000)

    @property
    def desired_samples(self) -> int:
        return int(self.sample_rate * self.clip_duration_ms / 1000)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


def _half_up(x: float) -> int:
    return int(Decimal(x).quantize(Decimal("1"), rounding=ROUND_HALF_UP))


def load_pcm16(path, sample_rate: int = 16000) -> np.ndarray:
    """Read a mono 16-bit PCM WAV as int16 (what ``librosa.load`` scales by 1/32768 in the reference,
    ``audio_processor.py:145``).  Stereo is averaged like librosa's mono mix-down; a different sample
    rate is refused (librosa would resample with soxr, which cannot be reproduced bit for bit here)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise AudioProcessingError(f"{path}: only 16-bit PCM WAV is supported (sample width {w.getsampwidth()})")
        if w.getframerate() != sample_rate:
            raise AudioProcessingError(f"{path}: sample rate {w.getframerate()} != {sample_rate}; resampling is not implemented")
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        ch = w.getnchannels()
    if ch > 1:
        data = data.reshape(-1, ch)
        return data  # caller mixes down in float
    return data


def load_audio(path, sample_rate: int = 16000, resample: bool = False) -> np.ndarray:
    """Decode a WAV file to what ``librosa.load(path, sr=sample_rate)`` hands the reference (``audio_processor.py:145``):
    float32 mono in [-1, 1).  Integer PCM of 8 / 16 / 24 / 32 bits is scaled like libsndfile does (u8: (v - 128)/128,
    otherwise v / 2**(bits-1)), IEEE float32 / float64 is taken as is, WAVE_FORMAT_EXTENSIBLE is read through its
    sub-format; channels are averaged in float32 (``librosa.to_mono``).

    A file at another sample rate is refused unless ``resample=True``: librosa resamples with soxr ('soxr_hq'), which is
    not available here and cannot be reproduced bit for bit; with ``resample=True`` a Kaiser-windowed polyphase filter
    (``scipy.signal.resample_poly``) is used instead -- PARITY UNPINNED against the reference for such files."""
    import struct

    with open(str(path), "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioProcessingError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise AudioProcessingError(f"{path}: missing fmt or data chunk")
    tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the first two bytes of the sub-format GUID are the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    if ch < 1:
        raise AudioProcessingError(f"{path}: no channels")
    if tag == 1 and bits == 8:
        x = (np.frombuffer(payload, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(payload[: len(payload) // 2 * 2], dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 24:
        raw = np.frombuffer(payload[: len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(payload[: len(payload) // 4 * 4], dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(payload[: len(payload) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(payload[: len(payload) // 8 * 8], dtype="<f8").astype(np.float32)
    else:
        raise AudioProcessingError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    x = x[: len(x) // ch * ch]
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1, dtype=np.float32)
    if rate != sample_rate:
        if not resample:
            raise AudioProcessingError(f"{path}: sample rate {rate} != {sample_rate}; pass resample=True for a polyphase "
                                       "resampler (librosa's soxr resampler cannot be reproduced: parity unpinned)")
        from math import gcd

        from scipy.signal import resample_poly

        g = gcd(int(rate), int(sample_rate))
        x = resample_poly(x.astype(np.float64), sample_rate // g, rate // g, window=("kaiser", 14.0)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def _to_float_mono(pcm: np.ndarray) -> np.ndarray:
    x = pcm.astype(np.float32) / np.float32(32768.0)
    return x.mean(axis=1, dtype=np.float32) if x.ndim == 2 else x


def fix_length(x: np.ndarray, size: int) -> np.ndarray:
    """Trim or zero-pad the last axis to ``size`` (``librosa.util.fix_length``, ``audio_processor.py:148``)."""
    n = x.shape[-1]
    if n >= size:
        return x[..., :size]
    return np.concatenate([x, np.zeros(x.shape[:-1] + (size - n,), dtype=x.dtype)], axis=-1)


class AudioProcessor:
    """WAV -> fixed length -> (augment) -> MFCC, with the MFCC on the GPU."""

    def __init__(self, dataset_path: Optional[Path], config: Optional[AudioConfig] = None, device: int = 0,
                 precise: bool = False, resample: bool = False):
        """``precise=True`` selects the float64 front end (``KWS_FE_F64``: everything after framing in float64, as psf
        computes it) instead of the fast float32 kernel; geometries the fast kernel is not built for (``nfft != 512``,
        i.e. ``frame_length * sample_rate > 512``) use it anyway."""
        self.config = config if config is not None else AudioConfig()
        self.dataset_path = Path(dataset_path) if dataset_path is not None else None
        self.device = device
        self.precise = bool(precise)
        self.resample = bool(resample)  # files at another rate: polyphase resampler instead of a refusal (parity unpinned)
        self._ctx = None
        self._ctx_key = None
        try:
            self.background_data = self._load_background_data()
        except Exception as e:  # same wrapping as the reference (:96-101)
            handle_error(e, AudioProcessingError, "Failed to initialize audio processor")

    # ------------------------------------------------------------------ host side
    def _load_background_data(self) -> List[np.ndarray]:
        if self.dataset_path is None:
            return []
        folder = self.dataset_path / BACKGROUND_NOISE_DIR
        if not folder.exists():
            return []
        clips = []
        for wav_path in sorted(folder.glob("*.wav")):
            try:
                clips.append(load_audio(wav_path, self.config.sample_rate, self.resample))
            except Exception:
                continue  # unreadable background files are skipped, as in the reference (:119-123)
        return clips

    def _apply_time_shift(self, audio: np.ndarray) -> np.ndarray:
        """Random shift by up to +-time_shift samples, zero filled (``:172-188``)."""
        limit = self.config.time_shift
        amount = np.random.randint(-limit, limit) if limit > 0 else 0
        n = self.config.desired_samples
        out = np.zeros(n, dtype=audio.dtype)
        if amount >= 0:
            out[amount:] = audio[: n - amount]
        else:
            out[: n + amount] = audio[-amount:n]
        return out

    def _add_background_noise(self, audio: np.ndarray, label: int, background_data: List[np.ndarray]) -> np.ndarray:
        """Mix a random background segment at a random volume (``:190-233``)."""
        if not background_data:
            return audio
        n = self.config.desired_samples
        bg = random.choice(background_data)
        if len(bg) <= n:
            bg = np.tile(bg, int(np.ceil(n / len(bg))) + 1)
        start = np.random.randint(0, len(bg) - n)
        if label == SILENCE_INDEX:
            volume = np.random.uniform(0, 1)
        elif np.random.uniform(0, 1) < self.config.background_frequency:
            volume = np.random.uniform(0, self.config.background_volume)
        else:
            volume = 0
        return audio + bg[start:start + n] * volume

    def transform(self, filepath: str, label: int) -> np.ndarray:
        """One file -> MFCC ``[frames, numcep]`` with the reference's augmentation order (``:130-170``)."""
        try:
            audio = load_audio(filepath, self.config.sample_rate, self.resample)  # float32 mono, as librosa.load returns it
            audio = fix_length(audio, self.config.desired_samples)
            if label == SILENCE_INDEX:
                audio = np.zeros_like(audio)
            audio = self._apply_time_shift(audio)
            if self.config.use_background_noise or label == SILENCE_INDEX:
                audio = self._add_background_noise(audio, label, self.background_data)
            return self.extract_features(audio, winlen=self.config.frame_length, winstep=self.config.frame_step)
        except Exception as e:
            raise AudioProcessingError(f"Error processing audio file {filepath}: {str(e)}") from e

    # ------------------------------------------------------------------ device side
    def _context(self, n_samples: int, samplerate: int, numcep: int, winlen: float, winstep: float, nfilt: int):
        from kws import _native

        nfft = max(self.config.fft_size, int(winlen * samplerate))  # the reference overrides its nfft argument (:268)
        key = (n_samples, samplerate, numcep, nfft, _half_up(winlen * samplerate), _half_up(winstep * samplerate), nfilt)
        if self._ctx is None:
            self._ctx = _native.Context(self.device, AudioProcessingError)
            self._ctx.set_frontend_math(_native.FE_F64 if self.precise else _native.FE_F32)
        if key != self._ctx_key:
            self._ctx.set_frontend(sample_rate=samplerate, n_samples=n_samples, frame_len=key[4], frame_step=key[5],
                                   nfft=nfft, nfilt=nfilt, numcep=numcep, preemph=0.97, ceplifter=22)
            self._ctx_key = key
        return self._ctx

    def extract_features(self, signal: np.ndarray, samplerate: int = None, numcep: int = None, nfft: int = None,
                         winlen: float = None, winstep: float = None, nfilt: int = None) -> np.ndarray:
        """MFCC of one clip: float signal in [-1, 1] (or int16 PCM) -> ``float64[frames, numcep]``.

        Falsy arguments fall back to the config exactly as in the reference (``:260-268``); ``nfft`` is
        accepted and ignored there too.  Values are computed on the GPU (float32 kernel, or float64 with
        ``precise=True`` / for ``nfft != 512``), cross the boundary as float32 and are returned as float64 to keep the
        reference's dtype."""
        import torch

        c = self.config
        samplerate = samplerate or c.sample_rate
        numcep = numcep or c.num_cepstral_coeffs
        winlen = winlen or c.frame_length
        winstep = winstep or c.frame_step
        nfilt = nfilt or c.num_mel_filters
        sig = np.asarray(signal)
        if sig.ndim != 1:
            raise AudioProcessingError("extract_features expects a 1-D signal (use extract_features_batch for batches)")
        ctx = self._context(sig.shape[0], samplerate, numcep, winlen, winstep, nfilt)
        dev = torch.device("cuda", self.device)
        if sig.dtype == np.int16:
            x = torch.from_numpy(np.ascontiguousarray(sig)).to(dev)[None]
        else:
            x = torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float32)).to(dev)[None]
        return self.extract_features_batch(x, _ctx=ctx)[0, 0].double().cpu().numpy()

    def transform_batch(self, pcm, labels):
        """Batched, on-device version of ``transform`` for clips already decoded to ``int16[B, n]``:
        silence zeroing, random time shift, background mix (``kws_augment_i16``) and MFCC (``kws_mfcc_f32``).
        The random draws are made on the host with the same NumPy / random calls as the reference
        (``:172-233``); returns ``float32[B,1,frames,numcep]`` on the device."""
        import torch

        c = self.config
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        labels = np.asarray(labels)
        B, n = pcm.shape
        if n != c.desired_samples:
            raise AudioProcessingError(f"transform_batch expects clips of {c.desired_samples} samples")
        limit = c.time_shift
        shift = np.array([np.random.randint(-limit, limit) if limit > 0 else 0 for _ in range(B)], dtype=np.int32)
        silence = (labels == SILENCE_INDEX).astype(np.uint8)
        dev = torch.device("cuda", self.device)
        ctx = self._context(n, c.sample_rate, c.num_cepstral_coeffs, c.frame_length, c.frame_step, c.num_mel_filters)
        bg = bg_off = bg_vol = None
        if self.background_data and (c.use_background_noise or silence.any()):
            pool, starts = getattr(self, "_bg_pool", None), getattr(self, "_bg_starts", None)
            if pool is None:
                clips = [np.tile(b, int(np.ceil(n / len(b))) + 1) if len(b) <= n else b for b in self.background_data]
                starts = np.cumsum([0] + [len(b) for b in clips[:-1]])
                pool = torch.from_numpy(np.concatenate(clips).astype(np.float32)).to(dev)
                self._bg_pool, self._bg_starts, self._bg_lens = pool, starts, [len(b) for b in clips]
            off, vol = np.zeros(B, np.int32), np.zeros(B, np.float32)
            for i in range(B):
                if not (c.use_background_noise or silence[i]):
                    continue
                k = random.randrange(len(self.background_data))
                off[i] = self._bg_starts[k] + np.random.randint(0, self._bg_lens[k] - n)
                if silence[i]:
                    vol[i] = np.random.uniform(0, 1)
                elif np.random.uniform(0, 1) < c.background_frequency:
                    vol[i] = np.random.uniform(0, c.background_volume)
            bg, bg_off, bg_vol = self._bg_pool, torch.from_numpy(off).to(dev), torch.from_numpy(vol).to(dev)
        out = torch.empty((B, n), dtype=torch.float32, device=dev)
        ctx.use_torch_stream()
        ctx.augment_i16(torch.from_numpy(pcm).to(dev), out, shift=torch.from_numpy(shift).to(dev), bg=bg, bg_off=bg_off,
                        bg_vol=bg_vol, silence=torch.from_numpy(silence).to(dev))
        return self.extract_features_batch(out, _ctx=ctx)

    def extract_features_batch(self, signals, _ctx=None):
        """``int16[B,n]`` (PCM) or ``float32[B,n]`` device tensor -> ``float32[B,1,frames,numcep]`` device tensor:
        the batch the reference builds with ``__getitem__`` + default collate (``kws/libs/data_loader.py:96-105``)."""
        import torch

        if not signals.is_cuda:
            raise AudioProcessingError("extract_features_batch needs a CUDA/ROCm tensor (there is no CPU path)")
        if signals.dim() != 2:
            raise AudioProcessingError("extract_features_batch expects [B, n_samples]")
        c = self.config
        ctx = _ctx or self._context(signals.shape[1], c.sample_rate, c.num_cepstral_coeffs, c.frame_length, c.frame_step,
                                    c.num_mel_filters)
        frames, numcep = ctx.frontend_shape()
        signals = signals.contiguous()
        out = torch.empty((signals.shape[0], 1, frames, numcep), dtype=torch.float32, device=signals.device)
        ctx.use_torch_stream()
        if signals.dtype == torch.int16:
            ctx.mfcc_i16(signals, out)
        elif signals.dtype == torch.float32:
            ctx.mfcc_f32(signals, out)
        else:
            raise AudioProcessingError(f"unsupported signal dtype {signals.dtype}: use int16 PCM or float32")
        return out
