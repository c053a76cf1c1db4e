"""Signal-processing operators on device tensors.

Drop-in for ``kws/libs/speech_features/sigproc.py`` (``framesig`` ``:14-52``, ``magspec`` ``:55-77``,
``powspec`` ``:80-90``, ``preemphasis`` ``:93-103``): same names, arguments and shapes, executed by the
HIP kernels behind ``kws_framesig_f32`` / ``kws_spec_f32`` / ``kws_preemphasis_f32``.

Differences that follow from running on the GPU, stated rather than hidden:
  * tensors must live on a CUDA/ROCm device -- there is no CPU implementation in this package;
  * values cross the boundary as float32 (other floating dtypes are cast in and the result cast back); the transform
    itself is float32 for ``NFFT == 512`` (the hot path's wavefront FFT) and float64 for every other length;
  * ``NFFT`` may be any power of two in [64, 4096] or any value in [2, 2048].
"""
from __future__ import annotations

import math
from typing import Callable

import torch

from kws.common.errors import AudioProcessingError


def _ctx_for(t: torch.Tensor):
    from kws import _native

    if not t.is_cuda:
        raise AudioProcessingError("sigproc operators need a CUDA/ROCm tensor (no CPU implementation in this package)")
    ctx = _native.default_context(t.device.index or 0)
    ctx.use_torch_stream()
    return ctx


def _f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_floating_point():
        raise AudioProcessingError(f"expected a floating tensor, got {t.dtype}")
    return t.to(torch.float32).contiguous()


def framesig(signal: torch.Tensor, frame_len: int, frame_step: int, winfunc: Callable = torch.hann_window) -> torch.Tensor:
    """Overlapping, zero-padded, windowed frames ``[num_frames, frame_len]``."""
    if signal.dim() != 1:
        raise AudioProcessingError("framesig expects a 1-D signal")
    ctx = _ctx_for(signal)
    n = signal.shape[0]
    num_frames = 1 if n <= frame_len else 1 + math.ceil((n - frame_len) / frame_step)
    sig = _f32(signal)
    win = _f32(winfunc(frame_len, device=signal.device))
    frames = torch.empty((num_frames, frame_len), dtype=torch.float32, device=signal.device)
    ctx.framesig_f32(sig, frame_len, frame_step, win, frames)
    return frames.to(signal.dtype)


def _spec(frames: torch.Tensor, NFFT: int, power: bool) -> torch.Tensor:
    if frames.dim() != 2:
        raise AudioProcessingError("expected frames of shape [num_frames, frame_len]")
    ctx = _ctx_for(frames)
    fr = _f32(frames)
    spec = torch.empty((fr.shape[0], NFFT // 2 + 1), dtype=torch.float32, device=frames.device)
    ctx.spec_f32(fr, int(NFFT), power, spec)
    return spec.to(frames.dtype)


def magspec(frames: torch.Tensor, NFFT: int) -> torch.Tensor:
    """``|rfft(frame, NFFT)|`` per frame, ``[num_frames, NFFT//2 + 1]``."""
    return _spec(frames, NFFT, power=False)


def powspec(frames: torch.Tensor, NFFT: int) -> torch.Tensor:
    """``1/NFFT * |rfft(frame, NFFT)|**2`` per frame."""
    return _spec(frames, NFFT, power=True)


def preemphasis(signal: torch.Tensor, coeff: float = 0.97) -> torch.Tensor:
    """``y[0] = x[0]; y[n] = x[n] - coeff * x[n-1]``."""
    ctx = _ctx_for(signal)
    sig = _f32(signal).reshape(-1)
    out = torch.empty_like(sig)
    ctx.preemphasis_f32(sig, coeff, out)
    return out.reshape(signal.shape).to(signal.dtype)
