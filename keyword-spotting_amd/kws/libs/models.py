"""Keyword-spotting models whose forward runs in the fused gfx950 kernel.

Drop-in for the reference's ``kws/libs/models.py``: ``KeywordSpottingModel`` (``forward`` / ``save`` /
``load``, ``:13-72``), ``DepthwiseSeparableConvBlock`` (``:75-119``) and
``DepthwiseSeparableConv(num_classes=12, input_channels=1)`` (``:122-183``) with identical parameter
names, shapes and initialisation, so reference checkpoints (bare ``state_dict`` or the trainer's
``{"model_state_dict": ...}``) load unchanged.  The modules only *hold* parameters; the arithmetic of
``forward`` is ``kws_forward_f32`` (include/kws_hip.h): LDS-resident activations, depthwise 3x3 on the
VALU, pointwise 1x1 and conv1 on the matrix cores.  Inference only -- no autograd graph is built.
"""
from __future__ import annotations

import abc
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from kws.common.errors import ModelError

FEATURE_SHAPE = (1, 99, 10)  # [C, T, F] the fused kernel is built for (reference defaults)


class KeywordSpottingModel(nn.Module, abc.ABC):
    """Interface every KWS model implements."""

    def __init__(self, num_classes: int, **kwargs):
        super().__init__()
        self.num_classes = num_classes

    @abc.abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """``x [B,1,T,F]`` -> logits ``[B,num_classes]``."""

    def save(self, path: str) -> None:
        try:
            torch.save(self.state_dict(), path)
        except Exception as e:
            raise ModelError(f"Failed to save model to {path}: {str(e)}") from e

    def load(self, path: str, device: Optional[torch.device] = None) -> None:
        try:
            if device is None:
                device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
            state = torch.load(path, map_location=device, weights_only=True)
            if isinstance(state, dict) and "model_state_dict" in state:  # trainer checkpoints (training.py:199-216)
                state = state["model_state_dict"]
            self.load_state_dict(state)
            self.to(device)
        except Exception as e:
            raise ModelError(f"Failed to load model from {path}: {str(e)}") from e


class DepthwiseSeparableConvBlock(nn.Module):
    """One block: kxk depthwise + 1x1 pointwise + ReLU (the pointwise keeps the reference's ``padding=padding`` --
    the relu(bias) ring, ``models.py:104-106``).  Inside ``DepthwiseSeparableConv`` the four blocks run fused in the
    DS-CNN kernel and this module only holds their parameters; called on its own, ``forward`` is the general-shape
    operator ``kws_dsblock_forward_f32`` (any ``[B, C_in, H, W]``, kernel size, stride and padding)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, padding: int = 1):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                   groups=in_channels)
        self.pointwise = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=padding)
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self._ctx = None
        self._dev_params = None  # (fingerprint, tensors on the device)

    def sync_weights(self) -> None:
        """Force a re-upload at the next forward (needed only after edits through ``p.data``)."""
        self._dev_params = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """``float32[B, C_in, H, W]`` on the GPU -> ``float32[B, C_out, Ho + 2p, Wo + 2p]`` (``models.py:108-119``)."""
        from kws import _native

        if not x.is_cuda:
            raise ModelError("DepthwiseSeparableConvBlock.forward needs a CUDA/ROCm tensor: the forward is a HIP kernel "
                             "and has no CPU fallback")
        c_in = self.depthwise.in_channels
        if x.dim() != 4 or x.shape[1] != c_in:
            raise ModelError(f"expected input [B,{c_in},H,W], got {tuple(x.shape)}")
        dev = x.device.index or 0
        if self._ctx is None or self._ctx.device != dev:
            self._ctx = _native.Context(dev, ModelError)
            self._dev_params = None
        params = (self.depthwise.weight, self.depthwise.bias, self.pointwise.weight, self.pointwise.bias)
        fp = tuple((p.data_ptr(), p._version) for p in params)
        if self._dev_params is None or self._dev_params[0] != fp:
            self._dev_params = (fp, tuple(p.detach().to(x.device, torch.float32).contiguous() for p in params))
        dw_w, dw_b, pw_w, pw_b = self._dev_params[1]
        k, s, p = self.kernel_size, self.stride, self.padding
        B, _, H, W = x.shape
        if H + 2 * p < k or W + 2 * p < k:
            raise ModelError(f"kernel size {k} exceeds the padded input {H + 2 * p} x {W + 2 * p}")
        ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        self._ctx.use_torch_stream()
        x = x.detach().to(torch.float32).contiguous()
        out = torch.empty((B, self.pointwise.out_channels, ho + 2 * p, wo + 2 * p), dtype=torch.float32, device=x.device)
        self._ctx.dsblock_forward_f32(x, dw_w, dw_b, pw_w, pw_b, k, s, p, out)
        return out


class DepthwiseSeparableConv(KeywordSpottingModel):
    """DS-CNN: conv1 (1->64, 10x10, s2, p2) + 4 depthwise-separable blocks + global pool + Linear."""

    def __init__(self, num_classes: int = 12, input_channels: int = 1):
        super().__init__(num_classes)
        if not 1 <= input_channels <= 64:
            raise ModelError("input_channels must be in [1, 64]")
        self.input_channels = int(input_channels)
        self.conv1 = nn.Conv2d(input_channels, 64, kernel_size=10, stride=2, padding=2)
        self.dsconv1 = DepthwiseSeparableConvBlock(64, 64)
        self.dsconv2 = DepthwiseSeparableConvBlock(64, 64)
        self.dsconv3 = DepthwiseSeparableConvBlock(64, 64)
        self.dsconv4 = DepthwiseSeparableConvBlock(64, 64)
        self.fc = nn.Linear(64, num_classes)
        self._initialize_weights()
        self._ctx = None
        self._uploaded = None  # fingerprint of the parameters currently on the device

    def _initialize_weights(self):
        # same scheme as the reference (:149-158)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ device plumbing
    def packed_weights(self) -> np.ndarray:
        """The 20 ``state_dict`` tensors, in order, as one float32 vector (``kws_load_dscnn`` layout)."""
        return np.concatenate([v.detach().to("cpu", torch.float32).reshape(-1).numpy() for v in self.state_dict().values()])

    def sync_weights(self) -> None:
        """Force a re-upload of the parameters at the next forward.  The device copy is refreshed automatically when
        a parameter is replaced or modified through autograd-visible in-place operations (``p.copy_``, ``p.add_``,
        ``load_state_dict``: they bump ``p._version``); edits made behind torch's back through ``p.data`` (e.g.
        ``p.data.clamp_()``) do not, and need this call."""
        self._uploaded = None

    def _context(self, device_index: int):
        from kws import _native

        if self._ctx is None or self._ctx.device != device_index:
            self._ctx = _native.Context(device_index, ModelError)
            self._uploaded = None
        fp = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if fp != self._uploaded:
            self._ctx.load_dscnn(self.packed_weights(), self.num_classes, self.input_channels)
            self._uploaded = fp
        self._ctx.use_torch_stream()
        return self._ctx

    def _check_input(self, x: torch.Tensor, what: str):
        if not x.is_cuda:
            raise ModelError(f"{what} needs a CUDA/ROCm tensor: the forward is a HIP kernel and has no CPU fallback")

    def forward(self, x: torch.Tensor, return_labels: bool = False):
        """``float32[B,C,T,F]`` on the GPU -> logits ``float32[B,num_classes]`` (and argmax labels).  Any ``T x F`` the
        reference's forward accepts (``models.py:160-183``: the pooling is adaptive): 99 x 10 runs the fused LDS-resident
        kernel, any other map the composed path (``kws_forward_map_f32``)."""
        self._check_input(x, "DepthwiseSeparableConv.forward")
        if x.dim() != 4 or x.shape[1] != self.input_channels:
            raise ModelError(f"expected input [B,{self.input_channels},T,F], got {tuple(x.shape)}")
        ctx = self._context(x.device.index or 0)
        x = x.detach().to(torch.float32).contiguous()
        logits = torch.empty((x.shape[0], self.num_classes), dtype=torch.float32, device=x.device)
        labels = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
        if tuple(x.shape[2:]) == FEATURE_SHAPE[1:]:
            ctx.forward_f32(x, logits, labels)
        else:
            ctx.forward_map_f32(x, logits, labels)
        return (logits, labels) if return_labels else logits

    def infer_pcm16(self, wav: torch.Tensor):
        """Fused path: ``int16[B,16000]`` PCM on the GPU -> (logits, labels); MFCC + DS-CNN back to back
        on one stream, features never leave the device (``kws_infer_i16``)."""
        self._check_input(wav, "DepthwiseSeparableConv.infer_pcm16")
        if self.input_channels != 1:
            raise ModelError("infer_pcm16 needs input_channels=1: the MFCC front end yields one channel")
        if wav.dtype != torch.int16 or wav.dim() != 2:
            raise ModelError("infer_pcm16 expects an int16 tensor [B, n_samples]")
        ctx = self._context(wav.device.index or 0)
        wav = wav.contiguous()
        logits = torch.empty((wav.shape[0], self.num_classes), dtype=torch.float32, device=wav.device)
        labels = torch.empty((wav.shape[0],), dtype=torch.int32, device=wav.device)
        ctx.infer_i16(wav, logits, labels)
        return logits, labels


class DepthwiseSeparableConvBN(KeywordSpottingModel):
    """Build-defined model-zoo member (the reference has no BatchNorm): the same DS-CNN
    with an inference-mode BatchNorm2d after conv1, after every depthwise and after every pointwise convolution
    (conv -> BN -> ReLU where the plain model has conv -> ReLU).  ``fold()`` folds every BatchNorm into the
    convolution before it -- w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta -- which
    yields an ordinary ``DepthwiseSeparableConv`` whose forward is the fused kernel.  The relu(bias) ring of the
    padded 1x1 convolutions stays exact: a BatchNorm acts per channel, ring included.  Inference only."""

    def __init__(self, num_classes: int = 12, eps: float = 1e-5):
        super().__init__(num_classes)
        self.plain = DepthwiseSeparableConv(num_classes)
        self.bn_conv1 = nn.BatchNorm2d(64, eps=eps)
        self.bn_dw = nn.ModuleList([nn.BatchNorm2d(64, eps=eps) for _ in range(4)])
        self.bn_pw = nn.ModuleList([nn.BatchNorm2d(64, eps=eps) for _ in range(4)])
        object.__setattr__(self, "_folded", None)  # kept out of nn.Module's child registry: not a parameter holder
        self._folded_key = None

    @staticmethod
    def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        w = conv.weight.detach() * scale.reshape(-1, 1, 1, 1)
        b = (conv.bias.detach() - bn.running_mean) * scale + bn.bias.detach()
        return w, b

    def fold(self) -> DepthwiseSeparableConv:
        """The equivalent plain model (cached until a parameter or running statistic changes)."""
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if self._folded is not None and key == self._folded_key:
            return self._folded
        out = DepthwiseSeparableConv(self.num_classes)
        with torch.no_grad():
            w, b = self._fold(self.plain.conv1, self.bn_conv1)
            out.conv1.weight.copy_(w)
            out.conv1.bias.copy_(b)
            for i in range(4):
                src = getattr(self.plain, f"dsconv{i + 1}")
                dst = getattr(out, f"dsconv{i + 1}")
                w, b = self._fold(src.depthwise, self.bn_dw[i])
                dst.depthwise.weight.copy_(w)
                dst.depthwise.bias.copy_(b)
                w, b = self._fold(src.pointwise, self.bn_pw[i])
                dst.pointwise.weight.copy_(w)
                dst.pointwise.bias.copy_(b)
            out.fc.weight.copy_(self.plain.fc.weight)
            out.fc.bias.copy_(self.plain.fc.bias)
        object.__setattr__(self, "_folded", out)
        self._folded_key = key
        return out

    def sync_weights(self) -> None:
        """Drop the cached fold (needed only after edits through ``.data``, which do not bump ``_version``)."""
        self._folded_key = None

    def forward(self, x: torch.Tensor, return_labels: bool = False):
        return self.fold().forward(x, return_labels)

    def infer_pcm16(self, wav: torch.Tensor):
        return self.fold().infer_pcm16(wav)


class CnnTradFpool3(KeywordSpottingModel):
    """Build-defined model-zoo member: Sainath & Parada's cnn-trad-fpool3 on the reference's
    ``[1,99,10]`` MFCC map with SAME padding -- conv 64x(20x8)+ReLU, max-pool 1x3 over frequency, conv 64x(10x4)+ReLU,
    flatten, Linear 32, Linear 128+ReLU, Linear C.  The modules hold parameters; ``forward`` is
    ``kws_forward_cnn_trad_f32`` (both convolutions as implicit GEMMs on the bf16 matrix pipe with the exact
    three-way split, the dense tail batched on the VALU).  Inference only."""

    def __init__(self, num_classes: int = 12):
        super().__init__(num_classes)
        self.conv1 = nn.Conv2d(1, 64, kernel_size=(20, 8))
        self.conv2 = nn.Conv2d(64, 64, kernel_size=(10, 4))
        self.lin = nn.Linear(64 * 99 * 3, 32)
        self.dnn = nn.Linear(32, 128)
        self.fc = nn.Linear(128, num_classes)
        self._ctx = None
        self._uploaded = None

    def packed_weights(self) -> np.ndarray:
        return np.concatenate([v.detach().to("cpu", torch.float32).reshape(-1).numpy() for v in self.state_dict().values()])

    def sync_weights(self) -> None:
        """Force a re-upload at the next forward (needed only after edits through ``p.data``, which do not bump
        ``p._version``; see ``DepthwiseSeparableConv.sync_weights``)."""
        self._uploaded = None

    def _context(self, device_index: int):
        from kws import _native

        if self._ctx is None or self._ctx.device != device_index:
            self._ctx = _native.Context(device_index, ModelError)
            self._uploaded = None
        fp = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if fp != self._uploaded:
            self._ctx.load_cnn_trad(self.packed_weights(), self.num_classes)
            self._uploaded = fp
        self._ctx.use_torch_stream()
        return self._ctx

    def forward(self, x: torch.Tensor, return_labels: bool = False):
        if not x.is_cuda:
            raise ModelError("CnnTradFpool3.forward needs a CUDA/ROCm tensor: the forward is a HIP kernel and has no CPU fallback")
        if x.dim() != 4 or tuple(x.shape[1:]) != FEATURE_SHAPE:
            raise ModelError(f"expected input [B,1,99,10], got {tuple(x.shape)}")
        ctx = self._context(x.device.index or 0)
        x = x.detach().to(torch.float32).contiguous()
        logits = torch.empty((x.shape[0], self.num_classes), dtype=torch.float32, device=x.device)
        labels = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
        ctx.forward_cnn_trad_f32(x, logits, labels)
        return (logits, labels) if return_labels else logits

    def infer_pcm16(self, wav: torch.Tensor):
        """Fused path (BASELINE.json configs[2]): ``int16[B,16000]`` PCM on the GPU -> (logits, labels); MFCC +
        cnn-trad-fpool3 back to back on one stream (``kws_infer_cnn_trad_i16``)."""
        if not wav.is_cuda:
            raise ModelError("CnnTradFpool3.infer_pcm16 needs a CUDA/ROCm tensor: the path is HIP kernels and has no CPU fallback")
        if wav.dtype != torch.int16 or wav.dim() != 2:
            raise ModelError("infer_pcm16 expects an int16 tensor [B, n_samples]")
        ctx = self._context(wav.device.index or 0)
        wav = wav.contiguous()
        logits = torch.empty((wav.shape[0], self.num_classes), dtype=torch.float32, device=wav.device)
        labels = torch.empty((wav.shape[0],), dtype=torch.int32, device=wav.device)
        ctx.infer_cnn_trad_i16(wav, logits, labels)
        return logits, labels
