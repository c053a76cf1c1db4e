"""CPU oracle: the reference's depthwise-separable CNN forward, restated functionally.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) -- never imported by the
product path.

Follows ``kws/libs/models.py``:
  * conv1   Conv2d(1->64, k=10, stride=2, pad=2) + ReLU              (:135, :170)
  * 4 x block: depthwise Conv2d(64,64,k=3,s=1,p=1,groups=64)         (:96-103, :117)
               pointwise Conv2d(64,64,k=1,s=1,**padding=1**) + ReLU  (:104-106, :118-119)
    (the 1x1 conv with padding=1 grows H and W by 2 per block; the new ring
     equals relu(bias) -- reference behaviour, reproduced on purpose)
  * adaptive_avg_pool2d -> (1,1), flatten, Linear(64 -> num_classes) (:179-181)
  * prediction = argmax over logits, first max wins (kws/libs/training.py:371)

Parameters are a plain ``dict`` with the reference's ``state_dict`` key names
(SURVEY.md section 8 a14), so a reference checkpoint loads unchanged.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

N_BLOCKS = 4
CHANNELS = 64

STATE_KEYS = (
    ["conv1.weight", "conv1.bias"]
    + [f"dsconv{i}.{part}.{wb}" for i in range(1, N_BLOCKS + 1) for part in ("depthwise", "pointwise") for wb in ("weight", "bias")]
    + ["fc.weight", "fc.bias"]
)


def state_shapes(num_classes: int = 12, input_channels: int = 1) -> "OrderedDict[str, tuple]":
    shapes = OrderedDict()
    shapes["conv1.weight"] = (CHANNELS, input_channels, 10, 10)
    shapes["conv1.bias"] = (CHANNELS,)
    for i in range(1, N_BLOCKS + 1):
        shapes[f"dsconv{i}.depthwise.weight"] = (CHANNELS, 1, 3, 3)
        shapes[f"dsconv{i}.depthwise.bias"] = (CHANNELS,)
        shapes[f"dsconv{i}.pointwise.weight"] = (CHANNELS, CHANNELS, 1, 1)
        shapes[f"dsconv{i}.pointwise.bias"] = (CHANNELS,)
    shapes["fc.weight"] = (num_classes, CHANNELS)
    shapes["fc.bias"] = (num_classes,)
    return shapes


def random_state(seed: int, std: float = 0.1, num_classes: int = 12) -> "OrderedDict[str, torch.Tensor]":
    """Every parameter (biases too) ~ N(0, std): non-zero biases expose the
    relu(bias) ring, which the reference's default init (zero biases) hides."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for k, shp in state_shapes(num_classes).items():
        out[k] = torch.from_numpy((rs.standard_normal(shp) * std).astype(np.float32))
    return out


def forward(state: dict, x: torch.Tensor, return_layers: bool = False):
    """x [B,1,T,F] -> logits [B,num_classes]; dtype follows ``x`` (params are cast)."""
    p = {k: v.to(dtype=x.dtype) for k, v in state.items()}
    layers = OrderedDict()
    h = F.relu(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], stride=2, padding=2))
    layers["conv1"] = h
    for i in range(1, N_BLOCKS + 1):
        h = F.conv2d(h, p[f"dsconv{i}.depthwise.weight"], p[f"dsconv{i}.depthwise.bias"], stride=1, padding=1, groups=CHANNELS)
        layers[f"dsconv{i}.depthwise"] = h
        h = F.relu(F.conv2d(h, p[f"dsconv{i}.pointwise.weight"], p[f"dsconv{i}.pointwise.bias"], stride=1, padding=1))
        layers[f"dsconv{i}"] = h
    h = F.adaptive_avg_pool2d(h, (1, 1)).reshape(h.shape[0], -1)
    layers["pool"] = h
    logits = F.linear(h, p["fc.weight"], p["fc.bias"])
    if return_layers:
        return logits, layers
    return logits


def block_forward(params: dict, x: torch.Tensor, kernel_size: int = 3, stride: int = 1, padding: int = 1) -> torch.Tensor:
    """``DepthwiseSeparableConvBlock.forward`` on its own (kws/libs/models.py:108-119): depthwise (groups = channels,
    ``kernel_size``/``stride``/``padding``) -> pointwise 1x1 with the SAME ``padding`` -> ReLU.  ``params`` uses the
    block's state_dict names (depthwise.weight, depthwise.bias, pointwise.weight, pointwise.bias)."""
    c_in = x.shape[1]
    h = F.conv2d(x, params["depthwise.weight"], params["depthwise.bias"], stride=stride, padding=padding, groups=c_in)
    return F.relu(F.conv2d(h, params["pointwise.weight"], params["pointwise.bias"], stride=1, padding=padding))


def predict(logits: torch.Tensor) -> torch.Tensor:
    """``torch.max(outputs, 1)`` indices (kws/libs/training.py:371)."""
    return torch.max(logits, 1)[1]


def flatten_state(state: dict) -> np.ndarray:
    """The 20 tensors in ``state_dict`` order as one float32 vector (the blob
    layout ``kws_load_dscnn`` takes)."""
    return np.concatenate([np.asarray(state[k], dtype=np.float32).reshape(-1) for k in STATE_KEYS if k in state])


def forward_bn(state: dict, bn: dict, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """CPU definition of the build's BatchNorm variant (not in the reference): the DS-CNN
    with an inference-mode BatchNorm after conv1, every depthwise and every pointwise convolution, applied
    UNFOLDED.  bn[name] = (gamma, beta, running_mean, running_var) for name in conv1, dw1..4, pw1..4."""
    import torch.nn.functional as F

    def norm(y, name):
        g, b, m, v = bn[name]
        return F.batch_norm(y, m, v, g, b, training=False, eps=eps)

    y = F.relu(norm(F.conv2d(x, state["conv1.weight"], state["conv1.bias"], stride=2, padding=2), "conv1"))
    for i in range(1, 5):
        y = norm(F.conv2d(y, state[f"dsconv{i}.depthwise.weight"], state[f"dsconv{i}.depthwise.bias"], padding=1, groups=64), f"dw{i}")
        y = F.relu(norm(F.conv2d(y, state[f"dsconv{i}.pointwise.weight"], state[f"dsconv{i}.pointwise.bias"], padding=1), f"pw{i}"))
    y = F.adaptive_avg_pool2d(y, (1, 1)).view(y.size(0), -1)
    return F.linear(y, state["fc.weight"], state["fc.bias"])
