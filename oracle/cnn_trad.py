"""CPU definition of the build's cnn-trad-fpool3 (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Not a restatement of reference code: the reference only names this model (``test.py:80``, a checkpoint file name);
SURVEY.md section 8 f-4 makes it a build-defined member of the model zoo, with parity against the build's own
torch-CPU definition -- this file.  PARITY UNPINNED against the reference by construction.

Architecture: Sainath & Parada, "Convolutional Neural Networks for Small-footprint Keyword Spotting" (2015),
model cnn-trad-fpool3, on the reference's ``[1, 99, 10]`` MFCC map with SAME padding (TensorFlow's speech_commands
convention: total padding k-1, the extra row/column at the bottom/right):
    conv 64 x (20 time x 8 freq) + ReLU -> max-pool 1 x 3 over frequency (stride 3, floor) ->
    conv 64 x (10 x 4) + ReLU -> flatten (channel-major) -> Linear 32 -> Linear 128 + ReLU -> Linear C
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

FLAT = 64 * 99 * 3


def state_shapes(num_classes: int = 12) -> "OrderedDict[str, tuple]":
    return OrderedDict([
        ("conv1.weight", (64, 1, 20, 8)), ("conv1.bias", (64,)),
        ("conv2.weight", (64, 64, 10, 4)), ("conv2.bias", (64,)),
        ("lin.weight", (32, FLAT)), ("lin.bias", (32,)),
        ("dnn.weight", (128, 32)), ("dnn.bias", (128,)),
        ("fc.weight", (num_classes, 128)), ("fc.bias", (num_classes,)),
    ])


def random_state(seed: int, num_classes: int = 12) -> "OrderedDict[str, torch.Tensor]":
    """Fan-in scaled normal weights (so activations stay O(1) through the 2560- and 19008-wide sums), N(0, 0.1) biases."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in state_shapes(num_classes).items():
        if k.endswith("bias"):
            out[k] = torch.randn(shp, generator=g) * 0.1
        else:
            fan_in = int(np.prod(shp[1:]))
            out[k] = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
    return out


def forward(state: dict, x: torch.Tensor, return_layers: bool = False):
    """``x [B,1,99,10]`` float32 -> logits ``[B,C]``."""
    y1 = F.relu(F.conv2d(F.pad(x, (3, 4, 9, 10)), state["conv1.weight"], state["conv1.bias"]))
    yp = F.max_pool2d(y1, kernel_size=(1, 3), stride=(1, 3))
    y2 = F.relu(F.conv2d(F.pad(yp, (1, 2, 4, 5)), state["conv2.weight"], state["conv2.bias"]))
    h = F.linear(y2.flatten(1), state["lin.weight"], state["lin.bias"])
    d = F.relu(F.linear(h, state["dnn.weight"], state["dnn.bias"]))
    logits = F.linear(d, state["fc.weight"], state["fc.bias"])
    if return_layers:
        return logits, {"pool": yp, "conv2": y2, "lin": h, "dnn": d}
    return logits


def flatten_state(state: dict) -> np.ndarray:
    return np.concatenate([state[k].detach().to(torch.float32).reshape(-1).numpy() for k in state_shapes(state["fc.bias"].numel())])


def unflatten_state(blob: np.ndarray, num_classes: int = 12) -> "OrderedDict[str, torch.Tensor]":
    """Inverse of ``flatten_state``: the ten tensors from one float32 vector in state_dict order."""
    out, o = OrderedDict(), 0
    for k, shp in state_shapes(num_classes).items():
        n = int(np.prod(shp))
        out[k] = torch.from_numpy(np.asarray(blob[o:o + n], dtype=np.float32).reshape(shp).copy())
        o += n
    assert o == len(blob)
    return out
