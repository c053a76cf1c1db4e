"""Shared test plumbing.

* ``gpu`` marker: tests that need an MI355X (run by the driver with ``-m gpu``).
* ``keyword-spotting_amd/`` is put on ``sys.path`` so the drop-in package
  imports under the reference's own name (``import kws``).
* Nothing here (or in any test) reads /root/reference at run time.
"""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO, "keyword-spotting_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG_ROOT, GOLDEN):  # GOLDEN: the speech-like clip generator shared with make_golden.py
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels are launched)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def sigproc_golden():
    return np.load(os.path.join(GOLDEN, "sigproc_golden.npz"))


@pytest.fixture(scope="session")
def dscnn_golden():
    return np.load(os.path.join(GOLDEN, "dscnn_golden.npz"))


@pytest.fixture(scope="session")
def e2e_golden():
    """Diverse clips + signal-preserving weights -> logits / labels of the imported reference model (make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "e2e_golden.npz"))


@pytest.fixture(scope="session")
def stress_golden():
    """16 speech-like clips + three weight tags (he, he5: 5x classifier gain, raw: conv1 not pre-divided by the MFCC maps' RMS)
    -> logits / labels of the imported reference model for the 48 diverse clips of e2e_golden followed by the 16 (make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "stress_golden.npz"))


@pytest.fixture(scope="session")
def anymap_golden():
    """DepthwiseSeparableConv on maps other than 99 x 10 (149 x 10, 99 x 13, 20 x 8, 3-channel 50 x 12): inputs, weights, logits,
    labels and stage outputs of the imported reference model (make_golden.py anymap)."""
    return np.load(os.path.join(GOLDEN, "anymap_golden.npz"))


@pytest.fixture(scope="session")
def dsblock_golden():
    """DepthwiseSeparableConvBlock on its own: inputs, parameters and the imported reference module's outputs."""
    return np.load(os.path.join(GOLDEN, "dsblock_golden.npz"))


def synth_clips(batch: int, seed: int = 0, kind: str = "uniform") -> np.ndarray:
    """Synthetic int16 [batch,16000] clips of SURVEY.md section 8(d)."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(-32768, 32768, size=(batch, 16000), dtype=np.int16)
    if kind == "gauss":
        x = np.clip(np.round(rng.standard_normal((batch, 16000)) * 3000.0), -32768, 32767).astype(np.int16)
        x[::16] = 0  # 1/16 of the clips all-zero (silence class path)
        return x
    raise ValueError(kind)
