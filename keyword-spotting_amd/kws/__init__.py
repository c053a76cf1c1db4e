"""MI355X-native drop-in for the inference hot path of z430/keyword-spotting.

Same import names as the reference (``kws.libs.audio_processor``, ``kws.libs.speech_features.sigproc``,
``kws.libs.models``, ``kws.libs.data_loader``, ``kws.inference``, ``kws.common.errors``); the arithmetic
runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/kws_hip.h``
(``kws/_native/libkws_hip.so``, loaded with ctypes).  PyTorch-ROCm tensors are used as device buffers
only.  There is no CPU fallback: without the library or a GPU the compute entry points raise.
"""

__all__ = ["__version__"]
__version__ = "0.1.0"
