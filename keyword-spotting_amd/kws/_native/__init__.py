"""ctypes binding of libkws_hip.so (C ABI: include/kws_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C keyword-spotting_amd/csrc``; if it is missing, importing this module still works (so the
host-only classes are usable) but the first compute call raises ``KWSError`` -- loudly, never a
silent fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from kws.common.errors import AudioProcessingError, KWSError, ModelError

# KWS_HIP_LIB selects another build of the same ABI (A/B timing of kernel variants); default is the in-tree library
LIB_PATH = os.environ.get("KWS_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkws_hip.so")

KWS_OK, KWS_EINVAL, KWS_ENOMEM, KWS_EHIP, KWS_ESTATE, KWS_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
KWS_CT_F16_PAIR, KWS_CT_BF16_TRIPLE = 0, 1  # kws_set_cnn_trad_math
KWS_K_MFCC, KWS_K_DSCNN, KWS_K_CNNTRAD_CONV, KWS_K_CNNTRAD_DENSE, KWS_K_STREAM_FRAME, KWS_K_MFCC_F64, KWS_K_MFCC_REFINE = 0, 1, 2, 3, 4, 5, 6
FE_REFINE_SPAN_DEFAULT = 10.2  # KWS_FE_REFINE_SPAN_DEFAULT: log(largest bin power / weakest mel band) beyond which a frame is redone in float64
FE_F32, FE_F64 = 0, 1  # KWS_FE_F32 (default: the fast float32 front end) / KWS_FE_F64 (float64 after framing, as psf)
ACT_FLOATS_PER_CLIP = 64 * (141 + 141 + 245 + 357) + 64 + 64 * 477  # KWS_ACT_FLOATS_PER_CLIP
PW_F32 = 1          # KWS_PW_F32: pointwise convolutions on v_mfma_f32_32x32x2_f32
PW_SPLIT_BF16 = 4   # KWS_PW_SPLIT_BF16: exact three-way bf16 split, six bf16 MFMAs per f32 product
PW_PAIR_F16 = 5     # KWS_PW_PAIR_F16 (default): f16 pairs with per-clip power-of-two scales, three f16 MFMAs per f32 product
PW_DEFAULT = PW_PAIR_F16

_c_ctx = C.c_void_p
_i16p, _f32p, _i32p = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers

# name -> (restype, argtypes): every symbol include/kws_hip.h declares
SIGNATURES = {
    "kws_abi_version": (C.c_int, []),
    "kws_create": (C.c_int, [C.POINTER(_c_ctx), C.c_int]),
    "kws_destroy": (None, [_c_ctx]),
    "kws_set_stream": (C.c_int, [_c_ctx, C.c_void_p, C.c_int]),
    "kws_sync": (C.c_int, [_c_ctx]),
    "kws_last_error": (C.c_char_p, [_c_ctx]),
    "kws_set_frontend": (C.c_int, [_c_ctx] + [C.c_int] * 7 + [C.c_float, C.c_int]),
    "kws_set_frontend_math": (C.c_int, [_c_ctx, C.c_int]),
    "kws_frontend_math": (C.c_int, [_c_ctx]),
    "kws_set_frontend_refine": (C.c_int, [_c_ctx, C.c_float]),
    "kws_frontend_stats": (C.c_int, [_c_ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "kws_spec_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "kws_frontend_shape": (C.c_int, [_c_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kws_mfcc_i16": (C.c_int, [_c_ctx, _i16p, C.c_int, _f32p]),
    "kws_mfcc_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p]),
    "kws_load_dscnn": (C.c_int, [_c_ctx, C.POINTER(C.c_float), C.c_size_t, C.c_int]),
    "kws_load_dscnn_ex": (C.c_int, [_c_ctx, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_int]),
    "kws_forward_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p, _i32p]),
    "kws_forward_map_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p]),
    "kws_forward_map_debug_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _f32p]),
    "kws_infer_i16": (C.c_int, [_c_ctx, _i16p, C.c_int, _f32p, _i32p]),
    "kws_dsblock_forward_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, _f32p]),
    "kws_infer_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p, _i32p]),
    "kws_infer_host_i16": (C.c_int, [_c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kws_infer_host_submit_i16": (C.c_int, [_c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "kws_infer_host_wait": (C.c_int, [_c_ctx, C.c_uint64]),
    "kws_ingest_config": (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_int]),
    "kws_reserve": (C.c_int, [_c_ctx, C.c_int]),
    "kws_set_pointwise_math": (C.c_int, [_c_ctx, C.c_int]),
    "kws_forward_debug_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p, _i32p, _f32p, C.c_int]),
    "kws_stream_open": (C.c_int, [_c_ctx, C.c_int]),
    "kws_stream_close": (C.c_int, [_c_ctx]),
    "kws_stream_cluster": (C.c_int, [_c_ctx, C.c_int]),
    "kws_stream_host_results": (C.c_int, [_c_ctx, C.c_int]),
    "kws_stream_wait_host": (C.c_int, [_c_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "kws_stream_push_host_i16": (C.c_int, [_c_ctx, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "kws_stream_push_i16": (C.c_int, [_c_ctx, _i16p, _f32p, _i32p, C.c_int]),
    "kws_stream_state": (C.c_int, [_c_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "kws_stream_copy_features": (C.c_int, [_c_ctx, _f32p]),
    "kws_load_cnn_trad": (C.c_int, [_c_ctx, C.POINTER(C.c_float), C.c_size_t, C.c_int]),
    "kws_stream_vad_f32": (C.c_int, [_c_ctx, C.c_float, C.c_int, C.c_int, _i32p]),
    "kws_forward_cnn_trad_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p, _i32p]),
    "kws_set_cnn_trad_math": (C.c_int, [_c_ctx, C.c_int]),
    "kws_infer_cnn_trad_i16": (C.c_int, [_c_ctx, _i16p, C.c_int, _f32p, _i32p]),
    "kws_softmax_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, _f32p]),
    "kws_stream_smooth_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, _f32p, _i32p]),
    "kws_augment_i16": (C.c_int, [_c_ctx, _i16p, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_void_p, _f32p, C.c_void_p, _f32p]),
    "kws_forward_stamps_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, _f32p, C.c_void_p, C.c_int]),
    "kws_preemphasis_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_float, _f32p]),
    "kws_framesig_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p]),
    "kws_spec512_f32": (C.c_int, [_c_ctx, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "kws_prof_enable": (C.c_int, [_c_ctx, C.c_int]),
    "kws_prof_reset": (C.c_int, [_c_ctx]),
    "kws_prof_read": (C.c_int, [_c_ctx, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "kws_kernel_name": (C.c_char_p, [C.c_int]),
    "kws_host_mel_edges": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "kws_host_mel_dense": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "kws_host_mel_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "kws_host_dct_lifter": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load libkws_hip.so once; raise KWSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise KWSError(
                    f"native library missing: {LIB_PATH} (build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C keyword-spotting_amd/csrc`); there is no CPU fallback"
                )
            try:
                h = C.CDLL(LIB_PATH)
            except OSError as e:  # pragma: no cover - depends on the host
                raise KWSError(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(h, name)
                fn.restype = res
                fn.argtypes = args
            _lib = h
    return _lib


def _ptr(t) -> int:
    return int(t.data_ptr())


class Context:
    """One kws_ctx: a GPU, its stream, front-end tables, model weights, workspace."""

    def __init__(self, device: int = 0, error_cls=KWSError):
        self._h = _c_ctx()
        self._lib = lib()
        rc = self._lib.kws_create(C.byref(self._h), int(device))
        if rc != KWS_OK:
            msg = self._lib.kws_last_error(None).decode()
            self._h = _c_ctx()
            raise error_cls(f"kws_create failed ({rc}): {msg}")
        self.device = int(device)
        self.num_classes = None

    # -- plumbing -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.kws_destroy(self._h)
            self._h = _c_ctx()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, error_cls=KWSError):
        if rc != KWS_OK:
            raise error_cls(f"{self._lib.kws_last_error(self._h).decode()} (code {rc})")

    def use_torch_stream(self):
        """Enqueue on torch's current stream for this device, so torch sees the work in order."""
        import torch

        self._check(self._lib.kws_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), 1))

    def use_own_stream(self):
        self._check(self._lib.kws_set_stream(self._h, None, 0))

    def sync(self):
        self._check(self._lib.kws_sync(self._h))

    # -- front end ------------------------------------------------------------------------------
    def set_frontend(self, sample_rate=16000, n_samples=16000, frame_len=400, frame_step=160, nfft=512, nfilt=26,
                     numcep=10, preemph=0.97, ceplifter=22):
        self._check(
            self._lib.kws_set_frontend(self._h, sample_rate, n_samples, frame_len, frame_step, nfft, nfilt, numcep,
                                       float(preemph), ceplifter),
            AudioProcessingError,
        )

    def set_frontend_math(self, math: int):
        self._check(self._lib.kws_set_frontend_math(self._h, int(math)), AudioProcessingError)

    def frontend_math(self) -> int:
        """FE_F32 or FE_F64: the arithmetic kws_mfcc_* actually uses for the configured geometry."""
        return int(self._lib.kws_frontend_math(self._h))

    def set_frontend_refine(self, log_span: float = FE_REFINE_SPAN_DEFAULT):
        """Frames whose log-mel values span more than ``log_span`` are recomputed in float64 (0 switches it off)."""
        self._check(self._lib.kws_set_frontend_refine(self._h, float(log_span)), AudioProcessingError)

    def frontend_stats(self):
        """(frames through the float32 front end, frames recomputed in float64, frames the last batched call listed)."""
        total, refined, last = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self._lib.kws_frontend_stats(self._h, C.byref(total), C.byref(refined), C.byref(last)), AudioProcessingError)
        return int(total.value), int(refined.value), int(last.value)

    def frontend_shape(self):
        nf, nc = C.c_int(), C.c_int()
        self._check(self._lib.kws_frontend_shape(self._h, C.byref(nf), C.byref(nc)), AudioProcessingError)
        return nf.value, nc.value

    def mfcc_i16(self, wav, out):
        self._check(self._lib.kws_mfcc_i16(self._h, _ptr(wav), int(wav.shape[0]), _ptr(out)), AudioProcessingError)

    def mfcc_f32(self, wav, out):
        self._check(self._lib.kws_mfcc_f32(self._h, _ptr(wav), int(wav.shape[0]), _ptr(out)), AudioProcessingError)

    # -- model ----------------------------------------------------------------------------------
    def load_dscnn(self, blob: np.ndarray, num_classes: int, input_channels: int = 1):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(
            self._lib.kws_load_dscnn_ex(self._h, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, int(num_classes),
                                        int(input_channels)),
            ModelError,
        )
        self.num_classes = int(num_classes)

    def forward_f32(self, feat, logits, label=None):
        self._check(
            self._lib.kws_forward_f32(self._h, _ptr(feat), int(feat.shape[0]), _ptr(logits), _ptr(label) if label is not None else None),
            ModelError,
        )

    def forward_map_f32(self, feat, logits, label=None, layers=None):
        """``feat`` float32 [B, C, T, F] of any size (99 x 10: the fused kernel; otherwise conv1 -> blocks -> pool + fc through
        HBM).  ``layers`` (diagnostics): every stage's output, stage after stage (``kws_forward_map_debug_f32``)."""
        B, _, T, F = (int(v) for v in feat.shape)
        lab = _ptr(label) if label is not None else None
        if layers is None:
            self._check(self._lib.kws_forward_map_f32(self._h, _ptr(feat), B, T, F, _ptr(logits), lab), ModelError)
        else:
            self._check(self._lib.kws_forward_map_debug_f32(self._h, _ptr(feat), B, T, F, _ptr(logits), lab, _ptr(layers)), ModelError)

    def load_cnn_trad(self, blob: np.ndarray, num_classes: int):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(self._lib.kws_load_cnn_trad(self._h, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, int(num_classes)),
                    ModelError)

    def set_cnn_trad_math(self, math: int):
        """KWS_CT_F16_PAIR (0, default) or KWS_CT_BF16_TRIPLE (1): arithmetic of cnn-trad-fpool3's GEMM layers."""
        self._check(self._lib.kws_set_cnn_trad_math(self._h, int(math)), ModelError)

    def forward_cnn_trad_f32(self, feat, logits, label=None):
        self._check(self._lib.kws_forward_cnn_trad_f32(self._h, _ptr(feat), int(feat.shape[0]), _ptr(logits),
                                                       _ptr(label) if label is not None else None), ModelError)

    def infer_cnn_trad_i16(self, wav, logits, label=None):
        self._check(self._lib.kws_infer_cnn_trad_i16(self._h, _ptr(wav), int(wav.shape[0]), _ptr(logits),
                                                     _ptr(label) if label is not None else None), ModelError)

    def softmax_f32(self, logits, prob):
        self._check(self._lib.kws_softmax_f32(self._h, _ptr(logits), int(logits.shape[0]), int(logits.shape[1]), _ptr(prob)),
                    ModelError)

    def stream_smooth_f32(self, logits, window, smoothed, label=None):
        self._check(self._lib.kws_stream_smooth_f32(self._h, _ptr(logits), int(logits.shape[1]), int(window), _ptr(smoothed),
                                                    _ptr(label) if label is not None else None), ModelError)

    def stream_vad_f32(self, threshold, on_window, off_window, state):
        self._check(self._lib.kws_stream_vad_f32(self._h, float(threshold), int(on_window), int(off_window), _ptr(state)), ModelError)

    def set_pointwise_math(self, math):
        self._check(self._lib.kws_set_pointwise_math(self._h, int(math)), ModelError)

    def forward_debug_f32(self, feat, logits, label, act, use_mfma=PW_SPLIT_BF16):
        self._check(
            self._lib.kws_forward_debug_f32(self._h, _ptr(feat), int(feat.shape[0]), _ptr(logits),
                                            _ptr(label) if label is not None else None,
                                            _ptr(act) if act is not None else None, int(use_mfma)),
            ModelError,
        )

    def forward_stamps_f32(self, feat, logits, stamps, mode=PW_SPLIT_BF16):
        self._check(self._lib.kws_forward_stamps_f32(self._h, _ptr(feat), int(feat.shape[0]), _ptr(logits), _ptr(stamps), int(mode)), ModelError)

    def infer_i16(self, wav, logits, label=None):
        self._check(
            self._lib.kws_infer_i16(self._h, _ptr(wav), int(wav.shape[0]), _ptr(logits), _ptr(label) if label is not None else None),
            ModelError,
        )

    def dsblock_forward_f32(self, x, dw_w, dw_b, pw_w, pw_b, kernel_size, stride, padding, out):
        B, c_in, H, W = (int(v) for v in x.shape)
        self._check(self._lib.kws_dsblock_forward_f32(self._h, _ptr(x), B, c_in, H, W, _ptr(dw_w), _ptr(dw_b), _ptr(pw_w), _ptr(pw_b),
                                                      int(pw_w.shape[0]), int(kernel_size), int(stride), int(padding), _ptr(out)), ModelError)

    def infer_host_i16(self, wav: np.ndarray, logits: np.ndarray = None, label: np.ndarray = None):
        """Host ``int16[B,n]`` (numpy array or CPU torch tensor, pageable or pinned) -> host (logits float32[B,C], labels
        int32[B]) through the library's pack / H2D / compute / D2H pipeline (``kws_infer_host_i16``)."""
        logits, label, ticket, _keep = self.infer_host_submit_i16(wav, logits, label)
        self.infer_host_wait(ticket)
        return logits, label

    def infer_host_wait(self, ticket: int = 0):
        """Block until every submitted batch up to ``ticket`` (0: all) has its results in its host arrays."""
        self._check(self._lib.kws_infer_host_wait(self._h, C.c_uint64(int(ticket))), ModelError)

    def infer_host_submit_i16(self, wav: np.ndarray, logits: np.ndarray = None, label: np.ndarray = None):
        """Enqueue one host batch without waiting for its results (``kws_infer_host_submit_i16``).  Returns (logits, label,
        ticket, keepalive): the arrays are filled once ``infer_host_wait(ticket)`` has returned; keep ``keepalive`` (the
        source buffer -- a pinned one is read by the DMA until then) referenced until that wait."""
        if self.num_classes is None:
            raise ModelError("no model loaded (kws_load_dscnn)")
        if hasattr(wav, "data_ptr"):  # CPU torch tensor (possibly pinned)
            if wav.is_cuda or wav.dtype.__str__() != "torch.int16" or not wav.is_contiguous() or wav.dim() != 2:
                raise ModelError("infer_host_i16 expects a contiguous CPU int16 tensor [B, n_samples]")
            src, B = int(wav.data_ptr()), int(wav.shape[0])
        else:
            wav = np.ascontiguousarray(wav, dtype=np.int16)
            if wav.ndim != 2:
                raise ModelError("infer_host_i16 expects int16 [B, n_samples]")
            src, B = wav.ctypes.data, int(wav.shape[0])
        if logits is None:
            logits = np.empty((B, self.num_classes), np.float32)
        if label is None:
            label = np.empty((B,), np.int32)
        assert logits.dtype == np.float32 and logits.flags.c_contiguous and logits.shape == (B, self.num_classes)
        assert label.dtype == np.int32 and label.flags.c_contiguous and label.shape == (B,)
        ticket = C.c_uint64(0)
        self._check(self._lib.kws_infer_host_submit_i16(self._h, C.c_void_p(src), B, C.c_void_p(logits.ctypes.data),
                                                        C.c_void_p(label.ctypes.data), C.byref(ticket)), ModelError)
        return logits, label, int(ticket.value), wav

    def ingest_config(self, chunk_clips: int = 0, n_slots: int = 0, pack_threads: int = 0):
        self._check(self._lib.kws_ingest_config(self._h, int(chunk_clips), int(n_slots), int(pack_threads)), ModelError)

    def infer_f32(self, wav, logits, label=None):
        self._check(
            self._lib.kws_infer_f32(self._h, _ptr(wav), int(wav.shape[0]), _ptr(logits), _ptr(label) if label is not None else None),
            ModelError,
        )

    # -- streaming -------------------------------------------------------------------------------
    def stream_open(self, n_streams: int):
        self._check(self._lib.kws_stream_open(self._h, int(n_streams)), AudioProcessingError)

    def stream_cluster(self, workgroups_per_stream: int = 0):
        """Workgroups per stream of the one-launch push: 1, or 2 / 4 time tiles; 0 = by stream count (``kws_stream_cluster``)."""
        self._check(self._lib.kws_stream_cluster(self._h, int(workgroups_per_stream)), ModelError)

    def stream_host_results(self, enable: bool = True):
        """Zero-copy delivery of the one-launch push's logits / labels to pinned host memory (``kws_stream_host_results``)."""
        self._check(self._lib.kws_stream_host_results(self._h, 1 if enable else 0), ModelError)
        self._host_views = None

    def stream_push_host_i16(self, hop: np.ndarray, n_streams: int):
        """Host int16 [S, hop] -> (logits, labels) host views in one call (``kws_stream_push_host_i16``): no H2D submission,
        no synchronise, no D2H copy."""
        if hop.dtype != np.int16 or not hop.flags.c_contiguous:
            raise ModelError("stream_push_host_i16 expects a C-contiguous int16 array")
        pl, py = C.c_void_p(), C.c_void_p()
        self._check(self._lib.kws_stream_push_host_i16(self._h, hop.ctypes.data, C.byref(pl), C.byref(py)), ModelError)
        return self._host_views_of(pl, py, n_streams)

    def stream_wait_host(self, n_streams: int):
        """Spin until the newest push's results are in host memory; returns numpy VIEWS (logits float32[S,C], labels int32[S])
        of the context's pinned arrays -- valid until the next push."""
        pl, py = C.c_void_p(), C.c_void_p()
        self._check(self._lib.kws_stream_wait_host(self._h, C.byref(pl), C.byref(py)), ModelError)
        return self._host_views_of(pl, py, n_streams)

    def _host_views_of(self, pl, py, n_streams):
        if getattr(self, "_host_views", None) is None or self._host_views[0] != (pl.value, py.value, n_streams):
            lg = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_float)), shape=(n_streams, self.num_classes))
            lb = np.ctypeslib.as_array(C.cast(py, C.POINTER(C.c_int32)), shape=(n_streams,))
            self._host_views = ((pl.value, py.value, n_streams), lg, lb)
        return self._host_views[1], self._host_views[2]

    def stream_close(self):
        self._check(self._lib.kws_stream_close(self._h))

    def stream_push_i16(self, hop, logits=None, label=None, use_graph=False):
        self._check(
            self._lib.kws_stream_push_i16(self._h, _ptr(hop), _ptr(logits) if logits is not None else None,
                                          _ptr(label) if label is not None else None, 1 if use_graph else 0),
            ModelError,
        )

    def stream_state(self):
        """(device address of the feature ring, pushes so far)."""
        ring, hops = C.c_void_p(), C.c_int()
        self._check(self._lib.kws_stream_state(self._h, C.byref(ring), C.byref(hops)))
        return ring.value, hops.value

    def stream_copy_features(self, out):
        self._check(self._lib.kws_stream_copy_features(self._h, _ptr(out)))

    # -- augmentation ----------------------------------------------------------------------------
    def augment_i16(self, wav, out, shift=None, bg=None, bg_off=None, bg_vol=None, silence=None):
        p = lambda t: _ptr(t) if t is not None else None
        self._check(
            self._lib.kws_augment_i16(self._h, _ptr(wav), int(wav.shape[0]), p(shift), p(bg), int(bg.numel()) if bg is not None else 0,
                                      p(bg_off), p(bg_vol), p(silence), _ptr(out)),
            AudioProcessingError,
        )

    def reserve(self, max_batch: int):
        self._check(self._lib.kws_reserve(self._h, int(max_batch)))

    # -- sigproc operators ----------------------------------------------------------------------
    def preemphasis_f32(self, sig, coeff, out):
        self._check(self._lib.kws_preemphasis_f32(self._h, _ptr(sig), int(sig.numel()), float(coeff), _ptr(out)), AudioProcessingError)

    def framesig_f32(self, sig, frame_len, frame_step, window, frames):
        self._check(
            self._lib.kws_framesig_f32(self._h, _ptr(sig), int(sig.numel()), int(frame_len), int(frame_step),
                                       _ptr(window) if window is not None else None, _ptr(frames)),
            AudioProcessingError,
        )

    def spec_f32(self, frames, nfft, power, spec):
        self._check(
            self._lib.kws_spec_f32(self._h, _ptr(frames), int(frames.shape[0]), int(frames.shape[1]), int(nfft), 1 if power else 0, _ptr(spec)),
            AudioProcessingError,
        )

    def spec512_f32(self, frames, power, spec):
        self._check(
            self._lib.kws_spec512_f32(self._h, _ptr(frames), int(frames.shape[0]), int(frames.shape[1]), 1 if power else 0, _ptr(spec)),
            AudioProcessingError,
        )

    # -- measurement ----------------------------------------------------------------------------
    def prof_enable(self, on=True):
        """False/0 = off, True/1 = time every kernel launch, n > 1 = time every n-th launch of each kernel."""
        self._check(self._lib.kws_prof_enable(self._h, int(on)))

    def prof_reset(self):
        self._check(self._lib.kws_prof_reset(self._h))

    def prof_read(self, kernel_id: int):
        ms, n = C.c_double(), C.c_int()
        self._check(self._lib.kws_prof_read(self._h, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def kernel_name(kernel_id: int) -> str:
    return lib().kws_kernel_name(int(kernel_id)).decode()


# -- host-only helpers (no GPU) --------------------------------------------------------------------
def host_mel_edges(nfilt=26, nfft=512, sample_rate=16000) -> np.ndarray:
    out = (C.c_int * (nfilt + 2))()
    rc = lib().kws_host_mel_edges(nfilt, nfft, sample_rate, out)
    if rc != KWS_OK:
        raise KWSError(f"kws_host_mel_edges failed ({rc})")
    return np.array(out[:], dtype=np.int64)


def host_mel_dense(nfilt=26, nfft=512, sample_rate=16000) -> np.ndarray:
    out = np.zeros((nfilt, nfft // 2 + 1), np.float32)
    rc = lib().kws_host_mel_dense(nfilt, nfft, sample_rate, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != KWS_OK:
        raise KWSError(f"kws_host_mel_dense failed ({rc})")
    return out


def host_mel_layout(nfilt=26, nfft=512, sample_rate=16000):
    """(first_lane[nfilt+1], n_lanes[nfilt+1], lanes_used, row_safe) of the sparse mel evaluation."""
    first = (C.c_int * (nfilt + 1))()
    count = (C.c_int * (nfilt + 1))()
    used, safe = C.c_int(0), C.c_int(0)
    rc = lib().kws_host_mel_layout(nfilt, nfft, sample_rate, first, count, C.byref(used), C.byref(safe))
    if rc != KWS_OK:
        raise KWSError(f"kws_host_mel_layout failed ({rc})")
    return np.array(first[:], dtype=np.int64), np.array(count[:], dtype=np.int64), int(used.value), bool(safe.value)


def host_dct_lifter(nfilt=26, numcep=10, ceplifter=22) -> np.ndarray:
    out = np.zeros((numcep, nfilt), np.float32)
    rc = lib().kws_host_dct_lifter(nfilt, numcep, ceplifter, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != KWS_OK:
        raise KWSError(f"kws_host_dct_lifter failed ({rc})")
    return out


_contexts: dict = {}


def default_context(device: int = 0) -> Context:
    """Process-wide context per GPU, bound to torch's current stream at each use by the callers."""
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx
