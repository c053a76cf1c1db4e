"""wav -> label, the shape of the reference's ``inference(test_audio)``
(``kws/inference/inference_local.py:67-81``: load -> fix length to 1 s -> MFCC -> model -> argmax ->
word), on the fused MI355X path and for whole batches of files."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from kws.common.errors import ModelError
from kws.datasets.speech_commands import DEFAULT_WORDS, SpeechCommandDataset
from kws.libs.audio_processor import AudioConfig, fix_length, load_audio, load_pcm16
from kws.libs.models import DepthwiseSeparableConv

WANTED_WORDS = [SpeechCommandDataset.SILENCE_LABEL, SpeechCommandDataset.UNKNOWN_LABEL] + DEFAULT_WORDS


class KeywordSpotter:
    """Holds a model on one GPU and maps wav files / PCM batches to (index, word)."""

    def __init__(self, model: Optional[DepthwiseSeparableConv] = None, words: Sequence[str] = WANTED_WORDS,
                 config: Optional[AudioConfig] = None, device: int = 0):
        self.config = config or AudioConfig()
        self.words = list(words)
        self.model = model if model is not None else DepthwiseSeparableConv(num_classes=len(self.words))
        if self.model.num_classes != len(self.words):
            raise ModelError(f"model has {self.model.num_classes} classes but {len(self.words)} words were given")
        self.device = torch.device("cuda", device)

    def load_weights(self, path: str) -> None:
        self.model.load(path, device=torch.device("cpu"))  # parameters are packed from the host copy

    def infer_pcm16(self, pcm: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """``int16[B,n]`` host array -> (labels int32[B], logits float32[B,C]) through the library's host-ingest
        pipeline (``kws_infer_host_i16``: pack into pinned staging, H2D, MFCC + DS-CNN and D2H overlapped chunk by chunk)."""
        clips = fix_length(np.atleast_2d(np.asarray(pcm, dtype=np.int16)), self.config.desired_samples)
        ctx = self.model._context(self.device.index or 0)
        logits, labels = ctx.infer_host_i16(np.ascontiguousarray(clips))
        return labels, logits

    def infer_batches(self, batches: Iterable[np.ndarray], max_batch: Optional[int] = None
                      ) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        """Host ingest for many batches: every ``int16[B,n]`` host batch (numpy array, or a CPU
        torch tensor -- a pinned one is read by the DMA directly, without the pack stage) goes through the library's
        pipeline: a pool of host threads packs chunk k+1 into pinned staging while chunk k crosses PCIe on a copy stream,
        chunk k-1 runs MFCC + DS-CNN and the results of chunk k-2 return on a second copy stream.  The pipeline stays
        full ACROSS batches: batch k+1 is submitted (``kws_infer_host_submit_i16``) before batch k's results are waited
        for, so its pack and H2D run under batch k's kernels -- with the reference's own batch size (1028, ``train.py:110``)
        a batch is also cut into as many chunks as the ring has slots.  Yields ``(labels int32[B], logits float32[B,C])``
        per batch, in order.  ``max_batch``: clips per chunk of the staging ring (default 1024)."""
        n = self.config.desired_samples
        ctx = self.model._context(self.device.index or 0)
        if max_batch:
            ctx.infer_host_wait(0)
            ctx.ingest_config(chunk_clips=int(max_batch))
        pending = None  # (logits, labels, ticket, keepalive) of the batch submitted last
        try:
            for batch in batches:
                if torch.is_tensor(batch) and batch.dtype == torch.int16 and batch.dim() == 2 and batch.shape[1] == n \
                        and not batch.is_cuda and batch.is_contiguous():
                    src = batch
                else:
                    src = np.ascontiguousarray(fix_length(np.atleast_2d(np.asarray(batch, dtype=np.int16)), n))
                nxt = ctx.infer_host_submit_i16(src)
                if pending is not None:
                    ctx.infer_host_wait(pending[2])
                    yield pending[1], pending[0]
                pending = nxt
            if pending is not None:
                ctx.infer_host_wait(pending[2])
                yield pending[1], pending[0]
                pending = None
        finally:
            if pending is not None:  # the consumer stopped early: nothing may stay in flight into freed arrays
                ctx.infer_host_wait(0)

    def infer_files(self, paths: Sequence[str], resample: bool = False) -> List[Tuple[int, str]]:
        """wav files -> (index, word).  16-bit mono files at the configured rate go through the int16 path (the PCM
        itself is the device input); anything else -- other bit depths, float, stereo, and with ``resample=True`` other
        rates -- is decoded to float32 mono as ``librosa.load`` does and takes the float32 path (``kws_infer_f32``)."""
        n = self.config.desired_samples
        clips, all_i16 = [], True
        for p in paths:
            try:
                x = load_pcm16(p, self.config.sample_rate)
                if x.ndim == 2:
                    raise ValueError("stereo: mix down in float")
            except Exception:
                x = load_audio(p, self.config.sample_rate, resample)
                all_i16 = False
            clips.append(fix_length(x, n))
        if all_i16:
            labels, _ = self.infer_pcm16(np.stack(clips))
        else:
            f32 = np.stack([c.astype(np.float32) / np.float32(32768.0) if c.dtype == np.int16 else c for c in clips])
            labels, _ = self.infer_f32(f32)
        return [(int(i), self.words[int(i)]) for i in labels]

    def infer_f32(self, signals: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """``float32[B,n]`` host signals in [-1, 1] -> (labels int32[B], logits float32[B,C]) (``kws_infer_f32``)."""
        x = torch.from_numpy(np.ascontiguousarray(fix_length(np.atleast_2d(np.asarray(signals, dtype=np.float32)),
                                                             self.config.desired_samples))).to(self.device)
        ctx = self.model._context(self.device.index or 0)
        logits = torch.empty((x.shape[0], self.model.num_classes), dtype=torch.float32, device=self.device)
        labels = torch.empty((x.shape[0],), dtype=torch.int32, device=self.device)
        ctx.infer_f32(x, logits, labels)
        return labels.cpu().numpy(), logits.cpu().numpy()


_default: Optional[KeywordSpotter] = None


def inference(test_audio, spotter: Optional[KeywordSpotter] = None):
    """Classify one wav file; prints and returns ``(index, word)`` like the reference script prints."""
    global _default
    if spotter is None:
        if _default is None:
            _default = KeywordSpotter()
        spotter = _default
    idx, word = spotter.infer_files([test_audio])[0]
    print(idx, word)
    return idx, word


class StreamingSpotter:
    """Sliding-window spotting over concurrent live streams (10 ms hops).

    The reference's live path captures a VAD-segmented utterance, writes a wav and classifies it once
    (``kws/inference/inference_local.py:114-192``).  Here every ``push`` of ``frame_step`` new samples per
    stream adds one MFCC frame to a 99-frame ring on the GPU and re-classifies the last second of every
    stream (``kws_stream_push_i16``); with ``use_graph`` the two launches of a push replay as one hipGraph.
    """

    def __init__(self, n_streams: int, model: Optional[DepthwiseSeparableConv] = None, words: Sequence[str] = WANTED_WORDS,
                 config: Optional[AudioConfig] = None, device: int = 0, use_graph: bool = False, smooth_window: int = 0,
                 vad_log_energy: Optional[float] = None, vad_windows: Tuple[int, int] = (40, 80), host_results: bool = True):
        from kws import _native

        self.config = config or AudioConfig()
        self.words = list(words)
        self.model = model if model is not None else DepthwiseSeparableConv(num_classes=len(self.words))
        self.n_streams = int(n_streams)
        self.hop = int(round(self.config.frame_step * self.config.sample_rate))
        self.device = torch.device("cuda", device)
        self.use_graph = use_graph
        self._ctx = _native.Context(device, ModelError)
        cfg = self.config
        frame_len = int(round(cfg.frame_length * cfg.sample_rate))
        if (cfg.sample_rate, cfg.desired_samples, frame_len, self.hop, cfg.fft_size, cfg.num_mel_filters, cfg.num_cepstral_coeffs) != \
                (16000, 16000, 400, 160, 512, 26, 10):  # a non-default AudioConfig: tell the front end (nfft as audio_processor.py:268 derives it)
            self._ctx.set_frontend(cfg.sample_rate, cfg.desired_samples, frame_len, self.hop, max(cfg.fft_size, frame_len),
                                   cfg.num_mel_filters, cfg.num_cepstral_coeffs)
        self._ctx.load_dscnn(self.model.packed_weights(), self.model.num_classes)
        self._ctx.stream_open(self.n_streams)
        self._hop_buf = torch.zeros((self.n_streams, self.hop), dtype=torch.int16, device=self.device)
        self._logits = torch.zeros((self.n_streams, self.model.num_classes), dtype=torch.float32, device=self.device)
        self._labels = torch.zeros((self.n_streams,), dtype=torch.int32, device=self.device)
        # posterior smoothing: softmax of every hop's logits averaged over the last
        # `smooth_window` hops per stream on the device (kws_stream_smooth_f32); 0 = raw logits / argmax
        self.smooth_window = int(smooth_window)
        self._smoothed = torch.zeros_like(self._logits) if self.smooth_window > 0 else None
        # energy endpointer (stands in for the reference's webrtcvad gate,
        # kws/inference/inference_local.py:131-166): per stream, bit 0 of `vad_state` = inside an utterance,
        # bits 1-2 = opened (1) / closed (2) at the last hop; None = off
        self.vad_log_energy = vad_log_energy
        self.vad_windows = (int(vad_windows[0]), int(vad_windows[1]))
        self._vad = torch.zeros((self.n_streams,), dtype=torch.int32, device=self.device) if vad_log_energy is not None else None
        self.vad_state: Optional[np.ndarray] = None
        # zero-copy delivery (kws_stream_host_results): the push's own kernel writes logits and labels to pinned host memory
        # and raises a flag there, so a plain push returns without a stream synchronise or a device-to-host copy
        self._host = bool(host_results) and self.smooth_window == 0 and self._vad is None
        if self._host:
            self._ctx.stream_host_results(True)
        torch.cuda.synchronize(self.device)

    def load_model(self, model: DepthwiseSeparableConv) -> None:
        """Swap the classifier while the streams keep running (their PCM and feature rings are untouched); a captured
        hipGraph is re-captured on the next push (``kws_load_dscnn`` retires it: it holds the old weights)."""
        if model.num_classes != self.model.num_classes:
            raise ModelError(f"the streams were opened for {self.model.num_classes} classes, the new model has {model.num_classes}")
        self.model = model
        self._ctx.load_dscnn(model.packed_weights(), model.num_classes)

    def push(self, samples) -> Tuple[np.ndarray, np.ndarray]:
        """``int16[n_streams, hop]`` (host array or device tensor) -> (labels int32[S], logits float32[S,C]);
        with ``smooth_window`` > 0 the second array holds the smoothed posteriors and the labels are their argmax."""
        if self._host and not isinstance(samples, torch.Tensor):
            # host samples in, host results out, one call: the kernel reads the hop from pinned host memory and writes the
            # results back there (kws_stream_push_host_i16) -- no H2D copy, no synchronise, no D2H copy
            h = np.ascontiguousarray(samples, dtype=np.int16)
            if h.shape != (self.n_streams, self.hop):
                raise ModelError(f"push expects int16 [{self.n_streams}, {self.hop}]")
            lg, lb = self._ctx.stream_push_host_i16(h, self.n_streams)
            return lb.copy(), lg.copy()
        x = samples if isinstance(samples, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(samples, dtype=np.int16))
        if tuple(x.shape) != (self.n_streams, self.hop) or x.dtype != torch.int16:
            raise ModelError(f"push expects int16 [{self.n_streams}, {self.hop}]")
        self._hop_buf.copy_(x, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()  # the context runs on its own stream
        self._ctx.stream_push_i16(self._hop_buf, self._logits, self._labels, use_graph=self.use_graph)
        if self._host:
            lg, lb = self._ctx.stream_wait_host(self.n_streams)
            return lb.copy(), lg.copy()
        if self._vad is not None:
            self._ctx.stream_vad_f32(self.vad_log_energy, self.vad_windows[0], self.vad_windows[1], self._vad)
        if self.smooth_window > 0:
            self._ctx.stream_smooth_f32(self._logits, self.smooth_window, self._smoothed, self._labels)
            self._ctx.sync()
            self._fetch_vad()
            return self._labels.cpu().numpy(), self._smoothed.cpu().numpy()
        self._ctx.sync()
        self._fetch_vad()
        return self._labels.cpu().numpy(), self._logits.cpu().numpy()

    def _fetch_vad(self):
        if self._vad is not None:
            self.vad_state = self._vad.cpu().numpy()

    def features(self) -> np.ndarray:
        """The current windows, oldest frame first: float32 [S, 99, 10] (zeros where no frame exists yet)."""
        _, hops = self._ctx.stream_state()
        t, f = 99, 10
        tmp = torch.empty((self.n_streams, t, f), dtype=torch.float32, device=self.device)
        self._ctx.stream_copy_features(tmp)
        self._ctx.sync()
        host = tmp.cpu().numpy()
        frame_len = int(round(self.config.frame_length * self.config.sample_rate))
        k = -(-frame_len // self.hop)       # hops a frame spans (3 for 400 / 160): the newest frame is hops - k
        head = (hops - k + 1) % t
        return np.roll(host, -head, axis=1), hops

    def close(self):
        self._ctx.stream_close()
        self._ctx.close()
