"""wav -> label, the shape of the reference's ``inference(test_audio)``
(``kws/inference/inference_local.py:67-81``: load -> fix length to 1 s -> MFCC -> model -> argmax ->
word), on the fused MI355X path and for whole batches of files."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from kws.common.errors import ModelError
from kws.datasets.speech_commands import DEFAULT_WORDS, SpeechCommandDataset
from kws.libs.audio_processor import AudioConfig, fix_length, load_pcm16
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
        """``int16[B,n]`` host array -> (labels int32[B], logits float32[B,C])."""
        clips = fix_length(np.atleast_2d(np.asarray(pcm, dtype=np.int16)), self.config.desired_samples)
        wav = torch.from_numpy(np.ascontiguousarray(clips)).to(self.device)
        logits, labels = self.model.infer_pcm16(wav)
        return labels.cpu().numpy(), logits.cpu().numpy()

    def infer_files(self, paths: Sequence[str]) -> List[Tuple[int, str]]:
        clips = []
        for p in paths:
            x = load_pcm16(p, self.config.sample_rate)
            if x.ndim == 2:
                x = x.astype(np.int32).mean(axis=1).astype(np.int16)
            clips.append(fix_length(x, self.config.desired_samples))
        labels, _ = self.infer_pcm16(np.stack(clips))
        return [(int(i), self.words[int(i)]) for i in labels]


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
