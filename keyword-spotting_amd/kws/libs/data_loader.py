"""Dataset wrapper + batched device collate.

Drop-in for ``kws/libs/data_loader.py``: ``SpeechCommandsDataLoader(dataset, audio_processor, split)``
with ``__len__``, ``__getitem__ -> (float32[1,99,10], int)`` and ``get_class_mapping`` (``:14-123``).
``collate_pcm16`` is the batched path: it packs int16 clips into one pinned host buffer, copies once
and computes all MFCCs in a single kernel launch, producing the same ``float32[B,1,T,F]`` /
``int64[B]`` pair torch's default collate builds from per-sample ``__getitem__`` calls
(reference ``train.py:108-121``).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from kws.common.errors import DatasetError, handle_error
from kws.libs.audio_processor import AudioProcessor, fix_length, load_pcm16


class SpeechCommandsDataLoader(Dataset):
    VALID_SPLITS = ["training", "validation", "testing"]

    def __init__(self, dataset, audio_processor: AudioProcessor, split: str = "training") -> None:
        try:
            self.ap = audio_processor
            self.word_to_index = dataset.word_to_index
            if split not in self.VALID_SPLITS:
                raise DatasetError(f"Invalid split: {split}. Must be one of {self.VALID_SPLITS}")
            self.data = dataset.get_data(split)
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        except Exception as e:
            handle_error(e, DatasetError, f"Failed to initialize data loader for {split} split")

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, index: int) -> Tuple[torch.Tensor, int]:
        if index >= len(self.data):
            raise IndexError(f"Index {index} out of bounds for dataset of size {len(self.data)}")
        try:
            sample = self.data[index]
            label_idx = self.word_to_index[sample["label"]]
            feats = self.ap.transform(sample["file"], label_idx)
            return torch.tensor(feats, dtype=torch.float32).unsqueeze(0), label_idx
        except Exception as e:
            handle_error(e, DatasetError, f"Error processing sample at index {index}", re_raise=True)

    def get_class_mapping(self) -> Dict[int, str]:
        return {idx: word for word, idx in self.word_to_index.items()}

    # ------------------------------------------------------------------ batched path
    def collate_pcm16(self, indices: Sequence[int], device=None):
        """Un-augmented batch for evaluation/inference: files -> int16 [B,n] (pinned) -> one H2D copy ->
        one MFCC launch.  Returns (``float32[B,1,T,F]`` on the device, ``int64[B]`` labels)."""
        n = self.ap.config.desired_samples
        pcm = torch.empty((len(indices), n), dtype=torch.int16).pin_memory() if torch.cuda.is_available() else torch.empty((len(indices), n), dtype=torch.int16)
        labels: List[int] = []
        for row, i in enumerate(indices):
            sample = self.data[i]
            clip = load_pcm16(sample["file"], self.ap.config.sample_rate)
            if clip.ndim == 2:
                clip = clip.astype(np.int32).mean(axis=1).astype(np.int16)
            pcm[row] = torch.from_numpy(fix_length(clip, n).copy())
            labels.append(self.word_to_index[sample["label"]])
        dev = device or torch.device("cuda", self.ap.device)
        feats = self.ap.extract_features_batch(pcm.to(dev, non_blocking=True))
        return feats, torch.tensor(labels, dtype=torch.int64, device=dev)
