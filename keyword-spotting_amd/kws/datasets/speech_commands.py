"""Speech Commands label list and split rule (host side, no download).

Only what the inference path needs from ``kws/datasets/speech_commands.py``: the class order
(``[_silence_, _unknown_] + wanted_words``, ``:28-41,123-132``; ids 0/1 fixed by
``kws/common/types.py``) and the filename-hash split (``which_set``, ``:257-281``), plus a local
directory indexer.  There is no network here, so nothing is ever downloaded.
"""
from __future__ import annotations

import hashlib
import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List

from kws.common.errors import DatasetError
from kws.common.types import LabelIndex

DEFAULT_WORDS = ["yes", "no", "up", "down", "left", "right", "on", "off", "stop", "go"]


@dataclass
class DatasetConfig:
    silence_percentage: float = 10
    unknown_percentage: float = 10
    testing_percentage: float = 10
    validation_percentage: float = 10
    wanted_words: List[str] = field(default_factory=lambda: list(DEFAULT_WORDS))
    random_seed: int = 59185
    download_data: bool = False  # the reference defaults to True; this build never downloads


class SpeechCommandDataset:
    SILENCE_LABEL = "_silence_"
    UNKNOWN_LABEL = "_unknown_"
    BACKGROUND_NOISE_DIR = "_background_noise_"
    MAX_NUM_WAVS_PER_CLASS = 2 ** 27 - 1

    def __init__(self, config: DatasetConfig, root_dir: Path):
        if config.download_data:
            raise DatasetError("download_data=True is not available: this build has no network access")
        self.config = config
        self.root_dir = Path(root_dir)
        self.words_list = self.prepare_word_list(config.wanted_words)
        self.word_to_index: Dict[str, int] = {self.SILENCE_LABEL: int(LabelIndex.SILENCE_INDEX),
                                              self.UNKNOWN_LABEL: int(LabelIndex.UNKNOWN_WORD_INDEX)}
        for w in config.wanted_words:
            self.word_to_index[w] = len(self.word_to_index)
        self.data_index = {"training": [], "validation": [], "testing": []}
        if self.root_dir.exists():
            self._index_local_files()

    def prepare_word_list(self, wanted_words: List[str]) -> List[str]:
        return [self.SILENCE_LABEL, self.UNKNOWN_LABEL] + list(wanted_words)

    def _index_local_files(self) -> None:
        for wav in sorted(self.root_dir.glob("*/*.wav")):
            word = wav.parent.name.lower()
            if word == self.BACKGROUND_NOISE_DIR:
                continue
            label = word if word in self.config.wanted_words else self.UNKNOWN_LABEL
            self.data_index[self.which_set(str(wav))].append({"label": label, "file": str(wav)})

    def get_data(self, split: str) -> List[Dict]:
        if split not in self.data_index:
            raise DatasetError(f"Invalid split: {split}. Must be one of {list(self.data_index.keys())}")
        return self.data_index[split]

    def __len__(self) -> int:
        return sum(len(v) for v in self.data_index.values())

    def get_class_count(self) -> int:
        return len(self.words_list)

    def get_words_list(self) -> List[str]:
        return self.words_list

    def which_set(self, filename: str) -> str:
        """Stable split from the sha1 of the speaker part of the file name."""
        speaker = re.sub(r"_nohash_.*$", "", os.path.basename(filename))
        h = int(hashlib.sha1(speaker.encode()).hexdigest(), 16)
        pct = (h % (self.MAX_NUM_WAVS_PER_CLASS + 1)) * (100.0 / self.MAX_NUM_WAVS_PER_CLASS)
        if pct < self.config.validation_percentage:
            return "validation"
        if pct < self.config.testing_percentage + self.config.validation_percentage:
            return "testing"
        return "training"
