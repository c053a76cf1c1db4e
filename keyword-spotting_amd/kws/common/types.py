"""Label ids fixed by the reference (``kws/common/types.py:6-10``)."""
from enum import IntEnum


class LabelIndex(IntEnum):
    SILENCE_INDEX = 0
    UNKNOWN_WORD_INDEX = 1
