"""TEST INFRASTRUCTURE ONLY -- never imported by the product.

CPU statement of the energy endpointer behind ``kws_stream_vad_f32``.  It is build-defined:
the reference's live loop gates on ``webrtcvad`` (absent here, ``kws/inference/inference_local.py:34``), so **parity is
unpinned**; what is kept from the reference is the hysteresis of its endpointing, at hop (10 ms) instead of 30 ms
chunk granularity:

* a ring of voiced flags that starts as zeros (``inference_local.py:116-120``),
* the utterance opens when more than 80 % of the last ``on_window`` flags are voiced (``:148-152``),
* it closes when more than 90 % of the last ``off_window`` flags are unvoiced (``:158-163``),
* 400 ms / 800 ms windows (``:27-28``) = 40 / 80 hops.

The voiced decision itself is the frame's log energy (MFCC coefficient 0 with appendEnergy) above a threshold.
"""
from __future__ import annotations

from collections import deque
from typing import Tuple


class EnergyEndpointer:
    def __init__(self, threshold: float, on_window: int = 40, off_window: int = 80):
        assert 1 <= on_window <= off_window
        self.threshold, self.on_window, self.off_window = float(threshold), on_window, off_window
        self.flags = deque([0] * off_window, maxlen=off_window)  # newest last; history before the stream = unvoiced
        self.triggered = False

    def update(self, log_energy: float) -> Tuple[bool, int]:
        """One hop -> (inside an utterance, event: 0 none / 1 opened / 2 closed)."""
        self.flags.append(1 if log_energy > self.threshold else 0)
        recent = list(self.flags)
        event = 0
        if not self.triggered:
            if 10 * sum(recent[-self.on_window:]) > 8 * self.on_window:
                self.triggered, event = True, 1
        elif 10 * (self.off_window - sum(recent)) > 9 * self.off_window:
            self.triggered, event = False, 2
        return self.triggered, event
