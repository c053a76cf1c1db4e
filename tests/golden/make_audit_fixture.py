#!/usr/bin/env python3
"""Writes tests/golden/audit_hard_clips.npz: the clips of the front-end precision audit (tools/fe_precision_audit.py, 500 clips
per generator, seeds 1..3) in which a frame with a log-mel span between 11.5 and 12.0 missed 1e-4 while the refinement
threshold was 12.0 -- the evidence the threshold of 11.5 rests on (profiles/r03_precision_audit.txt), kept as a regression
fixture.  Inputs only: the expected cepstra are computed by the oracle when the test runs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import audit_clips  # noqa: E402

PICKS = [(1, "tones", 374, (57,)), (1, "mix", 274, (1, 32)), (2, "tones", 196, (81, 94)), (3, "mix", 3, (7,))]   # seed, generator, clip, frames

clips, names, frames = [], [], []
for seed in (1, 2, 3):
    sets = audit_clips.make_sets(500, seed)
    for s, g, i, fr in PICKS:
        if s == seed:
            clips.append(sets[g][i]); names.append(f"seed{s}_{g}_{i}"); frames.append(list(fr) + [-1] * (2 - len(fr)))
np.savez_compressed(os.path.join(HERE, "audit_hard_clips.npz"), clips=np.stack(clips), names=np.array(names), frames=np.array(frames, dtype=np.int32))
print(names, np.stack(clips).shape)

# The four frames of seeds 4..11 (2.38 M frames, audited after the threshold had been set) that miss 1e-4 AT the shipped threshold:
# spans 10.8-11.5, float32 errors 1.2e-4 .. 1.5e-4 (profiles/r03_precision_audit.txt).  Kept as the documented exceptions of the
# default front end: the test holds them under 2e-4 and holds KWS_FE_F64 to 1e-4 on them.
EXCEPTIONS = [(4, "mix", 130, 97), (5, "tones", 54, 1), (11, "tones", 362, 56), (11, "mix", 125, 85)]
clips, names, frames = [], [], []
for seed in (4, 5, 11):
    sets = audit_clips.make_sets(500, seed)
    for s, g, i, f in EXCEPTIONS:
        if s == seed:
            clips.append(sets[g][i]); names.append(f"seed{s}_{g}_{i}"); frames.append(f)
np.savez_compressed(os.path.join(HERE, "audit_exception_clips.npz"), clips=np.stack(clips), names=np.array(names), frames=np.array(frames, dtype=np.int32))
print(names, np.stack(clips).shape)
