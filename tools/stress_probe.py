#!/usr/bin/env python3
"""Diagnostics (GPU box): logit error of the fused path on the 48 diverse + 16 speech-like clips under the three stress weight
tags (tests/golden/stress_golden.npz: he, he5, raw), with the selective refinement on and off."""
import os, sys, numpy as np, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/keyword-spotting_amd")
from kws import _native
e=np.load(ROOT+"/tests/golden/e2e_golden.npz"); g=np.load(ROOT+"/tests/golden/stress_golden.npz")
clips=np.concatenate([e["clips"], g["speech_clips"]]); dev=torch.device("cuda",0)
names=[str(n) for n in e["names"]]+[str(n) for n in g["speech_names"]]
for span in (_native.FE_REFINE_SPAN_DEFAULT, 0.0):
  for tag in ("he","he5","raw"):
    c=_native.Context(0); c.set_frontend_refine(span); c.load_dscnn(g[tag+".blob"],12)
    wav=torch.from_numpy(clips).to(dev); lo=torch.empty((len(clips),12),dtype=torch.float32,device=dev); la=torch.empty((len(clips),),dtype=torch.int32,device=dev)
    c.infer_i16(wav,lo,la); c.sync()
    err=np.abs(lo.cpu().numpy()-g[tag+".logits"]).max(axis=1); w=int(err.argmax())
    print(f"refine span {span}: {tag}: max logit err {err.max():.2e} at {names[w]} (|logit| max {np.abs(g[tag+'.logits']).max():.1f}), labels equal {np.array_equal(la.cpu().numpy(), g[tag+'.label'])}, speech-only max {err[48:].max():.2e}")
    c.close()
