import sys, os, numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'keyword-spotting_amd'))
from kws import _native
from oracle import psf_mfcc as o
g = np.load(os.path.join(ROOT,'tests/golden/sigproc_golden.npz'))
clips = g['clips']; names = list(g['names'])
dev = torch.device('cuda',0)
ctx = _native.Context(0)
wav = torch.from_numpy(clips).to(dev)
out = torch.empty((len(clips),1,99,10), dtype=torch.float32, device=dev)
ctx.mfcc_i16(wav, out); ctx.sync()
got = out.cpu().numpy()[:,0]
for i,n in enumerate(names):
    want = o.extract_features_pcm16(clips[i])
    err = np.abs(got[i]-want)
    fr = np.argwhere(err > 1e-4)
    print(n, err.max(), 'bad frames', sorted(set(fr[:,0].tolist()))[:12], 'bad coefs', sorted(set(fr[:,1].tolist())))
    if len(fr):
        f = fr[0][0]; print('  frame', f, 'got', got[i][f], '\n  want', want[f])
