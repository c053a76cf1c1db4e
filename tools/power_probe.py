#!/usr/bin/env python3
"""Diagnostics (GPU box): socket power and shader clock while one kernel runs back to back.

    python tools/power_probe.py dscnn|mfcc [seconds = 4]

Samples the amdgpu hwmon files (power1_average / power1_input, freq1_input) and `rocm-smi` once per run; prints the idle
reading, the loaded readings and the power cap, so that "power-limited" in DESIGN.md is a measurement.
"""
import glob, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native

which = sys.argv[1] if len(sys.argv) > 1 else "dscnn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0

def read(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None

hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
def sample():
    out = []
    for h in hw:
        p = read(h + "/power1_average") or read(h + "/power1_input")
        f = read(h + "/freq1_input")
        cap = read(h + "/power1_cap")
        out.append((p / 1e6 if p else None, f / 1e6 if f else None, cap / 1e6 if cap else None))
    return out

def smi():
    try:
        return subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return f"rocm-smi failed: {e}"

print("hwmon dirs:", hw)
print("idle:", sample())
dev = torch.device("cuda", 0)
B = 4096
ctx = _native.Context(0); ctx.use_torch_stream()
ctx.load_dscnn(bench.bench_weights()[0], 12)
wav = torch.from_numpy(bench.synth_clips(B, 0)).to(dev)
feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
ctx.mfcc_i16(wav, feat)
logits = torch.empty((B, 12), dtype=torch.float32, device=dev)
labels = torch.empty((B,), dtype=torch.int32, device=dev)
stop = False
samples = []
def sampler():
    while not stop:
        samples.append((time.perf_counter(), sample()))
        time.sleep(0.1)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
smi_out = None
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        if which == "dscnn": ctx.forward_f32(feat, logits, labels)
        else: ctx.mfcc_i16(wav, feat)
    torch.cuda.synchronize(); n += 200
    if smi_out is None and time.perf_counter() - t0 > secs / 2:
        for _ in range(2000):
            if which == "dscnn": ctx.forward_f32(feat, logits, labels)
            else: ctx.mfcc_i16(wav, feat)
        smi_out = smi()  # taken while ~1 s of launches is queued
        torch.cuda.synchronize(); n += 2000
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{which}: {n} launches, {dt / n * 1e3:.4f} ms per launch (wall, incl. probe gaps)")
for t, s in samples[:: max(1, len(samples) // 12)]:
    print(f"  t={t - t0:5.2f}s  " + "  ".join(f"{p} W @ {f} MHz (cap {c} W)" for p, f, c in s))
print(smi_out)
