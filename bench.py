#!/usr/bin/env python3
"""Throughput of the keyword-spotting hot path on MI355X: 1 s / 16 kHz clips per second, end to end
(device-resident int16 PCM -> MFCC -> DS-CNN -> logits + label), BASELINE.json's metric.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused path (kws_infer_i16: MFCC kernel + DS-CNN kernel on one stream)
over one batch of 4096 synthetic clips per GPU (BASELINE.json configs[2], the end-to-end
configuration; the model is the reference's DS-CNN -- "cnn-trad-fpool3" does not exist in the
reference).  Clips are independent, so N GPUs = N shards with no collective on
the data path (weak scaling); torch.distributed is used only for the barrier and the max-over-ranks
of the timed region.  Rank 0 prints ONE JSON line.  Before the W warm-up steps the step runs `--spinup` more untimed
times (default 60): from idle the GPU needs ~30 steps for its clocks to settle, and the timed K steps should see the
steady state whatever W the caller picked.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "keyword-spotting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

N_SAMPLES = 16000
NUM_CLASSES = 12
BYTES_PER_CLIP = N_SAMPLES * 2                 # algorithmic HBM read: int16[16000]
DSCNN_FLOP_PER_CLIP = 2 * 6_603_008            # SURVEY.md 8a totals: 6 603 008 MAC, interior-only pointwise
MFCC_FLOP_PER_CLIP = 1.4e6
PEAK_F32_TFLOPS = 157.3                        # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_BPS = 8.0e12                          # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_BF16_TFLOPS = 2500.0                      # MI355X_MICROARCH.md: dense bf16 MFMA
# bf16 MFMA work the split path executes per clip: (42 block units x 48 + 10 conv1 units x 42) MFMAs of 32x32x16
DSCNN_EXECUTED_BF16_FLOP_PER_CLIP = (42 * 48 + 10 * 42) * 32 * 32 * 16 * 2
# cnn-trad-fpool3 (build-defined): the two convolutions of kws_cnntrad_conv_kernel / all five layers
CNNTRAD_CONV_FLOP_PER_CLIP = 2 * (99 * 10 * 64 * 160 + 297 * 64 * 2560)
CNNTRAD_FLOP_PER_CLIP = CNNTRAD_CONV_FLOP_PER_CLIP + 2 * (19008 * 32 + 32 * 128 + 128 * NUM_CLASSES)
# executed on the bf16 pipe: conv1 33 tiles x 2 channel tiles x 10 k-blocks, conv2 10 x 2 x 160, six products each
CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP = (33 * 2 * 10 + 10 * 2 * 160) * 6 * 32 * 32 * 16 * 2


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous shard [lo, hi) of `total` units for `rank` of `world` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step_fn, steps: int, barrier, device_sync, all_reduce_max):
    """The timed region of the bench contract: barrier + device sync on both sides of exactly `steps`
    calls, wall time = MAX over ranks.  Backend-agnostic so the N>1 logic is testable with gloo on CPU."""
    barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    device_sync()
    barrier()
    return all_reduce_max(time.perf_counter() - t0)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 read correction applied).  bench.py
    cannot collect counters itself; None if the file has no entry."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return float(json.load(f)["kernels"][kernel]["hbm_bytes"])
    except Exception:
        return None


def synth_weights(seed: int = 1, std: float = 0.1) -> np.ndarray:
    """Random-init DS-CNN in state_dict order, every parameter (biases too) ~ N(0, std)."""
    n = 6400 + 64 + 4 * (576 + 64 + 4096 + 64) + NUM_CLASSES * 64 + NUM_CLASSES
    return (np.random.RandomState(seed).standard_normal(n) * std).astype(np.float32)


def synth_cnn_trad_weights(seed: int = 1) -> np.ndarray:
    """Random-init cnn-trad-fpool3 in state_dict order (conv1, conv2, lin, dnn, fc; weight then bias): fan-in scaled
    normal weights so activations stay O(1) through the 2560- and 19008-wide sums, N(0, 0.1) biases."""
    rs = np.random.RandomState(seed)
    parts = []
    for shape in ((64, 1, 20, 8), (64, 64, 10, 4), (32, 64 * 99 * 3), (128, 32), (NUM_CLASSES, 128)):
        fan_in = int(np.prod(shape[1:]))
        parts.append(rs.standard_normal(int(np.prod(shape))) * (2.0 / fan_in) ** 0.5)
        parts.append(rs.standard_normal(shape[0]) * 0.1)
    return np.concatenate(parts).astype(np.float32)


def synth_clips(batch: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(-32768, 32768, size=(batch, N_SAMPLES), dtype=np.int16)


def cpu_baseline(clips: np.ndarray, blob: np.ndarray):
    """The CPU oracle (NumPy/SciPy psf-equivalent MFCC called per clip in a Python loop, the reference's
    structure, + torch-CPU DS-CNN) timed on this host.  Checker code, used here only as the baseline."""
    import torch

    from oracle import dscnn as o_dscnn
    from oracle import psf_mfcc as o_mfcc

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    state, off = {}, 0
    for k, shp in o_dscnn.state_shapes(NUM_CLASSES).items():
        n = int(np.prod(shp))
        state[k] = torch.from_numpy(blob[off:off + n].reshape(shp).copy())
        off += n
    o_mfcc.collate_pcm16(clips[:4])  # warm caches / imports
    t0 = time.perf_counter()
    feats = o_mfcc.collate_pcm16(clips)
    t1 = time.perf_counter()
    with torch.no_grad():
        logits = o_dscnn.forward(state, torch.from_numpy(feats))
        o_dscnn.predict(logits)
    t2 = time.perf_counter()
    n = len(clips)
    return {
        "value": n / (t2 - t0),
        "unit": "clips/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} of the step's clips once: per-clip NumPy MFCC loop (1 thread) {t1 - t0:.2f} s + "
                  f"torch-CPU DS-CNN batch forward ({cores} threads) {t2 - t1:.2f} s",
        "mfcc_clips_per_s": n / (t1 - t0),
        "dscnn_clips_per_s": n / (t2 - t1),
    }, logits.numpy()


def mfcc_only_line(args, world, B, elapsed, ctx, _native, clips, feat_out):
    """The JSON line of `--model mfcc-only` (BASELINE.json configs[1]: the MFCC kernel alone, vs the CPU)."""
    m_ms, m_n = ctx.prof_read(_native.KWS_K_MFCC)
    mfcc_s = (m_ms / max(m_n, 1)) * 1e-3
    value = B * world * args.steps / elapsed
    achieved = B * BYTES_PER_CLIP / mfcc_s / 1e9 if mfcc_s > 0 else 0.0
    out = {
        "metric": "1s 16kHz clips/sec, MFCC only (wav->features)", "value": value, "unit": "clips/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "C2: batch=4096/GPU synthetic uniform int16 1s/16kHz clips, device-resident, MFCC(400/160/512, "
                        "26 mel, 10 cep) -> float32 [B,1,99,10]",
            "clips_per_gpu_per_step": B,
            "sharding": f"{world} independent shard(s), no data-path collective",
            "spinup_steps": args.spinup,
        },
        "roofline": {
            "kernel": _native.kernel_name(_native.KWS_K_MFCC), "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_BPS / 1e9,
            "unit": "GB/s", "frac": achieved / (PEAK_HBM_BPS / 1e9), "traffic": pmc_traffic(_native.kernel_name(_native.KWS_K_MFCC)),
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
            "algorithmic_bytes_per_launch": B * BYTES_PER_CLIP, "avg_kernel_ms": mfcc_s * 1e3, "launches": m_n,
            "note": "algorithmic HBM read (32 000 B per clip) over the kernel's duration against the 8 TB/s spec peak, the "
                    "roofline BASELINE.json declares; the kernel is LDS/VALU-bound (DESIGN.md 4.1): "
                    f"{B / mfcc_s * MFCC_FLOP_PER_CLIP / 1e12 if mfcc_s > 0 else 0.0:.1f} TFLOP/s of 157.3 f32",
        },
    }
    if world == 1 and args.cpu_sample > 0:
        from oracle import psf_mfcc as o_mfcc

        n = min(args.cpu_sample, B)
        o_mfcc.collate_pcm16(clips[:4])
        t0 = time.perf_counter()
        want = o_mfcc.collate_pcm16(clips[:n])
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "clips/s", "cores": 1, "kind": "port",
                               "sample": f"{n} of the step's clips once: per-clip NumPy MFCC loop (the reference's structure)"}
        out["parity_vs_cpu_max_abs_mfcc_err"] = float(np.abs(feat_out[:n].cpu().numpy() - want).max())
    return out


def cnn_trad_line(args, world, B, elapsed, ctx, _native, clips, state, logits):
    """The JSON line of `--model cnn-trad-fpool3` (BASELINE.json configs[2] read literally; not the driver's line)."""
    c_ms, c_n = ctx.prof_read(_native.KWS_K_CNNTRAD_CONV)
    d_ms, d_n = ctx.prof_read(_native.KWS_K_CNNTRAD_DENSE)
    m_ms, m_n = ctx.prof_read(_native.KWS_K_MFCC)
    conv_s, dense_s, mfcc_s = (c_ms / max(c_n, 1)) * 1e-3, (d_ms / max(d_n, 1)) * 1e-3, (m_ms / max(m_n, 1)) * 1e-3
    value = B * world * args.steps / elapsed
    executed = CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP * B / conv_s / 1e12 if conv_s > 0 else 0.0
    out = {
        "metric": "1s 16kHz clips/sec end-to-end (wav->label)", "value": value, "unit": "clips/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "configs[2] read literally: batch=4096/GPU synthetic uniform int16 1s/16kHz clips, device-resident, "
                        "MFCC + cnn-trad-fpool3 (build-defined, SAME padding on the 99x10 map, 12 classes, random-init) "
                        "-> logits+label",
            "clips_per_gpu_per_step": B,
            "sharding": f"{world} independent shard(s), no data-path collective",
            "spinup_steps": args.spinup,
        },
        "roofline": {
            "kernel": _native.kernel_name(_native.KWS_K_CNNTRAD_CONV), "bound": "mfma",
            "achieved": executed, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": executed / PEAK_BF16_TFLOPS,
            "traffic": None, "avg_kernel_ms": conv_s * 1e3, "launches": c_n,
            "flop_per_clip_executed_bf16": CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP,
            "flop_per_clip_algorithmic": CNNTRAD_CONV_FLOP_PER_CLIP,
            "algorithmic_tflops": CNNTRAD_CONV_FLOP_PER_CLIP * B / conv_s / 1e12 if conv_s > 0 else 0.0,
            "math": "f32 in / f32 out; both convolutions on v_mfma_f32_32x32x16_bf16 as exact three-way bf16 splits "
                    "(6 MFMAs per f32 product); achieved/peak count the bf16 MFMA work executed against the dense bf16 peak",
        },
        "other_kernels_ms": {_native.kernel_name(_native.KWS_K_MFCC): mfcc_s * 1e3,
                             _native.kernel_name(_native.KWS_K_CNNTRAD_DENSE): dense_s * 1e3},
    }
    if world == 1 and args.cpu_sample > 0:
        import torch

        from oracle import cnn_trad as o_ct
        from oracle import psf_mfcc as o_mfcc

        n = min(args.cpu_sample, B, 256)
        t0 = time.perf_counter()
        feats = o_mfcc.collate_pcm16(clips[:n])
        want = o_ct.forward(o_ct.unflatten_state(state, NUM_CLASSES), torch.from_numpy(feats)).numpy()
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n} of the step's clips: oracle MFCC (NumPy, per clip) + torch-CPU cnn-trad-fpool3"}
        scale = max(1.0, float(np.abs(want).max()))
        out["parity_vs_cpu_max_abs_logit_err_over_scale"] = float(np.abs(logits[:n].cpu().numpy() - want).max() / scale)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=60,
                    help="untimed steps before the warm-up: from idle the GPU's clocks take ~30 steps (20 ms) to settle "
                         "(per-step time 0.82 -> 0.67 ms, tools/ramp_probe.py); reported as config.spinup_steps")
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="clips timed on the CPU baseline (0 = skip)")
    ap.add_argument("--model", choices=["ds-cnn", "cnn-trad-fpool3", "mfcc-only"], default="ds-cnn",
                    help="ds-cnn: the reference's model (the driver's line); cnn-trad-fpool3: the build-defined model "
                         "BASELINE.json configs[2] names (parity unpinned against the reference, DESIGN.md 4.5); "
                         "mfcc-only: BASELINE.json configs[1], the front end alone, priced against the HBM-read roofline")
    args = ap.parse_args()

    import torch

    from kws import _native

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is a HIP library with no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    # ---- per-rank shard of the job: `batch` clips per GPU per step (weak scaling) ------------------
    B = args.batch
    lo, hi = shard_bounds(B * world, world, rank)
    assert hi - lo == B
    clips = synth_clips(B, seed=rank)
    blob = synth_weights()
    ctx = _native.Context(local_rank)
    ct_state = None
    feat_out = None
    if args.model == "mfcc-only":
        feat_out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
        step = lambda: ctx.mfcc_i16(wav, feat_out)
    elif args.model == "cnn-trad-fpool3":
        ct_state = synth_cnn_trad_weights()
        ctx.load_cnn_trad(ct_state, NUM_CLASSES)
        step = lambda: ctx.infer_cnn_trad_i16(wav, logits, labels)
    else:
        ctx.load_dscnn(blob, NUM_CLASSES)
        step = lambda: ctx.infer_i16(wav, logits, labels)
    ctx.reserve(B)
    wav = torch.from_numpy(clips).to(dev)
    logits = torch.empty((B, NUM_CLASSES), dtype=torch.float32, device=dev)
    labels = torch.empty((B,), dtype=torch.int32, device=dev)

    for _ in range(args.spinup + args.warmup):
        step()
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()

    def reduce_max(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = timed_steps(step, args.steps, barrier, torch.cuda.synchronize, reduce_max)

    k_ms, k_n = ctx.prof_read(_native.KWS_K_DSCNN)
    m_ms, m_n = ctx.prof_read(_native.KWS_K_MFCC)
    ctx.prof_enable(False)

    if rank == 0 and args.model == "mfcc-only":
        print(json.dumps(mfcc_only_line(args, world, B, elapsed, ctx, _native, clips, feat_out)), flush=True)
    elif rank == 0 and args.model == "cnn-trad-fpool3":
        print(json.dumps(cnn_trad_line(args, world, B, elapsed, ctx, _native, clips, ct_state, logits)), flush=True)
    elif rank == 0:
        total_clips = B * world * args.steps
        value = total_clips / elapsed
        dscnn_s = (k_ms / max(k_n, 1)) * 1e-3
        mfcc_s = (m_ms / max(m_n, 1)) * 1e-3
        achieved = DSCNN_FLOP_PER_CLIP * B / dscnn_s / 1e12 if dscnn_s > 0 else 0.0
        out = {
            "metric": "1s 16kHz clips/sec end-to-end (wav->label)",
            "value": value,
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C3: batch=4096/GPU synthetic uniform int16 1s/16kHz clips, device-resident, "
                            "MFCC(400/160/512, 26 mel, 10 cep) + DS-CNN(12 classes, random-init N(0,0.1)) -> logits+label",
                "clips_per_gpu_per_step": B,
                "sharding": f"{world} independent shard(s), no data-path collective",
                "spinup_steps": args.spinup,
            },
            "roofline": {
                "kernel": _native.kernel_name(_native.KWS_K_DSCNN),
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_TFLOPS,
                "traffic": pmc_traffic(_native.kernel_name(_native.KWS_K_DSCNN)),
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
                "algorithmic_bytes_per_launch": B * (99 * 10 * 4 + 52),
                "avg_kernel_ms": dscnn_s * 1e3,
                "launches": k_n,
                "flop_per_clip": DSCNN_FLOP_PER_CLIP,
                "math": "f32 in / f32 out; conv1 and the four 1x1 convolutions run on v_mfma_f32_32x32x16_bf16 as exact "
                        "three-way bf16 splits (6 MFMAs per f32 product, f32 accumulate); peak/frac are priced against "
                        "the f32 MFMA peak the dtype names",
                "bf16_pipe": {
                    "executed_tflops": DSCNN_EXECUTED_BF16_FLOP_PER_CLIP * B / dscnn_s / 1e12 if dscnn_s > 0 else 0.0,
                    "peak": PEAK_BF16_TFLOPS,
                    "frac": DSCNN_EXECUTED_BF16_FLOP_PER_CLIP * B / dscnn_s / 1e12 / PEAK_BF16_TFLOPS if dscnn_s > 0 else 0.0,
                },
            },
            "hbm_read": {
                "bytes_per_clip": BYTES_PER_CLIP,
                "achieved_GBps_per_gpu": value / world * BYTES_PER_CLIP / 1e9,
                "frac_of_8TBps": value / world * BYTES_PER_CLIP / PEAK_HBM_BPS,
            },
            "mfcc_kernel": {
                "kernel": _native.kernel_name(_native.KWS_K_MFCC),
                "avg_kernel_ms": mfcc_s * 1e3,
                "clips_per_s": B / mfcc_s if mfcc_s > 0 else 0.0,
                "hbm_read_frac": (B / mfcc_s * BYTES_PER_CLIP / PEAK_HBM_BPS) if mfcc_s > 0 else 0.0,
                "f32_frac": (B / mfcc_s * MFCC_FLOP_PER_CLIP / (PEAK_F32_TFLOPS * 1e12)) if mfcc_s > 0 else 0.0,
            },
        }
        if world == 1 and args.cpu_sample > 0:
            n = min(args.cpu_sample, B)
            base, cpu_logits = cpu_baseline(clips[:n], blob)
            out["cpu_baseline"] = base
            out["parity_vs_cpu_max_abs_logit_err"] = float(np.abs(logits[:n].cpu().numpy() - cpu_logits).max())
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
