#!/usr/bin/env python3
"""Throughput of the keyword-spotting hot path on MI355X: 1 s / 16 kHz clips per second, end to end
(device-resident int16 PCM -> MFCC -> DS-CNN -> logits + label), BASELINE.json's metric.

    python bench.py                                   # 1 GPU, defaults that finish within minutes
    python bench.py --gpus 8                          # launches its own 8 ranks (one process per GPU)
    python bench.py --gpus 8 --total-batch 8192       # BASELINE configs[3] as written: 8 x 1024, strong scaling
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # the driver's launch: also fine

A "step" is one pass of the fused path (kws_infer_i16: MFCC kernel + DS-CNN kernel on one stream) over one batch of
synthetic clips per GPU: 4096 per GPU by default (weak scaling), or a contiguous shard of --total-batch (strong
scaling).  The model is the reference's DS-CNN, the only model the reference defines; its
weights are the signal-preserving golden set (tests/golden/e2e_golden.npz, data generated from the imported reference).
Clips are independent, so N GPUs = N shards with no collective on the data path; torch.distributed is used only for the
barrier and the max-over-ranks of the timed region (two scalars: gloo by default, --dist-backend nccl for RCCL).  Rank 0
prints ONE JSON line.

With --gpus N > 1 and no WORLD_SIZE in the environment the script is its own launcher: a parent process that never
touches the GPU starts N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays
rank 0's line and exits with the worst child code -- no exec of a process that has initialised HIP.

At N = 1 the line also carries, under "configs", one short measurement of every other BASELINE.json configuration
(each with its own clock spin-up): configs[1] MFCC only, configs[2] read literally (cnn-trad-fpool3, build-defined),
configs[3]'s per-GPU shape (DS-CNN, 1024 clips), configs[4] streaming (64 streams, p50 / p99 per 10 ms hop).

Before the W warm-up steps the step runs `--spinup` more untimed times (default 60): from idle the GPU needs ~30 steps
for its clocks to settle, and the timed K steps should see the steady state whatever W the caller picked.
"""
import argparse
import json
import os
import socket
import subprocess
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
# 16-bit MFMA work the product path executes per clip on f16 pairs (three piece products per f32 k-block; the bf16 triple
# executes six): (42 block units x 24 + 10 conv1 units x 21) MFMAs of 32x32x16
DSCNN_EXECUTED_BF16_FLOP_PER_CLIP = (42 * 24 + 10 * 21) * 32 * 32 * 16 * 2
# cnn-trad-fpool3 (build-defined): the two convolutions of kws_cnntrad_conv_kernel / all five layers
CNNTRAD_CONV_FLOP_PER_CLIP = 2 * (99 * 10 * 64 * 160 + 297 * 64 * 2560)
CNNTRAD_FLOP_PER_CLIP = CNNTRAD_CONV_FLOP_PER_CLIP + 2 * (19008 * 32 + 32 * 128 + 128 * NUM_CLASSES)
# executed on the bf16 pipe: conv1 33 tiles x 2 channel tiles x 10 k-blocks, conv2 10 x 2 x 160, six products each
CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP = (33 * 2 * 10 + 10 * 2 * 160) * 6 * 32 * 32 * 16 * 2
GOLDEN = os.path.join(ROOT, "tests", "golden", "e2e_golden.npz")
LAUNCH_TIMEOUT_S = 1800                        # self-launched N > 1 runs: the parent gives its children this long


# ------------------------------------------------------------------------------------------ host-only helpers
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
    dt = time.perf_counter() - t0  # this rank's K steps; the clock is read BEFORE the closing barrier (a gloo round trip is not step time)
    barrier()
    return all_reduce_max(dt)


def pmc_traffic(kernel: str, workload: str = "ds-cnn"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 read correction applied).  bench.py
    cannot collect counters itself; None if the file has no entry."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        if workload != "ds-cnn":
            d = d["workloads"][workload]
        return float(d["kernels"][kernel]["hbm_bytes"])
    except Exception:
        return None


def synth_weights(seed: int = 1, std: float = 0.1) -> np.ndarray:
    """Random-init DS-CNN in state_dict order, every parameter (biases too) ~ N(0, std).  (Round 1's bench weights;
    with them the logits barely depend on the audio.  Kept for A/B tools; the bench uses `bench_weights`.)"""
    n = 6400 + 64 + 4 * (576 + 64 + 4096 + 64) + NUM_CLASSES * 64 + NUM_CLASSES
    return (np.random.RandomState(seed).standard_normal(n) * std).astype(np.float32)


def signal_preserving_weights(seed: int = 2, in_scale: float = 15.0) -> np.ndarray:
    """DS-CNN blob in state_dict order with fan-in scaled convolutions (N(0, 2/fan_in); conv1 also divided by the
    MFCC maps' RMS ~ 15), N(0, 0.1) biases, N(0, 0.5) classifier: activations stay O(1..30) through the net."""
    rs = np.random.RandomState(seed)
    parts = []
    shapes = [("conv1", (64, 1, 10, 10))]
    for _ in range(4):
        shapes += [("dw", (64, 1, 3, 3)), ("pw", (64, 64, 1, 1))]
    shapes.append(("fc", (NUM_CLASSES, 64)))
    for kind, shp in shapes:
        if kind == "fc":
            w = rs.standard_normal(shp) * 0.5
        else:
            w = rs.standard_normal(shp) * np.sqrt(2.0 / int(np.prod(shp[1:])))
            if kind == "conv1":
                w = w / in_scale
        parts += [w.reshape(-1), rs.standard_normal(shp[0]) * 0.1]
    return np.concatenate(parts).astype(np.float32)


def bench_weights():
    """(blob, golden or None): the golden signal-preserving DS-CNN weights (data file; expected logits of 48 diverse
    clips from the imported reference model ride along for the parity figure), else weights of the same statistics."""
    try:
        g = np.load(GOLDEN)
        return np.ascontiguousarray(g["he.blob"], dtype=np.float32), g
    except Exception:
        return signal_preserving_weights(), None


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


def cpu_model_string() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_cores() -> int:
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def gpu_numa_cpus(index: int):
    """(numa node, set of CPUs local to it) of the index-th GPU, read from sysfs WITHOUT touching HIP (the caller pins
    itself before the runtime starts): KFD topology node -> PCI address -> numa_node / local_cpulist.  None when the
    host does not expose it (containers often hide /sys/class/kfd)."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = {}
            with open(os.path.join(base, node, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) > 0:  # a GPU node (CPU nodes have no SIMDs)
                gpus.append(props)
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if visible:
            order = [int(v) for v in visible.split(",") if v.strip().isdigit()]
            gpus = [gpus[i] for i in order if i < len(gpus)]
        pr = gpus[index]
        loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
        addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        with open(f"/sys/bus/pci/devices/{addr}/local_cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        return node, cpus
    except Exception:
        return None


# ------------------------------------------------------------------------------------------ CPU baseline (oracle)
def _mfcc_chunk(chunk):  # process-pool worker: module-level so that spawn can import it
    from oracle import psf_mfcc as o_mfcc

    return o_mfcc.collate_pcm16(chunk)


def _median(xs):
    return float(np.median(np.asarray(xs, dtype=np.float64)))


def cpu_baseline(clips: np.ndarray, blob: np.ndarray, runs: int = 5, budget_s: float = 25.0):
    """The CPU oracle timed on this host, as the reference runs: per-clip NumPy/SciPy psf-equivalent MFCC in a Python
    loop (single process, and in a pool of worker processes as the reference's DataLoader(num_workers=8) does,
    train.py:108-121) + torch-CPU DS-CNN.  Thread / worker counts 1, 8, 32 and all cores are tried and the best is
    reported with its count (256 oversubscribed threads were 6x slower than 8 in round 1); every figure is the median
    of `runs` runs.  Checker code, used here only as the baseline.  Returns (dict, logits of `clips`)."""
    import torch

    from oracle import dscnn as o_dscnn
    from oracle import psf_mfcc as o_mfcc

    t_begin = time.perf_counter()
    cores = host_cores()
    counts = sorted({c for c in (1, 8, 32, cores) if c <= cores})
    pool_sizes = sorted({c for c in (8, 32) if c <= cores})  # spawning hundreds of interpreters costs more than it could return
    state, off = {}, 0
    for k, shp in o_dscnn.state_shapes(NUM_CLASSES).items():
        n = int(np.prod(shp))
        state[k] = torch.from_numpy(blob[off:off + n].reshape(shp).copy())
        off += n
    n = len(clips)
    o_mfcc.collate_pcm16(clips[:4])  # warm caches / imports

    # --- MFCC: single process
    n1 = min(n, 256)
    t_single = []
    for _ in range(runs):
        t0 = time.perf_counter()
        o_mfcc.collate_pcm16(clips[:n1])
        t_single.append(time.perf_counter() - t0)
    mfcc_rates = {1: n1 / _median(t_single)}
    feats = o_mfcc.collate_pcm16(clips)
    # --- MFCC: worker processes (spawn: the parent has initialised HIP, so no fork)
    pool_note = None
    try:
        import multiprocessing as mp

        ctx = mp.get_context("spawn")
        for workers in pool_sizes:
            if time.perf_counter() - t_begin > budget_s * 0.4:
                break
            chunks = np.array_split(clips, workers * 2)
            pool = ctx.Pool(workers)
            try:
                pool.map(_mfcc_chunk, [c[:2] for c in chunks])  # start the workers, import numpy/scipy
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    pool.map(_mfcc_chunk, chunks)
                    ts.append(time.perf_counter() - t0)
            finally:  # close + join: no worker (or resource tracker child) outlives the bench command
                pool.close()
                pool.join()
            mfcc_rates[workers] = n / _median(ts)
    except Exception as e:  # a host that cannot spawn workers still reports the single-process figure
        pool_note = f"worker pool unavailable: {type(e).__name__}: {e}"
    mfcc_best = max(mfcc_rates, key=mfcc_rates.get)

    # --- DS-CNN: thread sweep on a 512-clip sample, then B = n and B = 1 at the best thread count
    x = torch.from_numpy(feats)
    sweep = {}
    xs = x[: min(n, 512)]
    with torch.no_grad():
        for th in counts:
            torch.set_num_threads(th)
            o_dscnn.forward(state, xs[:8])
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                o_dscnn.predict(o_dscnn.forward(state, xs))
                ts.append(time.perf_counter() - t0)
            sweep[th] = len(xs) / min(ts)
        th_best = max(sweep, key=sweep.get)
        torch.set_num_threads(th_best)
        t_full = []
        for _ in range(runs):
            t0 = time.perf_counter()
            logits = o_dscnn.forward(state, x)
            o_dscnn.predict(logits)
            t_full.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s and len(t_full) >= 2:
                break
        t_one = []
        for i in range(max(runs, 20)):
            t0 = time.perf_counter()
            o_dscnn.predict(o_dscnn.forward(state, x[i % n:i % n + 1]))
            t_one.append(time.perf_counter() - t0)
    dscnn_rate = n / _median(t_full)
    value = 1.0 / (1.0 / mfcc_rates[mfcc_best] + 1.0 / dscnn_rate)
    one_clip_s = 1.0 / mfcc_rates[1] + _median(t_one)
    out = {
        "value": value,
        "unit": "clips/s",
        "cores": max(mfcc_best, th_best),
        "kind": "port",
        "sample": f"{n} of the step's clips: per-clip NumPy MFCC loop in {mfcc_best} process(es) + torch-CPU DS-CNN batch "
                  f"forward on {th_best} thread(s), stages back to back; medians of {len(t_full)} runs (thread / process "
                  "counts 1, 8, 32, all tried; the best is reported)",
        "cpu_model": cpu_model_string(),
        "host_cores": cores,
        "threads_best": {"mfcc_processes": mfcc_best, "dscnn_threads": th_best},
        "runs": len(t_full),
        "mfcc_clips_per_s_by_processes": {str(k): v for k, v in sorted(mfcc_rates.items())},
        "dscnn_clips_per_s_by_threads_b512": {str(k): v for k, v in sorted(sweep.items())},
        "rows": {
            f"B={n}": {"clips_per_s": value, "mfcc_clips_per_s": mfcc_rates[mfcc_best], "dscnn_clips_per_s": dscnn_rate},
            "B=1": {"clips_per_s": 1.0 / one_clip_s, "latency_ms": one_clip_s * 1e3,
                    "note": "one clip: MFCC once + DS-CNN forward at batch 1 (median of >= 20)"},
        },
        "seconds_spent": time.perf_counter() - t_begin,
    }
    if pool_note:
        out["note"] = pool_note
    return out, logits.numpy()


# ------------------------------------------------------------------------------------------ launcher (no GPU touched)
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(n: int, argv) -> int:
    """Parent of a self-launched N-rank run: starts N fresh child interpreters of this script (one per GPU), relays
    rank 0's stdout, and returns the worst exit code.  It imports neither torch nor the HIP library."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    try:
        out0, _ = procs[0].communicate(timeout=LAUNCH_TIMEOUT_S)
        codes = [procs[0].returncode] + [p.wait(timeout=60) for p in procs[1:]]
    except subprocess.TimeoutExpired:
        for p in procs:  # exactly the children this launcher started, nothing matched by name
            if p.poll() is None:
                p.kill()
        sys.stderr.write(f"bench.py launcher: ranks still running after {LAUNCH_TIMEOUT_S} s were killed\n")
        return 124
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py launcher: ranks failed (rank, code): {bad}\n")
        return max(abs(c) for _, c in bad) or 1
    return 0


# ------------------------------------------------------------------------------------------ one-GPU measurement legs
def spin_and_time(step, sync, spinup, warmup, steps):
    for _ in range(spinup + warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return time.perf_counter() - t0


def hbm_roofline(_native, kid, launches, avg_ms, B, workload):
    s = avg_ms * 1e-3
    achieved = B * BYTES_PER_CLIP / s / 1e9 if s > 0 else 0.0
    name = _native.kernel_name(kid)
    return {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_BPS / 1e9, "unit": "GB/s",
            "frac": achieved / (PEAK_HBM_BPS / 1e9), "traffic": pmc_traffic(name, workload),
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
            "algorithmic_bytes_per_launch": B * BYTES_PER_CLIP, "avg_kernel_ms": avg_ms, "launches": launches, "launches_note": "launches timed with HIP events inside the timed region (every --prof-every-th launch)",
            "note": "algorithmic HBM read (32 000 B per clip) over the kernel's duration against the 8 TB/s spec peak, the "
                    "roofline BASELINE.json declares; the kernel is LDS/VALU-bound (DESIGN.md 4.1): "
                    f"{B / s * MFCC_FLOP_PER_CLIP / 1e12 if s > 0 else 0.0:.1f} TFLOP/s of 157.3 f32"}


def dscnn_roofline(_native, launches, avg_ms, B, workload="ds-cnn"):
    s = avg_ms * 1e-3
    achieved = DSCNN_FLOP_PER_CLIP * B / s / 1e12 if s > 0 else 0.0
    executed = DSCNN_EXECUTED_BF16_FLOP_PER_CLIP * B / s / 1e12 if s > 0 else 0.0
    name = _native.kernel_name(_native.KWS_K_DSCNN)
    return {
        "kernel": name, "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_F32_TFLOPS, "traffic": pmc_traffic(name, workload),
        "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
        "algorithmic_bytes_per_launch": B * (99 * 10 * 4 + 52), "avg_kernel_ms": avg_ms, "launches": launches, "launches_note": "launches timed with HIP events inside the timed region (every --prof-every-th launch)",
        "flop_per_clip": DSCNN_FLOP_PER_CLIP,
        "math": "f32 in / f32 out; conv1 and the four 1x1 convolutions run on v_mfma_f32_32x32x16_f16 with every operand as an f16 pair "
                "(hi + residual, 22 bits) after exact per-clip power-of-two scaling (3 MFMAs per f32 k-block, f32 accumulate; scales from "
                "measured maxima and rigorous bounds: no overflow for any input; KWS_PW_PAIR_F16, the bf16 triple stays selectable); "
                "achieved/peak/frac price the ALGORITHMIC f32 flops against the f32 MFMA peak the dtype names (a courtesy figure: this "
                "formulation could exceed it); bf16_pipe prices the executed 16-bit MFMA work against the pipe it runs on -- the engineering number",
        "bf16_pipe": {"executed_tflops": executed, "peak": PEAK_BF16_TFLOPS, "frac": executed / PEAK_BF16_TFLOPS},
        # the two fractions side by side: `frac` (= frac_f32_algorithmic) is the contract's definition, frac_bf16_pipe is the
        # utilisation of the matrix pipe the kernel actually runs on -- the engineering number
        "frac_f32_algorithmic": achieved / PEAK_F32_TFLOPS, "frac_bf16_pipe": executed / PEAK_BF16_TFLOPS,
    }


def refine_report(_native, ctx, B, steps_frames_before, steps):
    """What the selective float64 refinement of the float32 front end did during the timed steps."""
    total, refined, last = ctx.frontend_stats()
    r_ms, r_n = ctx.prof_read(_native.KWS_K_MFCC_REFINE)
    return {"kernel": _native.kernel_name(_native.KWS_K_MFCC_REFINE), "log_peak_to_weakest_band_threshold": _native.FE_REFINE_SPAN_DEFAULT,
            "frames_per_step": B * 99, "frames_recomputed_per_step": last,
            "frames_recomputed_frac": (refined - steps_frames_before[1]) / max(total - steps_frames_before[0], 1),
            "avg_kernel_ms": r_ms / max(r_n, 1), "launches_timed": r_n,
            "note": "frames whose weakest mel band lies more than the threshold (natural log of power) below their largest spectral bin are "
                    "listed by the float32 kernel and recomputed in float64 by this launch (one per MFCC call, empty lists included)"}


def leg_mfcc_only(args, _native, torch, dev, B, cpu_n, precise=False):
    """BASELINE.json configs[1]: the MFCC kernel alone on 4096 clips, vs the CPU loop.  precise: the float64 front end
    (KWS_FE_F64) instead of the fast float32 kernel -- what matching psf's float64 on every input costs."""
    ctx = _native.Context(dev.index)
    kid = _native.KWS_K_MFCC_F64 if precise else _native.KWS_K_MFCC
    if precise:
        ctx.set_frontend_math(_native.FE_F64)
    clips = synth_clips(B, seed=100)
    wav = torch.from_numpy(clips).to(dev)
    feat = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
    step = lambda: ctx.mfcc_i16(wav, feat)
    for _ in range(args.spinup + args.warmup):
        step()
    ctx.sync()
    ctx.prof_enable(args.prof_every)
    ctx.prof_reset()
    steps = max(10, args.config_steps // 5) if precise else args.config_steps
    stats0 = ctx.frontend_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    ms, n = ctx.prof_read(kid)
    refine = None if precise else refine_report(_native, ctx, B, stats0, steps)
    ctx.prof_enable(False)
    out = {"workload": f"configs[1]: batch={B} synthetic uniform int16 1s/16kHz clips, device-resident, MFCC(400/160/512, 26 mel, "
                       "10 cep) -> float32 [B,1,99,10]" + (", float64 front end (KWS_FE_F64)" if precise else ""),
           "value": B * steps / dt, "unit": "clips/s", "ms_per_step": dt / steps * 1e3,
           "steps": steps, "dtype": "f64" if precise else "f32",
           "roofline": hbm_roofline(_native, kid, n, ms / max(n, 1), B, "mfcc-only-f64" if precise else "mfcc-only")}
    if refine is not None:
        out["frames_recomputed"] = refine["frames_recomputed_per_step"]
        out["refinement"] = refine
        # the same call on inputs that need the refinement: the 48 diverse golden clips tiled to the batch (tones, gated
        # bursts, chirps, four noise levels), with and without it
        try:
            g = np.load(GOLDEN)["clips"]
            mix = torch.from_numpy(np.ascontiguousarray(np.tile(g, (B // len(g) + 1, 1))[:B])).to(dev)
            res = {}
            for tag, span in (("off", 0.0), ("on", _native.FE_REFINE_SPAN_DEFAULT)):
                ctx.set_frontend_refine(span)
                for _ in range(10):
                    ctx.mfcc_i16(mix, feat)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(max(20, steps // 2)):
                    ctx.mfcc_i16(mix, feat)
                ctx.sync()
                res[tag] = (time.perf_counter() - t0) / max(20, steps // 2) * 1e3
            out["refinement"]["golden_mix"] = {"ms_per_step_refinement_off": res["off"], "ms_per_step_refinement_on": res["on"],
                                               "frames_recomputed_per_step": ctx.frontend_stats()[2], "frames_per_step": B * 99}
            # and the step's own clips with the refinement off: what the flag + the second launch cost on noise
            ctx.set_frontend_refine(0.0)
            for _ in range(10):
                step()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(max(20, steps // 2)):
                step()
            ctx.sync()
            off_ms = (time.perf_counter() - t0) / max(20, steps // 2) * 1e3
            out["refinement"]["this_workload_refinement_off"] = {"ms_per_step": off_ms, "clips_per_s": B / (off_ms * 1e-3),
                                                                 "note": "the float32 kernel alone: uniform noise stays within 9e-5 of the float64 oracle anyway, a clean tone over a quiet floor misses 1e-4 by up to 6x"}
        except Exception as e:
            out["refinement"]["golden_mix"] = {"error": f"{type(e).__name__}: {e}"}
        ctx.set_frontend_refine(_native.FE_REFINE_SPAN_DEFAULT)
        step()  # `feat` holds the step's own clips again for the parity figure below
        ctx.sync()
    if cpu_n > 0:
        from oracle import psf_mfcc as o_mfcc

        k = min(cpu_n, B)
        o_mfcc.collate_pcm16(clips[:4])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            want = o_mfcc.collate_pcm16(clips[:k])
            ts.append(time.perf_counter() - t0)
        out["cpu_baseline"] = {"value": k / _median(ts), "unit": "clips/s", "cores": 1, "kind": "port", "cpu_model": cpu_model_string(),
                               "runs": 3, "sample": f"{k} of the step's clips: per-clip NumPy MFCC loop (the reference's structure), one process"}
        out["parity_vs_cpu_max_abs_mfcc_err"] = float(np.abs(feat[:k].cpu().numpy() - want).max())
    ctx.close()
    return out


def leg_cnn_trad(args, _native, torch, dev, B, cpu_n):
    """BASELINE.json configs[2] read literally: MFCC + cnn-trad-fpool3 (build-defined) fused, 4096 clips."""
    ctx = _native.Context(dev.index)
    state = synth_cnn_trad_weights()
    ctx.load_cnn_trad(state, NUM_CLASSES)
    ctx.reserve(B)
    clips = synth_clips(B, seed=101)
    wav = torch.from_numpy(clips).to(dev)
    logits = torch.empty((B, NUM_CLASSES), dtype=torch.float32, device=dev)
    labels = torch.empty((B,), dtype=torch.int32, device=dev)
    step = lambda: ctx.infer_cnn_trad_i16(wav, logits, labels)
    steps = max(10, args.config_steps // 4)
    runs = {}
    # the default arithmetic (f16 pairs: three MFMAs per f32 k-block) and, for comparison on the same context, the exact
    # three-way bf16 split (six)
    for tag, math, products in (("bf16_triple", _native.KWS_CT_BF16_TRIPLE, 6), ("f16_pair", _native.KWS_CT_F16_PAIR, 3)):
        ctx.set_cnn_trad_math(math)
        for _ in range(max(10, args.spinup // 3) + 3):
            step()
        ctx.sync()
        ctx.prof_enable(args.prof_every)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        dt = time.perf_counter() - t0
        c_ms, c_n = ctx.prof_read(_native.KWS_K_CNNTRAD_CONV)
        d_ms, d_n = ctx.prof_read(_native.KWS_K_CNNTRAD_DENSE)
        m_ms, m_n = ctx.prof_read(_native.KWS_K_MFCC)
        ctx.prof_enable(False)
        runs[tag] = {"clips_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "conv_ms": c_ms / max(c_n, 1), "dense_ms": d_ms / max(d_n, 1),
                     "mfcc_ms": m_ms / max(m_n, 1), "launches": c_n, "products": products, "logits": logits.cpu().numpy().copy()}
    best = runs["f16_pair"]
    conv_s = best["conv_ms"] * 1e-3
    per_clip_executed = CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP // 6 * best["products"]
    executed = per_clip_executed * B / conv_s / 1e12 if conv_s > 0 else 0.0
    name = _native.kernel_name(_native.KWS_K_CNNTRAD_CONV)
    tri = runs["bf16_triple"]
    sc = max(1.0, float(np.abs(tri["logits"]).max()))
    out = {
        "workload": f"configs[2] read literally: batch={B} synthetic uniform int16 clips, device-resident, MFCC + cnn-trad-fpool3 "
                    "(build-defined: the reference only names it; SAME padding on the 99x10 map, 12 classes, random-init) -> logits+label",
        "value": best["clips_per_s"], "unit": "clips/s", "ms_per_step": best["ms_per_step"], "steps": steps, "dtype": "f32",
        "roofline": {
            "kernel": name, "bound": "mfma", "achieved": executed, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": executed / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(name, "cnn-trad-fpool3"),
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/pmc_traffic.json)",
            "algorithmic_bytes_per_launch": B * (99 * 10 * 4 + 64 * 297 * 4), "avg_kernel_ms": conv_s * 1e3, "launches": best["launches"],
            "flop_per_clip_executed_f16": per_clip_executed, "flop_per_clip_algorithmic": CNNTRAD_CONV_FLOP_PER_CLIP,
            "algorithmic_tflops": CNNTRAD_CONV_FLOP_PER_CLIP * B / conv_s / 1e12 if conv_s > 0 else 0.0,
            "math": "f32 in / f32 out; both convolutions and the 19008 -> 32 layer on v_mfma_f32_32x32x16_f16 with every operand as an f16 "
                    "pair hi + lo' 2^-11 after exact power-of-two scaling (3 MFMAs per f32 k-block, per-clip scales from rigorous bounds: "
                    "no overflow for any input); achieved/peak count the f16 MFMA work executed against the dense 16-bit peak"},
        "other_kernels_ms": {_native.kernel_name(_native.KWS_K_MFCC): best["mfcc_ms"],
                             _native.kernel_name(_native.KWS_K_CNNTRAD_DENSE): best["dense_ms"]},
        "bf16_triple_same_context": {"clips_per_s": tri["clips_per_s"], "ms_per_step": tri["ms_per_step"], "conv_ms": tri["conv_ms"],
                                     "dense_ms": tri["dense_ms"],
                                     "note": "the exact three-way bf16 split (six MFMAs per k-block), kws_set_cnn_trad_math(KWS_CT_BF16_TRIPLE)"},
        "f16_pair_vs_bf16_triple_max_abs_logit_diff_over_scale": float(np.abs(best["logits"] - tri["logits"]).max() / sc),
        "parity": "build-defined model: parity unpinned against the reference; checked against its own CPU definition (oracle/cnn_trad.py)",
    }
    if cpu_n > 0:
        from oracle import cnn_trad as o_ct
        from oracle import psf_mfcc as o_mfcc

        k = min(cpu_n, B, 128)
        torch.set_num_threads(min(8, host_cores()))
        t0 = time.perf_counter()
        feats = o_mfcc.collate_pcm16(clips[:k])
        want = o_ct.forward(o_ct.unflatten_state(state, NUM_CLASSES), torch.from_numpy(feats)).numpy()
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / dtc, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port", "runs": 1,
                               "cpu_model": cpu_model_string(),
                               "sample": f"{k} of the step's clips: oracle MFCC (NumPy, per clip) + torch-CPU cnn-trad-fpool3"}
        scale = max(1.0, float(np.abs(want).max()))
        out["parity_vs_cpu_max_abs_logit_err_over_scale"] = float(np.abs(logits[:k].cpu().numpy() - want).max() / scale)
    ctx.close()
    return out


def leg_batch1(args, _native, torch, dev, blob, cpu_check):
    """BASELINE.json configs[0]: batch = 1 -- the reference's own CPU-runnable case ("10-keyword cnn-trad-fpool3, batch=1 on
    reference CPU path").  GPU latency of ONE clip wav -> label (host wall time from the call to the label being complete,
    the clip already in device memory) for the reference's model (DS-CNN) and for the build-defined cnn-trad-fpool3, next
    to the same clip on the CPU oracle (per-clip NumPy MFCC + torch-CPU forward at batch 1)."""
    out = {"workload": "configs[0]: batch=1, one synthetic 1s/16kHz clip, wav -> label; GPU latency (device-resident clip, call -> label "
                       "complete) beside the CPU path at batch 1"}
    clips = synth_clips(64, seed=103)
    wav = torch.from_numpy(clips).to(dev)
    logits = torch.empty((1, NUM_CLASSES), dtype=torch.float32, device=dev)
    labels = torch.empty((1,), dtype=torch.int32, device=dev)
    ct_state = synth_cnn_trad_weights()
    for model in ("ds-cnn", "cnn-trad-fpool3"):
        ctx = _native.Context(dev.index)
        if model == "ds-cnn":
            ctx.load_dscnn(blob, NUM_CLASSES)
            call = lambda i: ctx.infer_i16(wav[i:i + 1], logits, labels)
        else:
            ctx.load_cnn_trad(ct_state, NUM_CLASSES)
            call = lambda i: ctx.infer_cnn_trad_i16(wav[i:i + 1], logits, labels)
        ctx.reserve(1)
        for i in range(300):
            call(i % 64)
        ctx.sync()
        lat = []
        for i in range(400):
            t0 = time.perf_counter()
            call(i % 64)
            ctx.sync()
            lat.append((time.perf_counter() - t0) * 1e6)
        lat = np.array(lat)
        row = {"gpu_latency_us_p50": float(np.percentile(lat, 50)), "gpu_latency_us_p99": float(np.percentile(lat, 99)),
               "gpu_clips_per_s_at_batch_1": 1e6 / float(np.percentile(lat, 50))}
        ctx.prof_enable(1)
        ctx.prof_reset()
        for i in range(50):
            call(i % 64)
        ctx.sync()
        kern = {}
        for kid in range(7):
            ms, n = ctx.prof_read(kid)
            if n:
                kern[_native.kernel_name(kid)] = ms / n * 1e3
        row["kernel_us"] = kern
        ctx.prof_enable(False)
        got = logits.cpu().numpy().copy()  # clip 49 (the last call)
        if cpu_check:
            import torch as _t

            from oracle import psf_mfcc as o_mfcc

            _t.set_num_threads(1)
            if model == "ds-cnn":
                from oracle import dscnn as o_net

                state, off = {}, 0
                for k, shp in o_net.state_shapes(NUM_CLASSES).items():
                    n = int(np.prod(shp))
                    state[k] = _t.from_numpy(blob[off:off + n].reshape(shp).copy())
                    off += n
                fwd = lambda x: o_net.forward(state, x)
            else:
                from oracle import cnn_trad as o_net

                state = o_net.unflatten_state(ct_state, NUM_CLASSES)
                fwd = lambda x: o_net.forward(state, x)
            ts = []
            with _t.no_grad():
                for i in range(12):
                    t0 = time.perf_counter()
                    want = fwd(_t.from_numpy(o_mfcc.collate_pcm16(clips[49:50])))
                    ts.append(time.perf_counter() - t0)
            row["cpu_latency_ms_median"] = _median(ts[2:]) * 1e3
            row["cpu_threads"] = 1
            row["max_abs_logit_err_vs_cpu_over_scale"] = float(np.abs(got - want.numpy()).max() / max(1.0, float(want.abs().max())))
        out[model] = row
        ctx.close()
    out["value"] = out["ds-cnn"]["gpu_latency_us_p50"]
    out["unit"] = "us p50 per clip (DS-CNN, batch 1)"
    out["higher_is_better"] = False
    out["cpu_model"] = cpu_model_string()
    out["note"] = ("latency, not throughput: one clip occupies 5 workgroups of the MFCC kernel and ONE of the DS-CNN kernel (1 of 256 CUs); "
                   "the throughput configurations are C2-C4")
    return out


def leg_dscnn_shard(args, _native, torch, dev, blob, B):
    """BASELINE.json configs[3]'s per-GPU shape: DS-CNN fused on one 1024-clip shard of the 8192-clip batch."""
    ctx = _native.Context(dev.index)
    ctx.load_dscnn(blob, NUM_CLASSES)
    ctx.reserve(B)
    wav = torch.from_numpy(synth_clips(B, seed=102)).to(dev)
    logits = torch.empty((B, NUM_CLASSES), dtype=torch.float32, device=dev)
    labels = torch.empty((B,), dtype=torch.int32, device=dev)
    step = lambda: ctx.infer_i16(wav, logits, labels)
    for _ in range(4 * args.spinup + args.warmup):  # a 1024-clip step is a quarter of the headline's: same spin-up time
        step()
    ctx.sync()
    ctx.prof_enable(args.prof_every)
    ctx.prof_reset()
    steps = args.config_steps * 2
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    k_ms, k_n = ctx.prof_read(_native.KWS_K_DSCNN)
    m_ms, m_n = ctx.prof_read(_native.KWS_K_MFCC)
    ctx.prof_enable(False)
    ctx.close()
    return {"workload": f"configs[3] per-GPU shape: DS-CNN 12-class, one {B}-clip shard of the 8192-clip batch (8 x 1024), fused wav->label, "
                        "device-resident; the 8-GPU figure is `python bench.py --gpus 8 --total-batch 8192`",
            "value": B * steps / dt, "unit": "clips/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f32",
            "roofline": dscnn_roofline(_native, k_n, k_ms / max(k_n, 1), B, "ds-cnn-1024"),
            "mfcc_kernel_ms": m_ms / max(m_n, 1),
            "note": "1024 clips = 4 workgroups per CU for the DS-CNN kernel (one resident at a time): the tail round is a quarter of the launch"}


def leg_stream(args, _native, torch, dev, blob, S, hops, cpu_check):
    """BASELINE.json configs[4]: S concurrent streams, 10 ms hops, latency of one push (host wall time from
    kws_stream_push_i16 to the labels being complete, the hop's 160 samples per stream already in device memory)."""
    hops = hops + 60  # the eager run's last 60 pushes carry profiling events and are not in its latency figures
    pcm_host = np.random.default_rng(7).integers(-32768, 32768, size=(hops, S, 160), dtype=np.int16)
    pcm_host[:, 1] //= 64  # one quiet stream
    pcm = torch.from_numpy(pcm_host).to(dev)
    res = {}
    final_logits = None
    # "eager": push -> kws_sync (results complete in DEVICE memory); "host_results": push -> kws_stream_wait_host (the kernel
    # wrote logits + labels to pinned host memory and raised a flag there: results readable by the HOST, no synchronise, no copy)
    for mode, use_graph, cluster in (("eager", False, 0), ("host_results", False, 0), ("one_workgroup_per_stream", False, 1)):
        ctx = _native.Context(dev.index)
        ctx.load_dscnn(blob, NUM_CLASSES)
        ctx.stream_open(S)
        ctx.stream_cluster(cluster)
        host = mode == "host_results"
        if host:
            ctx.stream_host_results(True)
        hop = torch.empty((S, 160), dtype=torch.int16, device=dev)
        logits = torch.empty((S, NUM_CLASSES), dtype=torch.float32, device=dev)
        labels = torch.empty((S,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        primary = mode == "eager"
        warm, extra = 40, 60  # the last 60 pushes carry per-kernel events (not in the latencies)
        lat = []
        for t in range(hops):
            hop.copy_(pcm[t])
            torch.cuda.synchronize()
            if t == hops - extra and extra:
                ctx.prof_enable(args.prof_every)
                ctx.prof_reset()
            t0 = time.perf_counter()
            ctx.stream_push_i16(hop, logits, labels, use_graph=use_graph)
            if host:
                h_logits, h_labels = ctx.stream_wait_host(S)
            else:
                ctx.sync()
            lat.append((time.perf_counter() - t0) * 1e6)
        lat = np.array(lat[warm:hops - extra])
        res[mode] = {"p50_us": float(np.percentile(lat, 50)), "p90_us": float(np.percentile(lat, 90)),
                     "p99_us": float(np.percentile(lat, 99)), "mean_us": float(lat.mean()), "hops_timed": int(len(lat))}
        k_ms, k_n = ctx.prof_read(_native.KWS_K_DSCNN)
        f_ms, f_n = ctx.prof_read(_native.KWS_K_STREAM_FRAME)
        ctx.prof_enable(False)
        # one launch per push: the one-frame front end runs in the prologue of the DS-CNN kernel (f_n == 0)
        res[mode]["kernel_us"] = {_native.kernel_name(_native.KWS_K_DSCNN): k_ms / max(k_n, 1) * 1e3}
        if f_n:
            res[mode]["kernel_us"][_native.kernel_name(_native.KWS_K_STREAM_FRAME)] = f_ms / f_n * 1e3
        res[mode]["launches_per_push"] = 2 if f_n else 1
        if primary:
            res[mode]["workgroups_per_stream"] = 4 if S <= 64 else (2 if S <= 128 else 1)
            dscnn_ms, dscnn_n = k_ms / max(k_n, 1), k_n
            final_logits = logits.cpu().numpy()
        elif host:
            ctx.sync()
            res[mode]["workgroups_per_stream"] = res["eager"]["workgroups_per_stream"]
            res[mode]["host_arrays_equal_device_arrays"] = bool(np.array_equal(h_logits, logits.cpu().numpy()) and np.array_equal(h_labels, labels.cpu().numpy())
                                                                and np.array_equal(h_logits, final_logits))
        else:
            res[mode]["workgroups_per_stream"] = 1
            res[mode]["max_abs_logit_diff_vs_clustered"] = float(np.abs(final_logits - logits.cpu().numpy()).max())
        ctx.stream_close()
        ctx.close()
    best = min(res, key=lambda m: res[m]["p50_us"])
    out = {"workload": f"configs[4]: {S} concurrent streams, 10 ms hop (160 samples @ 16 kHz), per hop one MFCC frame per stream + DS-CNN over "
                       "every stream's last 99 frames; latency = push -> labels complete (host_results: readable in host memory; eager: "
                       "complete in device memory)",
           "value": res[best]["p50_us"], "unit": "us p50 per hop", "higher_is_better": False, "p99_us": res[best]["p99_us"], "mode": best,
           "eager": res["eager"], "host_results": res["host_results"], "one_workgroup_per_stream": res["one_workgroup_per_stream"],
           "hipgraph": "not used for the one-launch push: replaying a one-node graph is 8 us slower than the plain launch on this stack "
                       "(tools/graph_overhead.hip, profiles/r03_graph_overhead.txt)",
           "real_time_factor_p50": 10000.0 / res[best]["p50_us"], "dtype": "f32",
           "roofline": dscnn_roofline(_native, dscnn_n, dscnn_ms, S, "stream"),
           "note": f"latency-bound: one launch per push; up to 64 streams every stream's DS-CNN is cut into 4 time tiles (one workgroup each, halos "
                   "recomputed, pooled partial sums combined by the last workgroup to arrive), so 64 streams occupy 256 CUs; the stream's new MFCC "
                   "frame is computed in the prologue of its last tile's workgroup; the roofline fraction is reported for completeness"}
    if cpu_check:
        import torch as _t

        from oracle import dscnn as o_dscnn
        from oracle import psf_mfcc as o_mfcc

        state, off = {}, 0
        for k, shp in o_dscnn.state_shapes(NUM_CLASSES).items():
            n = int(np.prod(shp))
            state[k] = _t.from_numpy(blob[off:off + n].reshape(shp).copy())
            off += n
        pick = [0, 1, S // 2, S - 1]
        want = np.zeros((len(pick), 1, 99, 10), np.float32)
        newest = hops - 3  # newest complete frame of the continuous signal after `hops` pushes
        for i, s in enumerate(pick):
            sig = o_mfcc.pcm16_to_float(np.ascontiguousarray(pcm_host[:, s].reshape(-1)))
            allf = o_mfcc.mfcc(sig, o_mfcc.FrontendSpec(n_samples=len(sig)))
            want[i, 0] = allf[newest - 98:newest + 1]
        ref = o_dscnn.forward(state, _t.from_numpy(want)).numpy()
        out["parity_vs_cpu_max_abs_logit_err"] = float(np.abs(final_logits[pick] - ref).max())
        out["parity_streams_checked"] = pick
    return out


# ------------------------------------------------------------------------------------------ worker (one rank)
def worker(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    numa = None
    if args.ingest == "host" and not args.selftest_cpu:
        # host-fed run: this rank's host threads (pack pool) and its pinned staging rings belong on the GPU's NUMA node --
        # pin the process BEFORE torch / HIP start (threads and first-touch allocations inherit it)
        numa = gpu_numa_cpus(local_rank)
        if numa and numa[1] and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, numa[1] & os.sched_getaffinity(0) or os.sched_getaffinity(0))
            except OSError:
                numa = None
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: either unset WORLD_SIZE (bench.py launches its own ranks) "
                         f"or start it with torch.distributed.run --nproc-per-node {args.gpus}")
    import torch

    cpu_selftest = args.selftest_cpu
    if not cpu_selftest and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is a HIP library with no CPU fallback")
    dist = None
    use_nccl = (not cpu_selftest) and args.dist_backend == "nccl"
    if cpu_selftest:
        dev = torch.device("cpu")
        dev_index = -1
    else:
        # one GPU per rank; on a box with fewer GPUs than ranks (a rehearsal of the N > 1 path) ranks share devices
        n_dev = torch.cuda.device_count()
        dev_index = local_rank % max(n_dev, 1)
        if use_nccl and n_dev < world:
            raise SystemExit(f"--dist-backend nccl needs one GPU per rank ({n_dev} visible, {world} ranks)")
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        saved_stdout = os.dup(1)  # gloo's C++ side announces its connections on stdout: keep stdout for the one JSON line
        os.dup2(2, 1)
        try:
            if use_nccl:
                dist.init_process_group(backend="nccl", device_id=dev)
            else:
                dist.init_process_group(backend="gloo")
            dist.barrier(device_ids=[dev_index]) if use_nccl else dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    ctl = dev if use_nccl else torch.device("cpu")  # where the control-plane scalars live

    def barrier():
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    def reduce_max(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(x: float):
        if world == 1:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device=ctl)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [float(o.item()) for o in outs]

    # ---- per-rank shard of the job -------------------------------------------------------------------
    if args.total_batch:
        total = args.total_batch
        scaling = "strong"
    else:
        total = args.batch * world
        scaling = "weak"
    lo, hi = shard_bounds(total, world, rank)
    B = hi - lo
    if B <= 0:
        raise SystemExit(f"rank {rank}: empty shard of {total} clips over {world} ranks")
    clips = synth_clips(total if scaling == "strong" else B, seed=0 if scaling == "strong" else rank)
    if scaling == "strong":
        clips = clips[lo:hi]
    blob, golden = bench_weights()

    if cpu_selftest:
        # launcher / harness rehearsal on CPU (tests/test_multi_rank_cpu.py): no kernels, a checksum per shard as the "step"
        acc = {"sum": 0}

        def step():
            acc["sum"] = int(clips.astype(np.int64).sum())

        sync = lambda: None
        _native = ctx = None
    else:
        from kws import _native

        ctx = _native.Context(dev_index)
        if args.frontend_math == "f64":
            ctx.set_frontend_math(_native.FE_F64)
        feat_out = None
        ct_state = None
        wav = torch.from_numpy(np.ascontiguousarray(clips)).to(dev)
        logits = torch.empty((B, NUM_CLASSES), dtype=torch.float32, device=dev)
        labels = torch.empty((B,), dtype=torch.int32, device=dev)
        host_logits = host_labels = None
        if args.ingest == "host":
            if args.model != "ds-cnn":
                raise SystemExit("--ingest host is the DS-CNN wav -> label path (kws_infer_host_i16)")
            ctx.load_dscnn(blob, NUM_CLASSES)
            host_clips = np.ascontiguousarray(clips)              # pageable host memory, as a DataLoader hands batches over
            host_logits = np.empty((B, NUM_CLASSES), np.float32)
            host_labels = np.empty((B,), np.int32)
            step = lambda: ctx.infer_host_i16(host_clips, host_logits, host_labels)
        elif args.model == "mfcc-only":
            feat_out = torch.empty((B, 1, 99, 10), dtype=torch.float32, device=dev)
            step = lambda: ctx.mfcc_i16(wav, feat_out)
        elif args.model == "cnn-trad-fpool3":
            ct_state = synth_cnn_trad_weights()
            ctx.load_cnn_trad(ct_state, NUM_CLASSES)
            step = lambda: ctx.infer_cnn_trad_i16(wav, logits, labels)
        else:
            ctx.load_dscnn(blob, NUM_CLASSES)
            if args.pointwise_math == "triple":
                ctx.set_pointwise_math(_native.PW_SPLIT_BF16)
            step = lambda: ctx.infer_i16(wav, logits, labels)
        ctx.reserve(B)
        sync = torch.cuda.synchronize
        for _ in range(args.spinup + args.warmup):
            step()
        ctx.sync()
        ctx.prof_enable(args.prof_every)
        ctx.prof_reset()
        stats0 = ctx.frontend_stats()

    t_local0 = time.perf_counter()
    elapsed = timed_steps(step, args.steps, barrier, sync, reduce_max)
    local_elapsed = time.perf_counter() - t_local0
    per_rank = gather_floats(B * args.steps / local_elapsed)
    shard_sizes = [int(v) for v in gather_floats(float(B))]
    devices = [int(v) for v in gather_floats(float(dev_index))]

    if cpu_selftest:
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test (CPU, gloo): no kernel ran", "selftest": True, "invalid_for_measurement": True,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": scaling, "value": total * args.steps / elapsed,
                              "unit": "clips/s", "shards": shard_sizes, "per_rank_clips_per_s": per_rank,
                              "max_over_ranks_s": elapsed}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    k_ms, k_n = ctx.prof_read(_native.KWS_K_DSCNN)
    mfcc_kid = _native.KWS_K_MFCC_F64 if args.frontend_math == "f64" else _native.KWS_K_MFCC
    m_ms, m_n = ctx.prof_read(mfcc_kid)
    c_ms, c_n = ctx.prof_read(_native.KWS_K_CNNTRAD_CONV)
    d_ms, d_n = ctx.prof_read(_native.KWS_K_CNNTRAD_DENSE)
    ctx.prof_enable(False)

    if rank == 0:
        value = total * args.steps / elapsed
        dscnn_ms, mfcc_ms = k_ms / max(k_n, 1), m_ms / max(m_n, 1)
        shard_note = (f"{world} independent shard(s) of one {total}-clip batch, no data-path collective" if scaling == "strong"
                      else f"{world} independent shard(s) of {B} clips each, no data-path collective")
        common = {
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        }
        cfg = {"clips_per_gpu_per_step": shard_sizes if scaling == "strong" else B, "global_batch": total, "sharding": shard_note,
               "spinup_steps": args.spinup, "launcher": os.environ.get("KWS_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "single process")}
        multi = {"per_rank_clips_per_s": per_rank, "max_over_ranks_s": elapsed, "devices": devices,
                 "control_plane": ("nccl (RCCL)" if use_nccl else "gloo") + ": barrier + max-over-ranks of the timed region only"}
        if len(set(devices)) < world:
            multi["invalid_for_measurement"] = "ranks share a GPU: a rehearsal of the N > 1 path, not a scaling point"
        if args.model == "mfcc-only":
            out = {"metric": "1s 16kHz clips/sec, MFCC only (wav->features)", "value": value, "unit": "clips/s", **common,
                   "config": {"workload": f"configs[1]: batch={B}/GPU synthetic uniform int16 1s/16kHz clips, device-resident, "
                                          "MFCC(400/160/512, 26 mel, 10 cep) -> float32 [B,1,99,10]", **cfg},
                   "roofline": hbm_roofline(_native, mfcc_kid, m_n, mfcc_ms, B, "mfcc-only"), **multi}
            if world == 1 and args.cpu_sample > 0:
                from oracle import psf_mfcc as o_mfcc

                n = min(args.cpu_sample, B, 512)
                o_mfcc.collate_pcm16(clips[:4])
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    want = o_mfcc.collate_pcm16(clips[:n])
                    ts.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {"value": n / _median(ts), "unit": "clips/s", "cores": 1, "kind": "port", "runs": 3,
                                       "cpu_model": cpu_model_string(),
                                       "sample": f"{n} of the step's clips: per-clip NumPy MFCC loop (the reference's structure), one process"}
                out["parity_vs_cpu_max_abs_mfcc_err"] = float(np.abs(feat_out[:n].cpu().numpy() - want).max())
        elif args.model == "cnn-trad-fpool3":
            conv_s = (c_ms / max(c_n, 1)) * 1e-3
            executed = CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP * B / conv_s / 1e12 if conv_s > 0 else 0.0
            name = _native.kernel_name(_native.KWS_K_CNNTRAD_CONV)
            out = {"metric": "1s 16kHz clips/sec end-to-end (wav->label)", "value": value, "unit": "clips/s", **common,
                   "config": {"workload": f"configs[2] read literally: batch={B}/GPU synthetic uniform int16 1s/16kHz clips, device-resident, "
                                          "MFCC + cnn-trad-fpool3 (build-defined, SAME padding on the 99x10 map, 12 classes, random-init) "
                                          "-> logits+label", **cfg},
                   "roofline": {"kernel": name, "bound": "mfma", "achieved": executed, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": executed / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(name, "cnn-trad-fpool3"),
                                "avg_kernel_ms": conv_s * 1e3, "launches": c_n,
                                "flop_per_clip_executed_bf16": CNNTRAD_CONV_EXECUTED_BF16_FLOP_PER_CLIP,
                                "flop_per_clip_algorithmic": CNNTRAD_CONV_FLOP_PER_CLIP,
                                "algorithmic_tflops": CNNTRAD_CONV_FLOP_PER_CLIP * B / conv_s / 1e12 if conv_s > 0 else 0.0},
                   "other_kernels_ms": {_native.kernel_name(_native.KWS_K_MFCC): mfcc_ms,
                                        _native.kernel_name(_native.KWS_K_CNNTRAD_DENSE): d_ms / max(d_n, 1)}, **multi}
        elif args.ingest == "host":
            out = {"metric": "1s 16kHz clips/sec end-to-end (wav->label), HOST-FED: pageable host int16 in, host logits + labels out (PCIe inclusive)",
                   "value": value, "unit": "clips/s", **common,
                   "config": {"workload": f"DS-CNN end to end from HOST memory: batch={B}/GPU synthetic uniform int16 clips in pageable host memory -> "
                                          "kws_infer_host_i16 (pack threads -> pinned rings -> H2D || MFCC + DS-CNN || D2H) -> host logits+label", **cfg,
                              "ingest": "host", "numa": ({"node": numa[0], "cpus_pinned": len(numa[1])} if numa else "not exposed by this host")},
                   "not_the_headline": "the contract's `value` is the device-resident rate (run without --ingest host); this line is the PCIe-inclusive rate",
                   "h2d_GBps_per_gpu": value / world * BYTES_PER_CLIP / 1e9, **multi}
            out["labels_seen"] = {"n_classes": int(len(np.unique(host_labels)))}
        else:
            out = {
                "metric": "1s 16kHz clips/sec end-to-end (wav->label)", "value": value, "unit": "clips/s", **common,
                "config": {
                    "workload": (f"DS-CNN end to end: batch={total} synthetic uniform int16 1s/16kHz clips sharded over {world} GPU(s) "
                                 "(BASELINE configs[3] as written when 8192 over 8)" if scaling == "strong" else
                                 f"DS-CNN end to end: batch={B}/GPU synthetic uniform int16 1s/16kHz clips (the batch shape of BASELINE "
                                 "configs[2], the reference's own model -- configs[2]'s literal cnn-trad-fpool3 is under configs.C3)")
                                + ", device-resident, MFCC(400/160/512, 26 mel, 10 cep) + DS-CNN(12 classes, signal-preserving golden "
                                  "weights) -> logits+label" + (" [float64 front end]" if args.frontend_math == "f64" else "")
                                + (" [DS-CNN on the bf16 triple, not the default arithmetic: roofline.math and bf16_pipe describe the default]"
                                   if args.pointwise_math == "triple" else ""), **cfg},
                "roofline": dscnn_roofline(_native, k_n, dscnn_ms, B),
                "hbm_read": {"bytes_per_clip": BYTES_PER_CLIP, "achieved_GBps_per_gpu": value / world * BYTES_PER_CLIP / 1e9,
                             "frac_of_8TBps": value / world * BYTES_PER_CLIP / PEAK_HBM_BPS},
                "mfcc_kernel": {"kernel": _native.kernel_name(mfcc_kid), "avg_kernel_ms": mfcc_ms,
                                "clips_per_s": B / (mfcc_ms * 1e-3) if mfcc_ms > 0 else 0.0,
                                "hbm_read_frac": (B / (mfcc_ms * 1e-3) * BYTES_PER_CLIP / PEAK_HBM_BPS) if mfcc_ms > 0 else 0.0,
                                "f32_frac": (B / (mfcc_ms * 1e-3) * MFCC_FLOP_PER_CLIP / (PEAK_F32_TFLOPS * 1e12)) if mfcc_ms > 0 else 0.0},
                **multi,
            }
            if args.frontend_math != "f64":
                out["frontend_refinement"] = refine_report(_native, ctx, B, stats0, args.steps)
            lab = labels.cpu().numpy()
            out["labels_seen"] = {"n_classes": int(len(np.unique(lab))), "logit_std_across_clips": float(logits.std(dim=0).mean().item())}
            if golden is not None and not args.no_parity:  # the 48 diverse golden clips against the imported reference model's logits (data file)
                gw = torch.from_numpy(np.ascontiguousarray(golden["clips"])).to(dev)
                gl = torch.empty((gw.shape[0], NUM_CLASSES), dtype=torch.float32, device=dev)
                gy = torch.empty((gw.shape[0],), dtype=torch.int32, device=dev)
                ctx.infer_i16(gw, gl, gy)
                ctx.sync()
                out["parity_golden"] = {
                    "max_abs_logit_err": float(np.abs(gl.cpu().numpy() - golden["he.logits"][8:]).max()),
                    "labels_identical": bool(np.array_equal(gy.cpu().numpy(), golden["he.label"][8:])),
                    "n_classes": int(len(set(golden["he.label"][8:].tolist()))), "clips": int(gw.shape[0]),
                    "source": "tests/golden/e2e_golden.npz: imported reference DS-CNN on the oracle's MFCC of 48 diverse clips"}
            if world == 1 and not args.no_parity and ctx is not None:
                # the same step on the exact three-way bf16 split (six MFMAs per f32 k-block), same context, after the timed region
                pair_logits = logits.clone()
                ctx.set_pointwise_math(_native.PW_SPLIT_BF16)
                for _ in range(30):
                    ctx.infer_i16(wav, logits, labels)
                ctx.sync()
                ctx.prof_enable(args.prof_every)   # the same event sampling as the timed region
                ctx.prof_reset()
                t0 = time.perf_counter()
                for _ in range(40):
                    ctx.infer_i16(wav, logits, labels)
                ctx.sync()
                dt3 = time.perf_counter() - t0
                t_ms, t_n = ctx.prof_read(_native.KWS_K_DSCNN)
                ctx.prof_enable(False)
                ctx.set_pointwise_math(_native.PW_DEFAULT)
                sc = max(1.0, float(logits.abs().max().item()))
                out["bf16_triple_same_context"] = {
                    "clips_per_s": B * 40 / dt3, "ms_per_step": dt3 / 40 * 1e3, "dscnn_kernel_ms": t_ms / max(t_n, 1),
                    "max_abs_logit_diff_vs_f16_pair_over_scale": float((pair_logits - logits).abs().max().item()) / sc,
                    "labels_identical": bool(torch.equal(pair_logits.argmax(dim=1), logits.argmax(dim=1))),
                    "note": "kws_set_pointwise_math(KWS_PW_SPLIT_BF16): the arithmetic of rounds 1-2, every f32 operand as three bf16 pieces"}
                ctx.infer_i16(wav, logits, labels)  # leave the default arithmetic's results in the buffers
                ctx.sync()
            if world == 1 and args.cpu_sample > 0:
                n = min(args.cpu_sample, B)
                base, cpu_logits = cpu_baseline(clips[:n], blob)
                out["cpu_baseline"] = base
                out["parity_vs_cpu_max_abs_logit_err"] = float(np.abs(logits[:n].cpu().numpy() - cpu_logits).max())
            if world == 1 and args.configs == "all":
                ctx.close()
                ctx = None
                cpu_n = args.cpu_sample
                legs = {}
                for name, fn in (("C1_batch1", lambda: leg_batch1(args, _native, torch, dev, blob, cpu_n > 0)),
                                 ("C2_mfcc_only", lambda: leg_mfcc_only(args, _native, torch, dev, 4096, min(cpu_n, 256))),
                                 ("C2_mfcc_only_float64", lambda: leg_mfcc_only(args, _native, torch, dev, 4096, min(cpu_n, 256), precise=True)),
                                 ("C3_cnn_trad_fpool3", lambda: leg_cnn_trad(args, _native, torch, dev, 4096, min(cpu_n, 64))),
                                 ("C4_dscnn_shard_1024", lambda: leg_dscnn_shard(args, _native, torch, dev, blob, 1024)),
                                 ("C5_stream_64", lambda: leg_stream(args, _native, torch, dev, blob, 64, args.stream_hops, cpu_n > 0))):
                    try:
                        legs[name] = fn()
                    except Exception as e:  # a failing side leg must not cost the headline line
                        legs[name] = {"error": f"{type(e).__name__}: {e}"}
                out["configs"] = legs
        print(json.dumps(out), flush=True)

    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=60,
                    help="untimed steps before the warm-up: from idle the GPU's clocks take ~30 steps (20 ms) to settle "
                         "(per-step time 0.82 -> 0.67 ms, tools/ramp_probe.py); reported as config.spinup_steps")
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU per step (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling: one batch of this many clips sharded over the GPUs (8192 = BASELINE configs[3] as written)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="clips timed on the CPU baseline (0 = skip every CPU leg)")
    ap.add_argument("--model", choices=["ds-cnn", "cnn-trad-fpool3", "mfcc-only"], default="ds-cnn",
                    help="ds-cnn: the reference's model (the driver's line); cnn-trad-fpool3: the build-defined model "
                         "BASELINE.json configs[2] names (parity unpinned against the reference, DESIGN.md 4.5); "
                         "mfcc-only: BASELINE.json configs[1], the front end alone, priced against the HBM-read roofline")
    ap.add_argument("--pointwise-math", choices=["pair", "triple"], default="pair",
                    help="DS-CNN GEMM arithmetic of the timed steps: pair = KWS_PW_PAIR_F16 (the library's default), triple = KWS_PW_SPLIT_BF16 "
                         "(rounds 1-2; for counter passes and A/B runs -- the JSON line says which)")
    ap.add_argument("--configs", choices=["all", "none"], default="all",
                    help="all: at N = 1 the ds-cnn line also measures the other BASELINE configurations under `configs`")
    ap.add_argument("--config-steps", type=int, default=100, help="timed steps of each side configuration")
    ap.add_argument("--stream-hops", type=int, default=340, help="pushes per mode of the streaming configuration (first 40 untimed)")
    ap.add_argument("--dist-backend", choices=["gloo", "nccl"], default="gloo",
                    help="control plane of an N > 1 run (barrier and the max-over-ranks of the timed region; the data path has no "
                         "collective): gloo over 127.0.0.1 by default -- nothing to gain from RCCL for two scalars -- or nccl (= RCCL)")
    ap.add_argument("--frontend-math", choices=["f32", "f64"], default="f32",
                    help="f32: the fast front end (default, the headline); f64: KWS_FE_F64, float64 after framing as psf computes it")
    ap.add_argument("--ingest", choices=["device", "host"], default="device",
                    help="device: inputs resident in HBM when the timed region starts (the contract's `value`); host: the timed step is "
                         "kws_infer_host_i16 on pageable host memory (PCIe inclusive), each rank pinned to its GPU's NUMA node")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the extra launch on the golden clips after the timed region (profile runs: exact launch counts)")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="HIP events around every n-th launch of each kernel inside the timed region (roofline.avg_kernel_ms); "
                         "a pair per launch costs the stream ~7 us = 2 %% of a 4096-clip step (tools/prof_overhead.py)")
    ap.add_argument("--selftest-cpu", action="store_true", help=argparse.SUPPRESS)  # launcher rehearsal on CPU (gloo), tests only
    return ap.parse_args(argv)


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["KWS_BENCH_LAUNCHER"] = "bench.py (self-launched, one child process per GPU)"
        return launch_children(args.gpus, argv)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
