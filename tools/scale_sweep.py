#!/usr/bin/env python3
"""One command for the scaling table BASELINE.json's north_star asks for: "throughput on synthetic 16 kHz waveforms reported
at 1, 2, 4 and 8 GPUs as absolute clips/s and as fraction of the HBM-read roofline, next to the reference's own CPU path
timed on the host cores of the same box (core count stated)".

    python tools/scale_sweep.py                       # weak (4096 clips per GPU) and strong (8192 clips in all) at 1, 2, 4, 8 GPUs
    python tools/scale_sweep.py --gpus 1,2 --ingest host   # also the host-fed (PCIe-inclusive) rows
    python tools/scale_sweep.py --out profiles/r03_scale_sweep.json

Every point is `python bench.py --gpus N ...` started as a FRESH child process before anything touches a GPU (bench.py is
its own launcher for N > 1: one child interpreter per GPU, no exec after HIP init).  The N = 1 weak point is the plain
bench line, so the sweep checks itself against it.  The sweep never computes an efficiency the driver would also
compute -- it prints the absolute numbers and the ratio to N = 1 for the reader.  Points whose ranks share a GPU (a
rehearsal on a smaller box) are kept and marked invalid.  Imports neither torch nor the HIP library."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES_PER_CLIP = 32000
PEAK_HBM_BPS = 8.0e12


def run_point(n: int, extra, timeout_s: int):
    """One bench.py run as a fresh child; returns (parsed JSON line or None, seconds, stderr tail)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + list(extra)
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return None, time.perf_counter() - t0, f"timed out after {timeout_s} s"
    dt = time.perf_counter() - t0
    line = None
    for ln in reversed(p.stdout.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                line = json.loads(ln)
                break
            except json.JSONDecodeError:
                continue
    return line, dt, (p.stderr or "")[-400:] if (line is None or p.returncode) else ""


def row_of(tag: str, n: int, line, seconds: float, err: str):
    if line is None:
        return {"series": tag, "n_gpus": n, "error": err.strip().splitlines()[-1] if err.strip() else "no JSON line", "driver_seconds": seconds}
    per = line.get("per_rank_clips_per_s") or [line["value"]]
    row = {"series": tag, "n_gpus": n, "clips_per_s": line["value"], "ms_per_step": line.get("ms_per_step"),
           "per_rank_min": min(per), "per_rank_max": max(per),
           "hbm_read_frac_per_gpu": line["value"] / n * BYTES_PER_CLIP / PEAK_HBM_BPS,
           "scaling": line.get("scaling"), "steps": line.get("steps"), "driver_seconds": seconds}
    if line.get("invalid_for_measurement"):
        row["invalid_for_measurement"] = line["invalid_for_measurement"]
    if line.get("selftest"):
        row["selftest"] = True
    cb = line.get("cpu_baseline")
    if cb:
        row["cpu_baseline"] = {"clips_per_s": cb.get("value"), "cores": cb.get("cores"), "host_cores": cb.get("host_cores"), "cpu_model": cb.get("cpu_model")}
    return row


def table(rows) -> str:
    out = ["| series | GPUs | clips/s | x N=1 | per-rank min .. max | HBM-read roofline / GPU | ms/step | note |", "|---|---|---|---|---|---|---|---|"]
    base = {}
    for r in rows:
        if "error" not in r and r["n_gpus"] == 1:
            base[r["series"]] = r["clips_per_s"]
    for r in rows:
        if "error" in r:
            out.append(f"| {r['series']} | {r['n_gpus']} | - | - | - | - | - | FAILED: {r['error']} |")
            continue
        b = base.get(r["series"])
        note = "CPU self-test, no kernel" if r.get("selftest") else ("ranks shared a GPU: not a scaling point" if r.get("invalid_for_measurement") else "")
        out.append(f"| {r['series']} | {r['n_gpus']} | {r['clips_per_s']:,.0f} | {r['clips_per_s'] / b:.2f} | " if b else f"| {r['series']} | {r['n_gpus']} | {r['clips_per_s']:,.0f} | - | ")
        out[-1] += f"{r['per_rank_min']:,.0f} .. {r['per_rank_max']:,.0f} | {100 * r['hbm_read_frac_per_gpu']:.2f} % | {(r['ms_per_step'] if r['ms_per_step'] is not None else float('nan')):.4f} | {note} |"
    cpu = next((r["cpu_baseline"] for r in rows if r.get("cpu_baseline")), None)
    if cpu:
        out.append("")
        out.append(f"CPU path on the same host (the oracle's per-clip NumPy MFCC + torch-CPU DS-CNN): {cpu['clips_per_s']:,.0f} clips/s on "
                   f"{cpu['cores']} of {cpu.get('host_cores')} cores ({cpu.get('cpu_model')}).")
    return "\n".join(out)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", default="1,2,4,8", help="comma-separated GPU counts")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--total-batch", type=int, default=8192, help="the strong-scaling batch (BASELINE configs[3]); 0 = skip the strong series")
    ap.add_argument("--ingest", choices=["device", "host"], default="device", help="host: add the host-fed (PCIe-inclusive) weak series")
    ap.add_argument("--selftest-cpu", action="store_true", help="pass --selftest-cpu to bench.py (launcher rehearsal on CPU, no kernels; tests)")
    ap.add_argument("--timeout", type=int, default=1500, help="seconds per point")
    ap.add_argument("--out", default="", help="also write the rows as JSON here")
    args = ap.parse_args(argv)
    counts = [int(v) for v in args.gpus.split(",") if v.strip()]
    common = ["--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.selftest_cpu:
        common.append("--selftest-cpu")
    rows = []
    for n in counts:
        # the CPU baseline and the side configurations ride on the N = 1 weak point only (the plain bench line)
        extra = common + ([] if n == 1 else ["--cpu-sample", "0", "--configs", "none"])
        line, dt, err = run_point(n, extra, args.timeout)
        rows.append(row_of("weak (4096 clips/GPU)", n, line, dt, err))
        print(f"[scale_sweep] weak N={n}: {rows[-1].get('clips_per_s', rows[-1].get('error'))}", file=sys.stderr, flush=True)
    if args.total_batch:
        for n in counts:
            line, dt, err = run_point(n, common + ["--total-batch", str(args.total_batch), "--cpu-sample", "0", "--configs", "none"], args.timeout)
            rows.append(row_of(f"strong ({args.total_batch} clips in all)", n, line, dt, err))
            print(f"[scale_sweep] strong N={n}: {rows[-1].get('clips_per_s', rows[-1].get('error'))}", file=sys.stderr, flush=True)
    if args.ingest == "host" and not args.selftest_cpu:
        for n in counts:
            line, dt, err = run_point(n, ["--steps", str(max(10, args.steps // 10)), "--warmup", "3", "--spinup", "5", "--ingest", "host",
                                          "--batch", "16384", "--cpu-sample", "0", "--configs", "none"], args.timeout)
            rows.append(row_of("host-fed weak (16384 clips/GPU, PCIe inclusive)", n, line, dt, err))
            print(f"[scale_sweep] host-fed N={n}: {rows[-1].get('clips_per_s', rows[-1].get('error'))}", file=sys.stderr, flush=True)
    # self-check: the N = 1 weak point IS the plain bench line; two N = 1 points of one sweep must agree to a few per cent
    ones = [r for r in rows if r["n_gpus"] == 1 and "error" not in r and r["series"].startswith(("weak", "strong"))]
    check = None
    if len(ones) == 2 and not args.selftest_cpu:
        # strong at N = 1 is one 8192-clip step, weak one 4096-clip step: the same per-clip rate up to the tail round
        check = abs(ones[0]["clips_per_s"] / ones[1]["clips_per_s"] - 1.0)
    result = {"rows": rows, "n1_weak_vs_strong_rel_diff": check, "bytes_per_clip": BYTES_PER_CLIP, "peak_hbm_Bps": PEAK_HBM_BPS}
    print(table(rows))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)
    print(json.dumps(result))
    return 0 if all("error" not in r for r in rows) else 1


if __name__ == "__main__":
    sys.exit(main())
