#!/usr/bin/env python3
"""BASELINE config 5: 64 concurrent streams, 10 ms hops, per-hop latency on one MI355X.

    python tools/bench_stream.py [n_streams] [hops] [eager|host|hipgraph|both] [workgroups per stream: 0 = automatic, 1, 2, 4]

A hop = 160 new int16 samples per stream already resident in device memory; latency = host wall time from
kws_stream_push_i16 to the labels being complete (kws_sync), i.e. launch + frame kernel + hop counter +
DS-CNN over every stream's last second.  Reported for eager launches, for eager launches with
zero-copy result delivery (kws_stream_host_results: latency = push -> logits and labels readable in HOST memory, no
synchronise, no copy) and for hipGraph replay.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "keyword-spotting_amd"))
import bench
from kws import _native


def run(S, hops, use_graph, cluster=0, host=False):
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.load_dscnn(bench.bench_weights()[0], 12)
    ctx.stream_open(S)
    ctx.stream_cluster(cluster)
    if host:  # zero-copy delivery: the kernel writes logits + labels to pinned host memory and raises a flag there
        ctx.stream_host_results(True)
    pcm = torch.from_numpy(np.random.default_rng(0).integers(-32768, 32768, size=(hops, S, 160), dtype=np.int16)).to(dev)
    hop = torch.empty((S, 160), dtype=torch.int16, device=dev)
    logits = torch.empty((S, 12), dtype=torch.float32, device=dev)
    labels = torch.empty((S,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    lat = []
    for t in range(hops):
        hop.copy_(pcm[t]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.stream_push_i16(hop, logits, labels, use_graph=use_graph)
        if host:
            ctx.stream_wait_host(S)   # the results are in host memory when this returns
        else:
            ctx.sync()                # the results are in device memory; a D2H copy is still to come
        lat.append((time.perf_counter() - t0) * 1e6)
    kern = None
    if not use_graph:  # the kernel's own duration (HIP events on the stream), 60 more pushes
        ctx.prof_enable(1); ctx.prof_reset()
        for t in range(60):
            hop.copy_(pcm[t % hops]); torch.cuda.synchronize()
            ctx.stream_push_i16(hop, logits, labels)
        ms, n = ctx.prof_read(_native.KWS_K_DSCNN)
        kern = ms / max(n, 1) * 1e3
        ctx.prof_enable(0)
    ctx.stream_close(); ctx.close()
    lat = np.array(lat[min(20, len(lat) // 4):])  # drop warm-up (graph build, clocks)
    out = {"p50_us": float(np.percentile(lat, 50)), "p90_us": float(np.percentile(lat, 90)), "p99_us": float(np.percentile(lat, 99)),
           "mean_us": float(lat.mean()), "hops_timed": int(len(lat)), "workgroups_per_stream": cluster}
    if kern is not None:
        out["kernel_us"] = kern
    return out


def run_host_to_host(S, hops, cluster=0):
    """The whole hop, host memory to host memory: (a) kws_stream_push_host_i16; (b) what a caller without it does --
    H2D copy of the hop, push, synchronise, D2H copies of logits and labels."""
    dev = torch.device("cuda", 0)
    pcm = np.random.default_rng(0).integers(-32768, 32768, size=(hops, S, 160), dtype=np.int16)
    out = {}
    for mode in ("push_host", "copy_push_sync_copy"):
        ctx = _native.Context(0)
        ctx.load_dscnn(bench.bench_weights()[0], 12)
        ctx.stream_open(S)
        ctx.stream_cluster(cluster)
        hop = torch.empty((S, 160), dtype=torch.int16, device=dev)
        logits = torch.empty((S, 12), dtype=torch.float32, device=dev)
        labels = torch.empty((S,), dtype=torch.int32, device=dev)
        pin = torch.empty((S, 160), dtype=torch.int16).pin_memory()
        torch.cuda.synchronize()
        lat = []
        for t in range(hops):
            t0 = time.perf_counter()
            if mode == "push_host":
                lg, lb = ctx.stream_push_host_i16(pcm[t], S)
                res = (lb.copy(), lg.copy())
            else:
                pin.numpy()[...] = pcm[t]
                hop.copy_(pin, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                ctx.stream_push_i16(hop, logits, labels)
                ctx.sync()
                res = (labels.cpu().numpy(), logits.cpu().numpy())
            lat.append((time.perf_counter() - t0) * 1e6)
        last = res
        ctx.stream_close(); ctx.close()
        lat = np.array(lat[min(20, len(lat) // 4):])
        out[mode] = {"p50_us": float(np.percentile(lat, 50)), "p90_us": float(np.percentile(lat, 90)), "p99_us": float(np.percentile(lat, 99)),
                     "mean_us": float(lat.mean()), "hops_timed": int(len(lat))}
        out[mode + "_last"] = last
    same = bool(np.array_equal(out["push_host_last"][0], out["copy_push_sync_copy_last"][0]) and np.array_equal(out["push_host_last"][1], out["copy_push_sync_copy_last"][1]))
    return {"push_host": out["push_host"], "copy_push_sync_copy": out["copy_push_sync_copy"], "results_identical": same,
            "note": "latency = host int16 hop in a numpy array -> labels + logits in numpy arrays (copies included)"}


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    hops = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    which = sys.argv[3] if len(sys.argv) > 3 else "both"   # one mode only: exactly `hops` launches per kernel (profiles)
    cluster = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out = {"config": f"C5: {S} concurrent streams, 10 ms hop (160 samples @ 16 kHz), MFCC frame + DS-CNN over the last 99 frames per hop"}
    if which in ("eager", "both"):
        out["eager"] = run(S, hops, False, cluster)
    if which in ("host", "both"):
        out["eager_host_results"] = run(S, hops, False, cluster, host=True)
    if which in ("host", "both"):
        out["host_to_host"] = run_host_to_host(S, hops, cluster)
    if which in ("hipgraph", "both"):
        out["hipgraph"] = run(S, hops, True, cluster)
    out["real_time_factor_p50"] = 10000.0 / min(v["p50_us"] for k, v in out.items() if k in ("eager", "hipgraph", "eager_host_results"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
