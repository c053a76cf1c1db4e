"""Helper launched by tests/test_multi_rank_cpu.py under torch.distributed.run (gloo, CPU).

Each rank takes its contiguous shard of a seeded job, runs the CPU oracle on it as a stand-in for the
per-GPU kernel launch (the sharding/timing harness is what is under test, not the arithmetic), and
rank 0 checks that the per-shard results concatenated by index equal the unsharded run.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import dscnn as o_dscnn  # noqa: E402
from oracle import psf_mfcc as o_mfcc  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 10  # deliberately not a multiple of the world size
    clips = np.random.default_rng(123).integers(-32768, 32768, size=(total, 16000), dtype=np.int16)
    state = o_dscnn.random_state(seed=1)
    lo, hi = bench.shard_bounds(total, world, rank)
    result = {}

    def step():
        feats = torch.from_numpy(o_mfcc.collate_pcm16(clips[lo:hi]))
        result["labels"] = o_dscnn.predict(o_dscnn.forward(state, feats))

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_steps(step, 2, dist.barrier, lambda: None, reduce_max)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, result["labels"].tolist(), elapsed))
    if rank == 0:
        gathered.sort()
        assert [g[0] for g in gathered] == [bench.shard_bounds(total, world, r)[0] for r in range(world)]
        assert gathered[0][0] == 0 and gathered[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(gathered, gathered[1:]))  # contiguous, disjoint, complete
        labels = sum((g[2] for g in gathered), [])
        whole = o_dscnn.predict(o_dscnn.forward(state, torch.from_numpy(o_mfcc.collate_pcm16(clips)))).tolist()
        assert labels == whole
        assert len({g[3] for g in gathered}) == 1  # every rank reports the same (max) time
        print(json.dumps({"ok": True, "world": world, "shards": [[g[0], g[1]] for g in gathered]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
