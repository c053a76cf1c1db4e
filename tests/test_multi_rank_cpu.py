"""N>1 path on CPU: world_size-2 gloo run of the bench's sharding + timing harness."""
import json
import os
import subprocess
import sys

import pytest

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_partition():
    for total in (1, 7, 8, 4096, 8192, 10):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_bounds(8192, 8, 3) == (3072, 4096)  # BASELINE config 4: 8 x 1024


def test_two_rank_gloo_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(HERE, "_dist_worker.py")]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["world"] == 2 and out["shards"] == [[0, 5], [5, 10]]


@pytest.mark.parametrize("extra,scaling,shards", [([], "weak", [16, 16]), (["--total-batch", "25"], "strong", [13, 12])])
def test_bench_launches_its_own_ranks(extra, scaling, shards):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent starts two child ranks itself
    (rendezvous on 127.0.0.1, a free port), relays rank 0's single JSON line and exits 0.  Rehearsed on CPU with gloo
    (--selftest-cpu: no kernel runs; the line says so)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "16", "--selftest-cpu"] + extra
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), proc.stdout[-1000:]   # stdout carries the JSON line and nothing else
    out = json.loads(lines[0])
    assert out["selftest"] and out["n_gpus"] == 2 and out["scaling"] == scaling and out["shards"] == shards
    assert len(out["per_rank_clips_per_s"]) == 2 and out["max_over_ranks_s"] > 0
    assert out["value"] == pytest.approx(sum(shards) * 3 / out["max_over_ranks_s"])


def test_bench_refuses_a_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--selftest-cpu"],
                          env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "WORLD_SIZE=3" in (proc.stdout + proc.stderr)


def test_weights_and_clips_are_seeded():
    import numpy as np

    assert bench.synth_weights().shape == (26444,)
    assert bench.signal_preserving_weights().shape == (26444,)
    blob, golden = bench.bench_weights()
    assert blob.shape == (26444,) and golden is not None and np.array_equal(blob, golden["he.blob"])
    assert np.array_equal(bench.synth_clips(3, 5), bench.synth_clips(3, 5))
    assert bench.synth_clips(2, 0).dtype == np.int16


def test_cnn_trad_bench_weights_match_the_oracle_layout():
    """bench.py builds the cnn-trad-fpool3 blob itself (the oracle may only serve its cpu_baseline leg): same size and
    tensor order as the oracle's state_dict, so unflatten_state reads it back tensor by tensor."""
    import numpy as np

    from oracle import cnn_trad as o_ct

    blob = bench.synth_cnn_trad_weights(seed=3)
    shapes = o_ct.state_shapes(bench.NUM_CLASSES)
    assert blob.dtype == np.float32 and blob.size == sum(int(np.prod(s)) for s in shapes.values())
    state = o_ct.unflatten_state(blob, bench.NUM_CLASSES)
    assert list(state) == list(shapes) and all(tuple(state[k].shape) == shapes[k] for k in shapes)
    assert np.array_equal(o_ct.flatten_state(state), blob)
    assert abs(float(state["conv2.weight"].std()) - (2.0 / 2560) ** 0.5) < 2e-3  # fan-in scaled


def test_scale_sweep_builds_the_table_from_fresh_children(tmp_path):
    """tools/scale_sweep.py: every point a fresh `bench.py --gpus N` child (self-launched ranks for N > 1), the rows parsed
    from the one JSON line each prints, the north_star table rendered -- rehearsed on CPU with two --selftest-cpu points
    per series (no kernel runs; the rows say so)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    out_json = tmp_path / "sweep.json"
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "scale_sweep.py"), "--gpus", "1,2", "--selftest-cpu",
                           "--steps", "3", "--warmup", "1", "--total-batch", "50", "--out", str(out_json)],
                          env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    res = json.loads(out_json.read_text())
    rows = res["rows"]
    assert [(r["series"].split()[0], r["n_gpus"]) for r in rows] == [("weak", 1), ("weak", 2), ("strong", 1), ("strong", 2)]
    assert all(r.get("selftest") and r["clips_per_s"] > 0 and r["per_rank_min"] <= r["per_rank_max"] for r in rows)
    assert rows[1]["scaling"] == "weak" and rows[3]["scaling"] == "strong"
    table = proc.stdout
    assert "| series | GPUs | clips/s |" in table and table.count("CPU self-test, no kernel") == 4
