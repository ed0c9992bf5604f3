"""bench.py's output contract (one JSON line on stdout with the driver's keys, `roofline` and `cpu_baseline` objects)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_stdout_stays_one_json_line_with_rccl_initialised():
    """RCCL prints a version banner to stdout when its first communicator is created; the driver parses stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "3", "--warmup", "1",
                          "--force-dist", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    # BASELINE's 8-GPU configurations ride in the same line whenever a process group is up (here: one RCCL rank, every
    # collective forced), with the exchange timed by events around its collectives
    kp = d["keyframe_parallel"]
    assert "error" not in kp, kp
    assert kp["world_size_reported_by_backend"] == 1 and kp["backend"] == "nccl"
    c5, c4 = kp["mapper_step_c5"], kp["trainer_step_c4"]
    assert c5["iters_per_s"] > 0 and c5["frozen_anchor_segment_left_out"] and c5["exchanged_MB"] < c5["bucket_MB"]
    assert c5["exchange_ms_per_step"] > 0 and c5["collectives"] and c5["dropped_steps"] == 0
    assert c4["iters_per_s"] > 0 and c4["exchange_ms_per_step"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["raster", "trainer"])
def test_bench_starts_its_own_ranks_without_a_launcher(mode):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver calls it) must start its N ranks
    itself and still print one JSON line.  Rehearsed here with two gloo ranks sharing the one GPU of the box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "3",
                          "--warmup", "2", "--backend", "gloo", "--rehearse-on-one-gpu", "--mode", mode], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and "cpu_baseline" not in d
    if mode == "raster":
        kp = d["keyframe_parallel"]
        assert "error" not in kp and kp["world_size_reported_by_backend"] == 2, kp
        assert kp["mapper_step_c5"]["iters_per_s"] > 0 and kp["trainer_step_c4"]["iters_per_s"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
