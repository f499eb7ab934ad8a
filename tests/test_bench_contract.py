"""bench.py as the driver runs it: a fresh child process, ONE JSON line last on stdout (the multi-GPU
path must verify what it gathers; the build boxes have one GPU, so the path runs with a world of one rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_force_dist_bench_verifies_its_gather():
    d = _run(["--force-dist", "--steps", "3", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-sparse", "--no-other-configs"])
    assert d["gather_verified"] is True and d["mismatching_records"] == 0 and d["mismatching_values_c_abi_route"] == 0
    assert d["dist_world_size"] == 1 and d["y355_comm_world"] == 1 and d["records_compared"] == 64
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["unit"] == "images/sec" and d["higher_is_better"] is True
    assert d["config"]["input_batches_rotated"] == 4 and "gather" in d["config"]
    # at 256 detections per record the dense fixture's images are cut: the receiver can tell
    assert d["truncated_images"] == 64


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    d = _run(["--steps", "4", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-sparse", "--no-other-configs"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "same_input_every_step", "one_stream", "timing"):
        assert k in d, k
    rf = d["roofline"]
    assert list(rf)[0] == "whole_path_frac" and rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and 0 < rf["whole_path_frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    km = rf["kernel_ms"]
    assert len(km) == 13 and "pairs_kernel" in km and "conv1+conv2 (fused front end)" in km      # every launch of a step
    assert 3000 < rf["peak_measured"] < 5200                  # measured in this run, not a constant
    assert "gather_verified" not in d
