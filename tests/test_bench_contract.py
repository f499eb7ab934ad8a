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
    assert d["mismatching_records_full"] == 0 and d["dist_backend"] == "nccl"
    assert d["dist_world_size"] == 1 and d["y355_comm_world"] == 1 and d["records_compared"] == 64
    # full records (the engine's max_det per image) are timed and verified beside the capped ones: nothing is cut there
    fr = d["gather_full_records"]
    assert fr["truncated_images"] == 0 and fr["detections_per_image"] >= 3380 and fr["value"] > 0
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
    # every launch of a step: the fused front end, conv3_1 -> conv3_2 + pool3 in one launch (round 5), six more layers, head / NMS
    assert len(km) == 12 and "pairs_kernel" in km and "conv1+conv2 (fused front end)" in km and "conv3_1+conv3_2 (fused pair)" in km
    assert rf["launch_ms_rocprof"] is None or rf["launch_ms_rocprof"] > 0
    assert 3000 < rf["peak_measured"] < 5200                  # measured in this run, not a constant
    assert "gather_verified" not in d
    # round 6: the timed region goes through the product entry point, and the hand-scheduled figure is beside it
    assert "y355_pipeline_submit" in d["config"]["entry_point"] and d["config"]["streams_per_gpu"] == 4
    assert d["hand_scheduled"]["value"] > 0 and d["value"] >= 0.6 * d["hand_scheduled"]["value"]       # 4-step regions: noisy; the 50-step figures differ by 1-2 %
    assert rf["traffic_stale"] is False and rf["launch_ms_rocprof_stale"] is False       # profiles/ holds this round's passes


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.gpu
def test_world2_bench_logic_on_one_gpu():
    """bench.py's world > 1 logic (shard_inputs(r > 0), rank 0's re-run of the other ranks' shards, MAX over
    ranks, the failure broadcast) executed with TWO ranks, launched exactly as the driver launches a multi-GPU run
    (python -m torch.distributed.run, a fresh child process), both ranks on cuda:0, the packed records staged through the host
    for gloo.  Everything but the transport is the production path; the rate is not a performance number."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu",
           "--steps", "3", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-sparse", "--no-other-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.strip().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines                           # rank 0 prints, rank 1 does not
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["dist_world_size"] == 2 and d["dist_backend"] == "gloo" and "share_gpu" in d
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "batch-shard x2"
    assert d["gather_verified"] is True and d["records_compared"] == 128
    assert d["mismatching_records"] == 0 and d["mismatching_records_full"] == 0
    assert d["mismatching_values_c_abi_route"] is None and d["y355_comm_world"] is None     # RCCL needs one GPU per rank
    assert d["truncated_images"] == 128 and d["gather_full_records"]["truncated_images"] == 0
    assert d["value"] > 0 and d["scaling"] == "weak"
    # a first N-GPU run diagnoses itself -- every rank's own rate, the N = 1 sub-result of the same process,
    # the transport's version
    pr = d["per_rank"]
    assert len(pr["value_per_rank"]) == 2 and 0 < pr["min"] <= pr["max"] and abs(pr["sum"] - sum(pr["value_per_rank"])) < 1.0
    assert pr["min"] * 2 >= d["value"] * 0.5                # the aggregate cannot be far above twice the slowest rank's own rate
    assert d["n1_same_process"]["value"] > 0 and d["n1_same_process"]["ms_per_step"] > 0
    assert isinstance(d["rccl_version"], str) and d["rccl_version"]
