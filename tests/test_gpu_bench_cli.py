"""GPU (`-m gpu`): bench.py's command-line contract, end to end in subprocesses (short runs).

* N = 1: ONE JSON line on stdout with the driver's keys, the `roofline` / `cpu_baseline`-style objects present.
* N = 2 WITHOUT a launcher: `python bench.py --gpus 2` re-executes itself under torch.distributed.run, one rank per
  process, barriers and max-over-ranks timing, the gradient all-reduce over a two-rank group. On a one-GPU box the two
  ranks share the device (RL4CO_BENCH_SHARED_GPU=1, gloo: RCCL refuses two ranks per device); with two GPUs visible the
  same command runs one rank per GPU over RCCL.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {proc.stdout[:500]}"
    # the driver parses this line: r03's had grown to 27 KB (whole parity block + every leg's dict) and came back as
    # `parsed: null`. Contract since r04: <= 4 KB, everything else in the detail file the line names
    assert len(lines[0].encode()) <= 4096, len(lines[0].encode())
    return json.loads(lines[0])


def _detail(line):
    with open(os.path.join(ROOT, line["detail_file"])) as f:
        return json.load(f)


def test_single_gpu_line_has_the_contract_keys():
    line = _run(["--steps", "10", "--warmup", "2", "--legs", "c2_greedy", "--no-cpu-baseline", "--no-parity"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 10 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0.3 < roof["frac"] <= 1.0 and abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    assert abs(line["value"] - 4096 * 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    assert line["encoder_roofline"]["bound"] == "mfma" and line["config"]["launch"] == "pipeline"
    for k in ("kernel", "achieved", "peak", "unit", "traffic", "launch_ms_mean"):
        assert k in roof, k
    detail = _detail(line)
    assert detail["value"] == pytest.approx(line["value"], rel=1e-4) and detail["roofline"]["bytes_per_launch"] > 0


def test_default_legs_line_stays_small_and_carries_baseline_and_parity():
    """The driver's own command shape (all legs, parity block, cpu_baseline) must still fit the line."""
    line = _run(["--steps", "10", "--warmup", "2", "--leg-steps", "4"], timeout=900)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    # C1 (TSP-20 x 256, check_solution on / off) and the headline workload, 1 warm-up + >= 5 passes, median
    # and best, a one-thread figure
    assert cb["passes"] >= 5 and cb["best"] >= cb["value"] > 0 and cb["c1_best"] >= cb["c1_value"] > 0 and cb["c1_nocheck_value"] > 0
    assert cb["one_thread"]["c1"] > 0
    # spread over the repeated K-step regions, headline and legs
    rs = line["region_ms_per_step"]
    assert rs["n"] >= 5 and rs["min"] <= rs["median"] <= rs["max"]
    assert rs["min"] * (1 - 1e-3) <= line["ms_per_step"] <= rs["max"] * (1 + 1e-3)  # (the line's numbers carry four digits)
    assert all(len(v["ms_min_med"]) == 2 for v in line["legs"].values())
    assert set(line["legs"]) >= {"c2_sampling", "c3_greedy", "c5_sampling", "c4_train", "c2_greedy_fp32"}
    assert all(v["ms_per_step"] > 0 for v in line["legs"].values())
    assert line["parity"]["c2_fp32_tours"].endswith("/4096") and len(line["parity"]) <= 10
    assert line["legs"]["c2_greedy_fp32"]["parity_tours"].endswith("/4096")
    detail = _detail(line)
    assert "parity" in detail and "c4_train" in detail["legs"] and detail["legs"]["c4_train"]["roofline"]["bound"] == "mfma"


def test_two_ranks_self_spawned():
    two_gpus = torch.cuda.device_count() >= 2
    extra = {} if two_gpus else {"RL4CO_BENCH_SHARED_GPU": "1", "RL4CO_DIST_BACKEND": "gloo"}
    line = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--legs", "c2_greedy,c4_train", "--no-cpu-baseline", "--no-parity"],
                env_extra=extra)
    assert line["n_gpus"] == 2 and line["collective_backend"] == ("nccl" if two_gpus else "gloo")
    # "rccl_ranks" only when the collective really ran on RCCL; gloo ranks (two processes sharing one GPU) say so
    assert line["rccl_ranks" if two_gpus else "collective_ranks"] == 2 and ("rccl_ranks" in line) == two_gpus
    assert line["rank_ms_per_step"]["min"] <= line["rank_ms_per_step"]["max"] == pytest.approx(line["ms_per_step"], rel=1e-3)
    assert 0 < line["scaling_efficiency"] and line["n1_ms_per_step"] > 0 and line["allreduce_ms"] > 0
    train = _detail(line)["legs"]["c4_train"]
    assert train["collective"]["backend"] == ("nccl" if two_gpus else "gloo") and train["collective"]["ranks"] == 2
    assert train["roofline"]["bound"] == "mfma" and train["rollout_roofline"]["launch_ms_mean"] > 0
    # whole-job throughput: both ranks' instance-steps over the max-over-ranks wall time
    assert abs(line["value"] - 2 * 4096 * 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


@pytest.mark.parametrize("n", [4, 8])
def test_the_drivers_rank_counts_on_one_shared_gpu(n):
    """The rank arithmetic of the driver's N = 4 and N = 8 runs (instance sharding, max-over-ranks clock, per-rank and per-region
    spread, the solo reference, the gradient all-reduce) with all ranks time-sharing ONE device over gloo — what a one-GPU box
    can check of `bench.py --gpus 8`; the times themselves mean nothing here."""
    if torch.cuda.device_count() >= n:
        pytest.skip("enough GPUs for the real thing: test_all_visible_gpus_on_rccl")
    line = _run(["--gpus", str(n), "--steps", "4", "--warmup", "1", "--legs", "c2_greedy,c4_train", "--no-cpu-baseline", "--no-parity"],
                env_extra={"RL4CO_BENCH_SHARED_GPU": "1", "RL4CO_DIST_BACKEND": "gloo"}, timeout=600)
    assert line["n_gpus"] == n and line["collective_backend"] == "gloo" and line["collective_ranks"] == n and "rccl_ranks" not in line
    assert line["config"]["parallelism"].startswith(f"replicas x{n}")
    assert line["rank_ms_per_step"]["min"] <= line["rank_ms_per_step"]["max"] == pytest.approx(line["ms_per_step"], rel=1e-3)
    assert line["region_ms_per_step"]["n"] == 5 and line["n1_ms_per_step"] > 0 and 0 < line["scaling_efficiency"] <= 1.2
    assert abs(line["value"] - n * 4096 * 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    train = _detail(line)["legs"]["c4_train"]
    assert train["collective"]["ranks"] == n and train["collective"]["backend"] == "gloo" and line["allreduce_ms"] > 0


def test_all_visible_gpus_on_rccl():
    """Every GPU of the node, one rank each, the driver's own multi-GPU command shape — RCCL for real (the
    REINFORCE gradient all-reduce, rl4co/utils/trainer.py:83-86). Needs >= 2 GPUs: the one-GPU boxes skip it."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL with more than one rank needs a multi-GPU node")
    line = _run(["--gpus", str(n), "--steps", "6", "--warmup", "2", "--legs", "c2_greedy,c4_train", "--no-cpu-baseline", "--no-parity"],
                timeout=900)
    assert line["n_gpus"] == n and line["collective_backend"] == "nccl" and line["rccl_ranks"] == n
    assert 0.5 < line["scaling_efficiency"] <= 1.2 and line["allreduce_ms"] > 0
    assert line["legs"]["c4_train"]["scaling_efficiency"] > 0.5
    assert abs(line["value"] - n * 4096 * 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_a_dead_rank_fails_the_run_instead_of_hanging():
    """One rank exits before its first barrier: the job must END with a non-zero status well inside the timeout (the
    launcher tears the survivors down; bench.py's own --timeout is the backstop), not sit in a collective."""
    two_gpus = torch.cuda.device_count() >= 2
    env = dict(os.environ, RL4CO_BENCH_KILL_RANK="1")
    if not two_gpus:
        env.update(RL4CO_BENCH_SHARED_GPU="1", RL4CO_DIST_BACKEND="gloo")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--legs", "c2_greedy",
                           "--no-cpu-baseline", "--no-parity", "--timeout", "120"], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{")], "no result line from a broken run"
