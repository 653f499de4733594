"""CPU (`-m "not gpu"`): host logic added in round 5 — the encoder's layer-shape gate (a stack with another MLP
width must never be packed for the kernels), the bench's spread / collective keys, the CPU baseline's protocol
(BASELINE.md §3). No kernel is launched."""
import importlib.util
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("kw,ok", [({}, True), ({"feedforward_hidden": 256}, False), ({"feedforward_hidden": 0}, False),
                                   ({"num_encoder_layers": 6, "normalization": "instance"}, True)])
def test_packed_encoder_refuses_layer_shapes_the_kernels_do_not_have(kw, ok):
    """csrc/am_encoder*.hip have 8 heads x 16 and a 128 -> 512 -> 128 MLP compiled in; the constructor accepts other shapes
    (zoo/am/encoder.py:40-57). Such a stack must report unsupported BEFORE anything is packed — packed with the wrong shape
    the kernels would read `layer * 512 * 128` into a smaller block (out of bounds, silently wrong embeddings)."""
    from rl4co_amd.policy import AttentionModelPolicy

    pol = AttentionModelPolicy("tsp", **kw).eval()
    pe = pol._packed_encoder()
    assert pe._stack_shape_ok() is ok
    if not ok:
        td = {"action_mask": torch.ones(4, 20, dtype=torch.bool), "locs": torch.rand(4, 20, 2)}
        assert pe.supported(td) is False and pe.supported(td, torch.float32) is False


def test_spread_and_compact_line_keys():
    b = _bench()
    sp = b._spread([3.3, 3.1, 3.2, 3.5, 3.15], 20)
    assert sp == {"min": 3.1, "median": 3.2, "max": 3.5, "n": 5, "steps": 20}
    detail = {"metric": "decode_steps_per_sec", "value": 1.2e8, "unit": "instance·step/s", "n_gpus": 2, "steps": 20, "warmup": 5,
              "ms_per_step": 3.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
              "config": {"workload": "w", "batch_per_gpu": 4096, "cache_dtype": "bf16", "parallelism": "replicas x2"},
              "roofline": {"kernel": "am_decode kernel, STREAM", "bound": "hbm", "achieved": 6500.0, "peak": 8000.0, "unit": "GB/s",
                           "frac": 0.81, "traffic": 1.6e10, "launch_ms_mean": 2.45},
              "region_ms_per_step": sp, "collective_backend": "gloo", "collective_ranks": 2, "allreduce_ms": 0.3,
              "cpu_baseline": {"value": 6e4, "best": 6.1e4, "c1_value": 1.3e5, "c1_best": 1.4e5, "c1_nocheck_value": 1.3e5,
                               "one_thread": {"c1": 3e4, "headline": 1.6e4}, "unit": "instance·step/s", "cores": 8, "kind": "port",
                               "passes": 5, "sample": "s" * 300, "gpu_over_cpu": 2000.0}}
    results = {"c2_greedy": {"ms_per_step": 3.3, "value": 1.2e8},
               "c4_train": {"ms_per_step": 21.0, "value": 1.0e7, "region_ms_per_step": sp, "scaling_efficiency": 0.97}}
    line = b.compact_line(detail, "c2_greedy", results, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    assert line["region_ms_per_step"]["median"] == 3.2 and line["collective_backend"] == "gloo" and line["collective_ranks"] == 2
    assert "rccl_ranks" not in line  # gloo ranks are not reported as RCCL ranks
    assert line["legs"]["c4_train"]["ms_min_med"] == [3.1, 3.2] and line["legs"]["c4_train"]["scaling_efficiency"] == 0.97
    cb = line["cpu_baseline"]
    assert cb["c1_value"] == 1.3e5 and cb["one_thread"]["c1"] == 3e4 and cb["passes"] == 5 and len(cb["sample"]) <= 170
    assert len(json.dumps(line, separators=(",", ":")).encode()) < b.MAX_LINE_BYTES


def test_cpu_baseline_follows_the_protocol_on_a_small_sample():
    """BASELINE.md §3 on a sample small enough for the CPU suite: C1 exactly (TSP-20 x 256, check_solution on and off),
    the leg's workload, 1 warm-up + >= 5 timed passes, median and best, the thread count, a one-thread figure.
    In a SUBPROCESS: the baseline probes thread counts (torch.set_num_threads), and ATen's CPU reductions partition by
    thread count — run in this process it changes the last bits of every later fp32 sum, and the bit-exact golden tests
    that follow (tests/test_oracle_cpu.py) fail."""
    import subprocess
    import sys

    code = ("import importlib.util, json, os, sys\n"
            f"spec = importlib.util.spec_from_file_location('bench_module', os.path.join({ROOT!r}, 'bench.py'))\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "r = b.cpu_baseline('tsp', 20, 128, repeats=5, budget_s=1.0)\n"
            "print('RESULT ' + json.dumps(r))\n")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-1500:]
    r = json.loads(next(ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT "))[7:])
    assert r["kind"] == "port" and r["passes"] >= 5 and r["cores"] >= 1 and r["unit"] == "instance·step/s"
    assert r["best"] >= r["value"] > 0 and r["min_s"] <= r["median_s"]
    assert r["c1_best"] >= r["c1_value"] > 0 and r["c1_nocheck_value"] > 0 and r["one_thread"]["c1"] > 0
    assert set(r["c1"]) == {"check", "nocheck"} and r["c1"]["check"]["mean_reward"] == r["c1"]["nocheck"]["mean_reward"] < 0
    assert "TSP-20 x 256" in r["sample"]
