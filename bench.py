#!/usr/bin/env python
"""bench.py — the rollout hot path on N MI355X GPUs of one node (driver contract).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: re-executes itself under the one below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic instances already resident in HBM:
``env.reset`` -> AttentionModel policy rollout (fused MFMA encoder + cache fold, ONE persistent launch of the
fused decode kernel for all T decode steps, tour-length reward, validity check). Inference legs run the K steps as a
stream of batches with TWO captured rollouts in flight on two HIP streams (``--launch pipeline``, the default: the next
batch's launches fill the CUs the finishing decode waves of the previous one release; every step is submitted, finished
and read back inside the timed region); ``graph_ms_per_step`` (one captured graph per step, one stream) and
``eager_ms_per_step`` are timed in the same process and reported beside it.

HEADLINE (the top-level keys of the JSON line; `--steps K` / `--warmup W` apply to it): BASELINE.json
configs[1] — TSPEnv num_loc=100, batch 4096 per GPU, AM (3 layers, d=128, 8 heads), bf16 encoder GEMMs and
cache planes with fp32 arithmetic, greedy. The other BASELINE configs are timed by the same process right after it
(``--legs`` selects):

    c2_sampling  configs[1], sampling (in-kernel Philox)          c3_greedy  configs[2] CVRP-100 x 4096
    c5_sampling  configs[4] CVRP-500 x 1024 sampling               c4_train   configs[3] per-GPU share: POMO
                 (WIDE decode variant, token-tile encoder)                    TSP-100 x 4096 x 8 starts REINFORCE
    c2_greedy_fp16  configs[1] in the reference's default precision           step incl. the RCCL grad all-reduce
    c2_greedy_fp32  configs[1] in the BIT-IDENTICAL configuration (fp32 MFMA encoder, fp32 planes), with the count of
                    reference tours it reproduces on the trained weights (``parity_tours``)

THE STDOUT LINE IS <= 4 KB (the driver parses it; r03's 27 KB line came back ``parsed: null``): the contract keys, the
decode kernel's ``roofline`` {kernel, bound, achieved, peak, unit, frac, traffic, launch_ms_mean}, ``encoder_roofline``,
``cpu_baseline``, three numbers per leg under ``legs``, a <= 10-key ``parity`` summary and ``detail_file`` — the path of the
JSON file (default gpurun_out/bench_detail.json) that holds everything else: every leg's full dict, the rooflines with their
byte models, the whole parity block. Optional blocks are dropped rather than exceed the limit.

Instances shard across ranks with no data-path collective (weak scaling; SURVEY.md §8e: inference = replicas); the
training leg's one exchange step is the flat fp32 gradient all-reduce on "nccl" (= RCCL), initialised even at N = 1 so
that the collective path really executes. Rank 0 prints ONE JSON line. ``value`` = whole-job instance·decode-steps per
second (B·T·N_gpus·K / wall), wall = max over ranks of the barrier-bracketed timed region. At N > 1 the line adds
``rank_ms_per_step`` {min, max}, ``n1_ms_per_step`` (rank 0 runs the same K steps alone while the other ranks wait at the
barrier), ``scaling_efficiency`` = that / the joint time, ``rccl_ranks``, ``allreduce_ms``.

``roofline`` (decode kernel, HBM-bound): ``achieved`` = bytes the launch MUST move — the cache rows it streams are
counted by the kernel itself (rows of the currently feasible nodes only: masked nodes are exact zeros in the
reference's formulation and are never read), x 3 planes x 256 B, plus the gathered context rows, masks and outputs —
divided by the mean launch duration from HIP events on the launch stream, so ``frac`` <= 1 by construction. The
SURVEY.md §8(d) contract figure (every node's row at every step, what the reference's formulation reads) is kept
beside it as ``contract_GBs`` (detail: ``algorithmic_bytes_contract``). ``traffic`` = HBM bytes per launch from the separate
rocprofv3 --pmc passes of the same leg (profiles/pmc_traffic.json; a counter pass cannot run inside this process).
``cpu_baseline``: the reference path (torch restatement, pinned bit-exact to the reference's own source by
oracle/gen_golden.py; kind "port") timed on this box's host cores on a bounded sample.
``parity`` (detail file; summary in the line): for every inference leg, the fp32 configuration and the benchmarked bf16
configuration against the reference's own rollouts (fp32 and under bf16 autocast) of TRAINED weights at the leg's full
size (tests/golden/trained; tests/trained_parity.py): flips proven near-ties, identical tours, per-decision agreement.
"""
from __future__ import annotations

import argparse
import gc
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_*_f32: 256 FLOP / cycle / CU x 256 CUs x 2.4 GHz

LEGS = {
    # name: (env, num_loc, batch per GPU, decode, BASELINE.json config index)
    "c2_greedy": ("tsp", 100, 4096, "greedy", 1),
    "c2_sampling": ("tsp", 100, 4096, "sampling", 1),
    "c3_greedy": ("cvrp", 100, 4096, "greedy", 2),
    "c5_sampling": ("cvrp", 500, 1024, "sampling", 4),
    "c4_train": ("tsp", 100, 4096, "multistart_sampling", 3),
    # configs[1] once more in the reference's DEFAULT precision ("16-mixed" = fp16 autocast, utils/trainer.py:57): fused
    # encoder on v_mfma_f32_32x32x16_f16, fp16 planes in the streaming decode kernel
    "c2_greedy_fp16": ("tsp", 100, 4096, "greedy", 1),
    # configs[1] in the BIT-IDENTICAL configuration (north_star: "greedy tour lengths bit-identical to the reference"): fp32
    # encoder on v_mfma_f32_16x16x4_f32 (csrc/am_encoder_f32.hip), fp32 planes, fp32 decode arithmetic; its `parity_tours` =
    # reference tours reproduced on the trained weights at the full size (tests/golden/trained)
    "c2_greedy_fp32": ("tsp", 100, 4096, "greedy", 1),
    # (not in the default run) configs[4] in the reference's default precision: fp16 planes through the WIDE decode variant
    "c5_sampling_fp16": ("cvrp", 500, 1024, "sampling", 4),
}
DEFAULT_LEGS = "c2_greedy,c2_sampling,c3_greedy,c5_sampling,c4_train,c2_greedy_fp16,c2_greedy_fp32"


def leg_dtypes(leg: str, args) -> tuple[str, str]:
    """(cache planes, encoder operands) of a leg: the _fp16 / _fp32 legs fix theirs, the others follow the flags."""
    if leg.endswith("_fp16"):
        return "f16", "f16"
    if leg.endswith("_fp32"):
        return "f32", "f32"
    return args.cache_dtype, args.encoder_dtype


def _trained_parity():
    """tests/trained_parity.py — the measurement shared with tests/test_gpu_trained_parity.py (fixtures only: no oracle, no reference)."""
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    import trained_parity

    return trained_parity


def log(msg: str) -> None:
    """Progress goes to stderr; stdout carries exactly one JSON line."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes_per_instance_step(env_name: str, n: int, elem: int) -> int:
    """SURVEY.md §8(d) contract: 3·N·d·e (K_g,V_g,K_l) + N (mask read) + N (mask write) + context rows
    (TSP: 2·d·e, CVRP: d·e + 8) + d·4 (graph context) + 16 (action, logp, scalars)."""
    d = 128
    ctx = 2 * d * elem if env_name == "tsp" else d * elem + 8
    return 3 * n * d * elem + 2 * n + ctx + d * 4 + 16


def must_move_bytes(env_name: str, n: int, elem: int, rows_read: int, instance_steps: int, trajectories: int, ctx_elem: int = 4) -> int:
    """Bytes one decode launch has to move through HBM given what it actually visits: `rows_read` cache rows per
    plane (counted in-kernel: the feasible rows of every step), per step the gathered context rows (TSP two,
    CVRP one + the load scalar) and the action / log-prob it stores, per trajectory its mask in and out, graph
    context and state words (the mask lives in LDS between entry and exit)."""
    d = 128
    per_step = (2 * d * ctx_elem if env_name == "tsp" else d * ctx_elem + 8) + 12  # (r06: 16-bit context rows beside 16-bit planes)
    per_traj = 2 * n + d * 4 + 64
    return rows_read * 3 * d * elem + instance_steps * per_step + trajectories * per_traj


def state_hash(tensors: dict) -> str:
    """Same digest as oracle/gen_golden.py writes into tests/golden/MANIFEST.json."""
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(tensors[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def _cpu_rollout_passes(pol, env, data, batch: int, chunk: int, passes: int) -> tuple[list[float], int, float]:
    """1 warm-up + `passes` timed greedy rollouts of `batch` instances (in calls of `chunk`), the span EvalBase.__call__
    times (rl4co/tasks/eval.py:38-63): env.reset -> policy(td, env, decode_type="greedy") -> reward.
    Returns (seconds per timed pass, instance·steps per pass, mean reward)."""
    times, work, rewards = [], 0, []
    for i in range(passes + 1):
        t0 = time.perf_counter()
        rewards, work = [], 0
        for lo in range(0, batch, chunk):
            td = env.reset({k: v[lo : lo + chunk].clone() for k, v in data.items()})
            out = pol(td, env, phase="test", decode_type="greedy")
            rewards.append(out["reward"])
            work += out["actions"].shape[0] * out["actions"].shape[1]
        if i > 0:  # the first pass is the warm-up
            times.append(time.perf_counter() - t0)
    return times, work, float(torch.cat(rewards).mean())


def cpu_baseline(env_name: str, num_loc: int, sample_batch: int, repeats: int = 5, chunk: int = 512, budget_s: float = 12.0) -> dict:
    """The reference path on the host cores, by the protocol of BASELINE.md §3: the oracle restatement (stock ATen ops,
    fp32, the reference's ops in the reference's order: kind "port"), greedy rollout, span = reset -> policy -> reward
    under inference_mode, 1 warm-up + `repeats` (>= 5) timed rollouts, MEDIAN and MIN, thread count reported, a 1-thread
    figure beside it.

    * C1 = BASELINE configs[0] exactly: TSPEnv num_loc=20, batch 256, default AttentionModelPolicy, greedy — with
      ``check_solution`` on (the reference's default, envs/common/base.py:54) and off (configs/experiment/base.yaml:21).
    * the headline workload (C2: TSP-100) on a bounded sample of its 4096 instances (whole calls of 4096 take 13 s each
      on the host: the sample keeps the leg at ~10 - 30 s of CPU work), same seeds and generator as the GPU leg's shape.
    `value` = the headline workload's MEDIAN rate (instance·step/s), comparable with the GPU line's `value`."""
    from oracle import reference_torch as R
    import statistics

    repeats = max(5, repeats)
    avail = max(1, len(os.sched_getaffinity(0)))  # cgroup / affinity aware (os.cpu_count() reports the whole machine)
    default_threads = max(1, min(avail, torch.get_num_threads()))
    torch.manual_seed(0)
    pol = R.AttentionModelPolicy(env_name).eval()
    env = R.get_env(env_name, num_loc, check_solution=True)
    torch.manual_seed(1234)
    data = env.generate(sample_batch)
    t_leg = time.perf_counter()
    with torch.inference_mode():
        # the reference's many small ATen ops do not scale to 100+ threads: probe a few thread counts on a small batch and
        # time the baseline at the fastest one (stated in `cores`)
        probe_b = min(64, sample_batch)
        per_inst, threads = float("inf"), default_threads
        for cand in sorted({default_threads, *(c for c in (64, 32, 16, 8) if c <= avail)}, reverse=True):
            torch.set_num_threads(cand)
            best_c = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                pol(env.reset({k: v[:probe_b].clone() for k, v in data.items()}), env, phase="test", decode_type="greedy")
                best_c = min(best_c, (time.perf_counter() - t0) / probe_b)
            if best_c < per_inst:
                per_inst, threads = best_c, cand
        torch.set_num_threads(threads)
        # ---- the headline workload on a bounded sample: ~12 s for the 1 + repeats passes on any host --------------------
        budget_b = int(budget_s / (repeats + 1) / max(per_inst, 1e-6))
        if budget_b < sample_batch:
            sample_batch = max(probe_b, budget_b // 64 * 64)
            data = {k: v[:sample_batch] for k, v in data.items()}
        chunk = min(chunk, sample_batch)
        log(f"cpu_baseline: {threads} threads, sample {sample_batch} instances in calls of {chunk} (probe {per_inst * 1e3:.2f} ms/instance)")
        times, work, mean_reward = _cpu_rollout_passes(pol, env, data, sample_batch, chunk, repeats)
        med, best = statistics.median(times), min(times)
        # ---- C1 = BASELINE configs[0], exactly: TSP-20 x 256, check_solution on / off --------------------------------
        c1 = {}
        torch.manual_seed(0)
        pol1 = R.AttentionModelPolicy("tsp").eval()
        for check in (True, False):
            env1 = R.get_env("tsp", 20, check_solution=check)
            torch.manual_seed(1234)
            d1 = env1.generate(256)
            t1, w1, r1 = _cpu_rollout_passes(pol1, env1, d1, 256, 256, repeats)
            c1["check" if check else "nocheck"] = {"median": w1 / statistics.median(t1), "best": w1 / min(t1),
                                                   "median_ms": statistics.median(t1) * 1e3, "min_ms": min(t1) * 1e3,
                                                   "instances_per_sec": 256 / statistics.median(t1), "mean_reward": r1}
        # ---- one thread (BASELINE.md §3 "also report 1-thread"): C1 as is, the headline workload on 64 instances --------
        torch.set_num_threads(1)
        t1, w1, _ = _cpu_rollout_passes(pol1, env1, d1, 256, 256, 2)
        one = {"c1": w1 / statistics.median(t1)}
        if time.perf_counter() - t_leg < 40.0:
            small = {k: v[:64] for k, v in data.items()}
            t2, w2, _ = _cpu_rollout_passes(pol, env, small, 64, 64, 1)
            one["headline"] = w2 / min(t2)
        torch.set_num_threads(default_threads)
    return {
        "value": work / med, "best": work / best, "unit": "instance·step/s",
        "median_s": med, "min_s": best, "passes": repeats,
        "cores": threads, "kind": "port",
        "c1_value": c1["check"]["median"], "c1_best": c1["check"]["best"], "c1_nocheck_value": c1["nocheck"]["median"],
        "one_thread": one,
        "c1": c1,
        "sample": f"{env_name.upper()}-{num_loc}: {sample_batch} of the leg's instances in calls of {chunk}; C1 = TSP-20 x 256 whole; "
                  f"greedy rollout reset->policy->reward, fp32 torch CPU, 1 warm-up + {repeats} passes, median (best beside it)",
        "mean_reward": mean_reward,
        "wall_s": time.perf_counter() - t_leg,
        "note": "the oracle restatement (the reference package itself cannot be installed on the box), BASELINE.md §3 protocol; "
                "thread count chosen by probing; the headline rate moves 0.7 - 1.6e5 between boxes (host CPU model / load)",
    }


TRAIN_PMC_FILE = "r06_c4_train_pmc.json"  # counters of the training kernels as they are in THIS tree (tools/train_pmc.sh)
MAX_LINE_BYTES = 4096  # the driver parses the ONE stdout line; r03's 27 KB line came back as `parsed: null`


def _r(x, digits=4):
    """Numbers of the compact line carry a few significant digits (the detail file keeps the full values)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _spread(ms: list, steps: int) -> dict:
    """min / median / max of the per-step time over the repeated K-step regions of one leg."""
    import statistics

    return {"min": min(ms), "median": statistics.median(ms), "max": max(ms), "n": len(ms), "steps": steps}


def compact_roofline(r: dict | None) -> dict | None:
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms_mean")
    out = {"kernel": r["kernel"].split(":")[0].split(" (")[0][:48]}
    out.update({k: _r(r.get(k)) for k in keep})
    # the SURVEY.md §8(d) figure reproducible from the line alone: contract bytes per launch, the rate and the (> 1) "fraction"
    # they give — the kernel never reads masked rows, so this is NOT a utilisation; `frac` (must-move bytes) is
    for k in ("contract_GBs", "frac_contract", "contract_bytes", "bytes_per_launch", "hbm_read_probe_GBs", "frac_large_batch"):
        if r.get(k) is not None:
            out[k] = _r(r[k])
    if r.get("traffic_source"):
        out["traffic_source"] = str(r["traffic_source"])[:96]
    return out


def compact_parity(par: dict | None) -> dict | None:
    """<= 10 keys: identical tours of the fp32 (bit-identical) configuration and of the benchmarked 16-bit one, per leg."""
    if not par:
        return None
    out = {}

    def frac(rec):
        return f"{rec['identical']}/{rec['of']}"

    for leg, tag in (("c2_greedy", "c2"), ("c3_greedy", "c3")):
        rec = par.get(leg)
        if rec:
            out[f"{tag}_fp32_tours"] = frac(rec["fp32"])
            out[f"{tag}_bf16_vs_ref_bf16"] = frac(rec["bf16_vs_reference_bf16_autocast"])
    rec = par.get("c2_greedy")
    if rec:
        out["c2_fp32_flip_regret_max"] = _r(rec["fp32"].get("flip_regret_max"))
        out["c2_bf16_step_agreement"] = _r(rec["bf16_vs_reference_bf16_autocast"]["step_agreement"])
        out["c2_bf16_reward_rel_gap"] = _r(rec["bf16_vs_reference_bf16_autocast"]["reward_rel_gap"])
        out["ref_bf16_vs_ref_fp32_tours"] = rec.get("reference_bf16_vs_reference_fp32_identical")
    rec = par.get("c2_sampling")
    if rec:
        out["c2_sampling_fp32_tours"] = frac(rec["fp32_reference_noise"])
    rec = par.get("c5_sampling")
    if rec and "greedy_fp32" in rec:
        out["c5_greedy_fp32_tours"] = frac(rec["greedy_fp32"])
    return out


def compact_line(detail: dict, head_name: str, results: dict, detail_path: str) -> dict:
    """The ONE stdout line: the driver's contract keys, the dominant kernel's roofline, cpu_baseline, one number per
    leg and a parity summary — under MAX_LINE_BYTES. Everything else goes to the detail file."""
    cfg = detail["config"]
    line = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {
        "workload": cfg["workload"][:160], "leg": head_name, "batch_per_gpu": cfg["batch_per_gpu"], "cache_dtype": cfg["cache_dtype"],
        "launch": (detail.get("launch") or "").split(":")[0],
        "input": "one synthetic batch resident in HBM, reused by every step", "parallelism": cfg["parallelism"],
        # what really runs before the clock starts (the echoed `warmup` is only the first group of it)
        "untimed_steps": detail.get("untimed_steps"), "untimed_extra": "1 s HBM read loop before the first leg",
    }
    line["roofline"] = compact_roofline(detail.get("roofline"))
    if detail.get("encoder_roofline"):
        line["encoder_roofline"] = compact_roofline(detail["encoder_roofline"])
    for k in ("node_steps_per_sec", "instances_per_sec", "graph_ms_per_step", "eager_ms_per_step", "train_ms_per_step", "rccl_ranks",
              "collective_ranks", "collective_backend", "n1_ms_per_step", "scaling_efficiency", "allreduce_ms"):
        if detail.get(k) is not None:
            line[k] = _r(detail[k])
    if detail.get("placement"):  # the start-up self-check's result: ranks and DISTINCT devices they sit on
        line["distinct_gpus"] = detail["placement"]["distinct_devices"]
    if detail.get("rank_ms_per_step"):
        line["rank_ms_per_step"] = {k: _r(v) for k, v in detail["rank_ms_per_step"].items()}
    if detail.get("region_ms_per_step"):  # min / median / max over the repeated K-step regions (ms_per_step = the first one)
        line["region_ms_per_step"] = {k: _r(v) for k, v in detail["region_ms_per_step"].items()}
    legs = {}
    for name, r in results.items():
        if name == head_name:
            continue
        legs[name] = {"ms_per_step": _r(r["ms_per_step"]), "value": _r(r["value"])}
        if r.get("region_ms_per_step"):
            legs[name]["ms_min_med"] = [_r(r["region_ms_per_step"]["min"]), _r(r["region_ms_per_step"]["median"])]
        if r.get("roofline"):
            legs[name]["frac"] = _r(r["roofline"]["frac"], 3)
        if r.get("scaling_efficiency") is not None:
            legs[name]["scaling_efficiency"] = _r(r["scaling_efficiency"], 3)
        if r.get("parity_tours"):
            legs[name]["parity_tours"] = r["parity_tours"]
    line["legs"] = legs
    if detail.get("parity"):
        line["parity"] = compact_parity(detail["parity"])
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "best": _r(cb.get("best")), "c1_value": _r(cb.get("c1_value")),
                                "c1_best": _r(cb.get("c1_best")), "c1_nocheck_value": _r(cb.get("c1_nocheck_value")),
                                "one_thread": {k: _r(v) for k, v in (cb.get("one_thread") or {}).items()},
                                "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "passes": cb.get("passes"),
                                "sample": cb["sample"][:170], "gpu_over_cpu": _r(cb.get("gpu_over_cpu"))}
    line["detail_file"] = os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path
    return line


class Bench:
    def __init__(self, args, rank: int, world: int, device: torch.device):
        from rl4co_amd import dist as D

        self.args, self.rank, self.world, self.device, self.D = args, rank, world, device, D
        self.cache_dtype = torch.bfloat16 if args.cache_dtype == "bf16" else torch.float32
        self.enc_dtype = torch.bfloat16 if args.encoder_dtype == "bf16" else None
        self.elem = 2 if args.cache_dtype == "bf16" else 4

    def barrier(self) -> None:
        torch.cuda.synchronize()
        self.D.barrier()
        torch.cuda.synchronize()

    def traffic(self, leg: str):
        """HBM bytes per decode launch of this leg from the separate rocprofv3 --pmc passes (tools/profile_legs.sh
        + tools/profile_parse.py write profiles/pmc_traffic.json); (None, None) when the leg was not profiled."""
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            return None, None
        key = f"{leg}/{leg_dtypes(leg, self.args)[0]}"
        row = table.get(key)
        return (row["traffic_bytes_per_launch"], row.get("source")) if row else (None, None)

    # -- inference rollout legs ---------------------------------------------------------------------------------
    def rollout_leg(self, leg: str, steps: int, warmup: int) -> dict:
        from rl4co_amd import kernels as K
        from rl4co_amd.envs import get_env
        from rl4co_amd.policy import AttentionModelPolicy

        a = self.args
        env_name, num_loc, batch, decode, cfg_idx = LEGS[leg]
        if a.batch is not None and leg == "c2_greedy":
            batch = a.batch
        half = leg.endswith("_fp16")
        full = leg.endswith("_fp32")
        cache_dtype, enc_dtype, elem = ((torch.float16, torch.float16, 2) if half else
                                        (torch.float32, None, 4) if full else (self.cache_dtype, self.enc_dtype, self.elem))
        torch.manual_seed(0)  # random-init weights of the reference architecture, identical on every rank
        policy = AttentionModelPolicy(env_name=env_name, cache_dtype=cache_dtype,
                                      encoder_autocast=enc_dtype).to(self.device).eval()
        if os.environ.get("RL4CO_BENCH_DECODE_VARIANT"):  # probe: pin the decode kernel variant (stream / lds / wide / ms)
            policy.decode_variant = os.environ["RL4CO_BENCH_DECODE_VARIANT"]
        env = get_env(env_name, generator_params=dict(num_loc=num_loc, device=self.device), device=self.device,
                      check_solution=not a.no_check_solution)
        torch.manual_seed(1234 + self.rank)  # each rank owns its shard of the synthetic instances
        data = env.generator(batch_size=[batch])
        torch.cuda.synchronize()

        def step():
            return policy(env.reset(data), env, phase="test", decode_type=decode)

        log(f"{leg}: rank {self.rank}/{self.world}, {batch} instances resident, warming up")
        rows = inst_steps = 0
        use_graph = a.launch in ("graph", "pipeline")
        graph_ms = None
        untimed = warmup  # every step the leg runs BEFORE its timed region (reported: `--warmup` is only the first group)
        with torch.inference_mode():
            for _ in range(warmup):
                out = step()
            torch.cuda.synchronize()
            eager_ms = None
            if use_graph:
                # kernel durations for the roofline: an eager pass with HIP events around the two big launches (events
                # cannot bracket nodes of a captured graph); the timed region below replays the SAME launches as one graph
                ev_steps = max(10, steps // 8)
                policy.decode_events, policy.encode_events = [], []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(ev_steps):
                    step()
                torch.cuda.synchronize()
                eager_ms = (time.perf_counter() - t0) / ev_steps * 1e3
                untimed += ev_steps
                decode_ms = [x.elapsed_time(y) for x, y in policy.decode_events]
                encode_ms = [x.elapsed_time(y) for x, y in policy.encode_events]
                policy.decode_events = policy.encode_events = None
                from rl4co_amd.graph import GraphedRollout

                try:
                    graphed = GraphedRollout(policy, env, data, decode_type=decode)
                    step = lambda: graphed(data)  # noqa: E731
                    for _ in range(max(2, warmup)):  # replays before the clock starts (the first ones run slower)
                        out = step()
                    untimed += max(2, warmup)
                    if a.launch == "pipeline":
                        # one stream first (reported as graph_ms_per_step), then two captured rollouts in flight on two
                        # streams: the next batch's launches fill the CUs this batch's finishing decode waves release
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(ev_steps):
                            step()
                        torch.cuda.synchronize()
                        graph_ms = (time.perf_counter() - t0) / ev_steps * 1e3
                        untimed += ev_steps
                        from rl4co_amd.graph import PipelinedRollout

                        pipe = PipelinedRollout(policy, env, data, decode_type=decode, depth=a.pipeline_depth)
                        tickets = []

                        def step():
                            if len(tickets) == pipe.depth:
                                o = pipe.collect(tickets.pop(0))
                            else:
                                o = None
                            tickets.append(pipe.submit(data))
                            return o

                        for _ in range(max(4, warmup)):
                            step()
                        untimed += max(4, warmup)
                except Exception as exc:  # a launch sequence that cannot be captured stays on the eager path, said so
                    log(f"{leg}: HIP graph capture failed ({type(exc).__name__}: {exc}); timing the eager path")
                    use_graph = False
                    torch.cuda.synchronize()
                    policy.decode_events, policy.encode_events = [], []
            else:
                policy.decode_events, policy.encode_events = [], []
            gc.collect()
            gc.freeze()  # (see train_leg: a full collection over torch's module graph is a 60 ms host stall)
            # ... which the chip idles through: the untimed steps directly in front of the clock are repeated after it (r05: the
            # first region of every leg was its slowest, 1 - 7 % above the median of the four after it — it opened on
            # clocks that had ramped down during the collection)
            for _ in range(warmup):
                out = step()
            untimed += warmup
            if not use_graph:  #  the eager leg's kernel durations come from the timed regions only, as in train_leg
                torch.cuda.synchronize()
                policy.decode_events, policy.encode_events = [], []
            pipelined = use_graph and a.launch == "pipeline"
            if pipelined:  # drain: the timed region starts and ends with nothing in flight
                while tickets:
                    out = pipe.collect(tickets.pop(0))
            def region(k):
                """K steps, all submitted, finished and read back; returns (wall, cache rows read, instance-steps, last output)."""
                nonlocal step
                rows_ = steps_ = 0
                o_last = None
                t0 = time.perf_counter()
                if pipelined:
                    done = 0
                    for _ in range(k):
                        o = step()
                        if o is not None:
                            o_last, done = o, done + 1
                            rows_ += policy.last_rows_read
                            steps_ += policy.last_instance_steps
                    while tickets:  # the K submitted steps are all finished (and read back) inside the timed region
                        o_last, done = pipe.collect(tickets.pop(0)), done + 1
                        rows_ += policy.last_rows_read
                        steps_ += policy.last_instance_steps
                    assert done == k
                else:
                    for _ in range(k):
                        o_last = step()
                        rows_ += policy.last_rows_read
                        steps_ += policy.last_instance_steps
                torch.cuda.synchronize()
                return time.perf_counter() - t0, rows_, steps_, o_last

            solo_ms = None
            if self.world > 1 and a.solo:
                # the N = 1 reference of THIS invocation: rank 0 runs the same K steps alone while the other ranks wait at
                # the barrier — weak scaling, so efficiency = that time / the joint time
                self.barrier()
                if self.rank == 0:
                    solo_ms = region(steps)[0] / steps * 1e3
            self.barrier()
            own_wall, rows, inst_steps, out = region(steps)
            self.barrier()
            wall = own_wall
            # spread: REGIONS - 1 further barrier-bracketed K-step regions right after the contract's one (same steps, same
            # data); `value` / `ms_per_step` stay those of the FIRST region, min / median / max over all of them go beside it
            region_ms = [own_wall / steps * 1e3]
            for _ in range(a.regions - 1):
                self.barrier()
                region_ms.append(region(steps)[0] / steps * 1e3)
            self.barrier()
        if not use_graph:
            decode_ms = [x.elapsed_time(y) for x, y in policy.decode_events]
            encode_ms = [x.elapsed_time(y) for x, y in policy.encode_events]
            policy.decode_events = policy.encode_events = None
        t_steps = out["actions"].shape[1]
        # the captured graph (its private memory pool, its streams) is released before the next leg: with two ranks
        # sharing ONE device (the RL4CO_BENCH_SHARED_GPU test mode) a lingering graph made the following training leg
        # crawl (3 s per step); nothing should outlive its leg anyway
        step = graphed = pipe = None  # noqa: F841
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        n_nodes = num_loc + (0 if env_name == "tsp" else 1)
        wall = self.D.reduce_scalar(own_wall, "max", self.device)
        wall_min = -self.D.reduce_scalar(-own_wall, "max", self.device)
        total_inst_steps = int(self.D.reduce_scalar(inst_steps, "sum", self.device))
        res = {"wall": wall}
        # per region the slowest rank's time (what the contract's max-over-ranks clock would have shown for that region)
        region_ms = [self.D.reduce_scalar(x, "max", self.device) for x in region_ms]
        if self.rank != 0:
            return res
        res["region_ms_per_step"] = _spread(region_ms, steps)
        if self.world > 1:
            res["rank_ms_per_step"] = {"min": wall_min / steps * 1e3, "max": wall / steps * 1e3}
            if solo_ms is not None:
                res["n1_ms_per_step"] = solo_ms
                res["scaling_efficiency"] = solo_ms / (wall / steps * 1e3)
        mean_decode_ms = sum(decode_ms) / len(decode_ms)
        per_launch_rows, per_launch_steps = rows / steps, inst_steps / steps
        need = must_move_bytes(env_name, n_nodes, elem, per_launch_rows, per_launch_steps, batch,
                               ctx_elem=int(getattr(policy, "last_ctx_elem_bytes", 4)))
        contract = algorithmic_bytes_per_instance_step(env_name, n_nodes, elem) * per_launch_steps
        achieved = need / (mean_decode_ms * 1e-3) / 1e9
        traffic, traffic_source = self.traffic(leg)
        variant = {1: "STREAM (1 wave / trajectory)", 2: "LDS-resident", 3: "WIDE (4 waves / trajectory)", 4: "MS"}.get(
            K.decode_variant(n_nodes, cache_dtype, t_steps, batch), "?")
        value = total_inst_steps / wall
        res.update({
            "workload": (f"BASELINE configs[{cfg_idx}]: {env_name.upper()}Env num_loc={num_loc} batch={batch}/GPU "
                         f"AttentionModel(3L,d128,h8) {decode} rollout, " +
                         ("fp16 encoder GEMMs, fp16 cache (the reference's default 16-mixed regime)" if half else
                          "fp32 encoder (fp32 MFMA), fp32 cache: the bit-identical configuration" if full else
                          f"{a.encoder_dtype} encoder GEMMs, {a.cache_dtype} cache")),
            "value": value, "unit": "instance·step/s", "steps": steps, "warmup": warmup,
            "ms_per_step": wall / steps * 1e3,
            "launch": (("pipeline: two captured rollouts (hip graphs) in flight on two HIP streams over the stream of batches "
                        "(rl4co_amd/graph.py PipelinedRollout); every one of the K steps is submitted, finished and read back "
                        "inside the timed region") if (use_graph and a.launch == "pipeline") else
                       ("hip_graph: reset + encoder + decode + check + reward captured once, one hipGraphLaunch and one "
                        "24-byte read-back per step (rl4co_amd/graph.py)") if use_graph else "eager: one host launch per kernel"),
            "eager_ms_per_step": eager_ms, "graph_ms_per_step": graph_ms,
            "untimed_steps": untimed,
            "untimed_note": "steps of this leg executed before the timed region: --warmup eager steps, the HIP-event pass for the "
                            "kernel durations, graph replays, the one-graph timing pass, the pipeline fill, --warmup again after the "
                            "garbage collection; plus (once per process) 1 s of plain HBM reads before the first leg",
            "node_steps_per_sec": value * n_nodes,
            "instances_per_sec": batch * self.world * steps / wall,
            "decode_steps_longest": t_steps, "instance_steps_per_launch": per_launch_steps,
            "mean_reward": float(out["reward"].mean()),
            "roofline": {
                "kernel": f"am_decode kernel, {variant}: fused persistent rollout, all decode steps in one launch",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": need,
                "bytes_model": "3 planes x 128 x elem x cache rows streamed (counted in-kernel: feasible rows only) + "
                               "per step gathered fp32 context rows and outputs + per trajectory mask in/out and state",
                "cache_rows_per_launch": per_launch_rows,
                "algorithmic_bytes_contract": contract,
                "contract_note": "SURVEY.md §8(d): every node's row at every step (the reference's formulation); the kernel "
                                 "skips masked rows, so contract bytes / time may exceed the HBM peak — it is not a fraction",
                "contract_GBs": contract / (mean_decode_ms * 1e-3) / 1e9,
                "frac_contract": contract / (mean_decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "contract_bytes": contract,
                "launch_ms_mean": mean_decode_ms, "launch_ms_min": min(decode_ms),
                "us_per_decode_step": mean_decode_ms * 1e3 / t_steps, "launches_timed": len(decode_ms),
                "hbm_utilisation_from_traffic": (traffic / (mean_decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            },
        })
        if encode_ms:  # second kernel of the step: fused encoder + cache fold on the matrix cores
            n_, d_, ff_, layers_ = n_nodes, 128, 512, 3
            flop_inst = layers_ * (2 * n_ * d_ * 3 * d_ + 4 * n_ * n_ * d_ + 2 * n_ * d_ * d_ + 4 * n_ * d_ * ff_) \
                + (5 if env_name == "tsp" else 4) * 2 * n_ * d_ * d_
            enc_ms = sum(encode_ms) / len(encode_ms)
            tf = flop_inst * batch / (enc_ms * 1e-3) / 1e12
            exact = full or enc_dtype is None  # (--encoder-dtype f32: the exact-fp32 kernels on any leg)
            peak = MFMA_F32_PEAK_TFLOPS if exact else MFMA_PEAK_TFLOPS
            res["encoder_roofline"] = {
                "kernel": (("am_encoder_f32_kernel" if exact else "am_encoder_kernel") +
                           " (init embedding + 3 x [MHA, norm, FFN, norm] + cache fold, one workgroup per instance)") if n_nodes <= 128 else
                          ("token-tile encoder, " + ("am_tokens_f32.hip" if exact else "tok16_* + attn_flash") +
                           " (init embedding, 3 x [QKV, streamed attention, out-proj + norm + MLP + norm], fold: sum of the launches)"),
                # beyond 128 nodes the 16-bit encoder's time is its attention kernel (d_h = 16: one exponential per 32 matrix
                # FLOPs), bound by transcendental / VALU issue — priced against the MFMA peak only for continuity
                "bound": "mfma" if (exact or n_nodes <= 128) else "valu_exp",
                "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "flop_per_instance": flop_inst, "launch_ms_mean": enc_ms,
                "note": "algorithmic FLOPs at N nodes (padding excluded); " + ("fp32 MFMA peak" if exact else "16-bit dense MFMA peak") +
                        ("" if (exact or n_nodes <= 128) else "; the attention launches (59 % of the time) are exp-issue bound: "
                         f"{layers_ * 8 * n_ * n_ * batch / 1e9:.1f} G exponentials per pass at a quarter of the VALU rate"),
            }
        if full and self.world == 1 and not a.no_parity:
            TP = _trained_parity()

            try:
                rec = TP.compare(TP.TrainedCase("t2_tsp100_b4096_greedy"), "fp32", self.device, against="fp32")
                res["parity_tours"] = f"{rec['identical']}/{rec['of']}"
                res["parity"] = rec
            except (OSError, ValueError, KeyError) as exc:
                log(f"{leg}: no trained golden for the parity count ({exc})")
        res["host_gap_ms"] = res["ms_per_step"] - mean_decode_ms - (sum(encode_ms) / len(encode_ms) if encode_ms else 0.0)
        return res

    def large_batch_probe(self, leg: str, batch: int = 16384) -> dict | None:
        """HBM or Infinity Cache? The headline batch's planes (315 MB) shrink below the 256 MiB MALL after ~20
        decode steps and FETCH_SIZE counts MALL hits, so the decode launch's rate is measured once more IN THIS PROCESS at
        four times the batch (planes 1.26 GB: the feasible rows stay above 256 MiB for most of the rollout): same kernel,
        same byte model (rows counted in-kernel), HIP events on the launch stream. A lower rate here is the MALL's share."""
        a = self.args
        saved = (a.batch, a.launch, a.regions)
        a.batch, a.launch, a.regions = batch, "eager", 1
        try:
            r = self.rollout_leg(leg, 3, 1)
        except RuntimeError as exc:  # (out of memory on a smaller part: the probe is optional)
            log(f"large-batch probe skipped: {exc}")
            return None
        finally:
            a.batch, a.launch, a.regions = saved
        if self.rank != 0:
            return None
        roof = r["roofline"]
        return {"batch": batch, "GBs": roof["achieved"], "frac": roof["frac"], "launch_ms_mean": roof["launch_ms_mean"],
                "bytes_per_launch": roof["bytes_per_launch"], "launches_timed": roof["launches_timed"]}

    # -- the training leg: configs[3]'s per-GPU share ----------------------------------------------------------------
    def train_leg(self, steps: int, warmup: int) -> dict:
        import torch.distributed as dist

        from rl4co_amd.envs import get_env
        from rl4co_amd.policy import AttentionModelPolicy

        D = self.D
        env_name, num_loc, batch, _, cfg_idx = LEGS["c4_train"]
        starts = 8
        torch.manual_seed(0)
        policy = AttentionModelPolicy(env_name, num_encoder_layers=6, normalization="instance", use_graph_context=False,
                                      cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                                      train_decode_type="multistart_sampling").to(self.device).train()
        policy.encoder.net.fused_stack = self.args.train_encoder == "stack"
        env = get_env(env_name, generator_params=dict(num_loc=num_loc, device=self.device), device=self.device,
                      check_solution=False)  # configs/experiment/base.yaml:21 trains with the check off
        # one multi-tensor launch for the whole update (torch's own fused implementation of the same Adam step: the default
        # "foreach" form is eight launches, 0.19 ms of the step — tools/probes/train_glue_probe.py)
        opt = torch.optim.Adam(policy.parameters(), lr=1e-4, fused=True)
        bucket = D.FlatGradBucket(policy)
        torch.manual_seed(1234 + self.rank)
        data = env.generator(batch_size=[batch])
        ar_events = []

        def step(i, collective=True):
            out = policy(env.reset(data), env, phase="train", seed=1000 * i + self.rank, num_starts=starts)
            reward = out["reward"].view(starts, batch).t()
            ll = out["log_likelihood"].view(starts, batch).t()
            adv = reward - reward.mean(dim=1, keepdim=True)  # SharedBaseline over the starts (pomo/model.py:88-111)
            loss = -(adv.detach() * ll).mean()
            bucket.release()  # fresh gradients, gathered into the flat bucket by one multi-tensor copy below
            loss.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if collective:
                work = bucket.allreduce_mean(async_op=True)  # ONE flat fp32 message over RCCL (utils/trainer.py:83-86)
                if work is not None:
                    work.wait()
                e1.record()
                ar_events.append((e0, e1))
            else:  # the solo (N = 1) reference region of a multi-rank run: the other ranks are not there to answer
                bucket._rebind()
            torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
            opt.step()
            return out

        from rl4co_amd import teacher as T

        log(f"c4_train: rank {self.rank}/{self.world}, warming up")
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        ar_events.clear()
        # HIP events around the two dominant launches of the step (the multistart rollout and the teacher-forced backward):
        # their durations feed the leg's `roofline` objects; rocprofv3's kernel stats of the same command must agree
        policy.decode_events, T.backward_events = [], []
        # the interpreter's long-lived objects (torch's module graph: ~1e6 of them) leave the cyclic collector's view:
        # one full collection walking them is a 60 ms host stall (tools/train_steps.py), and the rollout's 16-byte
        # status read-back keeps the host at most one step ahead of the GPU, so the stall lands on the step time.
        # The collector stays on for what the loop itself allocates
        gc.collect()
        gc.freeze()
        for i in range(min(warmup, 3)):  # the chip idled through the collection: the region must not open on ramped-down clocks
            step(i)
        torch.cuda.synchronize()
        ar_events.clear()
        policy.decode_events, T.backward_events = [], []
        solo_ms = None
        if self.world > 1 and self.args.solo:
            # N = 1 reference of this invocation: rank 0 alone, the same K steps WITHOUT the collective (nobody is there to
            # answer: the region leaves out the all-reduce's launch and wait, stated in the detail file). Its optimizer steps
            # are undone afterwards — weights and Adam state restored — so the replicas enter the joint region identical
            self.barrier()
            if self.rank == 0:
                import copy

                snap_p = [p.detach().clone() for p in policy.parameters()]
                snap_o = copy.deepcopy(opt.state_dict())
                t0 = time.perf_counter()
                for i in range(steps):
                    step(warmup + i, collective=False)
                torch.cuda.synchronize()
                solo_ms = (time.perf_counter() - t0) / steps * 1e3
                with torch.no_grad():
                    for p, q in zip(policy.parameters(), snap_p):
                        p.copy_(q)
                opt.load_state_dict(snap_o)
                del snap_p, snap_o
                policy.decode_events, T.backward_events = [], []
        region_ms = []
        for r in range(self.args.regions):  # the contract's region first; the further ones only feed `region_ms_per_step`
            self.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                o = step(warmup + (r + 1) * steps + i)
            torch.cuda.synchronize()
            region_ms.append((time.perf_counter() - t0) / steps * 1e3)
            if r == 0:
                own_wall, out = time.perf_counter() - t0, o
        self.barrier()
        policy.check_backward_errors()
        wall = D.reduce_scalar(own_wall, "max", self.device)
        wall_min = -D.reduce_scalar(-own_wall, "max", self.device)
        region_ms = [D.reduce_scalar(x, "max", self.device) for x in region_ms]
        res = {"wall": wall}
        if self.rank != 0:
            return res
        res["region_ms_per_step"] = _spread(region_ms, steps)
        if self.world > 1:
            res["rank_ms_per_step"] = {"min": wall_min / steps * 1e3, "max": wall / steps * 1e3}
            if solo_ms is not None:
                res["n1_ms_per_step"] = solo_ms
                res["scaling_efficiency"] = solo_ms / (wall / steps * 1e3)
                res["n1_note"] = "solo region: rank 0 alone, no collective launched (its launch + wait are in the joint region only)"
        t_steps = out["actions"].shape[1]
        traj = batch * starts * self.world * steps
        ar_ms = [x.elapsed_time(y) for x, y in ar_events]
        roll_ms = [x.elapsed_time(y) for x, y in policy.decode_events]
        back_ms = [x.elapsed_time(y) for x, y in T.backward_events]
        policy.decode_events = T.backward_events = None
        n_nodes, d = num_loc, 128
        decisions = batch * starts * (t_steps - 1)  # the first (imposed) start node of every trajectory is not decoded
        # matrix work per decoded trajectory-step (2 x MACs): rollout = scores, glimpse, logits over the N nodes;
        # backward = the same three products recomputed + their five gradient products (d logit key, d glimpse,
        # d glimpse value, d glimpse key, d query)
        flop_roll = 3 * 2 * n_nodes * d
        flop_back = 8 * 2 * n_nodes * d

        def chain_roofline(kernel, ms, flop_per_decision, pmc_key):
            mean = sum(ms) / len(ms)
            tf = flop_per_decision * decisions / (mean * 1e-3) / 1e12
            pmc = {}
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", TRAIN_PMC_FILE))).get(pmc_key, {})
            except (OSError, ValueError):
                pass
            return {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None, "launch_ms_mean": mean, "launch_ms_min": min(ms),
                    "launches_timed": len(ms), "flop_per_trajectory_step": flop_per_decision, "trajectory_steps_per_launch": decisions,
                    "trajectory_steps_per_sec": decisions / (mean * 1e-3),
                    "note": "algorithmic matrix FLOPs at N nodes against the dense bf16 MFMA peak. Both kernels are per-instance "
                            "LATENCY CHAINS (16-step column blocks: dependent MFMAs, softmax VALU and LDS hand-overs add up "
                            "instead of overlapping), not throughput kernels: see pipe_utilisation for where the cycles go",
                    "pipe_utilisation": pmc or None,
                    "pipe_utilisation_source": f"profiles/{TRAIN_PMC_FILE} (separate rocprofv3 --pmc passes, tools/train_pmc.sh)" if pmc else None}
        res.update({
            "workload": (f"BASELINE configs[{cfg_idx}] per-GPU share: POMO (6L, instance norm) REINFORCE step, TSPEnv num_loc={num_loc}, "
                         f"{batch} instances x {starts} starts per GPU: multistart sampling rollout (MS decode kernel), "
                         "teacher-forced backward (MMA), bf16 training-encoder kernels, flat fp32 grad all-reduce, clip, Adam (torch fused)"),
            "ms_per_step": wall / steps * 1e3, "steps": steps, "warmup": warmup,
            "value": traj * t_steps / wall, "unit": "instance·step/s (trajectory steps, rollout + backward)",
            "trajectories_per_sec": traj / wall,
            "collective": {
                "backend": dist.get_backend() if dist.is_initialized() else None,
                "ranks": dist.get_world_size() if dist.is_initialized() else 0,
                "message_bytes": bucket.nbytes, "allreduce_ms_mean": sum(ar_ms) / len(ar_ms), "allreduce_ms_max": max(ar_ms),
                "note": "one all-reduce(sum) of the flat gradient bucket per optimizer step; 'nccl' is RCCL on ROCm",
            },
            "mean_reward": float(out["reward"].mean()),
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
        })
        if back_ms:
            res["roofline"] = chain_roofline("am_teacher_mma_kernel (teacher-forced backward, MFMA 16-step blocks): the step's dominant kernel",
                                             back_ms, flop_back, "am_teacher_mma_kernel")
        if roll_ms:
            res["rollout_roofline"] = chain_roofline("am_decode_ms_kernel (multistart sampling rollout on MFMA column tiles)",
                                                     roll_ms, flop_roll, "am_decode_ms_kernel")
        return res

    # -- parity against the reference's tours, every inference leg --------------------------------------------------------
    def parity(self, legs) -> dict | None:
        """Policy-level parity on TRAINED weights (tests/golden/weights, trained by tools/train_sharp.py) against the
        reference's own rollouts of the same weights and seeded instances at the full size of each inference leg's
        BASELINE config (tests/golden/trained, made by oracle/gen_trained_golden.py from the verbatim reference source):

        * ``fp32``: the parity configuration (torch fp32 encoder on the GPU, fp32 planes) vs the reference's fp32 run —
          flips and the proof that each is a near-tie (``flip_regret_max``: the product's own log-prob gap between its
          choice and the reference's at the first divergent step);
        * ``bf16``: the benchmarked configuration (MFMA encoder, bf16 planes) vs the reference's run under
          ``torch.autocast(bfloat16)`` and vs its fp32 run — identical tours, per-decision agreement, mean-reward gap;
        * sampling legs: the reference's seeded multinomial stream injected into the kernel.
        The random-init golden of round 2 (every greedy step a near-tie) is kept as ``random_init`` for continuity."""
        TP = _trained_parity()

        cases = {"c2_greedy": ("t2_tsp100_b4096_greedy", "greedy"), "c2_sampling": ("t2_tsp100_b4096_sampling", "sampling"),
                 "c3_greedy": ("t3_cvrp100_b4096_greedy", "greedy"), "c5_sampling": ("t5_cvrp500_b1024_sampling", "sampling")}
        try:
            TP.manifest()
        except (OSError, ValueError):
            return None
        out = {"weights": "tests/golden/weights/*.safetensors (AM-3L trained by tools/train_sharp.py; loaded into the reference's "
                          "own policy class by oracle/gen_trained_golden.py)",
               "goldens": "tests/golden/trained/*.npz (verbatim reference source, CPU, fp32 and bf16 autocast)"}
        dev = self.device
        for leg in legs:
            if leg not in cases:
                continue
            name, decode = cases[leg]
            case = TP.TrainedCase(name)
            rec = {"case": name, "batch": case.batch}
            if decode == "greedy":
                rec["fp32"] = TP.compare(case, "fp32", dev, against="fp32")
                rec["bf16_vs_reference_bf16_autocast"] = TP.compare(case, "bf16", dev, against="bf16")
                rec["bf16_vs_reference_fp32"] = TP.compare(case, "bf16", dev, against="fp32")
                rec["reference_bf16_vs_reference_fp32_identical"] = case.meta.get("reference_bf16_vs_fp32_identical")
            else:
                rec["fp32_reference_noise"] = TP.compare(case, "fp32", dev, against="fp32", decode="sampling", regret=False)
                rec["bf16_reference_noise"] = TP.compare(case, "bf16", dev, against="fp32", decode="sampling", regret=False)
                if leg == "c5_sampling":  # configs[4] greedy at the benchmarked batch: the bf16 pipeline vs the reference's
                    g = TP.TrainedCase("t5_cvrp500_b1024_greedy")
                    rec["greedy_fp32"] = TP.compare(g, "fp32", dev, against="fp32")
                    rec["greedy_bf16_vs_reference_bf16_autocast"] = TP.compare(g, "bf16", dev, against="bf16")
            out[leg] = rec
            torch.cuda.empty_cache()
        if "c4_train" in legs and "t4_pomo_tsp100_b256_msgreedy" in TP.manifest():
            # configs[3]'s POLICY (POMO, trained by the product) at its evaluation protocol (zoo/pomo/model.py:99-140):
            # a greedy rollout from every start node, and the best of 8 dihedral augmentations x 100 starts
            case = TP.TrainedCase("t4_pomo_tsp100_b256_msgreedy")
            out["c4_train"] = {
                "case": case.name, "batch": case.batch, "num_starts": case.num_starts,
                "fp32": TP.compare(case, "fp32", dev, against="fp32", decode="multistart_greedy", regret=False),
                "bf16_vs_reference_bf16_autocast": TP.compare(case, "bf16", dev, against="bf16", decode="multistart_greedy", regret=False),
                "bf16_vs_reference_fp32": TP.compare(case, "bf16", dev, against="fp32", decode="multistart_greedy", regret=False),
                "reference_bf16_vs_reference_fp32_identical": case.meta.get("reference_bf16_vs_fp32_identical"),
                "augmented_fp32": TP.compare_augmented(case, "fp32", dev),
                "augmented_bf16": TP.compare_augmented(case, "bf16", dev),
            }
            torch.cuda.empty_cache()
        head = out.get("c2_greedy")
        if head:
            out["fp32_flips"] = head["fp32"]["flips"]
            out["fp32_flip_regret_max"] = head["fp32"]["flip_regret_max"]
            out["bf16_identical_trajectories"] = head["bf16_vs_reference_bf16_autocast"]["identical"]
            out["bf16_identical_frac"] = head["bf16_vs_reference_bf16_autocast"]["identical_frac"]
            out["bf16_step_agreement"] = head["bf16_vs_reference_bf16_autocast"]["step_agreement"]
        out["random_init"] = self.parity_random_init()
        return out

    def parity_random_init(self) -> dict | None:
        """Round 2's block: greedy tours of configs[1] on seeded RANDOM-INIT weights (a near-uniform policy: every step a
        near-tie at the 1e-2 level) against the reference's (tests/golden/c2_tsp100_b4096_greedy.npz)."""
        import numpy as np

        from rl4co_amd.envs import get_env
        from rl4co_amd.policy import AttentionModelPolicy

        gdir = os.path.join(ROOT, "tests", "golden")
        try:
            meta = {c["name"]: c for c in json.load(open(os.path.join(gdir, "MANIFEST.json")))["cases"]}["c2_tsp100_b4096_greedy"]
            z = np.load(os.path.join(gdir, "c2_tsp100_b4096_greedy.npz"))
        except (OSError, KeyError, ValueError):
            return None
        ref_actions = torch.from_numpy(z["actions"].astype(np.int64)).to(self.device)
        ref_reward = torch.from_numpy(z["reward"]).to(self.device)
        env_cpu = get_env("tsp", generator_params=dict(num_loc=100, device="cpu"), device="cpu")
        torch.manual_seed(meta["data_seed"])
        data = env_cpu.generator(batch_size=[4096])
        if state_hash({k: v for k, v in data.items()}) != meta["inputs_sha256"]:
            return {"error": "seeded inputs differ from the golden run"}
        data = data.to(self.device)
        env = get_env("tsp", generator_params=dict(num_loc=100, device=self.device), device=self.device)
        out = {"golden": "tests/golden/c2_tsp100_b4096_greedy.npz", "of": 4096}
        configs = {
            "fp32_fold_on": dict(cache_dtype=torch.float32),
            "fp32_fold_off": dict(cache_dtype=torch.float32, fold=False),
            "bf16": dict(cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16),
        }
        for name, kw in configs.items():
            torch.manual_seed(meta["weight_seed"])
            pol = AttentionModelPolicy(env_name="tsp", **kw).eval()
            if state_hash(pol.state_dict()) != meta["weights_sha256"]:
                return {"error": "seeded weights differ from the golden run"}
            pol = pol.to(self.device)
            with torch.inference_mode():
                o = pol(env.reset(data.clone()), env, phase="test", decode_type="greedy")
            same = (o["actions"] == ref_actions).all(1)
            out[name] = {"identical_trajectories": int(same.sum()), "flips": int((~same).sum()),
                         "mean_reward": float(o["reward"].mean()), "mean_reward_reference": float(ref_reward.mean())}
        return out


def main() -> None:
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to the process's stdout at communicator
    # creation, so file descriptor 1 is pointed at stderr for the run and the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("RL4CO_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack every N seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["RL4CO_BENCH_WATCHDOG"]), repeat=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps of the headline leg (3.5 s of GPU time at the default)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--legs", default=DEFAULT_LEGS, help=f"comma-separated subset of {sorted(LEGS)}; the first one is the headline")
    ap.add_argument("--leg-steps", type=int, default=0, help="timed steps of every further leg (default: steps // 8, at least 10)")
    ap.add_argument("--batch", type=int, default=None, help="override the instances per GPU of the c2_greedy leg")
    ap.add_argument("--cache-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--encoder-dtype", default="bf16", choices=["bf16", "f32"],
                    help="GEMM/attention input type of the encoder and cache-fold GEMMs (bf16 = MFMA rate, the "
                         "reference's mixed-precision regime; f32 = the parity configuration)")
    ap.add_argument("--launch", default="pipeline", choices=["pipeline", "graph", "eager"],
                    help="inference legs: two captured rollouts in flight on two streams (default), one captured HIP graph "
                         "replayed per step, or kernel-by-kernel launches")
    ap.add_argument("--no-solo", dest="solo", action="store_false",
                    help="N > 1: skip the N = 1 reference region (rank 0 alone, same K steps) that `scaling_efficiency` is taken from")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file the full per-leg / parity detail goes to (the stdout line stays under 4 KB)")
    ap.add_argument("--train-encoder", default="stack", choices=["stack", "blocks"],
                    help="c4_train: the encoder's training forward as one launch for the whole stack (default) or per sub-block")
    ap.add_argument("--pipeline-depth", type=int, default=2, help="captured rollouts in flight (--launch pipeline)")
    ap.add_argument("--regions", type=int, default=5,
                    help="barrier-bracketed K-step regions per leg: the first is the contract's (value, ms_per_step), min / median / "
                         "max over all of them are reported as region_ms_per_step")
    ap.add_argument("--timeout", type=float, default=float(os.environ.get("RL4CO_BENCH_TIMEOUT", "1500")),
                    help="seconds after which a rank dumps every thread's stack and exits non-zero instead of hanging (a peer that "
                         "died leaves the others in a barrier; torch.distributed.run then tears the job down)")
    ap.add_argument("--no-check-solution", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the decode launch at 16 384 instances (roofline.frac_large_batch: the HBM rate with planes 4 x the MALL)")
    ap.add_argument("--cpu-sample-batch", type=int, default=4096,
                    help="instances of the same workload timed on the host cores (shrunk to keep the leg within ~30 s)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rollout engine has no CPU path")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started directly: become the launcher. One rank per GPU under torch.distributed.run
        # (what Lightning's DDP strategy does for the reference, utils/trainer.py:73-86), rendezvous on 127.0.0.1; the
        # ranks inherit this process's stdout, rank 0 writes the one JSON line
        shared = os.environ.get("RL4CO_BENCH_SHARED_GPU") == "1"
        if torch.cuda.device_count() < args.gpus and not shared:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
        from rl4co_amd.dist import _free_port

        os.dup2(json_fd, 1)
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                                  os.path.abspath(__file__), *sys.argv[1:]])
    if args.timeout > 0 and not os.environ.get("RL4CO_BENCH_WATCHDOG"):  # fail loudly, never hang: stacks to stderr, exit code 1
        import faulthandler

        faulthandler.dump_traceback_later(args.timeout, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    legs = [x for x in args.legs.split(",") if x]
    for x in legs:
        if x not in LEGS:
            raise SystemExit(f"unknown leg {x!r}; choose from {sorted(LEGS)}")
    # RL4CO_BENCH_SHARED_GPU=1 (testing the multi-rank path on a one-GPU box): ranks share the visible devices
    if os.environ.get("RL4CO_BENCH_SHARED_GPU") == "1":
        local_rank %= torch.cuda.device_count()
        # two processes on ONE device: kernel by kernel only. With captured graphs alive in both processes the later
        # training leg's collective stalled for seconds (measured r02: 3 s per step, once a hang) — hardware-queue
        # oversubscription between the processes, not a property of the one-process-per-GPU layout this mode imitates
        args.launch = "eager"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    from rl4co_amd import dist as D

    # "nccl" == RCCL on ROCm; rendezvous on 127.0.0.1. Initialised at N = 1 as well when the training leg runs, so
    # that its gradient all-reduce really goes through RCCL. RCCL refuses two ranks on one device, so the shared-GPU
    # test mode above rides on gloo (RL4CO_DIST_BACKEND) — the data path has no collective either way
    if world > 1 or "c4_train" in legs:
        D.init_process_group(os.environ.get("RL4CO_DIST_BACKEND", "nccl"), device=device, single_process_ok=True)

    # start-up self-check, before anything is timed: exactly N ranks, one GPU each (distinct device identities), the current
    # device is the assigned one, and the collective backend sums over all N — the first 8-GPU run is the driver's
    placement = None
    if dist.is_initialized():
        placement = D.check_placement(device, args.gpus, allow_shared=os.environ.get("RL4CO_BENCH_SHARED_GPU") == "1")
        if world > 1 and dist.get_backend() == "nccl" and placement["distinct_devices"] != args.gpus:
            raise SystemExit(f"rank {rank}: {placement['distinct_devices']} distinct GPUs for --gpus {args.gpus}")
        if rank == 0:
            log(f"placement: {placement['world']} rank(s) on {placement['distinct_devices']} device(s), backend {placement['backend']}")

    if os.environ.get("RL4CO_BENCH_KILL_RANK") == str(rank) and world > 1:  # tests/test_gpu_bench_cli.py: a rank that dies
        os._exit(17)
    bench = Bench(args, rank, world, device)
    # untimed device warm-up before the first leg: ~1 s of plain HBM reads (the read-probe kernel) so that clocks, the power
    # state and the code-object loader have settled when the first timed region starts — between fresh boxes the first two
    # legs of a process otherwise varied by up to 9 % (r04: 3.27 - 3.58 ms on the headline leg for identical kernels)
    from rl4co_amd import kernels as _K

    _buf = torch.empty(256 << 20, dtype=torch.uint8, device=device)
    _sink = torch.zeros(1, device=device)
    _t0 = time.perf_counter()
    while time.perf_counter() - _t0 < 1.0:
        for _ in range(20):
            _K.hbm_read_probe(_buf, _sink)
        torch.cuda.synchronize()
    del _buf, _sink
    leg_steps = args.leg_steps or max(10, args.steps // 8)
    results = {}
    # execution order: the training leg runs FIRST, in a process that has not captured a HIP graph yet (legs[0] stays
    # the headline of the JSON line). Observed with two ranks sharing one device: a training leg that followed graphed
    # rollout legs stalled in its gradient all-reduce; the order removes the interaction whatever its cause
    order = sorted(range(len(legs)), key=lambda j: (legs[j] != "c4_train", j))
    for i in order:
        leg = legs[i]
        # (the further legs warm up for at least five steps: their first region otherwise opens on a chip that idled through the
        # previous leg's teardown — r05: c4 20.8 ms in the first region against 20.1 - 20.3 in the four after it)
        k, w = (args.steps, args.warmup) if i == 0 else (leg_steps, max(5, args.warmup))
        if leg == "c4_train":
            k = min(k, max(5, args.steps // 20)) if i else k
            # first leg of the process — clocks, allocator, code objects: ten warm-up steps (r05, five: the first 5-step region
            # 20.6 ms against 20.0 - 20.1 in the four after it on every box)
            results[leg] = bench.train_leg(k, max(w, args.warmup, 10))
        else:
            results[leg] = bench.rollout_leg(leg, k, w)
        if rank == 0:
            log(f"{leg}: {results[leg]['ms_per_step']:.3f} ms/step over {k} steps")
        torch.cuda.empty_cache()

    if rank == 0:
        head_name = legs[0]
        head = results[head_name]
        env_name, num_loc, batch, decode, cfg_idx = LEGS[head_name]
        # achievable-stream ceiling on this box (float4 grid-stride read of 2 GiB)
        from rl4co_amd import kernels as K

        probe = torch.empty(2 << 30, dtype=torch.uint8, device=device)
        sink = torch.zeros(1, device=device)
        K.hbm_read_probe(probe, sink)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            K.hbm_read_probe(probe, sink)
        e1.record()
        torch.cuda.synchronize()
        probe_gbs = 5 * probe.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del probe
        head_cache, head_enc = leg_dtypes(head_name, args)
        detail = {
            "metric": "decode_steps_per_sec",
            "value": head["value"],
            "unit": head["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": head_cache if head_cache == head_enc else "mixed",
            "dtype_detail": f"encoder GEMMs/attention: {head_enc} MFMA inputs, fp32 accumulate; cache planes: "
                            f"{head_cache}; decode arithmetic (scores, softmax, logits, log-probs, reward): fp32",
            "data": "synthetic",
            "config": {
                "workload": head["workload"], "leg": head_name,
                "env": env_name, "num_loc": num_loc, "batch_per_gpu": args.batch or batch,
                "decode_type": decode, "cache_dtype": head_cache, "encoder_dtype": head_enc,
                "check_solution": not args.no_check_solution, "parallelism": f"replicas x{world} (instances sharded)",
            },
        }
        for k in ("node_steps_per_sec", "instances_per_sec", "mean_reward", "roofline", "rollout_roofline", "encoder_roofline", "host_gap_ms", "launch",
                  "untimed_steps", "untimed_note",
                  "eager_ms_per_step", "graph_ms_per_step", "region_ms_per_step", "n1_note",
                  "trajectories_per_sec", "collective", "rank_ms_per_step", "n1_ms_per_step", "scaling_efficiency"):
            if k in head:
                detail[k] = head[k]
        if "roofline" in detail:
            detail["roofline"]["hbm_read_probe_GBs"] = probe_gbs
            if head_name == "c2_greedy" and world == 1 and args.batch is None and not args.no_large_batch:
                big = bench.large_batch_probe(head_name)
                if big:
                    detail["roofline"]["large_batch"] = big
                    detail["roofline"]["frac_large_batch"] = big["frac"]  # 16 384 instances: planes 4 x the Infinity Cache
        if placement:
            detail["placement"] = placement
        detail["legs"] = {name: {k: v for k, v in r.items() if k != "wall"} for name, r in results.items() if name != head_name}
        if "c4_train" in results and head_name != "c4_train":
            detail["train_ms_per_step"] = results["c4_train"]["ms_per_step"]
            coll = results["c4_train"]["collective"]
            detail["collective_backend"] = coll["backend"]
            # "nccl" IS RCCL on ROCm; any other backend (gloo: the shared-GPU test mode) is reported under its own name
            detail["rccl_ranks" if coll["backend"] == "nccl" else "collective_ranks"] = coll["ranks"]
            detail["allreduce_ms"] = results["c4_train"]["collective"]["allreduce_ms_mean"]
        if world == 1 and not args.no_parity:
            log("parity vs the reference's tours (trained weights, every inference leg)")
            detail["parity"] = bench.parity(legs)
        if world == 1 and not args.no_cpu_baseline and head_name != "c4_train":
            detail["cpu_baseline"] = cpu_baseline(env_name, num_loc, args.cpu_sample_batch, repeats=5)
            detail["cpu_baseline"]["gpu_over_cpu"] = head["value"] / detail["cpu_baseline"]["value"]
        line = compact_line(detail, head_name, results, args.detail)
        try:  # everything the compact line leaves out: per-leg dicts, rooflines with their notes, the whole parity block
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as exc:
            log(f"could not write {args.detail}: {exc}")
            line.pop("detail_file", None)
        encoded = json.dumps(line, separators=(",", ":"))
        for optional in ("parity", "encoder_roofline", "legs"):  # never again a line the driver cannot parse
            if len(encoded.encode()) <= MAX_LINE_BYTES:
                break
            log(f"stdout line is {len(encoded.encode())} bytes: dropping {optional!r} (kept in the detail file)")
            line.pop(optional, None)
            encoded = json.dumps(line, separators=(",", ":"))
        assert len(encoded.encode()) <= MAX_LINE_BYTES, f"stdout line grew to {len(encoded.encode())} bytes"
        os.write(json_fd, (encoded + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
