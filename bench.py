#!/usr/bin/env python
"""bench.py — the rollout hot path on N MI355X GPUs of one node (driver contract).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic instances already resident in
HBM: ``env.reset`` -> AttentionModel policy greedy rollout (encoder, cache fold, ONE persistent
launch of the fused decode kernel for all T decode steps, tour-length reward, validity check).
Workload at every N: BASELINE.json configs[1] — TSPEnv num_loc=100, batch 4096 per GPU, AM
(3 layers, d=128, 8 heads), bf16 cache planes with fp32 arithmetic, greedy. Instances shard
across ranks with no data-path collective (weak scaling; SURVEY.md §8e: inference = replicas).

Rank 0 prints ONE JSON line. ``value`` = whole-job instance·decode-steps per second
(B·T·N_gpus·K / wall), wall = max over ranks of the barrier-bracketed timed region.
``roofline``: the decode kernel's algorithmic bytes per launch (SURVEY.md §8d per instance-step
figure x B x T) / its mean launch duration, measured here with HIP events on the launch stream.
``cpu_baseline``: the reference path (torch restatement, pinned bit-exact to the reference's own
source by oracle/gen_golden.py) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(msg: str) -> None:
    """Progress goes to stderr; stdout carries exactly one JSON line."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes_per_instance_step(env_name: str, n: int, elem: int) -> int:
    """3·N·d·e (K_g,V_g,K_l) + N (mask read) + N (mask write) + context rows
    (TSP: 2·d·e, CVRP: d·e + 8) + d·4 (graph context) + 16 (action, logp, scalars)."""
    d = 128
    ctx = 2 * d * elem if env_name == "tsp" else d * elem + 8
    return 3 * n * d * elem + 2 * n + ctx + d * 4 + 16


def cpu_baseline(env_name: str, num_loc: int, sample_batch: int, repeats: int, chunk: int = 512) -> dict:
    """Reference path on the host cores: oracle restatement (stock ATen ops, fp32, same ops in
    the same order as the reference), greedy rollout, span = reset -> policy -> reward."""
    from oracle import reference_torch as R

    # host cores this process may actually run on (cgroup/affinity aware: os.cpu_count() can
    # report the whole machine inside a CPU-limited container and oversubscribe OpenMP)
    avail = max(1, len(os.sched_getaffinity(0)))
    default_threads = max(1, min(avail, torch.get_num_threads()))
    env = R.get_env(env_name, num_loc, check_solution=True)
    torch.manual_seed(0)
    pol = R.AttentionModelPolicy(env_name).eval()
    torch.manual_seed(1234)
    data = env.generate(sample_batch)
    times, steps = [], 0
    with torch.inference_mode():
        # the reference's many small ATen ops do not scale to 100+ threads: probe a few thread
        # counts on a small batch and time the baseline at the fastest one (stated in `cores`)
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
        # size the sample so that the whole leg stays within ~30 s of CPU work on any host
        budget_b = int(30.0 / (repeats + 1) / max(per_inst, 1e-6))
        if budget_b < sample_batch:
            sample_batch = max(probe_b, budget_b)
            data = {k: v[:sample_batch] for k, v in data.items()}
        # the host path is fastest at a few hundred instances per call (13.6 s for one call of 4096 vs 8 x 0.5 s
        # for the same instances in calls of 512, measured on the GPU box): the sample is rolled out chunk by chunk
        chunk = min(chunk, sample_batch)
        log(f"cpu_baseline: {threads} threads, sample {sample_batch} instances in calls of {chunk} "
            f"(probe {per_inst * 1e3:.2f} ms/instance)")
        for i in range(repeats + 1):
            t0 = time.perf_counter()
            rewards, work = [], 0
            for lo in range(0, sample_batch, chunk):
                td = env.reset({k: v[lo : lo + chunk].clone() for k, v in data.items()})
                out = pol(td, env, phase="test", decode_type="greedy")
                rewards.append(out["reward"])
                work += out["actions"].shape[0] * out["actions"].shape[1]  # instance·steps of this call
            dt = time.perf_counter() - t0
            if i > 0:  # first pass is the warm-up
                times.append(dt)
    best = min(times)
    return {
        "value": work / best,
        "unit": "instance·step/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{env_name.upper()}-{num_loc}, {sample_batch} instances in calls of {chunk}, greedy full rollout "
                  f"(encoder+decode+reward), fp32 torch CPU, best of {repeats} passes after 1 warm-up, {best:.3f} s each",
        "mean_reward": float(torch.cat(rewards).mean()),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--env", default="tsp", choices=["tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw"])
    ap.add_argument("--num-loc", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--cache-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--encoder-dtype", default="bf16", choices=["bf16", "f32"],
                    help="GEMM/attention input type of the encoder and cache-fold GEMMs (bf16 = MFMA rate, the "
                         "reference's mixed-precision regime; f32 = the parity configuration)")
    ap.add_argument("--decode", default="greedy", choices=["greedy", "sampling"])
    ap.add_argument("--no-check-solution", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=4096,
                    help="instances of the same workload timed on the host cores (shrunk to keep the leg within ~30 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rollout engine has no CPU path")
    # RL4CO_BENCH_SHARED_GPU=1 (testing the multi-rank path on a one-GPU box): ranks share the visible devices
    if os.environ.get("RL4CO_BENCH_SHARED_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    from rl4co_amd import dist as D

    if world > 1:
        # "nccl" == RCCL on ROCm; rendezvous on 127.0.0.1. RCCL refuses two ranks on one device, so the shared-GPU
        # test mode above rides on gloo (RL4CO_DIST_BACKEND) — the data path has no collective either way
        D.init_process_group(os.environ.get("RL4CO_DIST_BACKEND", "nccl"), device=device)

    from rl4co_amd import kernels as K
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    cache_dtype = torch.bfloat16 if args.cache_dtype == "bf16" else torch.float32
    torch.manual_seed(0)  # random-init weights of the reference architecture, identical on every rank
    enc_dtype = torch.bfloat16 if args.encoder_dtype == "bf16" else None
    policy = AttentionModelPolicy(env_name=args.env, cache_dtype=cache_dtype, encoder_autocast=enc_dtype).to(device).eval()
    env = get_env(args.env, generator_params=dict(num_loc=args.num_loc, device=device), device=device,
                  check_solution=not args.no_check_solution)
    torch.manual_seed(1234 + rank)  # each rank owns its shard of the synthetic instances
    data = env.generator(batch_size=[args.batch])
    torch.cuda.synchronize()

    def step():
        td = env.reset(data)
        return policy(td, env, phase="test", decode_type=args.decode)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: instances resident, warming up")
    with torch.inference_mode():
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        log("timing")
        policy.decode_events = []
        policy.encode_events = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        wall = time.perf_counter() - t0
    log(f"timed region done: {wall:.3f} s for {args.steps} steps")
    decode_ms = [a.elapsed_time(b) for a, b in policy.decode_events]
    encode_ms = [a.elapsed_time(b) for a, b in policy.encode_events]
    policy.decode_events = policy.encode_events = None
    t_steps = out["actions"].shape[1]
    n_nodes = args.num_loc + (0 if args.env == "tsp" else 1)

    # whole-job numbers: wall = max over ranks, work = sum over ranks (each rank owns its shard)
    wall = D.reduce_scalar(wall, "max", device)
    total_instance_steps = int(D.reduce_scalar(args.batch * t_steps * args.steps, "sum", device))

    if rank == 0:
        elem = 2 if args.cache_dtype == "bf16" else 4
        per_unit = algorithmic_bytes_per_instance_step(args.env, n_nodes, elem)
        # units one launch streams: trajectories stop reading the cache once done (CVRP), so the
        # kernel's own step count is used, not B x T_max (identical for TSP)
        units_per_launch = policy.last_instance_steps
        bytes_per_launch = per_unit * units_per_launch
        mean_decode_ms = sum(decode_ms) / len(decode_ms)
        achieved = bytes_per_launch / (mean_decode_ms * 1e-3) / 1e9
        # achievable-stream ceiling on this box (float4 grid-stride read of 2 GiB)
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
        value = total_instance_steps / wall
        workload = (f"BASELINE configs[1]: {args.env.upper()}Env num_loc={args.num_loc} batch={args.batch}/GPU "
                    f"AttentionModel(3L,d128,h8) {args.decode} rollout, {args.encoder_dtype} encoder GEMMs, {args.cache_dtype} cache")
        # HBM bytes per decode launch from the separate rocprofv3 --pmc passes (tools/profile_round.sh
        # + tools/profile_parse.py write profiles/pmc_traffic.json); null when this workload was not profiled
        traffic = None
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = table.get(workload, {}).get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            pass
        line = {
            "metric": "decode_steps_per_sec",
            "value": value,
            "unit": "instance·step/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if (args.cache_dtype == "bf16" and args.encoder_dtype == "bf16") else
                     ("f32" if (args.cache_dtype == "f32" and args.encoder_dtype == "f32") else "mixed"),
            "dtype_detail": f"encoder GEMMs/attention: {args.encoder_dtype} MFMA inputs, fp32 accumulate; cache planes: "
                            f"{args.cache_dtype}; decode arithmetic (scores, softmax, logits, log-probs, reward): fp32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "env": args.env, "num_loc": args.num_loc, "batch_per_gpu": args.batch, "decode_steps": t_steps,
                "decode_type": args.decode, "cache_dtype": args.cache_dtype, "encoder_dtype": args.encoder_dtype,
                "check_solution": not args.no_check_solution, "parallelism": f"replicas x{world} (instances sharded)",
            },
            "node_steps_per_sec": value * n_nodes,
            "instances_per_sec": args.batch * world * args.steps / wall,
            "mean_reward": float(out["reward"].mean()),
            "roofline": {
                "kernel": "am_decode_kernel (fused persistent rollout, all T decode steps in one launch)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_instance_step": per_unit,
                "instance_steps_per_launch": units_per_launch,
                "bytes_per_launch": bytes_per_launch,
                "launch_ms_mean": mean_decode_ms,
                "launch_ms_min": min(decode_ms),
                "us_per_decode_step": mean_decode_ms * 1e3 / t_steps,
                "launches_timed": len(decode_ms),
                "hbm_read_probe_GBs": probe_gbs,
                # the kernel skips the cache rows of masked nodes (exact zeros in the reference's
                # formulation), so the algorithmic figure can exceed what HBM really moves; the
                # PMC traffic over the same launch time is the true HBM utilisation
                "hbm_utilisation_from_traffic": (traffic / (mean_decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            },
        }
        if encode_ms:  # second kernel of the step: fused encoder + cache fold on the matrix cores
            n_, d_, ff_, layers_ = n_nodes, 128, 512, 3
            flop_inst = layers_ * (2 * n_ * d_ * 3 * d_ + 4 * n_ * n_ * d_ + 2 * n_ * d_ * d_ + 4 * n_ * d_ * ff_) \
                + (5 if args.env == "tsp" else 4) * 2 * n_ * d_ * d_
            enc_ms = sum(encode_ms) / len(encode_ms)
            tf = flop_inst * args.batch / (enc_ms * 1e-3) / 1e12
            line["encoder_roofline"] = {
                "kernel": "am_encoder_kernel (init embedding + 3 x [MHA, norm, FFN, norm] + cache fold, one workgroup per instance)",
                "bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                "flop_per_instance": flop_inst, "launch_ms_mean": enc_ms,
                "note": "algorithmic FLOPs at N nodes (the kernel pads to 128 tokens); bf16 dense MFMA peak",
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.env, args.num_loc, args.cpu_sample_batch, repeats=2)
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
