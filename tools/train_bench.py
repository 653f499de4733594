"""BASELINE configs[3]-shaped training step: TSP-100, POMO policy (6 layers, instance norm, no
graph context), S starts per instance, multistart sampling rollout in the fused kernel (no grad),
teacher-forced differentiable log-likelihood, REINFORCE with the shared (mean-over-starts)
baseline (reinforce/baselines.py:55-59, pomo/model.py:88-111), ONE flat gradient all-reduce over
RCCL (rl4co_amd/dist.py), grad-norm clip 1.0, Adam(1e-4) — the reference's recipe
(utils/trainer.py:57-86, configs/experiment/routing/pomo.yaml).

    python tools/train_bench.py --batch 512 --starts 8 --steps 5
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rl4co_amd import dist as D  # noqa: E402
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512, help="instances per GPU")
ap.add_argument("--starts", type=int, default=8)
ap.add_argument("--num-loc", type=int, default=100)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--model", default="pomo", choices=["pomo", "am"], help="pomo: 6L instance norm, shared baseline over starts; am: 3L batch norm + graph context, batch-mean baseline")
ap.add_argument("--env", default="tsp", choices=["tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw"])
ap.add_argument("--no-release", action="store_true", help="accumulate into the zeroed bucket views instead of gathering fresh gradients")
ap.add_argument("--no-fused-encoder", action="store_true", help="torch encoder (library GEMMs, SDPA, autograd norm) for comparison")
args = ap.parse_args()

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
rank, world = D.init_process_group(device=device)

torch.manual_seed(0)
if args.model == "pomo":
    policy = AttentionModelPolicy(args.env, num_encoder_layers=6, normalization="instance", use_graph_context=False,
                                  cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                                  train_decode_type="multistart_sampling").to(device).train()
else:
    policy = AttentionModelPolicy(args.env, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                                  train_decode_type="multistart_sampling" if args.starts > 1 else "sampling").to(device).train()
if args.no_fused_encoder:
    from rl4co_amd.policy import _EncoderLayer
    for m in policy.modules():
        if isinstance(m, _EncoderLayer):
            m.fused_train = False
env = get_env(args.env, generator_params=dict(num_loc=args.num_loc, device=device), device=device, check_solution=False)
opt = torch.optim.Adam(policy.parameters(), lr=1e-4)
bucket = D.FlatGradBucket(policy)
torch.manual_seed(1234 + rank)
data = env.generator(batch_size=[args.batch])
S, B = args.starts, args.batch


def step(i):
    td = env.reset(data)
    out = policy(td, env, phase="train", seed=1000 * i + rank, **(dict(num_starts=S) if S > 1 else {}))
    reward = out["reward"].view(S, B).t()           # unbatchify -> [B, S]
    ll = out["log_likelihood"].view(S, B).t()
    # SharedBaseline over the starts (POMO); a single start falls back to the batch mean
    adv = reward - (reward.mean(dim=1, keepdim=True) if S > 1 else reward.mean())
    loss = -(adv.detach() * ll).mean()
    if args.no_release:
        bucket.zero_()
    else:
        bucket.release()
    loss.backward()
    bucket.allreduce_mean()
    torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
    opt.step()
    return float(reward.mean()), out["actions"].shape[1]


for i in range(args.warmup):
    step(i)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
t0 = time.perf_counter()
for i in range(args.steps):
    r, t = step(args.warmup + i)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
wall = time.perf_counter() - t0
if rank == 0:
    traj = B * S * world * args.steps
    print(json.dumps({"workload": f"{args.model.upper()} REINFORCE train step, {args.env.upper()}-{args.num_loc}, {B} instances x {S} starts per GPU"
                                  + (" (torch encoder)" if args.no_fused_encoder else ""),
                      "n_gpus": world, "ms_per_step": wall / args.steps * 1e3,
                      "trajectories_per_sec": traj / wall, "instance_steps_per_sec": traj * t / wall,
                      "grad_bucket_bytes": bucket.nbytes, "mean_reward": r,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
if world > 1:
    torch.distributed.destroy_process_group()
