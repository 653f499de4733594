#!/usr/bin/env python
"""Train the SHARP weight sets the parity goldens are quoted on.

With random-init weights every greedy step of the AttentionModel is a near-tie (a near-uniform policy), so the
benchmarked bf16 configuration cannot be compared tour-by-tour with the reference. This tool trains the reference's
DEFAULT architecture (``AttentionModelPolicy(env_name)``: 3 layers, batch norm, graph context — zoo/am/policy.py:50-122)
with the product itself on one MI355X: multistart sampling rollouts (MS decode kernel), shared baseline over the
starts (zoo/pomo/model.py:88-111), teacher-forced MMA backward, bf16 training-encoder kernels, Adam. The resulting
``state_dict`` has the reference's keys; ``oracle/gen_golden.py`` loads it into the reference's own policy class
(imported verbatim) to produce the fp32 and bf16-autocast goldens.

    gpurun -- python tools/train_sharp.py --env tsp --steps 4000 --out gpurun_out/weights
    cp gpurun_out/weights/am_tsp100_sharp.safetensors tests/golden/weights/

Deterministic given (seed, steps) up to atomics in the batch-norm statistics kernels.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="tsp")
    ap.add_argument("--num-loc", type=int, default=100)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--starts", type=int, default=16)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/weights")
    ap.add_argument("--arch", default="am", choices=("am", "pomo"),
                    help="am: the AttentionModel default (3 layers, batch norm, graph context); pomo: BASELINE configs[3]'s "
                         "policy (zoo/pomo/model.py:45-60: 6 layers, instance norm, no graph context)")
    args = ap.parse_args()

    from safetensors.torch import save_file

    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(args.seed)
    arch = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False) if args.arch == "pomo" else {}
    policy = AttentionModelPolicy(args.env, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                                  train_decode_type="multistart_sampling", **arch).to(dev).train()
    env = get_env(args.env, generator_params=dict(num_loc=args.num_loc, device=dev), device=dev, check_solution=False)
    opt = torch.optim.Adam(policy.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(args.steps * 0.8), int(args.steps * 0.95)], gamma=0.3)
    torch.manual_seed(args.seed + 4321)
    val = env.generator(batch_size=[1024])
    b, s = args.batch, args.starts
    log = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        data = env.generator(batch_size=[b])
        out = policy(env.reset(data), env, phase="train", seed=1_000_003 * (args.seed + 1) + i, num_starts=s)
        reward = out["reward"].view(s, b).t()
        ll = out["log_likelihood"].view(s, b).t()
        adv = reward - reward.mean(dim=1, keepdim=True)
        loss = -(adv.detach() * ll).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
        opt.step()
        sched.step()
        if i % 250 == 0 or i == args.steps - 1:
            policy.eval()
            with torch.inference_mode():
                v = policy(env.reset(val), env, phase="test", decode_type="greedy")
            policy.train()
            rec = {"step": i, "train_reward": float(reward.mean()), "val_greedy_reward": float(v["reward"].mean()),
                   "elapsed_s": time.perf_counter() - t0}
            log.append(rec)
            print(json.dumps(rec), flush=True)
    policy.check_backward_errors()
    policy.eval()
    os.makedirs(args.out, exist_ok=True)
    name = f"{args.arch}_{args.env}{args.num_loc}_sharp"
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().contiguous() for k, v in policy.state_dict().items()}
    save_file(sd, os.path.join(args.out, name + ".safetensors"),
              metadata={"trained_by": "tools/train_sharp.py", "args": json.dumps(vars(args)),
                        "final": json.dumps(log[-1])})
    json.dump({"args": vars(args), "log": log}, open(os.path.join(args.out, name + ".train_log.json"), "w"), indent=1)
    print(f"wrote {name}.safetensors ({sum(v.numel() for v in sd.values())} values)")


if __name__ == "__main__":
    main()
