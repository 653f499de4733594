#!/usr/bin/env python
"""Time the exact-fp32 fused encoder (csrc/am_encoder_f32.hip) alone: HIP events around `PackedEncoder.encode`.

    python tools/enc_f32_bench.py [--env tsp] [--num-loc 100] [--batch 4096] [--layers 3] [--norm batch]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="tsp")
    ap.add_argument("--num-loc", type=int, nargs="+", default=[100])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--norm", default="batch")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    for num_loc in a.num_loc:
        torch.manual_seed(0)
        pol = AttentionModelPolicy(a.env, num_encoder_layers=a.layers, normalization=a.norm).cuda().eval()
        env = get_env(a.env, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
        td = env.reset(env.generator(batch_size=[a.batch]))
        pe = pol._packed_encoder()
        with torch.inference_mode():
            for _ in range(3):
                pe.encode(td, torch.float32, act_dtype=torch.float32)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                pe.encode(td, torch.float32, act_dtype=torch.float32)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        n = td["action_mask"].shape[-1]
        d, ff = 128, 512
        nb = 5 if a.env == "tsp" else 4
        flop = a.layers * (2 * n * d * 3 * d + 4 * n * n * d + 2 * n * d * d + 4 * n * d * ff) + nb * 2 * n * d * d
        npad = 16 * ((n + 15) // 16)
        flop_pad = a.layers * (2 * npad * d * 3 * d + 4 * npad * npad * d + 2 * npad * d * d + 4 * npad * d * ff) + nb * 2 * npad * d * d
        print(f"{a.env}-{num_loc} x {a.batch}, {a.layers} layers ({a.norm}): {ms:.3f} ms  "
              f"{flop * a.batch / ms / 1e9:.1f} TF/s algorithmic ({flop * a.batch / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak), "
              f"{flop_pad * a.batch / ms / 1e9:.1f} TF/s issued (tokens padded to {npad})")


if __name__ == "__main__":
    main()
