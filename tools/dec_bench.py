"""Micro-benchmark of the decode kernel alone (TSP-100 x 4096, bf16 planes), both variants."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
from rl4co_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variants = sys.argv[2:] or ["stream", "lds"]
torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[B])
with torch.inference_mode():
    cache, _ = pol._packed_encoder().encode(td, torch.bfloat16)
    for variant in variants:
        times = []
        for it in range(6):
            st = pol._initial_state(td, 0)
            actions = torch.zeros(B, 100, dtype=torch.int64, device="cuda")
            logps = torch.zeros(B, 100, device="cuda")
            err = K.new_error_word("cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.am_decode(cache, st, mode="greedy", max_steps=100, actions=actions, logps=logps, err=err, variant=variant)
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = min(times[1:])
        rew = K.tour_length(td["locs"], actions, negate=True).mean().item()
        print(f"B={B} {variant}: {ms:.3f} ms  ({ms*10:.1f} us/step, {B*100*78040/ms/1e6:.0f} GB/s algorithmic)  reward {rew:.4f}")
