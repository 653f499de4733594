"""Multistart (POMO) decode microbench: B_inst instances x S starts, 100 customers, bf16 planes.

    [ENV=tsp|cvrp|op|pctsp|pdp|cvrptw] [MODE=sampling|greedy] python tools/ms_bench.py B_inst S variant...
"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
from rl4co_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = sys.argv[3:] or ["stream"]
MODE = __import__("os").environ.get("MODE", "sampling")
ENV = __import__("os").environ.get("ENV", "tsp")
torch.manual_seed(0)
pol = AttentionModelPolicy(ENV, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                           num_encoder_layers=6, normalization="instance", use_graph_context=False).cuda().eval()
env = get_env(ENV, generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[B])
with torch.inference_mode():
    cache, _ = pol._packed_encoder().encode(td, torch.bfloat16)
    for variant in variants:
        times = []
        n = td["action_mask"].shape[-1]
        tmax = n if ENV in ("tsp", "pctsp", "pdp") else (n + 2 if ENV == "op" else 2 * n)
        for it in range(4):
            st = pol._initial_state(td, S)
            actions = torch.zeros(B * S, tmax, dtype=torch.int64, device="cuda")
            logps = torch.zeros(B * S, tmax, device="cuda")
            n_steps = torch.zeros(B * S, dtype=torch.int32, device="cuda")
            err = K.new_error_word("cuda")
            torch.manual_seed(1)
            first = env.select_start_nodes(td, S)
            actions[:, 0] = first
            pol._env_step_state(st, first, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.am_decode(cache, st, mode=MODE, max_steps=tmax - 1, t0=1, actions=actions, logps=logps, err=err,
                        philox_seed=7, variant=variant, n_steps=n_steps)
            e1.record(); torch.cuda.synchronize()
            if not __import__("os").environ.get("MS_NO_CHECK"): K.raise_if_error(err)
            times.append(e0.elapsed_time(e1))
        ms = min(times[1:])
        steps = int(n_steps.sum())
        print(f"{ENV} B={B} S={S} {MODE} {variant}: {ms:.3f} ms ({steps/ms/1e3:.1f} M trajectory-steps/s; "
              f"{steps / (B * S):.1f} steps per trajectory, longest {int(n_steps.max())})")
