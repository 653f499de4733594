"""Phase clocks of the multistart rollout (probe build -DRL4CO_MS_PROBE=7 of am_decode_ms.hip, tools/ms_variants.sh):
shader-clock sums per step segment for wave 0 and for the other waves of the first 256 workgroups (one round).
    RL4CO_AMD_LIB=tools/probes/_build/lib_ms_phases.so python tools/ms_phases.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd import _lib
from rl4co_amd import kernels as K
from rl4co_amd.envs import get_env
from rl4co_amd.policy import AttentionModelPolicy
B, S = 4096, 8
torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16, num_encoder_layers=6,
                           normalization="instance", use_graph_context=False).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[B])
lib = C.CDLL(os.environ["RL4CO_AMD_LIB"])
clk = (C.c_ulonglong * 16)()
with torch.inference_mode():
    cache, _ = pol._packed_encoder().encode(td, torch.bfloat16)
    for it in range(3):
        st = pol._initial_state(td, S)
        actions = torch.zeros(B * S, 100, dtype=torch.int64, device="cuda")
        logps = torch.zeros(B * S, 100, device="cuda")
        err = K.new_error_word("cuda")
        torch.manual_seed(1)
        first = env.select_start_nodes(td, S)
        actions[:, 0] = first
        pol._env_step_state(st, first, err)
        torch.cuda.synchronize()
        lib.rl4co_ms_probe_read(clk, 1)
        K.am_decode(cache, st, mode="sampling", max_steps=99, t0=1, actions=actions, logps=logps, err=err, philox_seed=7, variant="ms")
        torch.cuda.synchronize()
    lib.rl4co_ms_probe_read(clk, 0)
names = ["noise drawn", "query ready (ctx row wait)", "scores/softmax/glimpse", "wait B1", "logits/pieces", "wait B2", "selection/state"]
for cls, off, waves in (("wave 0", 0, 256), ("waves 1-7", 8, 256 * 7)):
    tot = sum(clk[off + i] for i in range(7))
    print(cls, "cycles per step:", round(tot / waves / 99))
    for i, n in enumerate(names):
        print(f"   {n:32s} {clk[off + i] / waves / 99:8.0f}  {100.0 * clk[off + i] / max(tot, 1):5.1f} %")
