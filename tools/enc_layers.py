"""GPU box: fused encoder time vs number of layers (per-layer cost and fixed cost: init embedding + fold + stores)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
env = get_env("tsp", generator_params=dict(num_loc=int(sys.argv[1]) if len(sys.argv) > 1 else 100, device="cuda"), device="cuda")
td = env.reset(batch_size=[4096])
for L in (1, 2, 3, 6):
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", num_encoder_layers=L, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
    pe = pol._packed_encoder()
    with torch.inference_mode():
        for _ in range(3): pe.encode(td, torch.bfloat16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): pe.encode(td, torch.bfloat16)
        e1.record(); torch.cuda.synchronize()
    print(f"layers {L}: {e0.elapsed_time(e1) / 10:.3f} ms")
