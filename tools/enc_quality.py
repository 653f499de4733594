"""GPU: numerical quality of the fused encoder (rel. Frobenius error vs the fp32 torch encoder, beside torch's own bf16
autocast path) and rollout tour quality of the three encoders over several instance sets (is a mean-reward gap noise?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env

def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
for seed in range(5):
    torch.manual_seed(100 + seed)
    td = env.reset(batch_size=[4096])
    with torch.inference_mode():
        cache, hidden = pol._packed_encoder().encode(td, torch.float32, want_hidden=True)
        h32, _ = pol.encoder(td)
        ref = pol.decoder.precompute_cache(h32, torch.float32, torch.float32)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h16, _ = pol.encoder(td)
        auto = pol.decoder.precompute_cache(h16, torch.float32, torch.bfloat16)
        r = {}
        pol.fused_encoder = True
        r["fused"] = float(pol(td, env, phase="test", decode_type="greedy")["reward"].mean())
        pol.fused_encoder = False
        r["torch_bf16"] = float(pol(td, env, phase="test", decode_type="greedy")["reward"].mean())
        pol.encoder_autocast = None
        pol.cache_dtype = torch.float32
        r["torch_fp32"] = float(pol(td, env, phase="test", decode_type="greedy")["reward"].mean())
        pol.encoder_autocast, pol.cache_dtype, pol.fused_encoder = torch.bfloat16, torch.bfloat16, True
    print(f"seed {seed}: hidden rel err fused {rel(hidden, h32):.4e} autocast {rel(h16, h32):.4e} | logit_key fused "
          f"{rel(cache.kvl[2], ref.kvl[2]):.4e} autocast {rel(auto.kvl[2], ref.kvl[2]):.4e} | glimpse_key fused "
          f"{rel(cache.kvl[0], ref.kvl[0]):.4e} autocast {rel(auto.kvl[0], ref.kvl[0]):.4e} | rewards {r}")
