"""Micro-benchmark of the teacher-forced backward kernel alone (TSP-N, B instances x S starts)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
from rl4co_amd import teacher
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                           cache_dtype=torch.bfloat16).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=N, device="cuda"), device="cuda", check_solution=False)
td = env.reset(batch_size=[B])
with torch.no_grad():
    out = pol(td, env, phase="train", decode_type="multistart_sampling" if S > 1 else "sampling", num_starts=S, seed=1)
    h, _ = pol.encoder(td)
    g = teacher.build_cache_autograd("tsp", h, pol.decoder)
    cache = teacher.detached_cache("tsp", g, torch.bfloat16)
for k in ("kvl", "ctx_first", "ctx_cur", "q_step0"):
    g[k] = g[k].detach().requires_grad_(True)
meta = dict(t0=1 if S > 1 else 0, mask_inner=True, mask_logits=True, tanh_clipping=10.0, temperature=1.0)
logps = torch.zeros(out["actions"].shape, device="cuda")
grad = torch.ones(out["actions"].shape, device="cuda")
steps = B * S * (N - (1 if S > 1 else 0))
for variant in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("replay", "mma")):
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        teacher.run_backward(cache, out["actions"], grad, meta, variant=variant)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"teacher backward [{variant}] B={B} S={S} N={N}: {ms:.2f} ms  ({steps/ms/1e3:.1f} M trajectory-steps/s)")


# (r06) the fused-fold layout of the training step: planes and context tables as 16-bit column blocks of ONE [B, N, 5 * 128]
# matrix, all five gradient blocks out of the kernel (ctx_dtype, d_ctx_in_planes) — and the two halves of that separately
if len(sys.argv) > 5:
    import dataclasses
    dt = torch.bfloat16
    b, n = cache.num_instances, cache.num_nodes
    big = torch.zeros(b, n, 5, 128, dtype=dt, device="cuda")
    big[:, :, :3] = cache.kvl.permute(1, 2, 0, 3)
    big[:, :, 3] = cache.ctx_first.to(dt)
    big[:, :, 4] = cache.ctx_cur.to(dt)
    kvl = big.permute(2, 0, 1, 3)[:3]
    forms = {
        "cols16+planes5": (dataclasses.replace(cache, kvl=kvl, ctx_cur=big[:, :, 4], ctx_first=big[:, :, 3]), 5),
        "cols16+planes3": (dataclasses.replace(cache, kvl=kvl, ctx_cur=big[:, :, 4], ctx_first=big[:, :, 3]), 3),
        "fp32ctx+planes5": (dataclasses.replace(cache, kvl=kvl), 5),
        "fp32ctx+planes3": (dataclasses.replace(cache, kvl=kvl), 3),
    }
    for name, (c, nb) in forms.items():
        dp = torch.empty(b, n, nb, 128, dtype=dt, device="cuda")
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            teacher.run_backward(c, out["actions"], grad, meta, variant="mma", d_planes=dp.permute(2, 0, 1, 3))
            e1.record(); torch.cuda.synchronize()
        print(f"teacher backward [mma, {name}]: {e0.elapsed_time(e1):.2f} ms")
