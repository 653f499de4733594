import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_teacher import _pomo_policy
from rl4co_amd.envs import get_env
for env_name, num_loc in (("tsp", 128), ("cvrp", 119), ("tsp", 50)):
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=False)
    torch.manual_seed(1)
    data = env.generator(batch_size=[64 if num_loc <= 50 else 16])
    kw = dict(normalization="instance", graph_context=False, env_name=env_name, dt=torch.bfloat16)
    ref_pol = _pomo_policy(fused=False, **kw)
    with torch.no_grad():
        out0 = ref_pol(env.reset(data), env, phase="train", num_starts=8, seed=3)
    acts = out0["actions"][:, 1:].contiguous()
    adv = torch.linspace(-1.0, 1.0, out0["actions"].shape[0], device="cuda")
    grads = {}
    for mode in ("torch", "blocks", "stack", "torch32"):
        pol = _pomo_policy(fused=mode in ("blocks", "stack"), **kw)
        pol.encoder.net.fused_stack = mode == "stack"
        if mode == "torch32":
            pol.encoder_autocast = None
            pol.cache_dtype = torch.float32
        out = pol(env.reset(data), env, phase="train", num_starts=8, actions=acts)
        (adv * out["log_likelihood"]).mean().backward()
        grads[mode] = torch.cat([p.grad.detach().float().flatten() for _, p in sorted(pol.named_parameters()) if p.grad is not None])
    cos = lambda a, b: float(a @ b) / float(a.norm() * b.norm())
    print(env_name, num_loc, {f"{a}~{b}": round(cos(grads[a], grads[b]), 5) for a, b in (("blocks", "torch"), ("stack", "torch"), ("stack", "blocks"),
          ("torch", "torch32"), ("blocks", "torch32"), ("stack", "torch32"))})
