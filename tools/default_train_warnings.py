"""Which RuntimeWarnings does a REINFORCE step of the reference's DEFAULT model (AttentionModelPolicy(): 3 layers, batch norm,
graph context) raise at TSP-100 — under fp16 autocast (Lightning's default "16-mixed"), bf16 autocast and without autocast?
  Also times the step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl4co_amd.envs import get_env
from rl4co_amd.policy import AttentionModelPolicy

env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda", check_solution=False)
torch.manual_seed(0)
data = env.generator(batch_size=[int(sys.argv[1]) if len(sys.argv) > 1 else 1024])
for regime in (torch.float16, torch.bfloat16, None):
    torch.manual_seed(1)
    pol = AttentionModelPolicy("tsp").cuda().train()
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
    def step(i):
        ctx = torch.autocast("cuda", dtype=regime) if regime is not None else torch.autocast("cuda", enabled=False)
        with ctx:
            out = pol(env.reset(data), env, phase="train", seed=i)
        adv = out["reward"] - out["reward"].mean()
        loss = -(adv.detach() * out["log_likelihood"]).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step(0)
    msgs = sorted({str(x.message)[:200] for x in w if issubclass(x.category, RuntimeWarning)})
    for _ in range(2):
        step(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        step(2 + i)
    torch.cuda.synchronize()
    print(f"regime {regime}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms / step; RuntimeWarnings: {len(msgs)}")
    for m in msgs:
        print("   -", m)
