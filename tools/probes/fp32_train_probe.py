import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from rl4co_amd.envs import get_env
from rl4co_amd.policy import AttentionModelPolicy
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda", check_solution=False)
torch.manual_seed(0)
data = env.generator(batch_size=[4096])
pol = AttentionModelPolicy("tsp").cuda().train()
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
def step(i):
    out = pol(env.reset(data), env, phase="train", seed=i)
    adv = out["reward"] - out["reward"].mean()
    loss = -(adv.detach() * out["log_likelihood"]).mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4): step(3 + i)
torch.cuda.synchronize(); print(f"fp32 default model TSP-100 x 4096: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms / step")
