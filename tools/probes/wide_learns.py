"""GPU box: a few REINFORCE (POMO, shared baseline) steps beyond 128 nodes on the r06 path: does the tour length go down?"""
import sys
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, batch, starts, steps = 150, 128, 8, 60
pol = AttentionModelPolicy("tsp", num_encoder_layers=3, normalization="instance", use_graph_context=False,
                           cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                           train_decode_type="multistart_sampling").to(dev).train()
env = get_env("tsp", generator_params=dict(num_loc=n, device=dev), device=dev, check_solution=False)
opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
hist = []
for i in range(steps):
    data = env.generator(batch_size=[batch])
    out = pol(env.reset(data), env, phase="train", seed=i, num_starts=starts)
    r = out["reward"].view(starts, batch).t()
    ll = out["log_likelihood"].view(starts, batch).t()
    loss = -((r - r.mean(1, keepdim=True)).detach() * ll).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)
    opt.step()
    hist.append(float(-r.mean()))
    assert torch.isfinite(loss)
print(f"TSP-{n} x {batch} x {starts} starts: mean tour length first 5 steps {sum(hist[:5]) / 5:.3f}, last 5 of {steps}: {sum(hist[-5:]) / 5:.3f}")
