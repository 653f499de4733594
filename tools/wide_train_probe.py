"""GPU box: a REINFORCE step beyond the backward kernels' node limit (CVRP-500 x 64 by default): where does the time go?

    python tools/wide_train_probe.py [env] [num_loc] [batch] [starts]
"""
import sys
import time
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

env_name = sys.argv[1] if len(sys.argv) > 1 else "cvrp"
num_loc = int(sys.argv[2]) if len(sys.argv) > 2 else 500
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
starts = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = AttentionModelPolicy(env_name, num_encoder_layers=6, normalization="instance", use_graph_context=False,
                              cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                              train_decode_type="multistart_sampling" if starts else "sampling").to(dev).train()
env = get_env(env_name, generator_params=dict(num_loc=num_loc, device=dev), device=dev, check_solution=False)
data = env.generator(batch_size=[batch])
opt = torch.optim.Adam(policy.parameters(), lr=1e-4, fused=True)


def sync_time(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


def step(i=[0]):
    i[0] += 1
    kw = dict(num_starts=starts) if starts else {}
    out = policy(env.reset(data), env, phase="train", seed=i[0], **kw)
    r, ll = out["reward"], out["log_likelihood"]
    loss = -((r - r.mean()).detach() * ll).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return out


with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    ms, out = sync_time(step)
    for x in {str(m.message)[:200] for m in w}:
        print("warning:", x)
print(f"{env_name}-{num_loc} x {batch} (starts {starts}): {ms:.1f} ms per REINFORCE step; horizon {out['actions'].shape[1]}")
with torch.no_grad():
    acts = out["actions"]
    ms_r, _ = sync_time(lambda: policy._replay(env.reset(data), acts, starts))
    print(f"  _replay alone: {ms_r:.1f} ms")
    ms_f, _ = sync_time(lambda: policy(env.reset(data), env, phase="test", decode_type="sampling", seed=3))
    print(f"  inference rollout (encoder + decode): {ms_f:.1f} ms")
