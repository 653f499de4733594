"""Per-step GPU time series of the c4_train step (HIP events around every step): shows clock / power drift over a run.

    python tools/train_steps.py [steps]
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rl4co_amd import dist as D  # noqa: E402
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
D.init_process_group(device=device, single_process_ok=True)
torch.manual_seed(0)
policy = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                              cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                              train_decode_type="multistart_sampling").to(device).train()
env = get_env("tsp", generator_params=dict(num_loc=100, device=device), device=device, check_solution=False)
opt = torch.optim.Adam(policy.parameters(), lr=1e-4)
bucket = D.FlatGradBucket(policy)
data = env.generator(batch_size=[4096])
S, B = 8, 4096
ev = []
walls = []
if os.environ.get('NOGC'):
    gc.collect(); gc.freeze(); gc.disable()
gc.callbacks.append(lambda ph, info: print('gc', ph, info) if info.get('generation') == 2 else None)
for i in range(steps):
    walls.append(time.perf_counter())
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    out = policy(env.reset(data), env, phase="train", seed=i, num_starts=S)
    e[1].record()
    reward = out["reward"].view(S, B).t()
    ll = out["log_likelihood"].view(S, B).t()
    loss = -((reward - reward.mean(dim=1, keepdim=True)).detach() * ll).mean()
    bucket.release()
    loss.backward()
    e[2].record()
    bucket.allreduce_mean()
    torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
    opt.step()
    e[3].record()
    ev.append(e)
torch.cuda.synchronize()
walls.append(time.perf_counter())
d = [(walls[i + 1] - walls[i]) * 1e3 for i in range(steps)]
print('host ms per step:', ' '.join(f'{x:.1f}' for x in d))
print('wall per step over steps 3..', (walls[-1] - walls[3]) / (steps - 3) * 1e3)
for i, e in enumerate(ev):
    if i < 12 or i % 8 == 0:
        print(f"step {i:3d}: forward {e[0].elapsed_time(e[1]):6.2f}  backward {e[1].elapsed_time(e[2]):6.2f}  "
              f"reduce+opt {e[2].elapsed_time(e[3]):5.2f}  total {e[0].elapsed_time(e[3]):6.2f} ms")
tot = [e[0].elapsed_time(e[3]) for e in ev]
print("first 10 mean", sum(tot[2:12]) / 10, " last 10 mean", sum(tot[-10:]) / 10)
