"""Which ATen ops (fills, copies, reductions, casts) a POMO training step of bench.py's c4 leg issues, by op and input
shape — torch.profiler over two steps after warm-up. Output: gpurun_out/train_glue.txt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rl4co_amd import dist as D  # noqa: E402
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

dev = torch.device("cuda:0")
starts, batch = 8, 4096
torch.manual_seed(0)
policy = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                              cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                              train_decode_type="multistart_sampling").to(dev).train()
env = get_env("tsp", generator_params=dict(num_loc=100, device=dev), device=dev, check_solution=False)
opt = torch.optim.Adam(policy.parameters(), lr=1e-4)
bucket = D.FlatGradBucket(policy)
data = env.generator(batch_size=[batch])


def step(i):
    out = policy(env.reset(data), env, phase="train", seed=1000 * i, num_starts=starts)
    reward = out["reward"].view(starts, batch).t()
    ll = out["log_likelihood"].view(starts, batch).t()
    adv = reward - reward.mean(dim=1, keepdim=True)
    loss = -(adv.detach() * ll).mean()
    bucket.release()
    loss.backward()
    bucket._rebind()
    torch.nn.utils.clip_grad_norm_(policy.parameters(), 1.0)
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(2):
        step(10 + i)
    torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "train_glue.txt"), "w") as f:
    f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=80, max_name_column_width=60,
                                                                max_shapes_column_width=70))
    f.write("\n\n")
    f.write(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=90))
print("written")
