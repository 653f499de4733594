"""Time rl4co_mlp_input_grad at the C4 shape (409 600 rows) against the two GEMM launches it replaces.
RL4CO_MLP_BWD_TILES=2|4 selects the workgroup's tile count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl4co_amd import train_ops as T

m, dt = 4096 * 100, torch.bfloat16
torch.manual_seed(0)
dy = (torch.randn(m, 128, device="cuda") * 0.5).to(dt)
h = torch.relu(torch.randn(m, 512, device="cuda")).to(dt)
w1 = (torch.randn(512, 128, device="cuda") * 0.08).to(dt)
w2 = (torch.randn(128, 512, device="cuda") * 0.05).to(dt)
w1_t, w2_t = w1.t().contiguous(), w2.t().contiguous()
p1, p2 = T._pack_stack(w1_t[None])[0], T._pack_stack(w2_t[None])[0]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def two():
    dh = T._gemm(dy, w2_t, mask=h)
    return T._gemm(dh, w1_t, residual=dy)


print("tiles", os.environ.get("RL4CO_MLP_BWD_TILES", "default"), "fused us", round(timeit(lambda: T.mlp_input_grad(dy, h, p1, p2)), 1),
      "two GEMM launches us", round(timeit(two), 1))
