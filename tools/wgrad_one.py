import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd import train_ops as T
M = 409600
d = torch.randn(M, 512, device="cuda").bfloat16(); x = torch.randn(M, 128, device="cuda").bfloat16()
d1 = torch.randn(M, 128, device="cuda").bfloat16()
w = (torch.randn(128, 128, device="cuda") * 0.1).bfloat16()
for _ in range(4):
    T._wgrad(d, x, with_bias=True)      # N = 512, K = 128: four tiles sharing X
    T._wgrad(d1, x, with_bias=True)     # single tile
    T._gemm(x, w, None)                 # the persistent K = N = 128 GEMM (6 TB/s) for comparison
torch.cuda.synchronize()
