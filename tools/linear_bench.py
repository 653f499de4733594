"""Micro-benchmarks of the training kernels on the training shapes (M = 4096 x 100 token rows): rl4co_linear vs hipBLASLt,
rl4co_wgrad (partial-sum reduction included), skip + instance norm and self-attention forward / backward."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rl4co_amd import train_ops as T

M = 409600
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

for k, n, relu in ((128, 384, False), (128, 128, False), (128, 512, True), (512, 128, False), (384, 128, False)):
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    us = bench(lambda: T._gemm(a, w, b, relu=relu))
    gb = (M * k + M * n) * 2 / 1e9
    lib = bench(lambda: torch.nn.functional.linear(a, w))
    print(f"linear  K={k:4d} N={n:4d}: {us:7.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s  ({gb*1e3:5.0f} MB)   hipBLASLt {lib:7.1f} us")
# backward-side epilogues: ReLU mask (d hidden), residual (skip gradient), the fold GEMM both ways
for k, n, kind in ((128, 512, "mask"), (512, 128, "residual"), (384, 128, "residual"), (128, 640, "plain"), (640, 128, "plain")):
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    aux = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    kw = {"mask": dict(mask=aux), "residual": dict(residual=aux), "plain": {}}[kind]
    us = bench(lambda: T._gemm(a, w, None, **kw))
    gb = (M * k + M * n * (1 if kind == "plain" else 2)) * 2 / 1e9
    print(f"linear  K={k:4d} N={n:4d} {kind:8s}: {us:7.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s  ({gb*1e3:5.0f} MB)")
for n, k in ((384, 128), (128, 128), (512, 128), (128, 512)):
    d = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    us = bench(lambda: T._wgrad(d, x, with_bias=True))
    gb = (M * k + M * n) * 2 / 1e9
    print(f"wgrad   N={n:4d} K={k:4d}: {us:7.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s")

# skip connection + instance norm (POMO), forward and backward, 4096 instances x 100 nodes
x = torch.randn(4096, 100, 128, device="cuda").to(torch.bfloat16).requires_grad_(True)
s_ = torch.randn(4096, 100, 128, device="cuda").to(torch.bfloat16).requires_grad_(True)
gamma = torch.ones(128, device="cuda", requires_grad=True)
beta = torch.zeros(128, device="cuda", requires_grad=True)
us_f = bench(lambda: T.skip_instance_norm(x, s_, gamma, beta))
out = T.skip_instance_norm(x, s_, gamma, beta)
dout = torch.randn_like(out)
us_fb = bench(lambda: torch.autograd.grad(T.skip_instance_norm(x, s_, gamma, beta), (x, s_, gamma, beta), dout))
mb = 4096 * 100 * 128 * 2 / 1e6
print(f"skip_inorm fwd: {us_f:7.1f} us ({4 * mb:.0f} MB -> {4 * mb / us_f:.2f} TB/s)   fwd+bwd: {us_fb:7.1f} us "
      f"(bwd alone ~{us_fb - us_f:.1f} us, {3 * mb:.0f} MB -> {3 * mb / (us_fb - us_f):.2f} TB/s)")

# training attention (8 heads, per instance), forward and backward
qkv = torch.randn(4096, 100, 384, device="cuda").to(torch.bfloat16).requires_grad_(True)
us_f = bench(lambda: T.attention(qkv))
o = T.attention(qkv)
do = torch.randn_like(o)
us_fb = bench(lambda: torch.autograd.grad(T.attention(qkv), (qkv,), do))
print(f"attention fwd: {us_f:7.1f} us   fwd+bwd: {us_fb:7.1f} us (bwd alone ~{us_fb - us_f:.1f} us)")
