"""GPU box: the persistent K = N = 128 linear kernel (csrc/am_train_ops.hip: linear_k128_n128_kernel) against fp32 torch, and its time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd import train_ops as T

for dt in (torch.bfloat16, torch.float16):
    for m in (409600, 409600 - 77, 100, 128 * 769 + 5):
        for relu in (False, True):
            torch.manual_seed(m)
            a = torch.randn(m, 128, device="cuda").to(dt)
            w = (torch.randn(128, 128, device="cuda") * 0.1).to(dt)
            b = torch.randn(128, device="cuda")
            out = T._gemm(a, w, b, relu=relu)
            ref = torch.nn.functional.linear(a.float(), w.float(), b)
            ref = ref.relu() if relu else ref
            err = float((out.float() - ref).norm() / ref.norm())
            assert out.shape == ref.shape and err < (4e-3 if dt == torch.bfloat16 else 6e-4), (dt, m, relu, err)
            exact = torch.equal(out, ref.to(dt))
            print(dt, m, relu, f"rel err {err:.2e}", "== fp32 result rounded once" if exact else f"differs in {(out != ref.to(dt)).float().mean():.2e} of the entries")
a = torch.randn(409600, 128, device="cuda").bfloat16()
w = (torch.randn(128, 128, device="cuda") * 0.1).bfloat16()
for _ in range(3):
    T._gemm(a, w, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    T._gemm(a, w, None)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"K = N = 128, M = 409600: {us:.1f} us, {409600 * 256 * 2 / us / 1e6:.2f} TB/s")
