"""GPU box: rl4co_linear_* (persistent stripes, csrc/am_train_ops.hip) on every shape / epilogue the training path uses, against
fp32 torch on the same 16-bit operands, incl. row counts that end inside a stripe and fewer stripes than workgroups."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd import train_ops as T

worst = 0.0
for dt in (torch.bfloat16, torch.float16):
    for m in (409600, 128 * 513 + 77, 100, 128 * 700):
        for k, n, kind in ((384, 128, "residual"), (512, 128, "residual"), (128, 512, "mask"), (128, 384, "bias"), (128, 128, "bias"), (512, 128, "relu"),
                           (640, 128, "plain"), (128, 640, "plain")):
            torch.manual_seed(m + k)
            a = torch.randn(m, k, device="cuda").to(dt)
            w = (torch.randn(n, k, device="cuda") * 0.1).to(dt)
            aux = torch.randn(m, n, device="cuda").to(dt)
            b = torch.randn(n, device="cuda")
            ref = a.float() @ w.float().t()
            if kind == "residual":
                out = T._gemm(a, w, None, residual=aux); ref = ref + aux.float()
            elif kind == "mask":
                out = T._gemm(a, w, None, mask=aux); ref = ref * (aux.float() > 0)
            elif kind == "bias":
                out = T._gemm(a, w, b); ref = ref + b
            elif kind == "relu":
                out = T._gemm(a, w, b, relu=True); ref = (ref + b).relu()
            else:
                out = T._gemm(a, w, None)
            err = float((out.float() - ref).norm() / ref.norm())
            worst = max(worst, err / (4e-3 if dt == torch.bfloat16 else 6e-4))
            assert err < (4e-3 if dt == torch.bfloat16 else 6e-4), (dt, m, k, n, kind, err)
print("all shapes ok; worst error / tolerance", round(worst, 3))
