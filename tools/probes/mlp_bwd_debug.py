import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl4co_amd import train_ops as T
from rl4co_amd.policy import _GraphAttentionNetwork

names = None
grads = {}
for fused in (True, False):
    torch.manual_seed(0)
    net = _GraphAttentionNetwork(8, 128, 3, "instance", 512).cuda().train()
    torch.manual_seed(1)
    x = torch.randn(64, 100, 128, device="cuda") * 0.7
    g = torch.randn(64, 100, 128, device="cuda")
    T.FUSED_MLP_INPUT_GRAD = fused
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xin = x.clone().requires_grad_()
        out = net(xin)
    (out.float() * g).sum().backward()
    names = ["x"] + [n for n, _ in net.named_parameters()]
    grads[fused] = [xin.grad.clone()] + [p.grad.clone() for p in net.parameters()]
for n, a, b in zip(names, grads[True], grads[False]):
    cos = float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
    print(f"{n:40s} cos {cos:.6f}  |a| {float(a.norm()):.4e} |b| {float(b.norm()):.4e}")
# same path twice: run-to-run determinism
torch.manual_seed(0)
