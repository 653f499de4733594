"""GPU box: error of csrc/am_train_attn.hip's backward against fp32 SDPA autograd on the same 16-bit q | k | v, and its time.
   RL4CO_AMD_LIB=<lib> python tools/attn_bwd_err.py"""
import ctypes as C
import os

import torch
import torch.nn.functional as F

lib = C.CDLL(os.environ.get("RL4CO_AMD_LIB", "rl4co_amd/lib/librl4co_amd.so"))
vp = C.c_void_p
for b, n, scale in ((4096, 100, 1.0), (256, 100, 0.5), (256, 50, 1.0), (256, 128, 2.0)):
    torch.manual_seed(0)
    qkv = (torch.randn(b, n, 384, device="cuda") * scale).bfloat16()
    go = torch.randn(b, n, 128, device="cuda").bfloat16()
    out = torch.empty(b, n, 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(b, 8, n, device="cuda")
    dqkv = torch.empty_like(qkv)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.rl4co_attn_fwd(1, vp(qkv.data_ptr()), b, n, vp(out.data_ptr()), vp(lse.data_ptr()), vp(s)) == 0

    def bwd():
        return lib.rl4co_attn_bwd(1, vp(qkv.data_ptr()), vp(out.data_ptr()), vp(go.data_ptr()), vp(lse.data_ptr()), b, n, vp(dqkv.data_ptr()), vp(s))

    assert bwd() == 0
    torch.cuda.synchronize()
    bb = min(b, 256)
    qr = qkv[:bb].float().requires_grad_(True)
    q, k, v = qr.view(bb, n, 3, 8, 16).permute(2, 0, 3, 1, 4).unbind(0)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(bb, n, 128)
    (gr,) = torch.autograd.grad(ref, [qr], go[:bb].float())
    errs = [float((dqkv[:bb, :, sl].float() - gr[..., sl]).norm() / gr[..., sl].norm()) for sl in (slice(0, 128), slice(128, 256), slice(256, 384))]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        bwd()
    e0.record()
    for _ in range(20):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    print(f"B {b} N {n} scale {scale}: rel err dq {errs[0]:.5f} dk {errs[1]:.5f} dv {errs[2]:.5f}   {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
