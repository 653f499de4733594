"""GPU box: where a workgroup of the attention backward spends its shader clocks (probe build: tools/kernel_variant.sh
am_train_attn.hip ab_p3 "-DRL4CO_ATTN_BWD_PROBE=3"):  RL4CO_AMD_LIB=tools/probes/_build/lib_ab_p3.so python tools/attn_bwd_phases.py"""
import ctypes as C
import os

import torch

lib = C.CDLL(os.environ["RL4CO_AMD_LIB"])
vp = C.c_void_p
for b, n in ((4096, 100), (256, 100), (4096, 50)):
    torch.manual_seed(0)
    qkv = torch.randn(b, n, 384, device="cuda").bfloat16()
    go = torch.randn(b, n, 128, device="cuda").bfloat16()
    out = torch.empty(b, n, 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(b, 8, n, device="cuda")
    dqkv = torch.empty_like(qkv)
    s = torch.cuda.current_stream().cuda_stream
    lib.rl4co_attn_fwd(1, vp(qkv.data_ptr()), b, n, vp(out.data_ptr()), vp(lse.data_ptr()), vp(s))
    bwd = lambda: lib.rl4co_attn_bwd(1, vp(qkv.data_ptr()), vp(out.data_ptr()), vp(go.data_ptr()), vp(lse.data_ptr()), b, n, vp(dqkv.data_ptr()), vp(s))  # noqa: E731
    for _ in range(3):
        bwd()
    torch.cuda.synchronize()
    clk = (C.c_ulonglong * 4)()
    lib.rl4co_attn_bwd_probe_read(clk, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    lib.rl4co_attn_bwd_probe_read(clk, 1)
    wgs = 2 * b * reps
    per = [c / wgs for c in clk]
    print(f"B {b} N {n}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch; clocks per workgroup: load k|v {per[0]:.0f}, query blocks {per[1]:.0f}, "
          f"barrier {per[2]:.0f}, stage + store {per[3]:.0f}  (sum {sum(per):.0f})")
