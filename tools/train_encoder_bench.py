#!/usr/bin/env python
"""Training encoder alone (POMO-6L, instance norm, TSP-100 x 4096, bf16 autocast): forward and backward, the fused stack
forward (rl4co_am_encoder_train_fwd) against the per-sub-block kernels; GPU time by HIP events, host time by the clock."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                           cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().train()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda", check_solution=False)
td = env.reset(env.generator(batch_size=[4096]))
go = torch.randn(4096, 100, 128, device="cuda", dtype=torch.bfloat16)
for fused in (True, False, True, False):
    pol.encoder.net.fused_stack = fused
    res = []
    for it in range(6):
        pol.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t0 = time.perf_counter()
        e[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h, _ = pol.encoder(td)
        e[1].record()
        t1 = time.perf_counter()
        h.backward(go)
        e[2].record()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        res.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3))
    f, b, hf, hb, tot = (sorted(x)[len(x) // 2] for x in zip(*res[2:]))
    print(f"fused_stack={fused}: GPU forward {f:.2f} ms, backward {b:.2f} ms | host issue forward {hf:.2f} ms, backward {hb:.2f} ms | "
          f"wall {tot:.2f} ms   (peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")
