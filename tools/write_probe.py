#!/usr/bin/env python
"""Calibrate rocprofv3's WRITE_SIZE on known byte counts before reading the fused encoder's figure as write amplification
(the guide: "WRITE_SIZE [is] uncalibrated: calibrate on a known byte count in your own access pattern").

    gpurun -- 'cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wp -- python $GRAFT_REPO_ROOT/tools/write_probe.py'

Launches, in this order: torch fill of 1 GiB fp32 (16-byte stores per lane), a 1 GiB device copy, a 512 MiB fp32 -> bf16
cast (reads 512 MiB, writes 256 MiB), then the fused encoder on TSP-100 x 4096 (algorithmic writes: three bf16 planes
+ two fp32 context tables + the graph-context rows = 736 MB). tools/write_probe_parse.py turns the counter CSV into ratios.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    dev = torch.device("cuda", 0)
    x = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    y = torch.empty_like(x)
    for _ in range(2):
        x.fill_(1.0)
        y.copy_(x)
        z = x[: 1 << 27].to(torch.bfloat16)
    torch.cuda.synchronize()
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).to(dev).eval()
    env = get_env("tsp", generator_params=dict(num_loc=100, device=dev), device=dev)
    td = env.reset(batch_size=[4096])
    pe = pol._packed_encoder()
    with torch.inference_mode():
        for _ in range(3):
            pe.encode(td, torch.bfloat16)
    torch.cuda.synchronize()
    del z


if __name__ == "__main__":
    main()
