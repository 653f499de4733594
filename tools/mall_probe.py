"""Three launches whose memory-side counters tell HBM from Infinity-Cache (MALL) service (tools/mall_probe.sh runs this under
rocprofv3 --pmc): (a) hbm_read_probe over 2 GiB — every read an HBM read; (b) the same kernel re-reading 128 MiB ten times —
after the first launch every read a MALL hit; (c) the decode launch of the headline leg (TSP-100 x 4096, bf16 planes).
The per-read fabric latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) of (c) sits between those of (b) and (a) in proportion
to the share of its reads the MALL serves."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd import kernels as K  # noqa: E402
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
dev = "cuda"
sink = torch.zeros(1, device=dev)
big = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
small = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
for _ in range(5):
    K.hbm_read_probe(big, sink)
torch.cuda.synchronize()
for _ in range(12):
    K.hbm_read_probe(small, sink)
torch.cuda.synchronize()
torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", cache_dtype=dt, encoder_autocast=None if dt == torch.float32 else dt).to(dev).eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device=dev), device=dev)
torch.manual_seed(1234)
data = env.generator(batch_size=[B])
with torch.inference_mode():
    for _ in range(4):
        out = pol(env.reset(data), env, phase="test", decode_type="greedy")
torch.cuda.synchronize()
print("reward", float(out["reward"].mean()))
