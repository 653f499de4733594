"""GPU box: does the fp32 encoder run UNDER the decode launches when it is issued on a high-priority stream of its own?

The fp32 rollout is a matrix-bound encoder launch (one 8-wave workgroup per CU: 118 KB of LDS, 2 x 184 registers per SIMD)
followed by an HBM-bound persistent decode launch (4096 one-wave workgroups, 128 registers each: 16 per CU when alone, 4 per
CU beside an encoder workgroup). Two plain streams lock into D D E E (tools/timeline_dump.py): the sum of the launches.
Here every rollout's encoder goes to ONE high-priority stream, back to back, and the decode launches (normal priority)
take what is left; the encoders may run ahead by `depth` batches.

    python tools/prio_pipeline_probe.py [fp32|bf16] [depth] [steps] [prio|plain]
"""
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from rl4co_amd.envs import get_env  # noqa: E402
from rl4co_amd.policy import AttentionModelPolicy  # noqa: E402

regime = sys.argv[1] if len(sys.argv) > 1 else "fp32"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
mode = sys.argv[4] if len(sys.argv) > 4 else "prio"
dev = torch.device("cuda:0")
kw = dict(cache_dtype=torch.float32, encoder_autocast=None) if regime == "fp32" else dict(cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16)
torch.manual_seed(0)
policy = AttentionModelPolicy(env_name="tsp", **kw).to(dev).eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device=dev), device=dev, check_solution="--nocheck" not in sys.argv)
torch.manual_seed(1234)
data = env.generator(batch_size=[4096])
lo, hi = -1, 0
print("stream priority range (greatest, least):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
streams = [torch.cuda.Stream(device=dev, priority=0) for _ in range(depth)]
if mode == "prio":
    policy.encoder_stream = torch.cuda.Stream(device=dev, priority=-1)
elif mode == "shared":  # one shared encoder stream of normal priority (is it the priority or the order that matters?)
    policy.encoder_stream = torch.cuda.Stream(device=dev, priority=0)

with torch.inference_mode():
    ref = policy(env.reset(data), env, phase="test", decode_type="greedy")
    torch.cuda.synchronize()
    pending = []

    def submit(i):
        s = streams[i % depth]
        with torch.cuda.stream(s):
            fin = policy(env.reset(data), env, phase="test", decode_type="greedy", _defer_finish=True)
        pending.append((s, fin))

    def collect():
        s, fin = pending.pop(0)
        with torch.cuda.stream(s):
            return fin()

    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for i in range(steps):
            if len(pending) == depth:
                outs.append(collect())
            submit(i)
        while pending:
            outs.append(collect())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        same = all(torch.equal(o["actions"], ref["actions"]) for o in outs)
        print(f"{regime} {mode} depth {depth}: {ms:.3f} ms per batch of 4096 (pass {rep}); tours identical to the single-stream rollout: {same}")
