#!/usr/bin/env python
"""Cross-batch overlap experiment: run the encoder of batch k+1 UNDER the decode of batch k.

Two ``GraphedRollout``s over the same policy, each with its own static buffers, replayed alternately on two HIP
streams, against the one-stream baseline — a stream of DISTINCT batches in both cases. The encoder is matrix-core /
latency bound, the decode HBM bound, so in principle they could share the chip; whether they do is a residency
question: the persistent decode launch holds 4096 waves = 16 per CU = 4 per SIMD at 124 VGPRs each (tools/
kernel_resources.py am_decode.hip), i.e. the whole register file, for its full 2.5 ms — an encoder workgroup (4 waves x
256 VGPRs, 70 KB LDS) can only start where a decode wave has RETIRED.

    gpurun -- python tools/overlap_bench.py            # prints one JSON object with both timings
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main() -> None:
    from rl4co_amd.envs import get_env
    from rl4co_amd.graph import GraphedRollout
    from rl4co_amd.policy import AttentionModelPolicy

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = {}
    for env_name, num_loc, batch in (("tsp", 100, 4096), ("tsp", 100, 2048), ("cvrp", 100, 4096)):
        torch.manual_seed(0)
        pol = AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).to(dev).eval()
        env = get_env(env_name, generator_params=dict(num_loc=num_loc, device=dev), device=dev)
        torch.manual_seed(1)
        data = [env.generator(batch_size=[batch]) for _ in range(4)]
        steps = 200
        # -- baseline: one stream, one graph, distinct batches back to back ------------------------------------------------
        g = GraphedRollout(pol, env, data[0], decode_type="greedy")
        for i in range(10):
            g(data[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = g(data[i % 4])
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / steps * 1e3
        want = [g(d)["reward"].clone() for d in data]
        # -- two graphs in flight on two streams -----------------------------------------------------------------------------
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        graphs = []
        for s in streams:
            with torch.cuda.stream(s):
                graphs.append(GraphedRollout(pol, env, data[0], decode_type="greedy"))
        torch.cuda.synchronize()

        def replay(k, i):
            """enqueue batch i on pipeline k without the read-back; returns the deferred finisher"""
            gr, s = graphs[k], streams[k]
            with torch.cuda.stream(s), torch.inference_mode():
                for key, v in gr.static_in.items():
                    v.copy_(data[i % 4][key], non_blocking=True)
                gr.graph.replay()
            return gr._finish

        for i in range(10):
            fin = replay(i % 2, i)
            with torch.cuda.stream(streams[i % 2]):  # the read-back must be ordered after the replay: same stream
                fin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = [None, None]
        ok = True
        for i in range(steps):
            k = i % 2
            if pending[k] is not None:
                with torch.cuda.stream(streams[k]):
                    o = pending[k][0]()  # read-back of the batch this pipeline ran two iterations ago
                    if i < 12:
                        ok = ok and torch.equal(o["reward"], want[pending[k][1] % 4])
            pending[k] = (replay(k, i), i)
        for k in range(2):
            if pending[k] is not None:
                with torch.cuda.stream(streams[k]):
                    pending[k][0]()
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / steps * 1e3
        res[f"{env_name}{num_loc}_b{batch}"] = {"one_stream_ms_per_batch": one, "two_streams_ms_per_batch": two,
                                                "speedup": one / two, "results_identical": bool(ok)}
        del g, graphs
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
