#!/usr/bin/env python
"""Where does an instance's time go inside the fused 16-bit encoder? (r04; the teacher kernel's clock probe of r03 for
am_encoder.hip.)

Builds, in the build container (hipcc cross-compiles; the GPU box only runs), an INSTRUMENTED variant of
csrc/am_encoder.hip — `s_memtime` stamps at the phase boundaries of wave 0 of every workgroup, summed per phase into the
tail of the (otherwise unused) `hidden` output — links it against the product's other objects into
tools/probes/_build/librl4co_probe.so, and on the GPU box runs the encoder with it and prints the shares.

    python tools/enc_clock_probe.py build      # here
    gpurun -- 'python tools/enc_clock_probe.py run'
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "_build")  # git-ignored; travels to the GPU box with the snapshot
PHASES = ["init embedding", "QKV projections", "attention", "out-proj + norm1", "FFN + norm2", "fold + stores", "graph context"]


def build():
    from rl4co_amd import build as B

    B.build_library()
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(B.CSRC, "am_encoder.hip")).read()
    head = '''
#define PROBE(i) do { if ((threadIdx.x >> 6) == 0) { const long long _t = __builtin_readcyclecounter(); if (_prev) _acc[i] += _t - _prev; _prev = _t; } } while (0)
'''
    src = src.replace('#include "common.h"\n', '#include "common.h"\n' + head, 1)
    marks = [
        ("  int tid = threadIdx.x;\n  int w = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;  // (not const: see the top of the layer loop)",
         None, "  long long _acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long _prev = 0; _prev = __builtin_readcyclecounter();\n"),
        ("  for (int layer = 0; layer < a.num_layers; ++layer) {\n    // The lane indices pass", "  PROBE(0);\n", None),
        ("    // ---- attention for heads 2w, 2w+1 over all queries, wave-private", "    PROBE(1);\n", None),
        ("    // ---- out-proj + residual + norm1", "    PROBE(2);\n", None),
        ("    // ---- FFN: hidden in 4 chunks of 128, FFN2 accumulates across chunks", "    PROBE(3);\n", None),
        ("  // ---- optional: final node embeddings h (fp32)", "  PROBE(4);\n", None),  # (after the layer loop: the last layer's FFN)
        ("  // ---- graph context: project_fixed_context(mean_j h_j)  (decoder.py:216-219)", "  PROBE(5);\n", None),
    ]
    for anchor, before, after in marks:
        assert src.count(anchor) == 1, anchor
        src = src.replace(anchor, (before or "") + anchor + ("\n" + after if after else ""))
    # FFN of layers 0 .. L-2 ends where the next iteration begins: stamp at the end of the loop body
    end_anchor = "    if (layer + 1 < a.num_layers) stage_biases(layer + 1);\n    __syncthreads();\n  }\n"
    assert src.count(end_anchor) == 1
    src = src.replace(end_anchor, "    if (layer + 1 < a.num_layers) stage_biases(layer + 1);\n    __syncthreads();\n    PROBE(4);\n  }\n")
    src = src.replace("    PROBE(1);\n", "    PROBE(1);\n", 1)
    # the kernel's last statement: write the sums (fused kernel only: first `}\n\ntemplate <typename E, int TT, int VR4>\nint launch_encoder`)
    tail_anchor = "      if (lane == r) a.q_bias[(int64_t)b * kD + 32 * w + r] = acc;\n    }\n  }\n}\n"
    assert src.count(tail_anchor) == 1
    src = src.replace(tail_anchor, "      if (lane == r) a.q_bias[(int64_t)b * kD + 32 * w + r] = acc;\n    }\n  }\n  PROBE(6);\n"
                      "  if (a.hidden && threadIdx.x == 0)\n    for (int i = 0; i < 7; ++i) atomicAdd(a.hidden + (int64_t)a.B * a.N * kD + i, (float)_acc[i]);\n}\n")
    # the first QKV stamp must not count the time before the loop: PROBE(0) closes the init phase; inside the loop the
    # QKV phase is closed by PROBE(1)
    vsrc = os.path.join(OUT, "am_encoder_probe.hip")
    open(vsrc, "w").write(src.replace('#include "common.h"', f'#include "{B.CSRC}/common.h"'))
    obj = os.path.join(OUT, "am_encoder_probe.o")
    subprocess.run([B._hipcc(), *B.COMPILE_FLAGS, f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c", vsrc, "-o", obj], check=True)
    objs = [str(B.OBJ_DIR / (n + ".o")) for n in B.SOURCES if n != "am_encoder.hip"]
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "librl4co_probe.so"), obj, *objs], check=True)
    print("built", os.path.join(OUT, "librl4co_probe.so"))


def run():
    os.environ["RL4CO_AMD_LIB"] = os.path.join(OUT, "librl4co_probe.so")
    import torch

    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    for env_name, layers, norm in (("tsp", 3, "batch"), ("cvrp", 3, "batch"), ("tsp", 6, "instance")):
        torch.manual_seed(0)
        pol = AttentionModelPolicy(env_name, num_encoder_layers=layers, normalization=norm, cache_dtype=torch.bfloat16,
                                   encoder_autocast=torch.bfloat16).cuda().eval()
        env = get_env(env_name, generator_params=dict(num_loc=100, device="cuda"), device="cuda")
        td = env.reset(env.generator(batch_size=[4096]))
        pe = pol._packed_encoder()
        n = td["action_mask"].shape[-1]
        import rl4co_amd.encoder as E

        orig_empty = torch.empty

        def empty(shape, *a, **k):  # the probe sums live behind the [B, N, 128] hidden rows: 8 more floats
            if tuple(shape) == (4096, n, 128) and k.get("dtype") == torch.float32:
                buf = orig_empty(4096 * n * 128 + 8, *a, **k)
                buf[-8:] = 0
                empty.last = buf
                return buf[:-8].view(4096, n, 128)
            return orig_empty(shape, *a, **k)

        with torch.inference_mode():
            E.torch.empty = empty  # EVERY call of the instrumented kernel writes its sums behind `hidden`
            try:
                pe.encode(td, torch.bfloat16, want_hidden=True, act_dtype=torch.bfloat16)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pe.encode(td, torch.bfloat16, want_hidden=True, act_dtype=torch.bfloat16)
                e1.record()
                torch.cuda.synchronize()
            finally:
                E.torch.empty = orig_empty
        # two fp32 [B,N,128] buffers were asked for in that call (ctx_cur / ctx_first / hidden): the LAST one is hidden
        sums = empty.last[-8:-1].double().cpu()
        tot = float(sums.sum())
        print(f"{env_name}-100 x 4096, {layers} layers ({norm}): {e0.elapsed_time(e1):.3f} ms (with the fp32 hidden copy); "
              f"wave-0 cycles per instance {tot / 4096:.0f}")
        for name, v in zip(PHASES, sums.tolist()):
            print(f"    {name:20s} {100 * v / tot:5.1f} %   {v / 4096:9.0f} cycles / instance")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
