#!/usr/bin/env python
"""Where do the cycles of one 16-step block of the teacher-forced backward go? (timing probe, r03)

Builds a variant of csrc/am_teacher_mma.hip with shader-clock reads (s_memtime) at the stage boundaries of the block
loop; lane 0 of every wave adds the elapsed cycles of each segment to an LDS table (ds_add_u64), flushed to a global one at the end,
which the run reads back and prints per wave as a share of the launch. The variant lives in tools/probes/bin/ (git-
ignored; travels with the gpurun snapshot) as a small library holding only the teacher entry point.

    python tools/teacher_clock_probe.py build          # build container
    gpurun -- python tools/teacher_clock_probe.py run  # MI355X
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "rl4co_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "bin")

SEGMENTS = ["0 query", "1 scores+softmax", "2 glimpse", "B1 wait", "3 logits+lse pieces", "B2 wait", "3b lse + d logits", "B3 wait",
            "4 dO, dKl", "5 softmax bwd, dV, dS, dQ, dKg", "6 ctx atomics", "B4 wait", "group setup (tables of <= 4 trajectories)"]

# (unique anchor in the source, clock index, insert "before" or "after" the anchor line)
MARKS = [
    ("      // ---- 1. scores^T and softmax numerators", 0, "before"),
    ("      // ---- 2. glimpse O_h^T = V_h^T P^T", 1, "before"),
    ("      rl4co::lds_barrier();  // B1: all heads' glimpses", 2, "before"),
    ("      rl4co::lds_barrier();  // B1: all heads' glimpses", 3, "after"),
    ("      rl4co::lds_barrier();  // B2: log-sum-exp pieces of all node tiles", 4, "before"),
    ("      rl4co::lds_barrier();  // B2: log-sum-exp pieces of all node tiles", 5, "after"),
    ("      rl4co::lds_barrier();  // B3: d logits of all node tiles", 6, "before"),
    ("      rl4co::lds_barrier();  // B3: d logits of all node tiles", 7, "after"),
    ("      // ---- 5. softmax backward of head h; d values, d keys, d query", 8, "before"),
    ("      // ---- 6. d query -> context rows, graph context", 9, "before"),
    ("      rl4co::lds_barrier();  // B4: the glimpse / d-logit blocks are rewritten", 10, "before"),
    ("      rl4co::lds_barrier();  // B4: the glimpse / d-logit blocks are rewritten", 11, "after"),
    ("    for (int tb = 0; tb < ntb; ++tb) {", 12, "before"),
]

PRELUDE = """
__device__ unsigned long long g_clk[8][16];
#define CLK(i) { const long long _n = __builtin_readcyclecounter(); if (lane == 0) __hip_atomic_fetch_add(&s_clk[w][i], (unsigned long long)(_n - _last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); _last = _n; }
extern "C" int rl4co_debug_clocks(unsigned long long* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(g_clk)) != hipSuccess) return 2;
  if (reset) { static unsigned long long z[8][16]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)) != hipSuccess) return 3; }
  return 0;
}
"""


PRIO = "  if (w >= 4) __builtin_amdgcn_s_setprio(1);\n"
VARIANTS = {"clk": None, "base": PRIO, "prio_none": "", "prio_flip": "  if (w < 4) __builtin_amdgcn_s_setprio(1);\n",
            "prio_odd": "  if (w & 1) __builtin_amdgcn_s_setprio(1);\n"}


def variant_source(kind="clk") -> str:
    text = open(os.path.join(SRC, "am_teacher_mma.hip")).read()
    assert text.count(PRIO) == 1
    if kind != "clk":
        return text.replace(PRIO, VARIANTS[kind])
    lines = text.split("\n")
    out = []
    for ln in lines:
        hits = [(i, pos) for a, i, pos in MARKS if ln.startswith(a)]
        for i, pos in hits:
            if pos == "before":
                out.append(f"      CLK({i});")
        out.append(ln)
        for i, pos in hits:
            if pos == "after":
                out.append(f"      CLK({i});")
    s = "\n".join(out)
    for a, i, pos in MARKS:
        assert sum(l.startswith(a) for l in lines) == 1, f"anchor not unique / missing: {a}"
    s = s.replace("namespace {\n\nconstexpr int kD", PRELUDE + "\nnamespace {\n\nconstexpr int kD", 1)
    assert "g_clk" in s
    # the running timestamp: scalar, starts at the top of every trajectory
    anchor = "  for (int s0 = 0; s0 < S; s0 += G) {\n"
    assert s.count(anchor) == 1
    s = s.replace(anchor, "  __shared__ unsigned long long s_clk[8][16];\n  if (tid < 128) s_clk[tid >> 4][tid & 15] = 0;\n  __syncthreads();\n"
                  "  long long _last = __builtin_readcyclecounter();\n" + anchor)
    # LDS accumulators (ds_add_u64, fire and forget) flushed once per workgroup: global atomics per segment would stall every
    # s_waitcnt vmcnt of the block loop behind them
    tail = "  if (errbits) atomicOr(a.err, (int)errbits);\n"
    assert s.count(tail) == 1
    s = s.replace(tail, "  __syncthreads();\n  if (tid < 128) atomicAdd(&g_clk[tid >> 4][tid & 15], s_clk[tid >> 4][tid & 15]);\n" + tail)
    return s


def build():
    from rl4co_amd import build as B

    os.makedirs(OUT, exist_ok=True)
    procs = []
    for kind in (sys.argv[2:] or VARIANTS):
        src = os.path.join(OUT, f"teacher_{kind}.hip")
        open(src, "w").write(variant_source(kind))
        extra = [os.path.join(SRC, n) for n in ("am_teacher.hip", "am_teacher_mma_f16.hip", "api.hip")]
        cmd = [B._hipcc(), *B.FLAGS, f"-I{B.INCLUDE}", f"-I{SRC}", "-o", os.path.join(OUT, f"libteacher_{kind}.so"), src, *extra]
        procs.append((kind, subprocess.Popen(cmd)))
    for kind, pr in procs:
        assert pr.wait() == 0, kind
        print("built", kind)


def run():
    import torch

    from rl4co_amd import _lib
    from rl4co_amd import teacher as T
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    handle = _lib.lib()
    name = "rl4co_am_teacher_backward"
    restype, argtypes = _lib.SYMBOLS[name]

    def swap(kind):
        lib = C.CDLL(os.path.join(OUT, f"libteacher_{kind}.so"))
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
        setattr(handle, name, fn)
        return lib

    torch.manual_seed(0)
    starts, batch = 8, 4096
    pol = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                               cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                               train_decode_type="multistart_sampling").cuda().train()
    env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda", check_solution=False)
    data = env.generator(batch_size=[batch])
    buf = (C.c_ulonglong * 128)()

    def step(i=[0]):
        i[0] += 1
        out = pol(env.reset(data), env, phase="train", seed=1000 * i[0], num_starts=starts)
        reward = out["reward"].view(starts, batch).t()
        ll = out["log_likelihood"].view(starts, batch).t()
        adv = reward - reward.mean(dim=1, keepdim=True)
        (-(adv.detach() * ll).mean()).backward()
        pol.zero_grad(set_to_none=True)

    timings = {}
    for kind in [k for k in VARIANTS if k != "clk" and os.path.exists(os.path.join(OUT, f"libteacher_{k}.so"))] * 2:
        swap(kind)
        for _ in range(2):
            step()
        T.backward_events = []
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        timings.setdefault(kind, []).append(round(sum(a.elapsed_time(b) for a, b in T.backward_events) / 5, 4))
    lib = swap("clk")
    dbg = lib.rl4co_debug_clocks
    dbg.restype, dbg.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
    for _ in range(2):
        step()
    assert dbg(buf, 1) == 0
    T.backward_events = []
    n = 3
    for _ in range(n):
        step()
    assert dbg(buf, 0) == 0
    ms = [a.elapsed_time(b) for a, b in T.backward_events]
    T.backward_events = None
    tab = [[buf[w * 16 + i] for i in range(16)] for w in range(8)]
    res = {"launch_ms_variants": timings, "launch_ms_with_probes": sum(ms) / len(ms), "launches": n, "segments": SEGMENTS, "cycles_per_wave": {}}
    for w in range(8):
        tot = sum(tab[w][:13])
        res["cycles_per_wave"][f"wave{w}"] = {"total_per_launch": tot / n,
                                              "share": {SEGMENTS[i]: round(tab[w][i] / tot, 4) for i in range(13)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
