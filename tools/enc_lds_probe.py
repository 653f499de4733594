#!/usr/bin/env python
"""Is the fused encoder bound by LDS operand bandwidth? (semantics-breaking timing probe, r03)

Every 32x32x16 MFMA of the kernel's GEMMs takes one 1 KiB activation fragment from LDS (gemm_t). This probe builds
variants of csrc/am_encoder.hip in which the odd token tiles (xhalf) or all but the first (xquarter) REUSE an earlier
tile's fragment through an opaque register copy — same MFMAs, same accumulators, half / a quarter of the LDS reads, wrong
numbers — and times them against the unmodified source. The variants are built on the host (no GPU needed) into
tools/probes/bin/ (git-ignored; travels with the gpurun snapshot) as small libraries holding only the encoder entry point;
the timing run swaps that entry point into the loaded product library.

    python tools/enc_lds_probe.py build          # build container
    gpurun -- python tools/enc_lds_probe.py run  # MI355X
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "rl4co_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "bin")
VARIANTS = tuple(v for v in ("base", "xhalf", "xquarter") if os.path.exists(os.path.join(OUT, f"libenc_{v}.so")) or len(sys.argv) > 1 and sys.argv[1] == "build")

LOAD = """  vec8<E> x[3][TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    x[0][tt] = load_x(xs, tt, 0, l31, hi);
    x[1][tt] = load_x(xs, tt, 1, l31, hi);
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks + 2 < 8) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) x[(ks + 2) % 3][tt] = load_x(xs, tt, ks + 2, l31, hi);
    }
"""


def variant_source(name: str) -> str:
    s = open(os.path.join(SRC, "am_encoder.hip")).read()
    assert LOAD in s, "gemm_t changed: update the probe"
    if name == "base":
        return s
    group = 2 if name == "xhalf" else 4
    def ld(slot, ks):
        return (f"      if (tt % {group} == 0) {{ x[{slot}][tt] = load_x(xs, tt, {ks}, l31, hi); }}\n"
                f"      else {{ x[{slot}][tt] = x[{slot}][tt - tt % {group}]; asm volatile(\"\" : \"+v\"(x[{slot}][tt])); }}\n")
    new = ("  vec8<E> x[3][TT];\n#pragma unroll\n  for (int tt = 0; tt < TT; ++tt) {\n" + ld("0", "0") + ld("1", "1") + "  }\n"
           "#pragma unroll\n  for (int ks = 0; ks < 8; ++ks) {\n    if (ks + 2 < 8) {\n#pragma unroll\n      for (int tt = 0; tt < TT; ++tt) {\n"
           + ld("(ks + 2) % 3", "ks + 2") + "      }\n    }\n")
    return s.replace(LOAD, new)


def build():
    from rl4co_amd import build as B

    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        src = os.path.join(OUT, f"enc_{v}.hip")
        open(src, "w").write(variant_source(v).replace('#include "common.h"', f'#include "{SRC}/common.h"'))
        cmd = [B._hipcc(), *B.FLAGS, f"-I{B.INCLUDE}", f"-I{SRC}", "-o", os.path.join(OUT, f"libenc_{v}.so"), src, os.path.join(SRC, "api.hip")]
        subprocess.run(cmd, check=True)
        print("built", v)


def run():
    import torch

    from rl4co_amd import _lib
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    handle = _lib.lib()
    restype, argtypes = _lib.SYMBOLS["rl4co_am_encoder"]
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
    td = env.reset(batch_size=[4096])
    pe = pol._packed_encoder()
    for v in VARIANTS + ("base",):
        lib = C.CDLL(os.path.join(OUT, f"libenc_{v}.so"))
        fn = lib.rl4co_am_encoder
        fn.restype, fn.argtypes = restype, argtypes
        handle.rl4co_am_encoder = fn
        with torch.inference_mode():
            for _ in range(3):
                pe.encode(td, torch.bfloat16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                pe.encode(td, torch.bfloat16)
            e1.record()
            torch.cuda.synchronize()
        print(f"{v:9s} {e0.elapsed_time(e1) / 20:.3f} ms per launch (TSP-100 x 4096)", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
