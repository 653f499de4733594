"""In-tree build of the HIP C-ABI library (``rl4co_amd/lib/librl4co_amd.so``).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only
build container as well as on the MI355X box. The build is keyed by a content hash of
the sources + flags (mtimes do not survive the snapshot copy to the GPU box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "librl4co_amd.so"
HASH_PATH = LIB_DIR / "librl4co_amd.so.hash"

SOURCES = ["api.hip", "env_step.hip", "tour_length.hip", "am_decode.hip", "am_decode_ms.hip", "am_encoder.hip", "am_teacher.hip", "am_teacher_mma.hip", "am_train_ops.hip", "am_train_attn.hip", "am_attn_flash.hip", "augment.hip",
           "am_train_ops_f16.hip", "am_train_attn_f16.hip", "am_attn_flash_f16.hip",
           "am_decode_ms_f16.hip", "am_teacher_mma_f16.hip"]
HEADERS = ["common.h", "rl4co_math.h", "elem16.h"]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # arithmetic order is part of the parity contract
    "-fPIC",
    "-shared",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librl4co_amd.so")
    return exe


def source_hash() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    h.update((INCLUDE / "rl4co_amd.h").read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    return LIB_PATH.exists() and HASH_PATH.exists() and HASH_PATH.read_text().strip() == source_hash()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source into one shared library; no-op when up to date."""
    if not force and is_fresh():
        return LIB_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    tmp = LIB_DIR / f"librl4co_amd.so.tmp.{os.getpid()}"  # several ranks may build at once: atomic replace
    cmd = [_hipcc(), *FLAGS, f"-I{INCLUDE}", "-o", str(tmp)] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, LIB_PATH)
    HASH_PATH.write_text(source_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
