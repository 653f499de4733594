"""In-tree build of the HIP C-ABI library (``rl4co_amd/lib/librl4co_amd.so``).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only
build container as well as on the MI355X box. Every source is compiled to its own object
(in parallel, each keyed by a content hash of the source, the headers and the flags: an
unchanged kernel is not recompiled) and the objects are linked into one shared library;
the library carries the hash of the whole source set (mtimes do not survive the snapshot
copy to the GPU box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_DIR = PKG_DIR / "lib"
OBJ_DIR = LIB_DIR / "obj"
LIB_PATH = LIB_DIR / "librl4co_amd.so"
HASH_PATH = LIB_DIR / "librl4co_amd.so.hash"

SOURCES = ["api.hip", "entry16.hip", "env_step.hip", "tour_length.hip", "am_decode.hip", "am_decode_ms.hip", "am_encoder.hip", "am_encoder_f32.hip", "am_tokens_f32.hip", "am_teacher.hip", "am_teacher_mma.hip", "am_train_ops.hip", "am_train_attn.hip", "am_attn_flash.hip", "am_cross_attn.hip", "am_logit_logp.hip", "augment.hip",
           "am_train_ops_f16.hip", "am_train_attn_f16.hip", "am_attn_flash_f16.hip", "am_cross_attn_f16.hip",
           "am_decode_ms_f16.hip", "am_teacher_mma_f16.hip"]
HEADERS = ["common.h", "rl4co_math.h", "elem16.h", "enc_f32.h"]
# *_f16.hip wrappers include their bf16 namesake: its text is part of their hash
INCLUDED = {name: [name.replace("_f16.hip", ".hip")] for name in SOURCES if name.endswith("_f16.hip")}
COMPILE_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # arithmetic order is part of the parity contract
    "-fPIC",
]
FLAGS = COMPILE_FLAGS + ["-shared"]  # (tools/ build probe variants with the same flags)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librl4co_amd.so")
    return exe


def _common_hash() -> "hashlib._Hash":
    h = hashlib.sha256()
    for name in HEADERS:
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    h.update((INCLUDE / "rl4co_amd.h").read_bytes())
    h.update(" ".join(COMPILE_FLAGS).encode())
    return h


def _object_hash(name: str) -> str:
    h = _common_hash()
    for part in [name] + INCLUDED.get(name, []):
        h.update(part.encode())
        h.update((CSRC / part).read_bytes())
    return h.hexdigest()


def source_hash() -> str:
    h = hashlib.sha256()
    for name in SOURCES:
        h.update(_object_hash(name).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    return LIB_PATH.exists() and HASH_PATH.exists() and HASH_PATH.read_text().strip() == source_hash()


def _compile(name: str, verbose: bool) -> Path:
    obj = OBJ_DIR / (name + ".o")
    stamp = OBJ_DIR / (name + ".o.hash")
    want = _object_hash(name)
    if obj.exists() and stamp.exists() and stamp.read_text().strip() == want:
        return obj
    tmp = OBJ_DIR / f"{name}.o.tmp.{os.getpid()}"
    cmd = [_hipcc(), *COMPILE_FLAGS, f"-I{INCLUDE}", "-c", str(CSRC / name), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {name} ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, obj)
    stamp.write_text(want + "\n")
    return obj


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source (in parallel) and link one shared library; no-op when up to date."""
    if not force and is_fresh():
        return LIB_PATH
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for f in OBJ_DIR.glob("*.o.hash"):
            f.unlink()
    jobs = max(1, min(len(SOURCES), len(os.sched_getaffinity(0)), int(os.environ.get("RL4CO_BUILD_JOBS", "8"))))
    # heaviest translation units first
    order = sorted(SOURCES, key=lambda n: -sum((CSRC / p).stat().st_size for p in [n] + INCLUDED.get(n, [])))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = dict(zip(order, pool.map(lambda n: _compile(n, verbose), order)))
    tmp = LIB_DIR / f"librl4co_amd.so.tmp.{os.getpid()}"  # several ranks may build at once: atomic replace
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp)] + [str(objs[n]) for n in SOURCES]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, LIB_PATH)
    HASH_PATH.write_text(source_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
