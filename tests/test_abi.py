"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol ``include/rl4co_amd.h`` declares (no compute is launched without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "rl4co_amd.h").read_text()


def declared_functions():
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|int64_t)\s+(rl4co_\w+)\s*\(", HEADER, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ["rl4co_am_decode", "rl4co_tour_length_f32", "rl4co_tsp_step", "rl4co_cvrp_step",
                 "rl4co_tsp_check_solution", "rl4co_cvrp_check_solution", "rl4co_gather_by_index_f32"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from rl4co_amd import _lib

    handle = _lib.lib()
    names = declared_functions()
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    for name in names:
        assert getattr(handle, name) is not None
    assert b"gfx950" in handle.rl4co_version()
    # the binding, the header and the built library agree on the ABI version (a stale .so under unchanged names is refused)
    assert int(re.search(r"#define RL4CO_ABI_VERSION (\d+)", HEADER).group(1)) == _lib.ABI_VERSION == handle.rl4co_abi_version()


def _c_fields(struct_name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct_name, struct_name), HEADER, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return re.findall(r"(\w+)\s*;", body)


def test_struct_layout_matches_header():
    """Field order of the ctypes mirrors == field order in the C structs."""
    from rl4co_amd import _lib
    from rl4co_amd.encoder import AmEncoderArgs

    assert _c_fields("rl4co_am_decode_args") == [f[0] for f in _lib.AmDecodeArgs._fields_]
    assert ctypes.sizeof(_lib.AmDecodeArgs) % 8 == 0
    assert _c_fields("rl4co_am_encoder_args") == [f[0] for f in AmEncoderArgs._fields_]
    assert ctypes.sizeof(AmEncoderArgs) == 8 * 4 + 32 * 8
    from rl4co_amd.teacher import AmTeacherArgs

    assert _c_fields("rl4co_am_teacher_args") == [f[0] for f in AmTeacherArgs._fields_]
    assert _c_fields("rl4co_env_replay_args") == [f[0] for f in _lib.EnvReplayArgs._fields_]
    # (several fields per line in this one: compare the flattened declaration order)
    body = re.search(r"typedef struct rl4co_cross_attn_args \{(.*?)\} rl4co_cross_attn_args;", HEADER, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n for decl in body.split(";") for n in re.findall(r"(\w+)\s*(?:,|$)", decl.strip())]
    assert names == [f[0] for f in _lib.CrossAttnArgs._fields_]


def test_library_exports_nothing_the_header_does_not_declare():
    import shutil
    import subprocess

    from rl4co_amd import _lib

    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    _lib.lib()
    from rl4co_amd import build as B

    out = subprocess.run([nm, "-D", "--defined-only", str(B.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T rl4co_" in line)
    assert exported == declared_functions()


def test_weight_packing_is_the_documented_fragment_order():
    """pack_weight: [out/32][in/16][lane = 32*hi + row][s] = W[32*tile + row][16*kstep + 8*hi + s]."""
    import torch

    from rl4co_amd.encoder import pack_weight

    w = torch.arange(64 * 48, dtype=torch.float32).view(64, 48) / 8.0  # exactly representable in bf16? use small ints
    w = (torch.arange(64 * 48) % 251).float().view(64, 48)
    p = pack_weight(w).float()
    assert p.shape == (2, 3, 64, 8)
    for tile, ks, lane, s in [(0, 0, 0, 0), (1, 2, 63, 7), (0, 1, 37, 3), (1, 0, 31, 5), (1, 1, 32, 0)]:
        row, hi = lane & 31, lane >> 5
        assert p[tile, ks, lane, s] == w[32 * tile + row, 16 * ks + 8 * hi + s]


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (status + message, no exception)."""
    from rl4co_amd import _lib

    handle = _lib.lib()
    args = _lib.AmDecodeArgs()
    st = handle.rl4co_am_decode(ctypes.byref(args), None)
    assert st == 1  # RL4CO_ERR_ARG
    assert b"requirement failed" in handle.rl4co_last_error()
    assert handle.rl4co_am_decode_lds_bytes(100, 0) == 128 * 8 * 4 + 128 * 4 + 32 + 128 * 2 + 256
    with pytest.raises(_lib.Rl4coLibraryError):
        _lib.check(st, "rl4co_am_decode")


def test_error_bits_map_to_reference_messages():
    from rl4co_amd import _lib

    with pytest.raises(AssertionError, match="Logits contain NaNs"):
        _lib.raise_for_error_bits(_lib.EBIT_NAN_LOGIT | _lib.EBIT_INFEASIBLE)
    with pytest.raises(AssertionError, match="Used more than capacity"):
        _lib.raise_for_error_bits(_lib.EBIT_CAPACITY)
    _lib.raise_for_error_bits(0)


def test_header_compiles_as_c99_and_layouts_match_ctypes(tmp_path):
    """The boundary is a C header: it must compile with a plain C compiler (``gcc -std=c99 -pedantic -Werror``, no HIP,
    no C++), and the ctypes mirrors the Python host marshals through must agree with the COMPILER's layout — sizeof and
    the offset of every field of every argument struct, not just the field order."""
    import shutil
    import subprocess

    from rl4co_amd import _lib
    from rl4co_amd.encoder import AmEncoderArgs
    from rl4co_amd.teacher import AmTeacherArgs

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    structs = {"rl4co_am_decode_args": _lib.AmDecodeArgs, "rl4co_am_encoder_args": AmEncoderArgs,
               "rl4co_am_teacher_args": AmTeacherArgs, "rl4co_am_train_save": _lib.AmTrainSave}
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "rl4co_amd.h"', "int main(void) {"]
    for cname, mirror in structs.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for field, _ in mirror._fields_:
            lines.append(f'  printf("{cname} {field} %zu\\n", offsetof({cname}, {field}));')
    # every declared function is a plain C symbol whose address can be taken from C
    lines.append("  typedef void (*any_fn)(void);")
    lines.append("  any_fn symbols[] = {" + ", ".join(f"(any_fn){name}" for name in declared_functions()) + "};")
    lines.append("  if (sizeof(symbols) == 0) return 2;")
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "abi_probe"
    lib_dir = ROOT / "rl4co_amd" / "lib"
    _lib.lib()  # make sure the library is built
    cmd = [gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
           f"-L{lib_dir}", "-lrl4co_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,--unresolved-symbols=ignore-in-shared-libs"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    # (the probe is compiled and LINKED against the product library; running it would load the HIP runtime, which the
    # CPU container cannot initialise, so the layout table is printed by a second, library-free build)
    cmd2 = [gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)]
    src.write_text("\n".join(l for l in lines if "any_fn" not in l and "symbols" not in l) + "\n")
    proc = subprocess.run(cmd2, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    table = {}
    for row in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        cname, field, value = row.split()
        table[(cname, field)] = int(value)
    for cname, mirror in structs.items():
        assert table[(cname, "sizeof")] == ctypes.sizeof(mirror), cname
        for field, _ in mirror._fields_:
            assert table[(cname, field)] == getattr(mirror, field).offset, (cname, field)
