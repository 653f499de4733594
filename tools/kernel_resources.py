#!/usr/bin/env python
"""Registers / LDS / occupancy of every kernel of one HIP source, from the compiler's own remarks (no GPU needed).

    python tools/kernel_resources.py am_decode.hip [filter]

Runs ``hipcc -c -Rpass-analysis=kernel-resource-usage`` with the library's flags and prints one line per kernel."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from rl4co_amd import build as B  # noqa: E402

src = B.CSRC / sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
cmd = [B._hipcc(), *flags, f"-I{B.INCLUDE}", "-c", str(src), "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    text = m.group(1)
    if text.startswith("Function Name:"):
        cur = {"name": text.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in text:
        k, v = text.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':90s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r, name in zip(rows, demangle):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name[:90]:90s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', r.get('SGPRs', '?')):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
