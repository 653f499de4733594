"""GPU box: is the fused encoder bound by the chip's POWER budget? Same kernel, same instruction stream, three data sets:
the normal random-init policy, all-zero weights and coordinates (no operand toggling: the matrix cores and the LDS draw far
less), and weights scaled down 1e-3 (tiny but non-zero mantissa activity). A kernel bound by issue, latency or bandwidth takes
the same time on all three; one clocked down by the power manager runs faster on the quiet data
(/opt/skills/guides/MI355X_MICROARCH.md, DVFS give-back: zero-filled inputs +19 % on a bf16 attention kernel)."""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env

DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16  # python tools/enc_power_probe.py [f32]
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[4096])


def bench(pol, td, iters=200):
    pe = pol._packed_encoder()
    with torch.inference_mode():
        for _ in range(5):
            pe.encode(td, DT, act_dtype=DT)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            pe.encode(td, DT, act_dtype=DT)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l))
    except Exception as exc:  # noqa: BLE001
        return f"rocm-smi: {exc}"


for name in ("random", "zeros", "tiny", "random"):
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", cache_dtype=DT, encoder_autocast=None if DT == torch.float32 else DT).cuda().eval()
    t = td
    if name == "zeros":
        with torch.no_grad():
            for p in pol.parameters():
                p.zero_()
        t = td.clone() if hasattr(td, "clone") else td
        t["locs"] = torch.zeros_like(td["locs"])
    elif name == "tiny":
        with torch.no_grad():
            for p in pol.parameters():
                p.mul_(1e-3)
    got = {}
    th = threading.Thread(target=lambda: got.update(smi=(time.sleep(0.25), smi())[1]))
    th.start()
    ms = bench(pol, t, iters=600 if DT != torch.float32 else 100)
    th.join()
    print(f"{name:7s} {ms:.3f} ms   [{got.get('smi')}]")
