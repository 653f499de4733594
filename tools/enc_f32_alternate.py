"""GPU box: why is the fp32 fused encoder 9 - 11 % slower inside the rollout (6.1 - 6.2 ms) than alone (5.56 ms)? The same
launch timed (HIP events) alone, alternating with a 5 ms HBM-streaming kernel (what the fp32 decode launch is to the chip),
and alternating with an idle gap of the same length; rocm-smi sampled meanwhile."""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
from rl4co_amd import kernels as K

torch.manual_seed(0)
pol = AttentionModelPolicy("tsp", cache_dtype=torch.float32, encoder_autocast=None).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[4096])
pe = pol._packed_encoder()
buf = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
sink = torch.zeros(1, device="cuda")


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    return " ".join(l.split(":")[-1].strip() for l in out.splitlines() if ("sclk" in l or "Power (W)" in l))


def run(mode, iters=40):
    evs = []
    with torch.inference_mode():
        for i in range(iters + 5):
            if mode == "stream":
                for _ in range(15):  # ~5 ms of HBM reads
                    K.hbm_read_probe(buf, sink)
            elif mode == "gap":
                torch.cuda.synchronize()
                time.sleep(0.005)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pe.encode(td, torch.float32, act_dtype=torch.float32)
            e1.record()
            if i >= 5:
                evs.append((e0, e1))
        torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], ms[0]


for mode in ("alone", "stream", "gap", "alone"):
    got = {}
    th = threading.Thread(target=lambda: got.update(s=(time.sleep(0.15), smi())[1]))
    th.start()
    med, best = run(mode)
    th.join()
    print(f"{mode:7s} encoder launch median {med:.3f} ms (min {best:.3f})   [{got.get('s')}]")
