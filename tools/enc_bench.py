"""Micro-benchmark of the fused encoder kernel alone (TSP-100 x 4096), HIP events."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
torch.manual_seed(0)
DT = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16  # python tools/enc_bench.py [f16]
pol = AttentionModelPolicy("tsp", cache_dtype=DT, encoder_autocast=DT).cuda().eval()
env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
td = env.reset(batch_size=[4096])
pe = pol._packed_encoder()
with torch.inference_mode():
    for _ in range(3): pe.encode(td, DT, act_dtype=DT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pe.encode(td, DT, act_dtype=DT)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
flops = 4096 * (3 * (128*128*384*2 + 2*8*128*128*16*2*1.5 + 128*128*128*2 + 2*128*128*512*2) + 5*128*128*128*2)
print(f"encoder [{DT}] {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s (padded-128 MFMA flops incl. 50% PV waste)")


