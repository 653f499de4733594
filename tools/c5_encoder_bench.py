"""GPU box: the 16-bit token-tile encoder alone at BASELINE configs[4] (CVRP-500 x 1024) and its attention launch
(csrc/am_attn_flash.hip) on both softmax paths, HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl4co_amd.policy import AttentionModelPolicy
from rl4co_amd.envs import get_env
from rl4co_amd import _lib
torch.manual_seed(0)
DT = torch.bfloat16
pol = AttentionModelPolicy("cvrp", cache_dtype=DT, encoder_autocast=DT).cuda().eval()
env = get_env("cvrp", generator_params=dict(num_loc=500, device="cuda"), device="cuda")
td = env.reset(batch_size=[1024])
pe = pol._packed_encoder()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.inference_mode():
    enc = timeit(lambda: pe.encode(td, DT, act_dtype=DT))
    B, N = 1024, 501
    qkv = (torch.randn(B, N, 384, device="cuda") * 0.5).to(DT)
    out = torch.empty(B, N, 128, device="cuda", dtype=DT)
    st = torch.cuda.current_stream().cuda_stream
    bound = torch.full((B, 8, 2), 4.0, device="cuda")  # |q|^2 |k|^2 = 16 <= 48^2: the max-free path
    fast = timeit(lambda: _lib.check(_lib.lib().rl4co_attn_flash_pre(1, qkv.data_ptr(), bound.data_ptr(), B, N, out.data_ptr(), st), "flash"))
    big = torch.full((B, 8, 2), 1e4, device="cuda")
    exact = timeit(lambda: _lib.check(_lib.lib().rl4co_attn_flash_pre(1, qkv.data_ptr(), big.data_ptr(), B, N, out.data_ptr(), st), "flash"))
print(f"C5 encoder {enc:.3f} ms   attention launch: max-free path {fast * 1e3:.0f} us, exact path {exact * 1e3:.0f} us")
