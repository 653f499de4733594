"""rl4co_attn_flash_bf16 at the C5 shape (1024 instances x 501 nodes): time per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl4co_amd import train_ops as T
torch.manual_seed(0)
qkv = torch.randn(1024, 501, 384, device="cuda").to(torch.bfloat16)
for _ in range(3): T.attention_flash(qkv)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): T.attention_flash(qkv)
e1.record(); torch.cuda.synchronize()
print(f"attn_flash 1024 x 501: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
