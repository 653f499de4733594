"""Plain ATen streams for reference: fill (write only), copy (1 read : 1 write), add (2 reads : 1 write), at the sizes of
the training activations (409 600 rows x 128 .. 512 bf16 columns)."""
import torch
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 409600
for cols in (128, 384, 512):
    x = torch.empty(M, cols, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(M, cols, device="cuda").to(torch.bfloat16)
    mb = M * cols * 2 / 1e6
    t = bench(lambda: x.fill_(1.0)); print(f"fill  {mb:5.0f} MB: {t:6.1f} us  {mb / t:5.2f} TB/s written")
    t = bench(lambda: x.copy_(y)); print(f"copy  {mb:5.0f} MB: {t:6.1f} us  {2 * mb / t:5.2f} TB/s moved")
    t = bench(lambda: torch.add(x, y, out=x)); print(f"add   {mb:5.0f} MB: {t:6.1f} us  {3 * mb / t:5.2f} TB/s moved")
