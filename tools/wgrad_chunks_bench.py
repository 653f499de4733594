import sys, torch, os
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/rl4co_amd") else os.environ["GRAFT_REPO_ROOT"])
from rl4co_amd import train_ops as T
M = 409600
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for wg in [int(x) for x in (sys.argv[1:] or ["1024", "512", "256", "2048"])]:
    T._WGRAD_MAX_WORKGROUPS = wg
    out = []
    for n, k in ((384, 128), (128, 128), (512, 128), (128, 512)):
        d = torch.randn(M, n, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, k, device="cuda").to(torch.bfloat16)
        out.append(bench(lambda: T._wgrad(d, x, with_bias=True)))
    print(wg, " ".join(f"{u:7.1f}" for u in out), f"sum {sum(out):.1f} us")
