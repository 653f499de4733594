import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import GoldenCase, fold_cache, max_horizon, rollout_state
from tests.test_gpu_decode_ms import _rollout, _c_ms_rollout
from rl4co_amd import kernels as K
name, starts = sys.argv[1], int(sys.argv[2])
g = GoldenCase(name); td0 = g.reset()
with torch.inference_mode():
    h, _ = g.policy.encoder(td0)
cache = fold_cache(g.policy, g.env_name, h, torch.bfloat16, device="cuda")
a_c, l_c = _c_ms_rollout(g, td0, cache.to("cpu"), starts, "greedy")
a_ms, l_ms, st, err, _ = _rollout(K, g, td0, cache, starts, "ms")
print("free run err", err, "identical", float((a_ms.cpu()[:, :a_c.shape[1]] == a_c[:, :a_ms.shape[1]]).all(1).float().mean()), a_ms.shape, a_c.shape)
a_ev, l_ev, st2, err2, _ = _rollout(K, g, td0, cache, starts, "ms", mode="evaluate", forced=a_c.cuda())
print("evaluate err", err2)
bad = (~torch.isfinite(l_ev.cpu())) | (l_ev.cpu() < -1000)
rows = bad.any(1).nonzero().flatten().tolist()
print("bad rows", rows[:10], "of", a_c.shape[0])
for r in rows[:3]:
    print("row", r, "inst", r % g.batch, "start", r // g.batch)
    print(" oracle  acts", a_c[r].tolist()); print(" kernel  acts", a_ms.cpu()[r].tolist())
    print(" oracle logp", [round(x, 3) for x in l_c[r].tolist()]); print(" eval   logp", [round(x, 3) for x in l_ev.cpu()[r].tolist()])
