"""GPU box: where the HOST time of one rollout step goes (cProfile over 300 steps of the c2_greedy loop)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl4co_amd.envs import get_env
from rl4co_amd.policy import AttentionModelPolicy

env_name, num_loc, batch = (sys.argv[1:] + ["tsp", 100, 4096])[:3]
num_loc, batch = int(num_loc), int(batch)
torch.manual_seed(0)
pol = AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
data = env.generator(batch_size=[batch])
def step():
    return pol(env.reset(data), env, phase="test", decode_type="greedy")
with torch.inference_mode():
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) * 10:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): step()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
