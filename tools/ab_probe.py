#!/usr/bin/env python
"""Same-process A/B of one kernel source: the committed version (git HEAD) against the working tree, timed alternately
through the entry point they share — box-to-box variation (±3 %) is larger than most single changes.

    python tools/ab_probe.py build am_decode_ms.hip             # build container: two small libraries in tools/probes/bin/
    gpurun -- python tools/ab_probe.py run train               # MI355X: training-step timing (rollout + backward events)
    gpurun -- python tools/ab_probe.py run decode              # MI355X: the streaming greedy decode launch alone
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "rl4co_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "bin")
# entry point -> the sources its library needs
GROUPS = {
    "rl4co_am_decode": ["am_decode.hip", "am_decode_ms.hip", "am_decode_ms_f16.hip", "api.hip"],
    "rl4co_am_teacher_backward": ["am_teacher.hip", "am_teacher_mma.hip", "am_teacher_mma_f16.hip", "api.hip"],
}


def build(changed: str):
    from rl4co_amd import build as B

    entry = next(e for e, srcs in GROUPS.items() if changed in srcs)
    os.makedirs(OUT, exist_ok=True)
    old = subprocess.run(["git", "-C", ROOT, "show", f"HEAD:rl4co_amd/csrc/{changed}"], capture_output=True, text=True, check=True).stdout
    procs = []
    for tag, text in (("head", old), ("tree", open(os.path.join(SRC, changed)).read())):
        d = os.path.join(OUT, f"ab_{tag}")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, changed), "w").write(text)
        srcs = [os.path.join(d, changed) if s == changed else os.path.join(SRC, s) for s in GROUPS[entry]]
        # wrappers that #include the changed file by name pick up the copy next to them first
        srcs = [s if not (s.endswith("_f16.hip") and changed in open(s).read()) else _copy(s, d) for s in srcs]
        cmd = [B._hipcc(), *B.FLAGS, f"-I{B.INCLUDE}", f"-I{d}", f"-I{SRC}", "-o", os.path.join(OUT, f"libab_{tag}.so"), *srcs]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0
    open(os.path.join(OUT, "ab_entry.txt"), "w").write(entry)
    print("built", entry)


def _copy(path, d):
    dst = os.path.join(d, os.path.basename(path))
    open(dst, "w").write(open(path).read())
    return dst


def run():
    import torch

    from rl4co_amd import _lib
    from rl4co_amd import teacher as T
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    entry = open(os.path.join(OUT, "ab_entry.txt")).read().strip()
    handle = _lib.lib()
    restype, argtypes = _lib.SYMBOLS[entry]
    fns = {}
    for tag in ("head", "tree"):
        fn = getattr(C.CDLL(os.path.join(OUT, f"libab_{tag}.so")), entry)
        fn.restype, fn.argtypes = restype, argtypes
        fns[tag] = fn
    torch.manual_seed(0)
    starts, batch = 8, 4096
    pol = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                               cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                               train_decode_type="multistart_sampling").cuda().train()
    env = get_env("tsp", generator_params=dict(num_loc=100, device="cuda"), device="cuda", check_solution=False)
    data = env.generator(batch_size=[batch])

    def step(i=[0]):
        i[0] += 1
        out = pol(env.reset(data), env, phase="train", seed=1000 * i[0], num_starts=starts)
        reward = out["reward"].view(starts, batch).t()
        ll = out["log_likelihood"].view(starts, batch).t()
        (-((reward - reward.mean(dim=1, keepdim=True)).detach() * ll).mean()).backward()
        pol.zero_grad(set_to_none=True)

    res = {"head": [], "tree": []}
    for rnd in range(4):
        for tag in ("head", "tree"):
            setattr(handle, entry, fns[tag])
            for _ in range(2):
                step()
            pol.decode_events, T.backward_events = [], []
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            ev = pol.decode_events if entry == "rl4co_am_decode" else T.backward_events
            res[tag].append(round(sum(a.elapsed_time(b) for a, b in ev) / len(ev), 4))
            pol.decode_events, T.backward_events = None, None
    print(entry, res)


def run_decode():
    """greedy decode launch alone (TSP-100 x 4096 and CVRP-100 x 4096, bf16 planes, streaming kernel), head vs tree"""
    import torch

    from rl4co_amd import _lib
    from rl4co_amd import kernels as K
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    entry = open(os.path.join(OUT, "ab_entry.txt")).read().strip()
    assert entry == "rl4co_am_decode"
    handle = _lib.lib()
    restype, argtypes = _lib.SYMBOLS[entry]
    fns = {}
    for tag in ("head", "tree"):
        fn = getattr(C.CDLL(os.path.join(OUT, f"libab_{tag}.so")), entry)
        fn.restype, fn.argtypes = restype, argtypes
        fns[tag] = fn
    for env_name, tmax in (("tsp", 100), ("cvrp", 200)):
        torch.manual_seed(0)
        pol = AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
        env = get_env(env_name, generator_params=dict(num_loc=100, device="cuda"), device="cuda")
        td = env.reset(batch_size=[4096])
        res, acts = {"head": [], "tree": []}, {}
        with torch.inference_mode():
            cache, _ = pol._packed_encoder().encode(td, torch.bfloat16)
            for rnd in range(4):
                for tag in ("head", "tree"):
                    setattr(handle, entry, fns[tag])
                    times = []
                    for it in range(5):
                        st = pol._initial_state(td, 0)
                        actions = torch.zeros(4096, tmax, dtype=torch.int64, device="cuda")
                        logps = torch.zeros(4096, tmax, device="cuda")
                        err = K.new_error_word("cuda")
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        K.am_decode(cache, st, mode="greedy", max_steps=tmax, actions=actions, logps=logps, err=err, variant="stream")
                        e1.record()
                        torch.cuda.synchronize()
                        times.append(e0.elapsed_time(e1))
                    res[tag].append(round(min(times[1:]), 4))
                    acts[tag] = (actions.clone(), logps.clone())
        same = torch.equal(acts["head"][0], acts["tree"][0]) and torch.equal(acts["head"][1], acts["tree"][1])
        print(env_name, res, "outputs bit-identical:", same)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[2] == "decode":
        run_decode()
    else:
        run()
