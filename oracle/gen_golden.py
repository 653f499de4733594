"""ORACLE (test infrastructure): pin the restatement against the REAL reference and emit goldens.

Run in the build container (needs ``/root/reference``; it does not exist on the GPU box):

    python -m oracle.gen_golden            # writes tests/golden/*.npz + tests/golden/MANIFEST.json

For every case below the reference's own source files (imported verbatim through
``oracle/ref_import.py``) run the rollout on CPU fp32; the same seeded inputs/weights are then
run through the restatement ``oracle/reference_torch.py`` and the two must agree BIT FOR BIT
(actions, rewards, log-likelihoods, and the seeded weights themselves) — otherwise this script
fails and nothing is written. The emitted fixtures hold only small arrays: the instance data,
the reference's actions / reward / log-likelihood, and a SHA-256 of the policy weights (which the
tests re-create from the seed and verify against the hash).
"""
from __future__ import annotations

import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import ref_import  # noqa: E402
from oracle import reference_torch as R  # noqa: E402

GOLDEN_DIR = ROOT / "tests" / "golden"

WEIGHT_SEED = 0
DATA_SEED = 1234
SAMPLE_SEED = 4321

# name, env, num_loc, batch, policy kind, decode_type, extra policy kwargs, extra forward kwargs
CASES = [
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    dict(name="c1_tsp20_b256_greedy", env="tsp", num_loc=20, batch=256, policy="am", decode="greedy"),
    dict(name="tsp20_b64_greedy_simple", env="tsp", num_loc=20, batch=64, policy="am", decode="greedy",
         pol_kw=dict(sdpa_fn_decoder="simple")),
    dict(name="tsp50_b64_greedy", env="tsp", num_loc=50, batch=64, policy="am", decode="greedy"),
    dict(name="tsp100_b64_greedy", env="tsp", num_loc=100, batch=64, policy="am", decode="greedy"),
    dict(name="tsp100_b64_sampling", env="tsp", num_loc=100, batch=64, policy="am", decode="sampling"),
    dict(name="cvrp20_b128_greedy", env="cvrp", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="cvrp100_b64_greedy", env="cvrp", num_loc=100, batch=64, policy="am", decode="greedy"),
    dict(name="cvrp100_b64_sampling", env="cvrp", num_loc=100, batch=64, policy="am", decode="sampling"),
    dict(name="pomo_tsp20_b16_msgreedy", env="tsp", num_loc=20, batch=16, policy="pomo",
         decode="multistart_greedy"),
    dict(name="pomo_tsp50_b8_mssampling", env="tsp", num_loc=50, batch=8, policy="pomo",
         decode="multistart_sampling", fw_kw=dict(num_starts=8)),
    dict(name="pomo_cvrp20_b16_msgreedy", env="cvrp", num_loc=20, batch=16, policy="pomo",
         decode="multistart_greedy"),
    # the orienteering problem shares the decode kernel (current-node + scalar context)
    dict(name="op20_b128_greedy", env="op", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="op50_b64_sampling", env="op", num_loc=50, batch=64, policy="am", decode="sampling"),
    dict(name="op100_b64_greedy", env="op", num_loc=100, batch=64, policy="am", decode="greedy"),
    dict(name="pctsp20_b128_greedy", env="pctsp", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="pctsp50_b64_sampling", env="pctsp", num_loc=50, batch=64, policy="am", decode="sampling"),
    dict(name="pctsp100_b64_greedy", env="pctsp", num_loc=100, batch=64, policy="am", decode="greedy"),
    # stochastic PCTSP: the init embedding sees the EXPECTED prize, the transition collects the REAL one
    dict(name="spctsp20_b128_greedy", env="spctsp", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="spctsp50_b64_sampling", env="spctsp", num_loc=50, batch=64, policy="am", decode="sampling"),
    # pickup and delivery (same row): precedence masks; POMO multistart from the pickups
    dict(name="pdp20_b128_greedy", env="pdp", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="pdp50_b64_sampling", env="pdp", num_loc=50, batch=64, policy="am", decode="sampling"),
    dict(name="pdp100_b64_greedy", env="pdp", num_loc=100, batch=64, policy="am", decode="greedy"),
    # CVRP with time windows (same row): distance/time masks, two context scalars, six init features
    dict(name="cvrptw20_b128_greedy", env="cvrptw", num_loc=20, batch=128, policy="am", decode="greedy"),
    dict(name="cvrptw50_b64_sampling", env="cvrptw", num_loc=50, batch=64, policy="am", decode="sampling"),
    dict(name="cvrptw100_b64_greedy", env="cvrptw", num_loc=100, batch=64, policy="am", decode="greedy"),
    # POMO multistart on the other depot environments (start node s % num_loc + 1, batchified state)
    dict(name="pomo_op20_b16_msgreedy", env="op", num_loc=20, batch=16, policy="pomo", decode="multistart_greedy"),
    dict(name="pomo_pctsp20_b16_msgreedy", env="pctsp", num_loc=20, batch=16, policy="pomo", decode="multistart_greedy"),
    dict(name="pomo_cvrptw20_b16_mssampling", env="cvrptw", num_loc=20, batch=16, policy="pomo", decode="multistart_sampling"),
    dict(name="pomo_pdp20_b16_msgreedy", env="pdp", num_loc=20, batch=16, policy="pomo", decode="multistart_greedy"),
    # non-default decoding arguments of ConstructivePolicy.forward (constructive/base.py:154-263, decoding.py:282-461):
    # best-of-starts selection, temperature / clipping, multisample. Policy-level fixtures: the decode-level tests
    # (kernel vs C oracle) drive the kernels directly and cover these arguments through their own parameters.
    dict(name="kw_tsp50_b32_ms_selectbest", env="tsp", num_loc=50, batch=32, policy="pomo", decode="multistart_greedy",
         fw_kw=dict(select_best=True), policy_only=True),
    dict(name="kw_tsp50_b32_sampling_temp", env="tsp", num_loc=50, batch=32, policy="am", decode="sampling",
         fw_kw=dict(temperature=0.7, tanh_clipping=8.0), policy_only=True),
    dict(name="kw_cvrp20_b32_multisample", env="cvrp", num_loc=20, batch=32, policy="am", decode="sampling",
         fw_kw=dict(num_samples=4), policy_only=True),
    dict(name="kw_tsp20_b32_sampling_entropy", env="tsp", num_loc=20, batch=32, policy="am", decode="sampling",
         fw_kw=dict(return_entropy=True), policy_only=True),
    dict(name="kw_cvrp20_b32_greedy_entropy", env="cvrp", num_loc=20, batch=32, policy="am", decode="greedy",
         fw_kw=dict(return_entropy=True), policy_only=True),
    dict(name="kw_cvrp20_b32_multisample_best", env="cvrp", num_loc=20, batch=32, policy="am", decode="sampling",
         fw_kw=dict(num_samples=4, select_best=True), policy_only=True),
    # BASELINE.json configs[3] / [4] shapes at a CPU-affordable batch: POMO 8-start sampling on
    # TSP-100, and CVRP-500 sampling (N = 501: the n >= 512 cascade of the tour-length sum)
    dict(name="c4_pomo_tsp100_b32_s8_sampling", env="tsp", num_loc=100, batch=32, policy="pomo",
         decode="multistart_sampling", fw_kw=dict(num_starts=8)),
    dict(name="c5_cvrp500_b16_sampling", env="cvrp", num_loc=500, batch=16, policy="am", decode="sampling",
         store_inputs=False),
    # BASELINE.json configs[1] / [2] at full size (inputs are re-created from the seed; only the
    # reference's actions (uint8) / rewards are stored)
    dict(name="c2_tsp100_b4096_greedy", env="tsp", num_loc=100, batch=4096, policy="am", decode="greedy",
         store_inputs=False),
    dict(name="c3_cvrp100_b1024_greedy", env="cvrp", num_loc=100, batch=1024, policy="am", decode="greedy",
         store_inputs=False),
    # round 2: configs[2] at its full batch, configs[4] at a batch that fills the WIDE decode variant's launch
    # (B <= 2048 picks it), and configs[1]'s sampling leg at full size
    dict(name="c3_cvrp100_b4096_greedy", env="cvrp", num_loc=100, batch=4096, policy="am", decode="greedy",
         store_inputs=False),
    dict(name="c5_cvrp500_b256_sampling", env="cvrp", num_loc=500, batch=256, policy="am", decode="sampling",
         store_inputs=False),
    dict(name="c2_tsp100_b4096_sampling", env="tsp", num_loc=100, batch=4096, policy="am", decode="sampling",
         store_inputs=False),
    # configs[3]'s per-GPU share at full size: POMO-6L, 4096 instances x 8 starts, multistart sampling (32 768 tours)
    dict(name="c4_pomo_tsp100_b4096_s8_sampling", env="tsp", num_loc=100, batch=4096, policy="pomo",
         decode="multistart_sampling", fw_kw=dict(num_starts=8), store_inputs=False),
]

POMO_KW = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)  # pomo/model.py:52-67


def state_hash(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def to_plain(td) -> dict:
    return {k: v.clone() for k, v in td.items() if torch.is_tensor(v)}


def run_case(ref, case: dict) -> dict:
    env_name, n, b = case["env"], case["num_loc"], case["batch"]
    pol_kw = dict(case.get("pol_kw", {}))
    if case["policy"] == "pomo":
        pol_kw.update(POMO_KW)
    fw_kw = dict(case.get("fw_kw", {}))

    # ---- the real reference -----------------------------------------------------------------
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "op": ref.OPEnv, "pctsp": ref.PCTSPEnv, "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv, "spctsp": ref.SPCTSPEnv}[env_name]
    gen_kw = dict(num_loc=n)
    if env_name == "op":  # the default prize sampler Uniform(1.0, 1.0) does not pass torch's argument validation;
        gen_kw["prize_distribution"] = "dist"  # "dist" takes no sampler and is what prize_type="dist" (the default) uses
    ref_env = env_cls(generator_params=gen_kw, seed=0)
    torch.manual_seed(WEIGHT_SEED)
    ref_pol = ref.AttentionModelPolicy(env_name=env_name, **pol_kw).eval()
    torch.manual_seed(DATA_SEED)
    data = ref_env.generator(batch_size=[b])
    td0 = ref_env.reset(data.clone())
    t0 = time.perf_counter()
    torch.manual_seed(SAMPLE_SEED)
    with torch.inference_mode():
        out_ref = ref_pol(td0.clone(), ref_env, phase="test", decode_type=case["decode"], **fw_kw)
    wall = time.perf_counter() - t0

    # ---- the restatement on the same seeds ---------------------------------------------------
    env = R.get_env(env_name, n)
    torch.manual_seed(WEIGHT_SEED)
    pol = R.AttentionModelPolicy(env_name=env_name, **pol_kw).eval()
    sd_ref, sd = ref_pol.state_dict(), pol.state_dict()
    assert list(sd_ref.keys()) == list(sd.keys()), "state_dict keys differ from the reference"
    for k in sd:
        assert torch.equal(sd_ref[k], sd[k]), f"seeded weight {k} differs from the reference"
    torch.manual_seed(DATA_SEED)
    data2 = env.generate(b)
    for k in data2:
        assert torch.equal(data2[k], data[k]), f"generated {k} differs from the reference generator"
    td1 = env.reset({k: v.clone() for k, v in data2.items()})
    for k in ("locs", "action_mask"):
        assert torch.equal(td1[k], td0[k]), f"reset state {k} differs"
    torch.manual_seed(SAMPLE_SEED)
    with torch.inference_mode():
        out = pol(td1, env, phase="test", decode_type=case["decode"], **fw_kw)
    for k in ("actions", "reward", "log_likelihood") + (("entropy",) if "entropy" in out_ref else ()):
        assert out[k].shape == out_ref[k].shape, (case["name"], k, out[k].shape, out_ref[k].shape)
        assert torch.equal(out[k], out_ref[k]), f"{case['name']}: restatement {k} differs from the reference"

    # ---- sampling: the explicit Exp(1) race == torch.multinomial on the same generator ---------
    noise = None
    if "sampling" in case["decode"]:
        rec: list = []
        td2 = env.reset({k: v.clone() for k, v in data2.items()})
        torch.manual_seed(SAMPLE_SEED)
        with torch.inference_mode():
            out_n = pol(td2, env, phase="test", decode_type=case["decode"], noise_recorder=rec, **fw_kw)
        assert torch.equal(out_n["actions"], out_ref["actions"]), "exponential-race sampling != multinomial"
        noise = torch.stack(rec, 0)
        # the whole noise tensor is one seeded stream of per-step [B,N] exponential_ draws
        torch.manual_seed(SAMPLE_SEED)
        redraw = torch.stack([torch.empty_like(rec[0]).exponential_(1) for _ in rec], 0)
        assert torch.equal(noise, redraw)

    actions = out_ref["actions"]
    fixture = {
        "actions": actions.numpy().astype(np.uint8 if int(actions.max()) < 256 else np.uint16),
        "reward": out_ref["reward"].numpy(),
        "log_likelihood": out_ref["log_likelihood"].numpy(),
    }
    if "entropy" in out_ref:
        fixture["entropy"] = out_ref["entropy"].numpy()
    if case.get("store_inputs", True):
        for k, v in data.items():
            fixture[f"in_{k}"] = v.numpy()
    meta = {
        "name": case["name"], "env": env_name, "num_loc": n, "batch": b, "policy": case["policy"],
        "decode_type": case["decode"], "policy_kwargs": pol_kw, "forward_kwargs": fw_kw,
        "weight_seed": WEIGHT_SEED, "data_seed": DATA_SEED, "sample_seed": SAMPLE_SEED,
        "weights_sha256": state_hash(sd_ref),
        "inputs_sha256": state_hash({k: v for k, v in data.items()}),
        "steps": int(actions.shape[1]), "mean_reward": float(out_ref["reward"].mean()),
        "reference_cpu_seconds": round(wall, 3), "torch": torch.__version__,
        "threads": torch.get_num_threads(), "policy_only": bool(case.get("policy_only", False)),
    }
    return fixture, meta


def main() -> None:
    if not ref_import.available():
        raise SystemExit("reference checkout not present: goldens can only be generated in the build container")
    ref = ref_import.load()
    GOLDEN_DIR.mkdir(parents=True, exist_ok=True)
    manifest = {"generator": "oracle/gen_golden.py", "reference": "ai4co/rl4co v0.6.0 (verbatim source via oracle/ref_import.py)",
                "cases": []}
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only:
            continue
        fixture, meta = run_case(ref, case)
        np.savez_compressed(GOLDEN_DIR / f"{case['name']}.npz", **fixture)
        manifest["cases"].append(meta)
        print(f"{case['name']:32s} T={meta['steps']:4d} mean_reward={meta['mean_reward']:.6f} "
              f"ref_cpu={meta['reference_cpu_seconds']:.2f}s  restatement == reference: OK", flush=True)
    path = GOLDEN_DIR / "MANIFEST.json"
    if only and path.exists():  # partial run: replace / append the regenerated cases, keep the others
        old = json.loads(path.read_text())
        fresh = {c["name"]: c for c in manifest["cases"]}
        order = [c["name"] for c in CASES]
        merged = {c["name"]: c for c in old["cases"]}
        merged.update(fresh)
        manifest["cases"] = [merged[n] for n in order if n in merged]
    path.write_text(json.dumps(manifest, indent=1) + "\n")


if __name__ == "__main__":
    main()
