"""ORACLE (test infrastructure): goldens of the BENCHMARKED configurations on TRAINED weights.

    python -m oracle.gen_trained_golden [case ...]     # build container only (needs /root/reference)

Why: with random-init weights the AttentionModel is a near-uniform policy — every greedy step
is a near-tie at the 1e-2 level — so the benchmarked bf16 configuration reproduced 0 of 4096 reference tours and
could only be checked on tour quality. Here the reference's OWN policy class (imported verbatim, oracle/ref_import.py)
is loaded with weight sets trained by the product (tools/train_sharp.py, committed under tests/golden/weights/ —
which also proves the state_dict round trip product -> reference) and rolled out on the CPU

  * in fp32 (the reference's arithmetic), and
  * under ``torch.autocast("cpu", dtype=torch.bfloat16)`` — the reference's mixed-precision regime
    (constructive/base.py:154-263 executed under autocast, as Lightning's precision plugin does),

at the full size of BASELINE configs[1] (TSP-100 x 4096), configs[2] (CVRP-100 x 4096) and configs[4]
(CVRP-500 x 1024, greedy and fixed-seed sampling), plus one "sharpened K_l" case: random-init
weights whose logit-key projection is scaled until the logits sit on the tanh plateau (exact ties at +-10 resolved by
index). The fp32 run is repeated through the restatement (oracle/reference_torch.py) and must agree bit for bit.

Note on the autocast op lists: torch's CPU autocast casts linear / matmul / bmm / SDPA to bf16 like the CUDA list,
but leaves ``log_softmax`` in the input dtype (the CUDA list promotes it to fp32). For greedy decoding the difference
is confined to additional exact ties of bf16 log-probs, all broken towards the lowest index on both devices.
"""
from __future__ import annotations

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
from oracle.gen_golden import DATA_SEED, SAMPLE_SEED, WEIGHT_SEED, state_hash  # noqa: E402

OUT_DIR = ROOT / "tests" / "golden" / "trained"
WEIGHT_DIR = ROOT / "tests" / "golden" / "weights"

CASES = [
    dict(name="t2_tsp100_b4096_greedy", env="tsp", num_loc=100, batch=4096, weights="am_tsp100_sharp", decode="greedy", bf16=True),
    dict(name="t3_cvrp100_b4096_greedy", env="cvrp", num_loc=100, batch=4096, weights="am_cvrp100_sharp", decode="greedy", bf16=True),
    dict(name="t5_cvrp500_b1024_greedy", env="cvrp", num_loc=500, batch=1024, weights="am_cvrp100_sharp", decode="greedy", bf16=True),
    dict(name="t5_cvrp500_b1024_sampling", env="cvrp", num_loc=500, batch=1024, weights="am_cvrp100_sharp", decode="sampling", bf16=False),
    dict(name="t2_tsp100_b4096_sampling", env="tsp", num_loc=100, batch=4096, weights="am_tsp100_sharp", decode="sampling", bf16=False),
    # tanh plateau: logit key x KL_SCALE on seeded random-init weights
    # (probed: x 40 still picks the unscaled arg-max, x 100 mixes plateau ties with knee decisions, from x 200 on nearly
    # every step is an exact tie and the tour degenerates to index order)
    dict(name="sharpkl100_tsp100_b1024_greedy", env="tsp", num_loc=100, batch=1024, weights=None, kl_scale=100.0, decode="greedy", bf16=False),
    dict(name="sharpkl400_tsp100_b512_greedy", env="tsp", num_loc=100, batch=512, weights=None, kl_scale=400.0, decode="greedy", bf16=False),
    # BASELINE configs[3]'s policy (POMO: 6 layers, instance norm, no graph context) at its evaluation protocol
    # (zoo/pomo/model.py:99-140): one greedy rollout from every start node, and the best of 8 dihedral augmentations x starts
    dict(name="t4_pomo_tsp100_b256_msgreedy", env="tsp", num_loc=100, batch=256, weights="pomo_tsp100_sharp", arch="pomo",
         decode="multistart_greedy", bf16=True, augment=8),
]
POMO_KW = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)  # zoo/pomo/model.py:52-67


def load_weights(name: str) -> dict:
    from safetensors.torch import load_file

    return load_file(str(WEIGHT_DIR / f"{name}.safetensors"))


def sharpen_logit_key(sd: dict, scale: float) -> dict:
    """Scale the logit-key third of project_node_embeddings (zoo/am/decoder.py:201-228: glimpse key | glimpse value |
    logit key) so that logits / sqrt(d) pass the fp32 tanh == 1.0 knee (9.011): exact ties at +-tanh_clipping."""
    sd = {k: v.clone() for k, v in sd.items()}
    w = sd["decoder.project_node_embeddings.weight"]
    d = w.shape[0] // 3
    w[2 * d:] *= scale
    return sd


def build_policies(ref, case):
    env_name = case["env"]
    torch.manual_seed(WEIGHT_SEED)
    arch = POMO_KW if case.get("arch") == "pomo" else {}
    ref_pol = ref.AttentionModelPolicy(env_name=env_name, **arch).eval()
    if case["weights"] is not None:
        sd = load_weights(case["weights"])
    else:
        sd = sharpen_logit_key(ref_pol.state_dict(), case["kl_scale"])
    missing = ref_pol.load_state_dict(sd, strict=True)  # product-trained checkpoint -> the reference's module tree
    assert not missing.missing_keys and not missing.unexpected_keys
    torch.manual_seed(WEIGHT_SEED)
    pol = R.AttentionModelPolicy(env_name=env_name, **arch).eval()
    pol.load_state_dict(sd, strict=True)
    return ref_pol.eval(), pol.eval(), sd


def run_case(ref, case):
    env_name, n, b = case["env"], case["num_loc"], case["batch"]
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv}[env_name]
    ref_env = env_cls(generator_params=dict(num_loc=n), seed=0)
    ref_pol, pol, sd = build_policies(ref, case)
    torch.manual_seed(DATA_SEED)
    data = ref_env.generator(batch_size=[b])
    fixture, meta = {}, {}
    t0 = time.perf_counter()
    torch.manual_seed(SAMPLE_SEED)
    with torch.inference_mode():
        out32 = ref_pol(ref_env.reset(data.clone()), ref_env, phase="test", decode_type=case["decode"])
    meta["reference_cpu_seconds_fp32"] = round(time.perf_counter() - t0, 2)
    # the restatement on the same weights / data: bit for bit
    env = R.get_env(env_name, n)
    torch.manual_seed(DATA_SEED)
    data2 = env.generate(b)
    for k in data2:
        assert torch.equal(data2[k], data[k])
    torch.manual_seed(SAMPLE_SEED)
    with torch.inference_mode():
        out_r = pol(env.reset({k: v.clone() for k, v in data2.items()}), env, phase="test", decode_type=case["decode"])
    for k in ("actions", "reward", "log_likelihood"):
        assert torch.equal(out_r[k], out32[k]), f"{case['name']}: restatement {k} differs from the reference"

    def pack(actions):
        return actions.numpy().astype(np.uint8 if int(actions.max()) < 256 else np.uint16)

    fixture["actions"] = pack(out32["actions"])
    fixture["reward"] = out32["reward"].numpy()
    fixture["log_likelihood"] = out32["log_likelihood"].numpy()
    meta.update(steps=int(out32["actions"].shape[1]), mean_reward=float(out32["reward"].mean()))
    if case["decode"] == "greedy":
        # how sharp the policy is: the share of greedy decisions taken with probability > 0.5 / > 0.9 is
        # exp(per-step log-prob); only the summed log-likelihood is returned, so report its mean per step
        meta["mean_logp_per_step"] = float((out32["log_likelihood"] / out32["actions"].shape[1]).mean())
    if case.get("augment"):
        # the reference's own augmentation module and best-of epilogue (data/transforms.py:105-151, pomo/model.py:112-140)
        import importlib

        tr = importlib.import_module("rl4co.data.transforms")
        n_aug, n_start = case["augment"], ref_env.get_num_starts(ref_env.reset(data.clone()))
        t0 = time.perf_counter()
        with torch.inference_mode():
            td_aug = tr.StateAugmentation(num_augment=n_aug, augment_fn="dihedral8")(ref_env.reset(data.clone()))
            out_aug = ref_pol(td_aug, ref_env, phase="test", decode_type=case["decode"], num_starts=n_start)
            rw = ref.ops.unbatchify(out_aug["reward"], (n_aug, n_start))           # [B, aug, starts]
            max_reward, _ = rw.max(dim=-1)                                          # best start per augmentation
            max_aug_reward, _ = max_reward.max(dim=1)                               # best augmentation
            base = ref.ops.unbatchify(out32["reward"], n_start).max(dim=-1).values  # no augmentation: best start
        meta["reference_cpu_seconds_augment"] = round(time.perf_counter() - t0, 2)
        fixture["aug_reward"] = out_aug["reward"].numpy()
        fixture["aug_max_reward"] = max_reward.numpy()
        fixture["aug_max_aug_reward"] = max_aug_reward.numpy()
        fixture["max_reward"] = base.numpy()
        meta.update(num_augment=n_aug, num_starts=int(n_start), mean_max_reward=float(base.mean()),
                    mean_max_aug_reward=float(max_aug_reward.mean()))
    if case.get("bf16"):
        t0 = time.perf_counter()
        with torch.inference_mode(), torch.autocast("cpu", dtype=torch.bfloat16):
            out16 = ref_pol(ref_env.reset(data.clone()), ref_env, phase="test", decode_type=case["decode"])
        meta["reference_cpu_seconds_bf16"] = round(time.perf_counter() - t0, 2)
        fixture["actions_bf16"] = pack(out16["actions"])
        fixture["reward_bf16"] = out16["reward"].float().numpy()
        fixture["log_likelihood_bf16"] = out16["log_likelihood"].float().numpy()
        t32, t16 = out32["actions"].shape[1], out16["actions"].shape[1]
        t = min(t32, t16)
        same = (out32["actions"][:, :t] == out16["actions"][:, :t]).all(1) if t32 == t16 else torch.zeros(b, dtype=torch.bool)
        meta.update(steps_bf16=int(t16), mean_reward_bf16=float(out16["reward"].float().mean()),
                    reference_bf16_vs_fp32_identical=int(same.sum()),
                    reference_bf16_dtypes={k: str(out16[k].dtype) for k in ("reward", "log_likelihood")})
    meta.update(name=case["name"], env=env_name, num_loc=n, batch=b, decode_type=case["decode"],
                weights=case["weights"], arch=case.get("arch"), kl_scale=case.get("kl_scale"), weight_seed=WEIGHT_SEED, data_seed=DATA_SEED,
                sample_seed=SAMPLE_SEED, weights_sha256=state_hash(sd), inputs_sha256=state_hash({k: v for k, v in data.items()}),
                torch=torch.__version__, threads=torch.get_num_threads())
    return fixture, meta


def main() -> None:
    if not ref_import.available():
        raise SystemExit("reference checkout not present: goldens can only be generated in the build container")
    ref = ref_import.load()
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    path = OUT_DIR / "MANIFEST.json"
    manifest = json.loads(path.read_text()) if path.exists() else {
        "generator": "oracle/gen_trained_golden.py",
        "reference": "ai4co/rl4co v0.6.0 (verbatim source via oracle/ref_import.py), weights trained by tools/train_sharp.py",
        "cases": []}
    merged = {c["name"]: c for c in manifest["cases"]}
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only:
            continue
        fixture, meta = run_case(ref, case)
        np.savez_compressed(OUT_DIR / f"{case['name']}.npz", **fixture)
        merged[case["name"]] = meta
        print(json.dumps(meta), flush=True)
    manifest["cases"] = [merged[c["name"]] for c in CASES if c["name"] in merged]
    path.write_text(json.dumps(manifest, indent=1) + "\n")


if __name__ == "__main__":
    main()
