"""Parity of the product against the reference's rollouts on TRAINED weights (tests/golden/trained).

Shared by ``tests/test_gpu_trained_parity.py`` (asserts) and ``bench.py``'s ``parity`` block (reports). Nothing here
touches ``oracle/`` or ``/root/reference``: the fixtures hold the reference's actions / rewards (made in the build
container by ``oracle/gen_trained_golden.py``), inputs are re-created from their seed and verified by hash, weights
come from ``tests/golden/weights/*.safetensors`` (trained by ``tools/train_sharp.py``). Lives under ``tests/`` (r06: test
infrastructure does not belong in the scratch directory ``tools/``); ``bench.py`` puts ``tests/`` on its path to import it.

What is measured for one (case, product configuration):

* ``identical`` / ``flips``: greedy tours equal to / different from the reference's, rewards bit-identical on the former;
* for every flip the **regret** at the first divergent step: the product, teacher-forced along the reference's tour,
  gives ``max_j logp_j - logp[reference's action]`` at that state. The common prefix makes it the same state in both
  rollouts, so a flip with regret < 1e-5 is a near-tie of the product's own arithmetic that the reference's summation
  order resolved the other way — not a different policy;
* ``step_agreement``: share of the reference's (live) decisions the product's arg-max reproduces when it is held on the
  reference's trajectory — the per-decision agreement that the all-or-nothing tour count compounds over ~100 steps.
"""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (this file lives in tests/)
TRAINED_DIR = os.path.join(ROOT, "tests", "golden", "trained")
WEIGHT_DIR = os.path.join(ROOT, "tests", "golden", "weights")


def state_hash(tensors: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(tensors[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def manifest() -> dict:
    with open(os.path.join(TRAINED_DIR, "MANIFEST.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


class TrainedCase:
    """One fixture: the reference's outputs + the seeded inputs and the weight set it was run on."""

    def __init__(self, name: str):
        self.name = name
        self.meta = m = manifest()[name]
        z = np.load(os.path.join(TRAINED_DIR, f"{name}.npz"))
        self.actions = torch.from_numpy(z["actions"].astype(np.int64))
        self.reward = torch.from_numpy(z["reward"])
        self.log_likelihood = torch.from_numpy(z["log_likelihood"])
        self.has_bf16 = "actions_bf16" in z.files
        if self.has_bf16:
            self.actions_bf16 = torch.from_numpy(z["actions_bf16"].astype(np.int64))
            self.reward_bf16 = torch.from_numpy(z["reward_bf16"])
        self.env_name, self.num_loc, self.batch = m["env"], m["num_loc"], m["batch"]
        self.num_starts = int(m.get("num_starts") or 0)  # multistart cases: rows are start-major over instances
        if "aug_reward" in z.files:  # POMO evaluation protocol: dihedral-8 x multistart (zoo/pomo/model.py:112-140)
            self.max_reward = torch.from_numpy(z["max_reward"])                  # [B] best start, no augmentation
            self.aug_reward = torch.from_numpy(z["aug_reward"])                  # [S * A * B] all rollouts
            self.aug_max_reward = torch.from_numpy(z["aug_max_reward"])          # [B, A] best start per augmentation
            self.aug_max_aug_reward = torch.from_numpy(z["aug_max_aug_reward"])  # [B] best augmentation
        self.n_nodes = self.num_loc + (0 if self.env_name == "tsp" else 1)

    # -- inputs / weights ---------------------------------------------------------------------------------------------
    def instances(self, device):
        from rl4co_amd.envs import get_env

        env_cpu = get_env(self.env_name, generator_params=dict(num_loc=self.num_loc, device="cpu"), device="cpu")
        torch.manual_seed(self.meta["data_seed"])
        data = env_cpu.generator(batch_size=[self.batch])
        if state_hash({k: v for k, v in data.items()}) != self.meta["inputs_sha256"]:
            raise RuntimeError(f"{self.name}: seeded inputs differ from the golden run")
        return data.to(device)

    def state_dict(self) -> dict:
        if self.meta["weights"] is not None:
            from safetensors.torch import load_file

            sd = load_file(os.path.join(WEIGHT_DIR, f"{self.meta['weights']}.safetensors"))
        else:  # seeded random-init weights with the logit-key projection scaled (oracle/gen_trained_golden.py)
            from rl4co_amd.policy import AttentionModelPolicy

            torch.manual_seed(self.meta["weight_seed"])
            sd = {k: v.clone() for k, v in AttentionModelPolicy(self.env_name).state_dict().items()}
            w = sd["decoder.project_node_embeddings.weight"]
            w[2 * (w.shape[0] // 3):] *= float(self.meta["kl_scale"])
        if state_hash(sd) != self.meta["weights_sha256"]:
            raise RuntimeError(f"{self.name}: weights differ from the golden run")
        return sd

    def policy(self, device, **kw):
        from rl4co_amd.policy import AttentionModelPolicy

        if self.meta.get("arch") == "pomo":  # zoo/pomo/model.py:52-67
            kw = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False,
                      train_decode_type="multistart_sampling", val_decode_type="multistart_greedy",
                      test_decode_type="multistart_greedy", **kw)
        pol = AttentionModelPolicy(self.env_name, **kw)
        pol.load_state_dict(self.state_dict(), strict=True)
        return pol.to(device).eval()

    def env(self, device):
        from rl4co_amd.envs import get_env

        return get_env(self.env_name, generator_params=dict(num_loc=self.num_loc, device=device), device=device)

    def reference_noise(self, steps: int) -> torch.Tensor:
        """The reference's sampling stream: one [B, N] exponential_ draw per decode step from manual_seed(sample_seed)
        (proved equal to torch.multinomial's by oracle/gen_golden.py)."""
        torch.manual_seed(self.meta["sample_seed"])
        return torch.stack([torch.empty(self.batch, self.n_nodes).exponential_(1) for _ in range(steps)], 0).contiguous()


CONFIGS = {
    "fp32": dict(cache_dtype=torch.float32),
    "fp32_fold_off": dict(cache_dtype=torch.float32, fold=False),
    "bf16": dict(cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16),
    "fp16": dict(cache_dtype=torch.float16, encoder_autocast=torch.float16),  # the reference's default "16-mixed"
}


def _pad(a: torch.Tensor, t: int) -> torch.Tensor:
    if a.shape[1] >= t:
        return a
    return torch.cat([a, a.new_zeros(a.shape[0], t - a.shape[1])], 1)


def compare(case: TrainedCase, config: str, device, against: str = "fp32", decode: str = "greedy", regret: bool = True) -> dict:
    """Roll the product out in `config` and compare with the reference's run `against` in {"fp32", "bf16"}."""
    ref_actions = (case.actions if against == "fp32" else case.actions_bf16).to(device)
    ref_reward = (case.reward if against == "fp32" else case.reward_bf16).to(device)
    pol, env, data = case.policy(device, **CONFIGS[config]), case.env(device), case.instances(device)
    kw = {}
    if decode == "sampling":
        t_noise = ref_actions.shape[1] + 64
        kw = dict(exp_noise=case.reference_noise(t_noise).to(device), max_steps=t_noise)
    if decode.startswith("multistart"):
        kw["num_starts"] = case.num_starts
    with torch.inference_mode():
        out = pol(env.reset(data.clone()), env, phase="test", decode_type=decode, **kw)
    acts = out["actions"]
    t = max(acts.shape[1], ref_actions.shape[1])
    a, r = _pad(acts, t), _pad(ref_actions, t)
    same = (a == r).all(1)
    rec = {
        "of": int(a.shape[0]), "identical": int(same.sum()), "flips": int((~same).sum()),
        "identical_frac": float(same.float().mean()),
        "rewards_bit_identical_on_identical": bool(torch.equal(out["reward"][same], ref_reward[same])) if bool(same.any()) else None,
        "mean_reward": float(out["reward"].mean()), "mean_reward_reference": float(ref_reward.mean()),
        "reward_rel_gap": abs(float(out["reward"].double().mean() - ref_reward.double().mean())) / abs(float(ref_reward.double().mean())),
        "common_prefix_steps_mean": float((a == r).long().cumprod(1).sum(1).float().mean()), "steps": int(ref_actions.shape[1]),
    }
    if decode.startswith("multistart") and against == "fp32" and hasattr(case, "max_reward"):
        best = out["reward"].view(case.num_starts, case.batch).max(0).values
        rec["best_of_starts_bit_identical"] = int((best == case.max_reward.to(device)).sum())
        rec["best_of_starts_rel_gap"] = abs(float(best.double().mean() - case.max_reward.double().mean())) / abs(float(case.max_reward.double().mean()))
    if regret and decode == "greedy":
        with torch.inference_mode():
            ev = pol(env.reset(data.clone()), env, phase="test", actions=ref_actions, calc_reward=False, return_all_logp=True)
        lp = ev["all_logp"][:, : ref_actions.shape[1]]                      # [B, T, N] along the reference's tours
        choice = lp.argmax(-1)
        # live decisions: up to and including the reference's last non-trivial action (CVRP pads finished rows with depot)
        steps = torch.arange(ref_actions.shape[1], device=device)
        last = ((ref_actions != 0) * steps).max(1).values
        live = steps[None, :] <= (last[:, None] + (0 if case.env_name == "tsp" else 1))
        if case.env_name == "tsp":
            live = torch.ones_like(live)
        rec["step_agreement"] = float(((choice == ref_actions) & live).sum() / live.sum())
        rec["live_decisions"] = int(live.sum())
        rows = (~same).nonzero().flatten()
        if rows.numel():
            first = (a[rows] == r[rows]).long().cumprod(1).sum(1).clamp(max=ref_actions.shape[1] - 1)
            at = lp[rows, first]                                            # the product's log-probs at the shared state
            reg = at.max(-1).values - at.gather(-1, ref_actions[rows, first][:, None]).squeeze(-1)
            rec.update({"flip_regret_max": float(reg.max()), "flip_regret_mean": float(reg.mean()),
                        "flips_with_regret_above_1e-5": int((reg > 1e-5).sum()),
                        "flips_with_regret_above_1e-4": int((reg > 1e-4).sum())})
        else:
            rec.update({"flip_regret_max": 0.0, "flip_regret_mean": 0.0, "flips_with_regret_above_1e-5": 0,
                        "flips_with_regret_above_1e-4": 0})
    return rec


def compare_augmented(case: TrainedCase, config: str, device) -> dict:
    """The POMO evaluation protocol on trained weights: ``rl4co_amd.data.pomo_evaluate`` (device augmentation kernel,
    multistart greedy rollouts, the fused best-of epilogue) against the reference's ``StateAugmentation`` +
    ``POMO.shared_step`` maxima (fixture keys ``aug_*``, fp32 reference run)."""
    from rl4co_amd.data import pomo_evaluate

    pol, env, data = case.policy(device, **CONFIGS[config]), case.env(device), case.instances(device)
    n_aug = int(case.meta["num_augment"])
    with torch.inference_mode():
        out = pomo_evaluate(pol, env, env.reset(data.clone()), num_augment=n_aug, num_starts=case.num_starts)
    ref_all = case.aug_reward.to(device)
    ref_ma, ref_best = case.aug_max_reward.to(device), case.aug_max_aug_reward.to(device)
    same_rollouts = out["reward"] == ref_all
    return {
        "of_rollouts": int(ref_all.numel()), "rollout_rewards_bit_identical": int(same_rollouts.sum()),
        "rollout_rewards_identical_frac": float(same_rollouts.float().mean()),
        "instances": int(ref_best.numel()),
        "max_reward_bit_identical": int((out["max_reward"] == ref_ma).sum()), "of_max_reward": int(ref_ma.numel()),
        "max_aug_reward_bit_identical": int((out["max_aug_reward"] == ref_best).sum()),
        "mean_max_aug_reward": float(out["max_aug_reward"].mean()), "mean_max_aug_reward_reference": float(ref_best.mean()),
        "max_aug_reward_rel_gap": abs(float(out["max_aug_reward"].double().mean() - ref_best.double().mean())) / abs(float(ref_best.double().mean())),
        "num_augment": n_aug, "num_starts": case.num_starts,
    }
