"""Rows N2 / N3: npz schema, dihedral-8 augmentation and the POMO evaluation
epilogue — checked against the REAL reference source (skipped where /root/reference is absent)
and against the oracle restatement; the rollout itself runs through the fake device (C oracle)."""
import numpy as np
import pytest
import torch

from oracle import ref_import
from oracle import reference_torch as R
from tests.fake_device import cpu_device  # noqa: F401
from tests.helpers import GoldenCase

needs_reference = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present")


def _td(g):
    from rl4co_amd.tensordict import TensorDict

    return TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])


def test_npz_roundtrip_reference_schema(tmp_path):
    from rl4co_amd.data import check_extension, load_npz_to_tensordict, save_tensordict_to_npz

    g = GoldenCase("cvrp20_b128_greedy")
    td = _td(g)
    fn = check_extension(str(tmp_path / "cvrp20"))
    assert fn.endswith(".npz")
    save_tensordict_to_npz(td, fn)
    assert sorted(np.load(fn).files) == ["capacity", "demand", "depot", "locs"]  # the reference's keys
    back = load_npz_to_tensordict(fn)
    assert back.batch_size[0] == g.batch
    for k in td.keys():
        assert torch.equal(back[k], td[k])


@needs_reference
def test_npz_matches_reference_loader(tmp_path):
    from rl4co_amd.data import save_tensordict_to_npz

    ref = ref_import.load()
    import importlib

    ref_utils = importlib.import_module("rl4co.data.utils")
    g = GoldenCase("tsp50_b64_greedy")
    fn = str(tmp_path / "tsp50.npz")
    save_tensordict_to_npz(_td(g), fn)
    td_ref = ref_utils.load_npz_to_tensordict(fn)
    assert np.array_equal(np.asarray(td_ref["locs"]), g.data["locs"].numpy())
    assert tuple(td_ref.batch_size) == (g.batch,)


@needs_reference
def test_dihedral8_matches_reference_transform():
    import importlib

    from rl4co_amd.data import StateAugmentation

    ref_import.install()
    tr = importlib.import_module("rl4co.data.transforms")
    TD = importlib.import_module("tensordict").TensorDict
    g = GoldenCase("tsp20_b64_greedy_simple")
    ours = StateAugmentation(8)(_td(g))
    theirs = tr.StateAugmentation(num_augment=8, augment_fn="dihedral8")(TD({"locs": g.data["locs"].clone()}, batch_size=[g.batch]))
    assert torch.equal(ours["locs"], theirs["locs"])
    assert ours["locs"].shape == (8 * g.batch, 20, 2)
    assert torch.equal(ours["locs"][: g.batch], g.data["locs"])  # first block = identity


def test_pomo_evaluate_epilogue(cpu_device):
    """best-of-starts then best-of-augmentations, shapes and values, vs the restatement's
    unbatchify/gather on the same rollout output (zoo/pomo/model.py:112-140)."""
    from rl4co_amd.data import pomo_evaluate
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    g = GoldenCase("pomo_tsp20_b16_msgreedy")
    pol = AttentionModelPolicy(env_name="tsp", **g.meta["policy_kwargs"]).eval()
    pol.load_state_dict(g.policy.state_dict())
    env = get_env("tsp", generator_params=dict(num_loc=20), device="cpu")
    with torch.inference_mode():
        out = pomo_evaluate(pol, env, env.reset(_td(g)), num_augment=8, num_starts=5)
    b = g.batch
    assert out["reward_per_aug_start"].shape == (b, 8, 5)
    assert out["max_reward"].shape == (b, 8) and out["max_aug_reward"].shape == (b,)
    assert out["actions"].shape == (b, 8, 5, 20)
    assert out["best_multistart_actions"].shape == (b, 8, 20) and out["best_aug_actions"].shape == (b, 20)
    # values: restatement ops on the flat outputs
    flat_reward = out["reward"]
    want = R.unbatchify(flat_reward, (8, 5))
    assert torch.equal(out["reward_per_aug_start"], want)
    assert torch.equal(out["max_aug_reward"], want.max(-1).values.max(-1).values)
    # the selected tour really has the selected reward (on the augmentation-0 coordinates all
    # symmetric copies have the same length up to rounding)
    tours = out["best_aug_actions"]
    lengths = R.get_tour_length(R.gather_by_index(g.data["locs"], tours))
    torch.testing.assert_close(-lengths, out["max_aug_reward"], rtol=1e-5, atol=1e-5)
    # augmentation can only help: best over 8 symmetries >= the identity block's best start
    assert bool((out["max_aug_reward"] >= out["max_reward"][:, 0] - 1e-6).all())
