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
def test_dihedral8_matches_reference_transform(cpu_device):
    import importlib

    from rl4co_amd.data import StateAugmentation

    ref_import.install()
    tr = importlib.import_module("rl4co.data.transforms")
    TD = importlib.import_module("tensordict").TensorDict
    g = GoldenCase("tsp20_b64_greedy_simple")
    ours = StateAugmentation(8, augment_fn="dihedral8")(_td(g))
    theirs = tr.StateAugmentation(num_augment=8, augment_fn="dihedral8")(TD({"locs": g.data["locs"].clone()}, batch_size=[g.batch]))
    assert torch.equal(ours["locs"], theirs["locs"])
    assert ours["locs"].shape == (8 * g.batch, 20, 2)
    assert torch.equal(ours["locs"][: g.batch], g.data["locs"])  # first block = identity


@needs_reference
@pytest.mark.parametrize("kw", [dict(num_augment=8), dict(num_augment=4, first_aug_identity=False),
                                dict(num_augment=5, normalize=True), dict(num_augment=8, feats=["locs", "depot2"]),
                                dict(num_augment=8, augment_fn="dihedral8", first_aug_identity=False)])
def test_state_augmentation_matches_reference_transform(kw, cpu_device):
    """rl4co/data/transforms.py:49-151 argument for argument: the symmetric (SymNCO) augmentation consumes the global
    torch generator like the reference (same seed -> the same rows, bit for bit), first-block identity,
    `first_aug_identity=False`'s own row restore, min-max normalisation, several features."""
    import importlib

    from rl4co_amd.data import StateAugmentation

    ref_import.install()
    tr = importlib.import_module("rl4co.data.transforms")
    TD = importlib.import_module("tensordict").TensorDict
    g = GoldenCase("tsp20_b64_greedy_simple")
    data = {"locs": g.data["locs"].clone(), "depot2": g.data["locs"][:, :3].clone()}
    torch.manual_seed(77)
    ours = StateAugmentation(**kw)(__import__("rl4co_amd.tensordict", fromlist=["TensorDict"]).TensorDict(
        {k: v.clone() for k, v in data.items()}, batch_size=[g.batch]))
    torch.manual_seed(77)
    theirs = tr.StateAugmentation(**kw)(TD({k: v.clone() for k, v in data.items()}, batch_size=[g.batch]))
    for k in data:
        assert ours[k].shape == theirs[k].shape == (kw["num_augment"] * g.batch, *data[k].shape[1:])
        assert torch.equal(ours[k], theirs[k]), k


def test_symmetric_augmentation_is_an_isometry_with_identity_first_block(cpu_device):
    """Size-independent properties (also what the GPU test checks at full size): rotations / reflections about the
    centre keep every pairwise distance; the first block is untouched; injected angles reproduce the draw."""
    from rl4co_amd.data import symmetric_augmentation

    torch.manual_seed(3)
    xy = torch.rand(6, 30, 2).repeat(4, 1, 1)
    torch.manual_seed(5)
    out = symmetric_augmentation(xy, num_augment=4)
    assert torch.equal(out[:6], xy[:6])
    pd = lambda a: (a[:, :, None, :] - a[:, None, :, :]).norm(dim=-1)  # noqa: E731  (cdist's GEMM form cancels badly)
    torch.testing.assert_close(pd(out), pd(xy), rtol=0, atol=2e-6)
    assert not torch.allclose(out[6:], xy[6:])
    torch.manual_seed(5)
    phi = torch.rand(24) * 4 * 3.141592653589793
    assert torch.equal(symmetric_augmentation(xy, num_augment=4, phi=phi), out)


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
    assert torch.equal(out["max_reward"], want.max(-1).values)
    assert torch.equal(out["max_aug_reward"], want.max(-1).values.max(-1).values)
    flat_actions = out["actions"].permute(2, 1, 0, 3).reshape(-1, 20)  # back to the rollout's row order
    acts = R.unbatchify(flat_actions, (8, 5))
    assert torch.equal(out["actions"], acts)
    idx = want.max(-1).indices
    best_ms = R.gather_by_index(acts, idx, dim=idx.dim())
    assert torch.equal(out["best_multistart_actions"], best_ms)
    assert torch.equal(out["best_aug_actions"], R.gather_by_index(best_ms, want.max(-1).values.max(1).indices, dim=1))
    # the selected tour really has the selected reward (on the augmentation-0 coordinates all
    # symmetric copies have the same length up to rounding)
    tours = out["best_aug_actions"]
    lengths = R.get_tour_length(R.gather_by_index(g.data["locs"], tours))
    torch.testing.assert_close(-lengths, out["max_aug_reward"], rtol=1e-5, atol=1e-5)
    # augmentation can only help: best over 8 symmetries >= the identity block's best start
    assert bool((out["max_aug_reward"] >= out["max_reward"][:, 0] - 1e-6).all())


@needs_reference
@pytest.mark.parametrize("a,s,b,t", [(8, 6, 16, 20), (1, 7, 5, 9), (8, 1, 4, 3), (3, 100, 7, 50), (70, 2, 3, 4)])
def test_pomo_best_restatement_equals_reference_epilogue(a, s, b, t):
    """oracle_pomo_best (the host restatement of rl4co_pomo_best, which the GPU tests pin the kernel to) against the
    reference's own unbatchify / max / gather_by_index (utils/ops.py:33-66 as used by zoo/pomo/model.py:112-140), with
    many exact ties in the rewards: the first maximum wins on both axes."""
    import importlib

    from oracle import c_oracle

    ref_import.install()
    ops = importlib.import_module("rl4co.utils.ops")
    torch.manual_seed(a * 1000 + s)
    reward = torch.randint(0, 4, (s * a * b,)).float() * -0.5  # four distinct values: ties everywhere
    if s * a * b > 40:  # and a few NaN rewards: torch.max propagates them (the first NaN wins), so must the epilogue
        reward[torch.randperm(s * a * b)[:5]] = float("nan")
    actions = torch.randint(0, 50, (s * a * b, t))
    got = c_oracle.pomo_best(reward, actions, a, s)
    r = ops.unbatchify(reward, (a, s))
    acts = ops.unbatchify(actions, (a, s))
    r = r.view(b, a, s)
    acts = acts.view(b, a, s, t)
    max_reward, idx = r.max(dim=-1)
    assert torch.equal(got["max_reward"].nan_to_num(7.0), max_reward.nan_to_num(7.0)) and torch.equal(got["best_start"], idx)
    best_ms = ops.gather_by_index(acts, idx, dim=idx.dim()).view(b, a, t)
    assert torch.equal(got["best_multistart_actions"], best_ms)
    max_aug, idx2 = max_reward.max(dim=1)
    assert torch.equal(got["max_aug_reward"].nan_to_num(7.0), max_aug.nan_to_num(7.0)) and torch.equal(got["best_aug"], idx2)
    assert torch.equal(got["best_aug_actions"], ops.gather_by_index(best_ms, idx2, dim=1).view(b, t))
    no_actions = c_oracle.pomo_best(reward, None, a, s)
    assert torch.equal(no_actions["max_aug_reward"].nan_to_num(7.0), max_aug.nan_to_num(7.0)) and "best_aug_actions" not in no_actions


def test_augmentation_refuses_cpu_tensors_without_the_device():
    """No CPU fallback: outside the fake-device fixture the augmentation kernels reject host tensors loudly."""
    from rl4co_amd import _lib
    from rl4co_amd.data import StateAugmentation
    from rl4co_amd.tensordict import TensorDict

    td = TensorDict({"locs": torch.rand(4, 5, 2)}, batch_size=[4])
    with pytest.raises(_lib.Rl4coLibraryError, match="no CPU fallback"):
        StateAugmentation(8, augment_fn="dihedral8")(td)


# ---------------------------------------------------------------------------------------------
# instance files and datasets (row N2): generate_data.py / dataset.py / RL4COEnvBase.dataset
# ---------------------------------------------------------------------------------------------

@needs_reference
@pytest.mark.parametrize("problem,size,dist", [("tsp", 20, None), ("vrp", 50, None), ("pdp", 20, None), ("op", 20, "const"),
                                               ("op", 50, "unif"), ("op", 100, "dist"), ("pctsp", 50, None)])
def test_generated_instance_files_equal_reference_generator(problem, size, dist):
    """The same numpy draws under the same seed: every array of the file equals the reference generator's, bit for bit."""
    import importlib

    from rl4co_amd import data as D

    ref_import.install()
    ref_gen = importlib.import_module("rl4co.data.generate_data")
    np.random.seed(4321)
    want = ref_gen.generate_env_data(problem, 64, size, dist)
    np.random.seed(4321)
    got = D.generate_env_data(problem, 64, size, dist)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k


@needs_reference
def test_generate_dataset_files_equal_reference(tmp_path):
    import importlib

    from rl4co_amd import data as D

    ref_import.install()
    ref_gen = importlib.import_module("rl4co.data.generate_data")
    kw = dict(name="val", problem="op", dataset_size=32, graph_sizes=[20, 50], seed=4321)
    ours = D.generate_dataset(data_dir=str(tmp_path / "ours"), **kw)
    ref_gen.generate_dataset(data_dir=str(tmp_path / "ref"), **kw)
    assert len(ours) == 6  # three prize distributions x two sizes
    for f in ours:
        rel = f[len(str(tmp_path / "ours")) + 1:]
        a, b = np.load(f), np.load(str(tmp_path / "ref" / rel))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (rel, k)
    assert D.generate_dataset(data_dir=str(tmp_path / "ours"), **kw) == []  # existing files are kept unless overwrite=True
    with pytest.raises(NotImplementedError):
        D.generate_env_data("mdpp", 4, 10)


def test_dataset_serves_batches_and_env_dataset_loads_files(cpu_device, tmp_path):
    from rl4co_amd import data as D
    from rl4co_amd.envs import get_env

    files = D.generate_dataset(data_dir=str(tmp_path), name="test", problem="vrp", dataset_size=40, graph_sizes=20, seed=1234)
    assert files == [str(tmp_path / "vrp" / "vrp20_test_seed1234.npz")]
    env = get_env("cvrp", generator_params=dict(num_loc=20), device="cpu", data_dir=str(tmp_path),
                  test_file="vrp/vrp20_test_seed1234.npz")
    ds = env.dataset(phase="test")
    assert len(ds) == 40 and sorted(ds.data.keys()) == ["capacity", "demand", "depot", "locs"]
    raw = np.load(files[0])
    assert np.array_equal(ds.data["demand"].numpy(), raw["demand"] / raw["capacity"][:, None])  # cvrp/env.py:179-186
    td0 = env.reset(ds.__getitems__(list(range(8))))  # a loaded batch resets like a generated one
    assert td0["action_mask"].shape == (8, 21) and bool(td0["action_mask"][:, 1:].all())
    batches = list(ds.batches(16))
    assert [b.batch_size[0] for b in batches] == [16, 16, 8]
    assert torch.equal(torch.cat([b["locs"] for b in batches]), ds.data["locs"])
    g = torch.Generator().manual_seed(0)
    shuffled = torch.cat([b["demand"] for b in ds.batches(16, shuffle=True, generator=g)])
    assert not torch.equal(shuffled, ds.data["demand"]) and torch.equal(shuffled.sort(0).values, ds.data["demand"].sort(0).values)
    ds.add_key("extra", torch.arange(40.0))
    b0 = ds.__getitems__([3, 5])
    assert b0["extra"].tolist() == [3.0, 5.0] and torch.equal(b0["locs"], ds.data["locs"][[3, 5]])
    stacked = ds.collate_fn([ds[i] for i in (3, 5)])
    assert torch.equal(stacked["locs"], b0["locs"])
    # no file configured / file missing: generated on the fly (base.py:240-262)
    assert len(env.dataset(batch_size=[12], phase="val")) == 12
    assert len(env.dataset(batch_size=[7], phase="val", filename=str(tmp_path / "missing.npz"))) == 7


def test_rollout_baseline_wraps_dataset_with_greedy_rewards(cpu_device):
    """RolloutBaseline.rollout / wrap_dataset (baselines.py:218-248): batch-wise greedy rewards == one big greedy
    rollout, attached to the dataset as "extra" and served with every batch."""
    from rl4co_amd import data as D
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    g = GoldenCase("tsp20_b64_greedy_simple")
    pol = AttentionModelPolicy(env_name="tsp", **{k: v for k, v in g.meta["policy_kwargs"].items() if k != "sdpa_fn_decoder"})
    pol.load_state_dict(g.policy.state_dict())
    pol.train()
    env = get_env("tsp", generator_params=dict(num_loc=20), device="cpu")
    ds = D.TensorDictDataset(TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch]))
    ds = D.wrap_dataset_with_baseline(pol, env, ds, batch_size=24)
    assert pol.training  # the mode is restored
    assert torch.equal(ds.data["extra"], g.reward)  # greedy rewards of the golden run, in dataset order
    batch = next(ds.batches(16))
    assert torch.equal(batch["extra"], g.reward[:16])
