"""GPU (`-m gpu`): rows N2 / N3 of SURVEY.md §8f on the device, against fixtures the REAL reference produced
(oracle/gen_aug_golden.py; generated in the build container, committed under tests/golden/).

N3: state augmentation (rl4co/data/transforms.py: dihedral-8 and the symmetric / SymNCO group) and the POMO
    evaluation epilogue (zoo/pomo/model.py:112-140: best start per augmentation, best augmentation per instance)
    around the fused rollout. Integer / gather work is exact; coordinates are compared at fp32 rounding (sin / cos
    on the device differ from the host's in the last ulp); rewards at the greedy-parity bar of the decode tests.
N2: instance files with the reference's npz schema loaded straight into HBM, served batch by batch from the device,
    rolled out greedily (RolloutBaseline.rollout / wrap_dataset): rewards equal the reference's on its own file.
"""
import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _td(d, b):
    from rl4co_amd.tensordict import TensorDict

    return TensorDict(d, batch_size=[b])


def test_dihedral8_on_device_is_exact():
    """rl4co_augment_dihedral8_f32 == its host restatement (which tests/test_data_cpu.py holds against the reference
    source) bit for bit, at the full C2 batch as well; block a of the output is the a-th symmetry of the square."""
    from oracle import c_oracle
    from rl4co_amd import kernels as K
    from rl4co_amd.data import StateAugmentation

    torch.manual_seed(0)
    for b, n in ((64, 100), (4096, 100), (3, 1), (17, 501)):
        locs = torch.rand(b, n, 2)
        dev = StateAugmentation(8, augment_fn="dihedral8")(_td({"locs": locs.cuda(), "other": torch.arange(b).cuda()}, b))
        assert dev["locs"].is_cuda and torch.equal(dev["locs"].cpu(), c_oracle.augment_dihedral8(locs))
        assert torch.equal(dev["locs"][:b].cpu(), locs)                                  # identity block
        assert torch.equal(dev["locs"][4 * b: 5 * b].cpu(), locs.flip(-1))               # (y, x)
        assert torch.equal(dev["locs"][3 * b: 4 * b].cpu(), 1 - locs)                    # (1 - x, 1 - y)
        assert torch.equal(dev["other"].cpu(), torch.arange(b).repeat(8))                # other keys: tiled copy-major
        assert torch.equal(K.augment_dihedral8(locs.cuda()), dev["locs"])


def test_symmetric_kernel_and_pomo_best_kernel_equal_their_host_restatements():
    """Same (cos, sin, swap) in -> same bits out (the arithmetic order is part of the contract: no fma); the fused
    best-of reduction + action gathers against oracle_pomo_best on rewards full of exact ties."""
    from oracle import c_oracle
    from rl4co_amd import kernels as K

    torch.manual_seed(1)
    for b, a, n in ((32, 8, 50), (5, 3, 7), (4096, 8, 100)):
        xy = torch.rand(b, n, 2)
        phi = torch.rand(a * b) * 4 * 3.141592653589793
        c, s_, sw = torch.cos(phi), torch.sin(phi), phi > 2 * 3.141592653589793
        want = c_oracle.augment_symmetric(xy, c, s_, sw)
        got = K.augment_symmetric(xy.cuda(), c.cuda(), s_.cuda(), sw.cuda())
        assert torch.equal(got.cpu(), want)
    for a, s, b, t in ((8, 6, 16, 20), (1, 7, 5, 9), (8, 1, 4, 3), (8, 100, 512, 100), (70, 2, 3, 4)):
        reward = torch.randint(0, 4, (s * a * b,)).float() * -0.5
        if s * a * b > 40:  # NaN rewards propagate like torch.max's
            reward[torch.randperm(s * a * b)[:5]] = float("nan")
        actions = torch.randint(0, 100, (s * a * b, t))
        want = c_oracle.pomo_best(reward, actions, a, s)
        got = K.pomo_best(reward.cuda(), actions.cuda(), a, s)
        assert sorted(got) == sorted(want)
        for k in want:
            g, w_ = got[k].cpu(), want[k]
            if g.is_floating_point():
                g, w_ = g.nan_to_num(7.0), w_.nan_to_num(7.0)
            assert torch.equal(g, w_), (k, a, s, b, t)
        lean = K.pomo_best(reward.cuda(), None, a, s)
        assert torch.equal(lean["max_aug_reward"].cpu().nan_to_num(7.0), want["max_aug_reward"].nan_to_num(7.0)) and "best_aug_actions" not in lean


@pytest.mark.parametrize("tag,kw", [("a8", dict(num_augment=8)),
                                    ("a4_noident_norm", dict(num_augment=4, first_aug_identity=False, normalize=True))])
def test_symmetric_augmentation_on_device_matches_reference(tag, kw):
    """The reference's own draw of angles (re-created from its seed by the generator script) fed to the product on the
    GPU: the augmented coordinates equal the reference's CPU result to fp32 rounding; isometry holds exactly as there."""
    from rl4co_amd.data import StateAugmentation, symmetric_augmentation

    z = np.load(GOLDEN_DIR / "n3_symmetric_tsp50.npz")
    locs, phi, want = torch.from_numpy(z["locs"]), torch.from_numpy(z[f"phi_{tag}"]), torch.from_numpy(z[f"aug_{tag}"])
    fn = lambda xy, n: symmetric_augmentation(xy, n, phi=phi)  # noqa: E731  (callable augment_fn: transforms.py:94-97)
    got = StateAugmentation(augment_fn=fn, **kw)(_td({"locs": locs.cuda()}, 32))["locs"]
    assert got.is_cuda and got.shape == want.shape
    torch.testing.assert_close(got.cpu(), want, rtol=0, atol=2e-6)
    if tag == "a8":
        assert torch.equal(got[:32].cpu(), locs)  # first block: phi = 0, the identity, bit for bit
        pd = lambda a: (a[:, :, None, :] - a[:, None, :, :]).norm(dim=-1)  # noqa: E731
        torch.testing.assert_close(pd(got[32:64]), pd(locs.cuda()), rtol=0, atol=2e-6)
    # the unseeded path draws on the device and is still a valid member of the group
    free = StateAugmentation(num_augment=8)(_td({"locs": locs.cuda()}, 32))["locs"]
    assert torch.equal(free[:32].cpu(), locs) and not torch.allclose(free[32:64].cpu(), locs)


def test_pomo_evaluate_on_device_matches_reference_epilogue():
    """Dihedral-8 x 6 starts on TSP-20 through the fused rollout (fp32 parity configuration): per-(augmentation, start)
    rewards, best start, best augmentation and the selected tours against the reference's (greedy near-tie flips
    bounded as in the decode tests; everything downstream of identical rollouts is exact gather / max work)."""
    from rl4co_amd.data import pomo_evaluate
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    z = np.load(GOLDEN_DIR / "n3_pomo_eval_tsp20.npz")
    n_aug, n_start = int(z["n_aug"]), int(z["n_start"])
    locs = torch.from_numpy(z["locs"])
    b = locs.shape[0]
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False).eval().cuda()
    env = get_env("tsp", generator_params=dict(num_loc=20, device="cuda"), device="cuda")
    with torch.inference_mode():
        out = pomo_evaluate(pol, env, env.reset(_td({"locs": locs.cuda()}, b)), num_augment=n_aug, num_starts=n_start)
    reward = out["reward_per_aug_start"].cpu()
    want = torch.from_numpy(z["reward"])
    assert reward.shape == want.shape == (b, n_aug, n_start)
    flat_same = (out["actions"].reshape(b, n_aug, n_start, -1).cpu() ==
                 torch.from_numpy(z["flat_actions"].astype(np.int64)).view(n_start, n_aug, b, -1).permute(2, 1, 0, 3)).all(-1)  # rows: s-major
    assert int((~flat_same).sum()) <= max(1, flat_same.numel() // 50), f"{int((~flat_same).sum())} of {flat_same.numel()} rollouts differ"
    assert torch.equal(reward[flat_same], want[flat_same])  # identical tours: bit-identical rewards
    inst_ok = flat_same.all(-1).all(-1)                      # instances whose 48 rollouts all coincide
    assert torch.equal(out["max_reward"].cpu()[inst_ok], torch.from_numpy(z["max_reward"])[inst_ok])
    assert torch.equal(out["max_aug_reward"].cpu()[inst_ok], torch.from_numpy(z["max_aug_reward"])[inst_ok])
    assert torch.equal(out["best_aug_actions"].cpu()[inst_ok], torch.from_numpy(z["best_aug_actions"].astype(np.int64))[inst_ok])
    torch.testing.assert_close(out["max_aug_reward"].cpu().mean(), torch.from_numpy(z["max_aug_reward"]).mean(), rtol=1e-3, atol=0)
    # the selected tour really has the selected reward on the ORIGINAL coordinates (symmetries preserve lengths)
    from rl4co_amd import kernels as K

    length = K.tour_length(locs.cuda(), out["best_aug_actions"].contiguous(), negate=True)
    torch.testing.assert_close(length, out["max_aug_reward"], rtol=1e-5, atol=1e-5)


def test_instance_file_to_device_dataset_and_greedy_baseline(tmp_path):
    """The reference's own vrp20 validation file (generate_data.py, seed 4321; arrays stored in the fixture): written
    with our generator -> identical bytes per array; loaded straight to the GPU; `env.load_data` normalises the demand
    as cvrp/env.py:180-186; the device-resident dataset serves batches without leaving HBM; the greedy rollout over it
    (RolloutBaseline.rollout) gives the reference's rewards; `wrap_dataset` attaches them as `extra`."""
    from rl4co_amd.data import (TensorDictDataset, generate_dataset, greedy_rollout_rewards, load_npz_to_tensordict,
                                wrap_dataset_with_baseline)
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    z = np.load(GOLDEN_DIR / "n2_vrp20_val.npz")
    fn = str(tmp_path / "vrp20_val.npz")
    generate_dataset(filename=fn, problem="vrp", dataset_size=64, graph_sizes=[20], seed=4321)
    mine = np.load(fn)
    for k in ("depot", "locs", "demand", "capacity"):
        assert np.array_equal(mine[k], z[f"file_{k}"]), k
    raw = load_npz_to_tensordict(fn, device="cuda")
    assert all(v.is_cuda for v in raw.values()) and raw.batch_size[0] == 64
    env = get_env("cvrp", generator_params=dict(num_loc=20, device="cuda"), device="cuda")
    td_all = env.load_data(fn, device="cuda")
    assert td_all["demand"].is_cuda and float(td_all["demand"].max()) <= 9 / 30 + 1e-6
    ds = TensorDictDataset(td_all)
    batches = list(ds.batches(24))
    assert [bt.batch_size[0] for bt in batches] == [24, 24, 16] and all(bt["locs"].is_cuda for bt in batches)
    torch.manual_seed(0)
    pol = AttentionModelPolicy("cvrp").eval().cuda()
    rewards = greedy_rollout_rewards(pol, env, ds, batch_size=24)
    want = torch.from_numpy(z["greedy_reward"])
    same = (rewards.cpu() == want)
    assert int((~same).sum()) <= 1, f"{int((~same).sum())} of 64 greedy rewards differ from the reference's"
    wrapped = wrap_dataset_with_baseline(pol, env, ds, batch_size=32)
    assert torch.equal(wrapped.data["extra"], rewards)
    first = next(iter(wrapped.batches(8)))
    assert first["extra"].shape == (8,) and first["extra"].is_cuda


def test_instance_generator_kernel_equals_host_restatement_and_is_uniform():
    """rl4co_uniform_f32 (instances drawn straight into HBM): bit-identical to its host restatement
    (oracle_uniform_f32: same Philox4x32-10 words), statistically U(low, high) like the reference's sampler
    (envs/common/utils.py:34-62), reproducible under torch.manual_seed, and CVRP's demands are the integers 1 .. 9 over
    the capacity (cvrp/generator.py:127-136)."""
    from oracle import c_oracle
    from rl4co_amd import kernels as K
    from rl4co_amd.envs import get_env

    for shape, low, high, seed, sid in [((4096, 100, 2), 0.0, 1.0, 1234567, 0), ((7, 3), -2.0, 5.0, 99, 3), ((1,), 0.0, 1.0, 1, 0),
                                        ((513, 101), 0.0, 9.0, 2**61 + 17, 1)]:
        dev = K.uniform(shape, low, high, seed, sid, "cuda")
        assert torch.equal(dev.cpu(), c_oracle.uniform(shape, low, high, seed, sid))
        assert float(dev.min()) >= low and float(dev.max()) < high
    dem = K.uniform((2048, 100), 0.0, 9.0, 5, 0, "cuda", demand_capacity=50.0)
    assert torch.equal(dem.cpu(), c_oracle.uniform((2048, 100), 0.0, 9.0, 5, 0, demand_capacity=50.0))
    vals = (dem.cpu() * 50.0).round()  # (on the host: ATen's GPU division by a scalar multiplies by the reciprocal)
    assert set(vals.unique().tolist()) == {float(i) for i in range(1, 10)} and torch.equal(vals / 50.0, dem.cpu())
    x = K.uniform((1 << 22,), 0.0, 1.0, 42, 0, "cuda")
    assert abs(float(x.mean()) - 0.5) < 1e-3 and abs(float(x.var()) - 1.0 / 12.0) < 1e-3
    hist = torch.histc(x, bins=64, min=0.0, max=1.0)
    assert float((hist - hist.mean()).abs().max()) < 6.0 * float(hist.mean()) ** 0.5  # every bin within 6 sigma
    env = get_env("cvrp", generator_params=dict(num_loc=100, device="cuda"), device="cuda")
    torch.manual_seed(7)
    a = env.generator(batch_size=[64])
    torch.manual_seed(7)
    b = env.generator(batch_size=[64])
    c = env.generator(batch_size=[64])
    assert torch.equal(a["locs"], b["locs"]) and torch.equal(a["demand"], b["demand"]) and not torch.equal(a["locs"], c["locs"])
    assert a["locs"].shape == (64, 100, 2) and a["depot"].shape == (64, 2) and a["locs"].is_cuda
    assert float(a["demand"].min()) >= 1 / 50 - 1e-7 and float(a["demand"].max()) <= 9 / 50 + 1e-7
