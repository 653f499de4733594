"""GPU (`-m gpu`): the BENCHMARKED configurations against the reference's rollouts on TRAINED weights.

Fixtures: tests/golden/trained/*.npz — the reference's own policy class (verbatim source) loaded with the weight sets
under tests/golden/weights (trained by the product, tools/train_sharp.py) and rolled out on the CPU in fp32 and under
bf16 autocast at the full size of BASELINE configs[1], [2] and [4] (oracle/gen_trained_golden.py). With sharp weights
the greedy decisions are no longer all near-ties, so tour-level identity is a meaningful, asserted number:

* fp32 planes (the parity configuration): the few flips are PROVEN near-ties — the product, held on the reference's
  trajectory, rates the reference's choice within `FP32_FLIP_REGRET` of its own best at the first divergent step;
* bf16 (MFMA encoder + bf16 planes, what bench.py times) against the reference's own bf16-autocast run: a floor on the
  share of identical tours, on the per-decision agreement, and a ceiling on the mean-reward gap. The reference's two
  regimes agree with EACH OTHER on 16 % of the tours (661 / 4096 at C2, MANIFEST `reference_bf16_vs_fp32_identical`):
  that is the scale an independent bf16 pipeline can reach, not 100 %.

Measured figures are written to gpurun_out/parity_measured.json (copied into profiles/ per round).
"""
import json
import os

import pytest
import torch

from trained_parity import TrainedCase, compare, compare_augmented

pytestmark = pytest.mark.gpu

# fp32: a flip must be a near-tie of the product's own log-probs — fp32 round-off of logits up to |10| summed in a
# different order (one ulp at 10 is 9.5e-7). Measured on MI355X (r03, profiles/r03_parity_measured.json): C2 and C3
# reproduce ALL 4096 reference tours (0 flips, also with fold=False); C5 (501 nodes, 726 steps, weights trained on 100
# nodes) 6 of 1024 with regrets up to 3.0e-5; the tanh-plateau case 5 of 1024 at 9.5e-7 (knee decisions).
FP32_FLIP_REGRET = {"t2_tsp100_b4096_greedy": 1e-5, "t3_cvrp100_b4096_greedy": 1e-5, "t5_cvrp500_b1024_greedy": 6e-5}
FP32_FLIP_BUDGET = 0.01  # share of tours


def _record(key, value):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_measured.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("name", ["t2_tsp100_b4096_greedy", "t3_cvrp100_b4096_greedy", "t5_cvrp500_b1024_greedy"])
def test_fp32_trained_tours_flip_only_at_proven_near_ties(name):
    case = TrainedCase(name)
    rec = compare(case, "fp32", "cuda", against="fp32")
    _record(f"trained/{name}/fp32", rec)
    print(name, "fp32:", rec)
    assert rec["rewards_bit_identical_on_identical"] is True
    assert rec["flips"] <= FP32_FLIP_BUDGET * rec["of"]
    assert rec["flip_regret_max"] <= FP32_FLIP_REGRET[name], "a flipped tour left the reference's at a decision that was NOT a near-tie"
    assert rec["reward_rel_gap"] <= (1e-5 if rec["flips"] == 0 else 2e-4)
    assert rec["step_agreement"] >= 0.9999


def test_fp32_reference_association_on_trained_weights():
    """fold=False (the reference's own association of the decoder) on the trained TSP weights."""
    case = TrainedCase("t2_tsp100_b4096_greedy")
    rec = compare(case, "fp32_fold_off", "cuda", against="fp32")
    _record("trained/t2_tsp100_b4096_greedy/fp32_fold_off", rec)
    print("fold off:", rec)
    assert rec["flips"] <= FP32_FLIP_BUDGET * rec["of"] and rec["flip_regret_max"] <= 1e-5


# bf16 (the benchmarked configuration) against the reference's own bf16-autocast run AND its fp32 run. Measured on MI355X
# (r03, profiles/r03_parity_measured.json), floors / ceilings = those measurements with margin:
#   case                     vs reference bf16-autocast                 vs reference fp32                  reference bf16 vs its own fp32
#   C2 TSP-100 x 4096        556 tours (13.6 %), 98.0 % of decisions,   1660 tours (40.5 %), 99.1 %,       661 tours (16.1 %)
#                            mean reward within 4.6e-4                  within 3.0e-4
#   C3 CVRP-100 x 4096       561 (13.7 %), 98.2 %, 5.5e-4               1506 (36.8 %), 99.1 %, 1.5e-4      675 (16.5 %)
#   C5 CVRP-500 x 1024       0 (720 steps), 89.9 %, 1.6e-2              0, 95.6 %, 5.3e-3                  0
# i.e. the product's bf16 pipeline (bf16 operands, fp32 accumulation and fp32 decode arithmetic) stays closer to the fp32
# reference than the reference's own autocast run does; two independent bf16 pipelines agree on ~14 % of 100-step tours.
BF16 = {
    #                           vs bf16-autocast: tours, decisions, reward gap | vs fp32: tours, decisions, reward gap
    "t2_tsp100_b4096_greedy": ((0.10, 0.97, 1e-3), (0.33, 0.985, 6e-4)),
    "t3_cvrp100_b4096_greedy": ((0.10, 0.97, 1.2e-3), (0.30, 0.985, 5e-4)),
    "t5_cvrp500_b1024_greedy": ((0.0, 0.87, 3e-2), (0.0, 0.94, 1.2e-2)),
}


@pytest.mark.parametrize("name", sorted(BF16))
def test_bf16_benchmarked_configuration_vs_reference_bf16_autocast(name):
    case = TrainedCase(name)
    rec = compare(case, "bf16", "cuda", against="bf16")
    vs32 = compare(case, "bf16", "cuda", against="fp32")
    _record(f"trained/{name}/bf16_vs_ref_bf16", rec)
    _record(f"trained/{name}/bf16_vs_ref_fp32", vs32)
    print(name, "bf16 vs reference bf16-autocast:", rec)
    print(name, "bf16 vs reference fp32:", vs32)
    for got, (floor_same, floor_agree, gap) in ((rec, BF16[name][0]), (vs32, BF16[name][1])):
        assert got["identical_frac"] >= floor_same
        if got["identical"]:
            assert got["rewards_bit_identical_on_identical"] is True
        assert got["step_agreement"] >= floor_agree
        assert got["reward_rel_gap"] <= gap
    # closer to the fp32 reference than to an independent bf16 pipeline
    assert vs32["step_agreement"] >= rec["step_agreement"]
    ref_self = case.meta["reference_bf16_vs_fp32_identical"] / case.batch
    assert vs32["identical_frac"] >= ref_self, "the product's bf16 run is further from the fp32 reference than the reference's own"


@pytest.mark.parametrize("name", ["sharpkl100_tsp100_b1024_greedy", "sharpkl400_tsp100_b512_greedy"])
def test_tanh_plateau_ties_resolve_like_the_reference(name):
    """Logit key scaled until the logits saturate the tanh clip: exact ties at +-10, lowest index wins."""
    case = TrainedCase(name)
    rec = compare(case, "fp32", "cuda", against="fp32")
    _record(f"trained/{name}/fp32", rec)
    print(name, rec)
    assert rec["flips"] <= max(2, 0.01 * rec["of"]) and rec["flip_regret_max"] <= 2e-5


@pytest.mark.parametrize("name", ["t2_tsp100_b4096_sampling", "t5_cvrp500_b1024_sampling"])
def test_fixed_seed_sampling_on_trained_weights(name):
    """The reference's seeded multinomial stream drives the kernel (C5 at the benchmarked batch of 1024)."""
    case = TrainedCase(name)
    rec = compare(case, "fp32", "cuda", against="fp32", decode="sampling", regret=False)
    _record(f"trained/{name}/fp32", rec)
    print(name, rec)
    assert rec["flips"] <= FP32_FLIP_BUDGET * rec["of"]
    assert rec["rewards_bit_identical_on_identical"] is True
    assert rec["reward_rel_gap"] <= (1e-5 if rec["flips"] == 0 else 1e-4)


@pytest.mark.parametrize("name", ["t2_tsp100_b4096_greedy", "t3_cvrp100_b4096_greedy"])
def test_fp16_configuration_vs_reference_fp32(name):
    """The reference's DEFAULT precision ("16-mixed" = fp16 autocast, utils/trainer.py:57) on the fp16 kernels (fused
    encoder on v_mfma_f32_32x32x16_f16, fp16 planes, fp32 decode arithmetic) against the reference's fp32 tours: with 11
    significant bits against bf16's 8 it must stay at least as close to fp32 as the bf16 configuration does."""
    case = TrainedCase(name)
    rec = compare(case, "fp16", "cuda", against="fp32")
    _record(f"trained/{name}/fp16_vs_ref_fp32", rec)
    print(name, "fp16 vs reference fp32:", rec)
    floor_same, floor_agree, gap = BF16[name][1]
    assert rec["identical_frac"] >= floor_same and rec["step_agreement"] >= floor_agree and rec["reward_rel_gap"] <= gap
    assert rec["rewards_bit_identical_on_identical"] is True


# BASELINE configs[3]'s policy — POMO (6 layers, instance norm, no graph context), trained by the product (tools/train_sharp.py
# --arch pomo) and loaded into the reference's own class — at its evaluation protocol (zoo/pomo/model.py:99-140): a greedy
# rollout from each of the 100 start nodes of 256 instances, and the best of 8 dihedral augmentations x 100 starts
# (204 800 rollouts). Measured on MI355X (r03, profiles/r03_parity_measured.json):
#   fp32   25 597 / 25 600 tours identical, best-of-starts reward bit-identical on 256 / 256 instances; augmented: 204 780 /
#          204 800 rollout rewards, 2047 / 2048 per-augmentation maxima and 255 / 256 final maxima bit-identical
#   bf16   53.4 % of the fp32 reference's tours, 17.0 % of its bf16-autocast run's (the reference's two regimes agree on
#          18.6 %); augmented best reward within 2.7e-5
#   fp16   92.1 % of the fp32 reference's tours; augmented best reward within 1.7e-4
POMO = "t4_pomo_tsp100_b256_msgreedy"


def test_pomo_multistart_fp32_reproduces_the_reference_on_trained_weights():
    case = TrainedCase(POMO)
    rec = compare(case, "fp32", "cuda", against="fp32", decode="multistart_greedy", regret=False)
    _record(f"trained/{POMO}/fp32", rec)
    print("pomo fp32:", rec)
    assert rec["of"] == case.batch * case.num_starts == 25600
    assert rec["rewards_bit_identical_on_identical"] is True
    assert rec["flips"] <= 0.001 * rec["of"]
    assert rec["best_of_starts_bit_identical"] >= case.batch - 1 and rec["reward_rel_gap"] <= 1e-6


def test_pomo_augmented_best_of_fp32_matches_the_reference_epilogue():
    """rl4co_amd.data.pomo_evaluate (augmentation kernel, multistart rollouts, fused best-of) against the reference's
    StateAugmentation + POMO.shared_step maxima."""
    case = TrainedCase(POMO)
    rec = compare_augmented(case, "fp32", "cuda")
    _record(f"trained/{POMO}/augmented_fp32", rec)
    print("pomo augmented fp32:", rec)
    assert rec["of_rollouts"] == 8 * 100 * 256
    assert rec["rollout_rewards_identical_frac"] >= 0.999
    assert rec["max_reward_bit_identical"] >= rec["of_max_reward"] - 4
    assert rec["max_aug_reward_bit_identical"] >= rec["instances"] - 3 and rec["max_aug_reward_rel_gap"] <= 2e-5


@pytest.mark.parametrize("config,floor_fp32,floor_bf16,gap,aug_gap", [("bf16", 0.45, 0.13, 5e-4, 1.5e-4), ("fp16", 0.88, None, 1e-4, 6e-4)])
def test_pomo_16bit_configurations_on_trained_weights(config, floor_fp32, floor_bf16, gap, aug_gap):
    case = TrainedCase(POMO)
    rec = compare(case, config, "cuda", against="fp32", decode="multistart_greedy", regret=False)
    _record(f"trained/{POMO}/{config}_vs_fp32", rec)
    assert rec["rewards_bit_identical_on_identical"] is True
    assert rec["identical_frac"] >= floor_fp32 and rec["reward_rel_gap"] <= gap
    if floor_bf16 is not None:
        r16 = compare(case, config, "cuda", against="bf16", decode="multistart_greedy", regret=False)
        _record(f"trained/{POMO}/{config}_vs_bf16", r16)
        assert r16["identical_frac"] >= floor_bf16 and r16["reward_rel_gap"] <= 1e-3
    aug = compare_augmented(case, config, "cuda")
    _record(f"trained/{POMO}/augmented_{config}", aug)
    print(f"pomo {config}:", rec, aug)
    assert aug["max_aug_reward_rel_gap"] <= aug_gap
