"""GPU parity: reward / env-step / solution-check kernels vs the oracle (bit-exact).

These are integer/byte/index kernels plus the tour-length reduction whose fp32 arithmetic is
restated to be BITWISE equal to ATen's CPU result, so every comparison is exact equality.
"""
import pytest
import torch

from oracle import reference_torch as R
from tests.helpers import clone_td, device_state, make_instances

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from rl4co_amd import kernels

    return kernels


def _random_tours(b, n, gen):
    return torch.stack([torch.randperm(n, generator=gen) for _ in range(b)])


@pytest.mark.parametrize("n", [2, 3, 5, 6, 7, 8, 9, 20, 21, 100, 101, 201, 500, 501, 513, 1001, 2500])
def test_tsp_reward_bit_exact(K, n):
    gen = torch.Generator().manual_seed(n)
    b = 64
    locs = torch.rand(b, n, 2, generator=gen)
    actions = _random_tours(b, n, gen)
    env = R.TSPEnv(n)
    want = env.get_reward({"locs": locs}, actions)
    got = K.tour_length(locs.cuda(), actions.cuda(), prepend_depot=False, negate=True).cpu()
    assert torch.equal(got, want), (got - want).abs().max()


@pytest.mark.parametrize("n_loc,t_extra", [(20, 7), (100, 15), (500, 31), (1000, 40)])
def test_cvrp_reward_bit_exact(K, n_loc, t_extra):
    """CVRP tours: T = customers + depot returns; the depot is prepended (cvrp/env.py:138-147)."""
    gen = torch.Generator().manual_seed(n_loc)
    b = 32
    n = n_loc + 1
    locs = torch.rand(b, n, 2, generator=gen)
    rows = []
    for _ in range(b):
        seq = (torch.randperm(n_loc, generator=gen) + 1).tolist()
        for _ in range(t_extra):  # sprinkle depot returns
            pos = int(torch.randint(0, len(seq) + 1, (1,), generator=gen))
            seq.insert(pos, 0)
        rows.append(torch.tensor(seq))
    actions = torch.stack(rows)
    env = R.CVRPEnv(n_loc, check_solution=False)
    want = env.get_reward({"locs": locs}, actions)
    got = K.tour_length(locs.cuda(), actions.cuda(), prepend_depot=True, negate=True).cpu()
    assert torch.equal(got, want)


def test_reward_multistart_layout(K):
    """s-major batchify: trajectory r reads instance r % B (ops.py:10-28)."""
    gen = torch.Generator().manual_seed(3)
    b, s, n = 16, 5, 50
    locs = torch.rand(b, n, 2, generator=gen)
    actions = _random_tours(b * s, n, gen)
    want = -R.get_tour_length(R.gather_by_index(R.batchify(locs, s), actions))
    got = K.tour_length(locs.cuda(), actions.cuda(), negate=True).cpu()
    assert torch.equal(got, want)


def test_gather_by_index(K):
    gen = torch.Generator().manual_seed(4)
    src = torch.rand(9, 33, 128, generator=gen)
    idx = torch.randint(0, 33, (9, 2), generator=gen)
    want = R.gather_by_index(src, idx)
    got = K.gather_by_index(src.cuda(), idx.cuda()).cpu()
    assert torch.equal(got, want)


def test_tsp_step_matches_oracle(K):
    env, data = make_instances("tsp", 20, 37)
    td = env.reset(clone_td(data))
    st = device_state("tsp", td, "cuda")
    gen = torch.Generator().manual_seed(5)
    err = K.new_error_word("cuda")
    for _ in range(20):
        action = torch.multinomial(td["action_mask"].float(), 1, generator=gen).squeeze(-1)
        td["action"] = action
        td = env.step(td)
        K.tsp_step(action.cuda(), st["action_mask"], st["first_node"], st["current_node"], st["i"], st["done"], err)
        assert torch.equal(st["action_mask"].cpu(), td["action_mask"])
        assert torch.equal(st["first_node"].cpu(), td["first_node"])
        assert torch.equal(st["current_node"].cpu(), td["current_node"])
        assert torch.equal(st["i"].cpu(), td["i"].reshape(-1))
        assert torch.equal(st["done"].cpu(), td["done"].reshape(-1))
    assert td["done"].all() and int(err.item()) == 0


@pytest.mark.parametrize("n_loc", [20, 100])
def test_cvrp_step_and_mask_match_oracle(K, n_loc):
    env, data = make_instances("cvrp", n_loc, 29)
    td = env.reset(clone_td(data))
    st = device_state("cvrp", td, "cuda")
    # a7: the reset mask is get_action_mask on the fresh state
    st["action_mask"].zero_()
    K.cvrp_step(None, st["demand"], st["used_capacity"], st["vehicle_capacity"], st["visited"],
                st["current_node"], st["action_mask"], None)
    assert torch.equal(st["action_mask"].cpu(), td["action_mask"])
    gen = torch.Generator().manual_seed(6)
    err = K.new_error_word("cuda")
    steps = 0
    while not td["done"].all():
        action = torch.multinomial(td["action_mask"].float(), 1, generator=gen).squeeze(-1)
        td["action"] = action
        td = env.step(td)
        K.cvrp_step(action.cuda(), st["demand"], st["used_capacity"], st["vehicle_capacity"], st["visited"],
                    st["current_node"], st["action_mask"], st["done"], err)
        assert torch.equal(st["action_mask"].cpu(), td["action_mask"])
        assert torch.equal(st["used_capacity"].cpu(), td["used_capacity"].reshape(-1))
        assert torch.equal(st["visited"].cpu(), td["visited"])
        assert torch.equal(st["current_node"].cpu(), td["current_node"].reshape(-1))
        assert torch.equal(st["done"].cpu(), td["done"].reshape(-1))
        steps += 1
        assert steps < 4 * n_loc
    assert int(err.item()) == 0


def test_check_solution_flags(K):
    from rl4co_amd import _lib

    gen = torch.Generator().manual_seed(7)
    good = _random_tours(8, 30, gen).cuda()
    err = K.new_error_word("cuda")
    K.tsp_check_solution(good, 30, err)
    assert int(err.item()) == 0
    bad = good.clone()
    bad[3, 5] = bad[3, 6]
    K.tsp_check_solution(bad, 30, err)
    assert int(err.item()) == _lib.EBIT_INVALID_TOUR
    with pytest.raises(AssertionError, match="Invalid tour"):
        K.raise_if_error(err)

    # CVRP: a valid greedy-feasible tour, a duplicated customer, and a capacity violation
    n_loc = 10
    demand = torch.full((2, n_loc), 0.3)
    cap = torch.ones(2)
    ok = torch.tensor([[1, 2, 3, 0, 4, 5, 6, 0, 7, 8, 9, 0, 10, 0]] * 2)
    env = R.CVRPEnv(n_loc)
    env.check_solution_validity({"demand": demand, "vehicle_capacity": cap[:, None]}, ok)
    err = K.new_error_word("cuda")
    K.cvrp_check_solution(ok.cuda(), demand.cuda(), cap.cuda(), err)
    assert int(err.item()) == 0
    over = torch.tensor([[1, 2, 3, 4, 0, 5, 6, 0, 7, 8, 9, 0, 10, 0]] * 2)  # 4 x 0.3 > 1
    K.cvrp_check_solution(over.cuda(), demand.cuda(), cap.cuda(), err)
    assert int(err.item()) == _lib.EBIT_CAPACITY
    err.zero_()
    dup = ok.clone()
    dup[1, 0] = 2
    K.cvrp_check_solution(dup.cuda(), demand.cuda(), cap.cuda(), err)
    assert int(err.item()) & _lib.EBIT_INVALID_TOUR


def test_select_start_nodes(K):
    class E:
        name = "tsp"
        num_loc = 20

    td = {"action_mask": torch.ones(6, 20, dtype=torch.bool)}
    want = R.select_start_nodes(td, E, 7)
    got = K.select_start_nodes(6, 7, 20, False, "cuda").cpu()
    assert torch.equal(got, want)
    E.name = "cvrp"
    want = R.select_start_nodes({"action_mask": torch.ones(6, 21, dtype=torch.bool)}, E, 20)
    got = K.select_start_nodes(6, 20, 20, True, "cuda").cpu()
    assert torch.equal(got, want)


def test_cpu_tensor_fails_loudly(K):
    from rl4co_amd._lib import Rl4coLibraryError

    with pytest.raises(Rl4coLibraryError, match="no CPU fallback"):
        K.tour_length(torch.rand(2, 5, 2), torch.zeros(2, 5, dtype=torch.int64))


# ---------------------------------------------------------------------------------------------
# orienteering problem: env kernels vs the C oracle and vs the restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_loc", [20, 100])
def test_op_env_kernels_match_oracle_and_restatement(K, n_loc):
    from oracle import c_oracle
    from oracle import reference_torch as R

    env = R.get_env("op", n_loc)
    torch.manual_seed(7)
    td = env.reset(env.generate(96))
    b, n = td["action_mask"].shape
    # entry-limit table: HIP == C oracle bit for bit == torch CPU (the reference's own arithmetic)
    torch.manual_seed(7)
    raw = env.generate(96)
    tab_hip = K.op_max_length(td["locs"].cuda(), raw["max_length"].cuda()).cpu()
    tab_c = c_oracle.op_max_length(td["locs"], raw["max_length"])
    assert torch.equal(tab_hip, tab_c) and torch.equal(tab_hip, td["max_length"])
    # random feasible walk: step kernel vs oracle vs restatement, state and mask bit for bit
    st = {k: td[k].clone() for k in ("tour_length", "current_node", "i", "done", "action_mask")}
    st["visited"] = td["visited"].to(torch.uint8)
    hip = {k: v.cuda() for k, v in st.items()}
    hip["current_node"] = hip["current_node"].reshape(-1).contiguous()
    ora = {k: v.clone() for k, v in st.items()}
    ora["current_node"] = ora["current_node"].reshape(-1).contiguous()
    locs_d, ml_d = td["locs"].cuda(), td["max_length"].cuda()
    tdr = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td.items()}
    gen = torch.Generator().manual_seed(3)
    for _ in range(n + 2):
        p = tdr["action_mask"].float()
        action = torch.multinomial(p, 1, generator=gen).squeeze(1)
        tdr["action"] = action
        tdr = env.step(tdr)
        K.op_step(action.cuda(), locs_d, ml_d, hip["tour_length"], hip["visited"], hip["current_node"], hip["i"],
                  hip["action_mask"], hip["done"])
        c_oracle.op_step(action, td["locs"], td["max_length"], ora["tour_length"], ora["visited"], ora["current_node"],
                         ora["i"], ora["action_mask"], ora["done"])
        for k in ("tour_length", "visited", "i", "action_mask", "done"):
            assert torch.equal(hip[k].cpu(), ora[k]), k
        assert torch.equal(hip["action_mask"].cpu(), tdr["action_mask"])
        assert torch.equal(hip["tour_length"].cpu(), tdr["tour_length"])
        assert torch.equal(hip["done"].cpu().bool(), tdr["done"])
    assert bool(tdr["done"].all())


@pytest.mark.parametrize("t", [1, 5, 8, 24, 101, 600])
def test_gather_sum_bit_exact_vs_aten(K, t):
    torch.manual_seed(t)
    prize = torch.rand(64, 101)
    actions = torch.randint(0, 101, (64 * 3, t))
    ref = prize.repeat(3, 1).gather(1, actions).sum(-1)
    got = K.gather_sum(prize.cuda(), actions.cuda()).cpu()
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))


def test_op_check_solution_flags(K):
    from oracle import reference_torch as R
    from rl4co_amd import _lib

    env = R.get_env("op", 20)
    torch.manual_seed(1)
    td = env.reset(env.generate(8))
    locs, ml = td["locs"].cuda(), td["max_length"].cuda()
    ok = torch.zeros(8, 6, dtype=torch.int64)
    ok[:, 0] = 1 + (td["locs"][:, 1:] - td["locs"][:, :1]).norm(dim=-1).argmin(1)  # depot -> nearest customer -> depot
    err = K.new_error_word("cuda")
    K.op_check_solution(ok.cuda(), locs, ml, err)
    assert int(err) == 0
    dup = ok.clone()
    dup[2, 1] = dup[2, 0]
    err = K.new_error_word("cuda")
    K.op_check_solution(dup.cuda(), locs, ml, err)
    assert int(err) & _lib.EBIT_DUPLICATES
    long = torch.arange(1, 21).repeat(8, 1)  # visit everything: far beyond max_length = 2
    err = K.new_error_word("cuda")
    K.op_check_solution(long.cuda(), locs, ml, err)
    assert int(err) & _lib.EBIT_MAX_LENGTH


# ---------------------------------------------------------------------------------------------
# prize-collecting TSP: env kernels vs the C oracle and vs the restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_loc,starts", [(20, 1), (100, 1), (50, 3)])
def test_pctsp_env_kernels_match_oracle_and_restatement(K, n_loc, starts):
    from oracle import c_oracle
    from oracle import reference_torch as R

    env = R.get_env("pctsp", n_loc)
    torch.manual_seed(7)
    td = env.reset(env.generate(96))
    rows = R.batchify({k: v for k, v in td.items() if torch.is_tensor(v)}, starts) if starts > 1 else td
    b, n = rows["action_mask"].shape
    st = {k: rows[k].clone().reshape(b, -1).squeeze(-1) if k != "action_mask" else rows[k].clone()
          for k in ("cur_total_prize", "current_node", "i", "done", "action_mask")}
    st["visited"] = rows["visited"].to(torch.uint8).clone()
    hip = {k: v.cuda() for k, v in st.items()}
    ora = {k: v.clone() for k, v in st.items()}
    rp = td["real_prize"].contiguous()  # instance data [96, N]: trajectory b reads row b % 96
    rp_d = rp.cuda()
    tdr = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in rows.items()}
    gen = torch.Generator().manual_seed(3)
    for _ in range(n):
        action = torch.multinomial(tdr["action_mask"].float(), 1, generator=gen).squeeze(1)
        tdr["action"] = action
        tdr = env.step(tdr)
        K.pctsp_step(action.cuda(), rp_d, hip["cur_total_prize"], hip["visited"], hip["current_node"], hip["i"],
                     hip["action_mask"], hip["done"])
        c_oracle.pctsp_step(action, rp, ora["cur_total_prize"], ora["visited"], ora["current_node"], ora["i"],
                            ora["action_mask"], ora["done"])
        for k in ("cur_total_prize", "visited", "i", "action_mask", "done", "current_node"):
            assert torch.equal(hip[k].cpu(), ora[k]), k
        assert torch.equal(hip["action_mask"].cpu(), tdr["action_mask"])
        assert torch.equal(hip["cur_total_prize"].cpu(), tdr["cur_total_prize"])
        assert torch.equal(hip["done"].cpu().bool(), tdr["done"])
    assert bool(tdr["done"].all())
    # mask-only call (action = NULL) leaves the state alone and reproduces the mask
    before = hip["action_mask"].clone()
    hip["action_mask"].zero_()
    K.pctsp_step(None, rp_d, hip["cur_total_prize"], hip["visited"], hip["current_node"], hip["i"], hip["action_mask"],
                 hip["done"])
    assert torch.equal(hip["action_mask"], before)


def test_pctsp_reward_and_check_solution(K):
    from oracle import reference_torch as R
    from rl4co_amd import _lib
    from tests.helpers import kernel_reward, oracle_reward

    env = R.get_env("pctsp", 20)
    torch.manual_seed(1)
    td = env.reset(env.generate(32))
    # a random feasible rollout through the restatement
    tdr = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td.items()}
    gen = torch.Generator().manual_seed(5)
    acts = []
    while not bool(tdr["done"].all()):
        a = torch.multinomial(tdr["action_mask"].float(), 1, generator=gen).squeeze(1)
        tdr["action"] = a
        tdr = env.step(tdr)
        acts.append(a)
    actions = torch.stack(acts, 1)
    want = env.get_reward(td, actions)  # includes the reference's validity check
    assert torch.equal(oracle_reward("pctsp", td, actions), want)
    assert torch.equal(kernel_reward(K, "pctsp", td, actions).cpu(), want)
    rp = td["real_prize"].cuda()
    err = K.new_error_word("cuda")
    K.pctsp_check_solution(actions.cuda(), rp, err)
    assert int(err) == 0
    dup = actions.clone()
    dup[3, 1] = dup[3, 0]
    err = K.new_error_word("cuda")
    K.pctsp_check_solution(dup.cuda(), rp, err)
    assert int(err) & _lib.EBIT_DUPLICATES
    short = torch.zeros(32, 4, dtype=torch.int64)
    short[:, 0] = 1  # one customer cannot carry a total prize of 1 (prizes are below 4/20)
    err = K.new_error_word("cuda")
    K.pctsp_check_solution(short.cuda(), rp, err)
    assert int(err) == _lib.EBIT_PRIZE
    every = torch.arange(1, 21).repeat(32, 1)  # visiting every customer is always valid
    err = K.new_error_word("cuda")
    K.pctsp_check_solution(every.cuda(), rp, err)
    assert int(err) == 0


# ---------------------------------------------------------------------------------------------
# pickup and delivery: env kernels vs the C oracle and vs the restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_loc,force", [(20, False), (100, False), (50, True)])
def test_pdp_env_kernels_match_oracle_and_restatement(K, n_loc, force):
    from oracle import c_oracle
    from oracle import reference_torch as R
    from tests.helpers import apply_step, rollout_state

    env = R.get_env("pdp", n_loc, force_start_at_depot=force)
    torch.manual_seed(7)
    td = env.reset(env.generate(96))
    ora = rollout_state("pdp", td)
    hip = {k: v.cuda() for k, v in rollout_state("pdp", td).items()}
    tdr = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td.items()}
    gen = torch.Generator().manual_seed(3)
    acts = []
    while not bool(tdr["done"].all()):
        action = torch.multinomial(tdr["action_mask"].float(), 1, generator=gen).squeeze(1)
        acts.append(action)
        tdr["action"] = action
        tdr = env.step(tdr)
        apply_step(K, "pdp", action.cuda(), hip)
        apply_step(c_oracle, "pdp", action, ora)
        for k in ("available", "to_deliver", "i", "action_mask", "done", "current_node"):
            assert torch.equal(hip[k].cpu(), ora[k]), k
        assert torch.equal(hip["action_mask"].cpu(), tdr["action_mask"])
        assert torch.equal(hip["available"].cpu().bool(), tdr["available"])
        assert torch.equal(hip["to_deliver"].cpu().bool(), tdr["to_deliver"])
        assert torch.equal(hip["done"].cpu().bool(), tdr["done"])
    actions = torch.stack(acts, 1)
    assert actions.shape[1] == n_loc + int(force)
    want = env.get_reward(td, actions)  # includes the reference's validity check
    got = K.tour_length(td["locs"].cuda(), actions.cuda(), prepend_depot=True, negate=True).cpu()
    assert torch.equal(got, want)
    # mask-only call
    before = hip["action_mask"].clone()
    hip["action_mask"].fill_(1)
    K.pdp_step(None, hip["available"], hip["to_deliver"], hip["current_node"], hip["i"], hip["action_mask"], hip["done"])
    assert torch.equal(hip["action_mask"], before)
    # validity kernel: the reference's three assertions, in its order
    from rl4co_amd import _lib

    def bits(a):
        err = K.new_error_word("cuda")
        K.pdp_check_solution(a.cuda().contiguous(), n_loc + 1, force, err)
        return int(err)

    assert bits(actions) == 0
    dup = actions.clone()
    dup[5, -1] = dup[5, -2]
    assert bits(dup) == _lib.EBIT_NOT_ALL_NODES
    swap = actions.clone()  # deliver before picking up: exchange a pickup with its delivery in one tour
    row = swap[7]
    p = int(row[row.ne(0) & row.le(n_loc // 2)][0])
    ip, idl = int((row == p).nonzero()), int((row == p + n_loc // 2).nonzero())
    row[ip], row[idl] = p + n_loc // 2, p
    assert bits(swap) == _lib.EBIT_NO_PICKUP
    for bad, bit in ((dup, "Not visiting all nodes"), (swap, "Deliverying without pick-up")):
        with pytest.raises(AssertionError, match=bit):
            env.check_solution_validity(td, bad)
    if force:
        mid = actions.clone()  # the depot moved from the first position into the tour
        mid[3, 0], mid[3, 4] = actions[3, 4], 0
        assert bits(mid) & _lib.EBIT_DEPOT_MIDDLE
        with pytest.raises(AssertionError, match="Going back to depot"):
            env.check_solution_validity(td, mid)


# ---------------------------------------------------------------------------------------------
# CVRP with time windows: env kernels vs the C oracle and vs the restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_loc,starts", [(20, 1), (100, 1), (50, 2)])
def test_cvrptw_env_kernels_match_oracle_and_restatement(K, n_loc, starts):
    from oracle import c_oracle
    from oracle import reference_torch as R
    from rl4co_amd import _lib
    from tests.helpers import apply_step, rollout_state

    env = R.get_env("cvrptw", n_loc)
    torch.manual_seed(7)
    td = env.reset(env.generate(64))
    ora = rollout_state("cvrptw", td, num_starts=starts if starts > 1 else 0)
    hip = {k: v.cuda() for k, v in rollout_state("cvrptw", td, num_starts=starts if starts > 1 else 0).items()}
    rows = R.batchify({k: v for k, v in td.items() if torch.is_tensor(v)}, starts) if starts > 1 else td
    tdr = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in rows.items()}
    gen = torch.Generator().manual_seed(3)
    acts = []
    for _ in range(2 * (n_loc + 1)):
        if bool(tdr["done"].all()):
            break
        action = torch.multinomial(tdr["action_mask"].float(), 1, generator=gen).squeeze(1)
        acts.append(action)
        tdr["action"] = action
        tdr = env.step(tdr)
        apply_step(K, "cvrptw", action.cuda(), hip)
        apply_step(c_oracle, "cvrptw", action, ora)
        for k in ("current_time", "used_capacity", "visited", "action_mask", "done", "current_node"):
            assert torch.equal(hip[k].cpu(), ora[k]), k
        assert torch.equal(hip["action_mask"].cpu(), tdr["action_mask"])
        assert torch.equal(hip["current_time"].cpu(), tdr["current_time"].reshape(-1))
        assert torch.equal(hip["done"].cpu().bool(), tdr["done"].reshape(-1))
    assert bool(tdr["done"].all())
    actions = torch.stack(acts, 1)
    want = env.get_reward(rows, actions)  # includes the reference's CVRP + time-window validity checks
    got = K.tour_length(td["locs"].cuda(), actions.cuda(), prepend_depot=True, negate=True).cpu()
    assert torch.equal(got, want)
    # mask-only call
    before = hip["action_mask"].clone()
    hip["action_mask"].fill_(1)
    K.cvrptw_step(None, hip["demand"], hip["locs"], hip["time_windows"], hip["durations"], hip["used_capacity"],
                  hip["vehicle_capacity"], hip["current_time"], hip["visited"], hip["current_node"], hip["action_mask"], None)
    assert torch.equal(hip["action_mask"], before)

    # the window part of the validity check
    def bits(a, tw=hip["time_windows"], dur=hip["durations"]):
        err = K.new_error_word("cuda")
        K.cvrptw_check_solution(a.cuda().contiguous(), hip["locs"], tw, dur, err)
        return int(err)

    padded = torch.cat([actions, torch.zeros(actions.shape[0], 5, dtype=torch.int64)], 1)
    assert bits(actions) == 0 and bits(padded) == 0
    late = actions.clone()  # two customers exchanged: with windows this tight some deadline is missed somewhere in the batch
    late[:, [0, 1]] = late[:, [1, 0]]
    assert (bits(late) & _lib.EBIT_TW_DEADLINE != 0) == ("deadline" in _first_tw_failure(R, rows, late))
    neg = hip["durations"].clone()
    neg[3, 4] = -1.0
    assert bits(actions, dur=neg) & _lib.EBIT_TW_DURATION
    empty = hip["time_windows"].clone()
    empty[2, 5, 1] = empty[2, 5, 0]
    assert bits(actions, tw=empty) & _lib.EBIT_TW_EMPTY


def _first_tw_failure(R, rows, actions) -> str:
    """Message of the reference's time-window assertions on `actions`, with its CVRP part switched off."""
    orig = R.CVRPEnv.__dict__["check_solution_validity"]
    R.CVRPEnv.check_solution_validity = staticmethod(lambda td, a: None)
    try:
        R.CVRPTWEnv.check_solution_validity(rows, actions)
        return ""
    except AssertionError as e:
        return str(e)
    finally:
        R.CVRPEnv.check_solution_validity = orig
