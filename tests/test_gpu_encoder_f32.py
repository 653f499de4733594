"""GPU: the exact-fp32 fused encoder + cache fold (csrc/am_encoder_f32.hip, v_mfma_f32_16x16x4_f32) vs the ORACLE's encoder.

This is the encoder of the bit-identical configuration (no autocast: fp32 encoder, fp32 planes, fp32 decode arithmetic).
Floating point => tolerance test, tolerance stated here: fp32 operands, fp32 accumulation (a k-ordered fmaf chain), so
the only difference to the reference's CPU run is the ORDER of fp32 sums — every output must be within 3e-6 relative
Frobenius error of the oracle restatement's CPU fp32 encoder (oracle/reference_torch.py, pinned bit for bit to the
reference's source; cache rows from its embeddings in float64), and no worse than 3x what torch's own fp32 GPU path
(rocBLAS / SDPA, the encoder this kernel replaces on the parity path) makes on the same inputs. Tour-level parity on
trained weights at the BASELINE sizes: tests/test_gpu_trained_parity.py (its "fp32" configuration runs on this kernel).
"""
import pytest
import torch

from tests.helpers import GoldenCase
from tests.test_gpu_encoder import _oracle_reference, _perturb_norm_stats, _policy, _rel, _td

pytestmark = pytest.mark.gpu

REL_TOL = 3e-6


def _torch_fp32(pol, td, fold=True):
    with torch.inference_mode():
        h, _ = pol.encoder(td)
        return h, pol.decoder.precompute_cache(h, torch.float32, torch.float32, fold=fold)


@pytest.mark.parametrize("name", ["tsp20_b64_greedy_simple", "tsp50_b64_greedy", "tsp100_b64_greedy", "cvrp20_b128_greedy",
                                  "cvrp100_b64_greedy", "pomo_tsp50_b8_mssampling", "pomo_cvrp20_b16_msgreedy"])
def test_fp32_encoder_matches_the_oracle_encoder(name):
    g = GoldenCase(name)
    pol = _policy(g, cache_dtype=torch.float32)
    _perturb_norm_stats(pol)
    env, td = _td(g)
    packed = pol._packed_encoder()
    assert packed.supported(td)
    with torch.inference_mode():
        cache, hidden = packed.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32)
    torch.cuda.synchronize()
    h_t, ref = _torch_fp32(pol, td)
    want = _oracle_reference(g, pol)
    checks = {"hidden": (hidden, want["hidden"], h_t)}
    for i, nm in enumerate(("glimpse_key", "glimpse_val", "logit_key")):
        checks[nm] = (cache.kvl[i], want[nm], ref.kvl[i])
    checks["ctx_cur"] = (cache.ctx_cur, want["ctx_cur"], ref.ctx_cur)
    if g.env_name == "tsp":
        checks["ctx_first"] = (cache.ctx_first, want["ctx_first"], ref.ctx_first)
    if "q_bias" in want:
        checks["q_bias"] = (cache.q_bias, want["q_bias"], ref.q_bias)
    else:
        assert cache.q_bias is None
    for nm, (got, wanted, torch_gpu) in checks.items():
        assert got.dtype == torch.float32 and torch.isfinite(got).all(), nm
        e_kernel, e_torch = _rel(got, wanted), _rel(torch_gpu, wanted)
        assert e_kernel <= REL_TOL, f"{nm}: kernel rel err {e_kernel:.3e}"
        assert e_kernel <= 3 * e_torch + 2e-7, f"{nm}: kernel {e_kernel:.3e} vs torch fp32 on the GPU {e_torch:.3e}"
    if g.env_name == "tsp":
        assert torch.equal(cache.q_step0, ref.q_step0)
    else:
        assert torch.equal(cache.w_cap, ref.w_cap)


@pytest.mark.parametrize("name", ["tsp50_b64_greedy", "cvrp20_b128_greedy"])
def test_fp32_encoder_normalisation_epsilon_matters(name):
    """Running variances of 2e-5 .. 2e-4: the 1e-5 epsilon changes alpha by 2 - 20 % (see tests/test_gpu_encoder.py)."""
    g = GoldenCase(name)
    pol = _policy(g, cache_dtype=torch.float32)
    gen = torch.Generator().manual_seed(9)
    for m in pol.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_var.copy_((torch.rand(m.running_var.shape, generator=gen) * 1.8e-4 + 2e-5).cuda())
            m.running_mean.copy_((torch.randn(m.running_mean.shape, generator=gen) * 0.05).cuda())
            m.weight.data.copy_((torch.rand(m.weight.shape, generator=gen) * 0.01 + 0.005).cuda())
    env, td = _td(g)
    with torch.inference_mode():
        cache, hidden = pol._packed_encoder().encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32)
    want = _oracle_reference(g, pol)
    assert _rel(hidden, want["hidden"]) <= 2 * REL_TOL and _rel(cache.kvl[0], want["glimpse_key"]) <= 2 * REL_TOL
    wrong = _oracle_reference(g, pol, eps=1e-3)
    assert _rel(wrong["hidden"], want["hidden"]) > 1e-2


@pytest.mark.parametrize("name", ["tsp50_b64_greedy", "cvrp100_b64_greedy"])
def test_fp32_encoder_unfolded_planes_are_the_reference_cache(name):
    """fold=False: planes = project_node_embeddings(h) chunked in three (zoo/am/decoder.py:201-228), node embeddings out."""
    g = GoldenCase(name)
    pol = _policy(g, cache_dtype=torch.float32, fold=False)
    env, td = _td(g)
    with torch.inference_mode():
        cache, hidden = pol._packed_encoder().encode(td, torch.float32, act_dtype=torch.float32, fold=False)
    want = _oracle_reference(g, pol)
    h64 = want["hidden"].double()
    w_node = pol.decoder.project_node_embeddings.weight.detach().double()
    assert cache.unfold and cache.ctx_cur is None and cache.ctx_first is None and cache.node_embed is hidden
    assert _rel(hidden, want["hidden"]) <= REL_TOL
    for i in range(3):
        assert _rel(cache.kvl[i], (h64 @ w_node[128 * i:128 * (i + 1)].t()).float()) <= REL_TOL, i
    assert _rel(cache.q_bias, want["q_bias"]) <= REL_TOL
    h_t, ref = _torch_fp32(pol, td, fold=False)
    assert torch.equal(cache.w_ctx_t, ref.w_ctx_t) and torch.equal(cache.w_out_t, ref.w_out_t)
    # and the whole rollout runs on it
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy")
    assert out["actions"].shape[0] == g.batch


@pytest.mark.parametrize("plane_dtype", [torch.bfloat16, torch.float16])
def test_fp32_encoder_16bit_planes_are_the_fp32_planes_rounded_once(plane_dtype):
    g = GoldenCase("cvrp100_b64_greedy")
    pol = _policy(g)
    env, td = _td(g)
    with torch.inference_mode():
        c32, _ = pol._packed_encoder().encode(td, torch.float32, act_dtype=torch.float32)
        c16, _ = pol._packed_encoder().encode(td, plane_dtype, act_dtype=torch.float32)
    assert c16.kvl.dtype == plane_dtype and torch.equal(c16.kvl, c32.kvl.to(plane_dtype))
    assert torch.equal(c16.ctx_cur, c32.ctx_cur) and torch.equal(c16.q_bias, c32.q_bias)


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 100), ("cvrp", 100), ("op", 20), ("pctsp", 20), ("pdp", 20), ("cvrptw", 20), ("tsp", 7), ("cvrp", 127)])
def test_fp32_policy_rollout_never_reaches_the_torch_encoder(env_name, num_loc):
    """No autocast => encoder, fold, decode, reward all on the library's own kernels: the torch encoder is unreachable
    (its forward raises), the embeddings agree with it to fp32 round-off, and the rollout is valid."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=True)
    torch.manual_seed(3)
    td = env.reset(env.generator(batch_size=[96]))
    with torch.inference_mode():
        h_torch, _ = pol.encoder(td)
    real = pol.encoder.forward
    pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached on the fp32 path"))
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy", return_hidden=True)
    pol.encoder.forward = real
    # (CVRPTW embeds raw time windows up to 480: activations of 1e2 and softmax logits of 1e2 - 1e3, where one ulp of a
    # score moves a probability by 1e-4 — two fp32 evaluations in different summation orders differ at that level)
    assert _rel(out["hidden"], h_torch) <= (1e-4 if env_name == "cvrptw" else REL_TOL)
    assert torch.isfinite(out["reward"]).all() and out["actions"].shape[0] == 96
    # the same rollout from the torch encoder's embeddings: a random-init policy is near-uniform (every step a near-tie at
    # the 1e-2 level), so a few tours may legitimately differ — most must not
    pol_t = AttentionModelPolicy(env_name, fused_encoder=False).cuda().eval()
    pol_t.load_state_dict(pol.state_dict())
    with torch.inference_mode():
        ref = pol_t(td, env, phase="test", decode_type="greedy")
    t = min(out["actions"].shape[1], ref["actions"].shape[1])
    same = (out["actions"][:, :t] == ref["actions"][:, :t]).all(1).float().mean()
    assert same >= 0.9, float(same)


def test_graphed_fp32_rollout_equals_eager_and_follows_weight_updates():
    from rl4co_amd.envs import get_env
    from rl4co_amd.graph import GraphedRollout
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp").cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=50, device="cuda"), device="cuda")
    torch.manual_seed(1)
    d1 = env.generator(batch_size=[256])
    g = GraphedRollout(pol, env, d1, decode_type="greedy")
    assert g._fused and g._act_dtype == torch.float32
    for _ in range(2):
        out = g(d1)
        with torch.inference_mode():
            want = pol(env.reset(d1), env, phase="test", decode_type="greedy")
        assert torch.equal(out["actions"], want["actions"]) and torch.equal(out["reward"], want["reward"])
        with torch.no_grad():
            for p in pol.parameters():
                p.mul_(1.03)


def _double_reference(pol, td):
    """The policy's own torch modules in float64 on the GPU: the exact value both fp32 evaluations are measured against."""
    import copy

    p64 = copy.deepcopy(pol).double()
    td64 = td.clone() if hasattr(td, "clone") else td
    td64 = type(td)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in td.items()}, batch_size=td.batch_size)
    with torch.inference_mode():
        h, _ = p64.encoder(td64)
        cache = p64.decoder.precompute_cache(h, torch.float64, torch.float64)
    return h, cache


@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 100, 64), ("cvrp", 100, 64), ("tsp", 150, 48), ("cvrp", 200, 32),
                                                    ("cvrp", 500, 16), ("pdp", 140, 16), ("tsp", 256, 8)])
def test_fp32_token_tile_encoder_matches_float64_and_the_fused_kernel(env_name, num_loc, batch):
    """csrc/am_tokens_f32.hip (any graph size) against the float64 evaluation of the same modules, against torch's fp32 GPU
    path, and — up to 128 nodes — against the fused kernel (same GEMM routine: only the attention's softmax order differs)."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name).cuda().eval()
    _perturb_norm_stats(pol)
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    torch.manual_seed(3)
    td = env.reset(env.generator(batch_size=[batch]))
    pe = pol._packed_encoder()
    assert pe.supported(td, torch.float32)
    with torch.inference_mode():
        cache, hidden = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32, tokens=True)
    torch.cuda.synchronize()
    h64, c64 = _double_reference(pol, td)
    h32, c32 = _torch_fp32(pol, td)
    checks = {"hidden": (hidden, h64, h32), "ctx_cur": (cache.ctx_cur, c64.ctx_cur, c32.ctx_cur), "q_bias": (cache.q_bias, c64.q_bias, c32.q_bias)}
    for i in range(3):
        checks[f"plane{i}"] = (cache.kvl[i], c64.kvl[i], c32.kvl[i])
    if env_name == "tsp":
        checks["ctx_first"] = (cache.ctx_first, c64.ctx_first, c32.ctx_first)
    for nm, (got, exact, torch32) in checks.items():
        assert torch.isfinite(got).all(), nm
        e_k = float((got.double() - exact).norm() / exact.norm())
        e_t = float((torch32.double() - exact).norm() / exact.norm())
        assert e_k <= REL_TOL and e_k <= 3 * e_t + 2e-7, f"{nm}: kernel {e_k:.3e}, torch fp32 {e_t:.3e}"
    if td["action_mask"].shape[-1] <= 128:
        with torch.inference_mode():
            fused, hf = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32, tokens=False)
        assert _rel(hidden, hf) <= 1e-6 and _rel(cache.kvl, fused.kvl) <= 1e-6


def test_fp32_rollout_beyond_128_nodes_never_reaches_the_torch_encoder():
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("cvrp").cuda().eval()
    env = get_env("cvrp", generator_params=dict(num_loc=300, device="cuda"), device="cuda", check_solution=True)
    torch.manual_seed(3)
    td = env.reset(env.generator(batch_size=[32]))
    pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached on the fp32 path"))
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy")
        out16 = AttentionModelPolicy("cvrp", cache_dtype=torch.bfloat16).cuda().eval()  # fp32 encoder, bf16 planes
        out16.load_state_dict(pol.state_dict())
        out16.encoder.forward = pol.encoder.forward
        o2 = out16(td, env, phase="test", decode_type="greedy")
    assert torch.isfinite(out["reward"]).all() and torch.isfinite(o2["reward"]).all()
    assert abs(float(out["reward"].mean() - o2["reward"].mean())) <= 0.02 * abs(float(out["reward"].mean()))


@pytest.mark.parametrize("num_loc,batch", [(129, 3), (128, 2), (255, 1), (256, 5), (17, 4)])
def test_token_tile_paths_at_tile_boundaries(num_loc, batch):
    """Graph sizes around the 128-node tile (CVRP: N = num_loc + 1) and tiny batches: the fp32 and the 16-bit token-tile
    launches against float64 / each other; fp32 planes out of the 16-bit tiles; the raw (fold=False) planes beyond 128 nodes."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(num_loc)
    pol = AttentionModelPolicy("cvrp").cuda().eval()
    _perturb_norm_stats(pol)
    env = get_env("cvrp", generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    td = env.reset(env.generator(batch_size=[batch]))
    pe = pol._packed_encoder()
    with torch.inference_mode():
        c32, h32 = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32, tokens=True)
        c16, h16 = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.bfloat16, tokens=True)  # fp32 planes, bf16 tiles
    h64, c64 = _double_reference(pol, td)
    assert float((h32.double() - h64).norm() / h64.norm()) <= REL_TOL
    assert float((c32.kvl.double() - c64.kvl).norm() / c64.kvl.norm()) <= REL_TOL
    assert c16.kvl.dtype == torch.float32 and float((c16.kvl.double() - c64.kvl).norm() / c64.kvl.norm()) <= 3e-2
    assert float((h16.double() - h64).norm() / h64.norm()) <= 3e-2 and _rel(c16.ctx_cur, c32.ctx_cur) <= 3e-2
    assert _rel(c16.q_bias, c32.q_bias) <= 3e-2
    if num_loc + 1 > 128:
        with torch.inference_mode():
            raw, hid = pe.encode(td, torch.float32, act_dtype=torch.float32, fold=False, tokens=True)
        w_node = pol.decoder.project_node_embeddings.weight.detach().double()
        assert raw.unfold and raw.node_embed is hid and _rel(hid, h32) <= 1e-6
        for i in range(3):
            assert float((raw.kvl[i].double() - h64 @ w_node[128 * i:128 * (i + 1)].t()).norm() / (h64 @ w_node[128 * i:128 * (i + 1)].t()).norm()) <= REL_TOL


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("b,n,blocks,graph", [(7, 100, 2, True), (3, 501, 1, True), (16, 129, 2, False), (1, 5, 1, True)])
def test_fold_tables_kernel_matches_float64(b, n, blocks, graph, dt):
    """rl4co_am_fold_tables_f32 (context tables + graph context from any encoder's embeddings, 16-bit rows widened on
    load, fp32 MFMA) against the float64 products of the SAME (rounded) embeddings: fp32 round-off only."""
    from rl4co_amd.cache import _fold_tables_f32

    gen = torch.Generator().manual_seed(b * 1000 + n)
    h = torch.randn(b, n, 128, generator=gen).to(dt).cuda()
    ws = [(torch.randn(128, 128, generator=gen) * 0.1).cuda() for _ in range(blocks)]
    w_fixed = (torch.randn(128, 128, generator=gen) * 0.1).cuda() if graph else None
    outs, q_bias = _fold_tables_f32(h, ws, w_fixed)
    torch.cuda.synchronize()
    h64 = h.double()
    for o, w in zip(outs, ws):
        want = h64 @ w.double().t()
        assert o.dtype == torch.float32 and float((o.double() - want).norm() / want.norm()) <= 2e-6
    if graph:
        want = h64.mean(1) @ w_fixed.double().t()
        assert float((q_bias.double() - want).norm() / want.norm()) <= 3e-6
    else:
        assert q_bias is None


@pytest.mark.parametrize("regime", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("env_name,num_loc", [("tsp", 100), ("cvrp", 50), ("op", 20), ("pctsp", 20), ("pdp", 20), ("cvrptw", 20), ("cvrp", 200)])
def test_return_init_embeds_is_served_by_the_kernels_own_init_embedding(env_name, num_loc, regime):
    """AttentionModelPolicy.forward(return_init_embeds=True) (zoo/am/encoder.py:84-103) no longer sends inference to the torch
    encoder: `init_embeds` is one more launch of the init-embedding routine the encoder kernels run
    (rl4co_am_encoder_init_embeds_f32 / _16), in the activations' type. Against the module's own init embedding in fp32:
    fp32 to round-off (K <= 6 products in a different order), 16-bit within one rounding of the output."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    dt = {"fp32": None, "bf16": torch.bfloat16, "f16": torch.float16}[regime]
    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name, encoder_autocast=dt, **({} if dt is None else {"cache_dtype": dt})).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    torch.manual_seed(3)
    td = env.reset(env.generator(batch_size=[48]))
    with torch.inference_mode():
        ref = pol.encoder.init_embedding(td).float()
    real = pol.encoder.forward
    pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
    try:
        with torch.inference_mode():
            out = pol(td, env, phase="test", decode_type="greedy", return_init_embeds=True, return_hidden=True)
    finally:
        pol.encoder.forward = real
    ie = out["init_embeds"]
    assert ie.shape == ref.shape and ie.dtype == (torch.float32 if dt is None else dt)
    tol = {"fp32": 2e-6, "bf16": 2.0 ** -8, "f16": 2.0 ** -11}[regime]
    assert _rel(ie.float(), ref) <= tol, _rel(ie.float(), ref)
    assert torch.isfinite(out["reward"]).all()
