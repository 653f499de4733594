"""CPU (`-m "not gpu"`): host logic added in round 4 — fp32 weight packing, the stacked packing of the training forward,
the deferred-reduction arena of the stack backward, the compact bench line and its size contract. No kernel is launched."""
import importlib.util
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fp32_weight_packing_is_the_documented_fragment_order():
    """[out/16 tiles][in/16 chunks][64 lanes][4]: lane = 16 g + c, element s = W[16 tile + c][16 chunk + 4 g + s]
    (include/rl4co_amd.h: rl4co_am_encoder_f32; csrc/enc_f32.h: load_w)."""
    from rl4co_amd.encoder import pack_weight_f32

    torch.manual_seed(0)
    for out_f, in_f in ((128, 128), (384, 128), (128, 512), (512, 128)):
        w = torch.randn(out_f, in_f)
        p = pack_weight_f32(w)
        assert p.shape == (out_f // 16, in_f // 16, 64, 4) and p.dtype == torch.float32 and p.is_contiguous()
        for tile, chunk, lane, s in ((0, 0, 0, 0), (1, 3, 17, 2), (out_f // 16 - 1, in_f // 16 - 1, 63, 3), (2, 1, 37, 1)):
            c, g = lane & 15, lane >> 4
            assert p[tile, chunk, lane, s] == w[16 * tile + c, 16 * chunk + 4 * g + s]


def test_stacked_16bit_packing_equals_the_per_layer_packing():
    """train_ops._pack_stack packs all layers of a stack at once: the bytes of encoder.pack_weight per layer."""
    from rl4co_amd.encoder import pack_weight
    from rl4co_amd.train_ops import _pack_stack

    torch.manual_seed(1)
    for out_f, in_f in ((384, 128), (128, 128), (512, 128), (128, 512)):
        w = torch.randn(6, out_f, in_f).to(torch.bfloat16)
        want = torch.stack([pack_weight(w[l].float(), torch.bfloat16) for l in range(6)])
        got = _pack_stack(w)
        assert got.is_contiguous() and got.numel() == want.numel()
        assert torch.equal(got.reshape(-1).view(torch.int16), want.reshape(-1).view(torch.int16))


def test_grad_arena_defers_and_sums_per_kind():
    """_GradArena: every (kind, layer) slot is a view of ONE buffer per kind; finish() reduces each kind once; the
    handles give the same tensors the immediate path returns (weight | bias split, norm pair)."""
    from rl4co_amd.train_ops import _GradArena

    torch.manual_seed(2)
    arena = _GradArena(3)
    n, k, chunks = 4, 5, 7
    parts, norms = {}, {}
    for layer in (2, 0, 1):  # (the backward walks the layers in reverse; any order must work)
        slot = arena.slot(("w1", layer), chunks, n * k + n, "cpu")
        assert slot.shape == (chunks, n * k + n)
        parts[layer] = torch.randn(chunks, n * k + n)
        slot.copy_(parts[layer])
        ns = arena.norm_slot(("norm1", layer), 11, 8, "cpu")
        assert ns.shape == (2, 11, 8)
        norms[layer] = torch.randn(2, 11, 8)
        ns.copy_(norms[layer])
    assert set(arena.buf) == {"w1", "norm1"} and arena.buf["w1"].shape == (3, chunks, n * k + n)
    arena.finish()
    for layer in range(3):
        dw, db = arena.wgrad((("w1", layer), n, k, True))
        want = parts[layer].sum(0)
        assert torch.allclose(dw, want[: n * k].view(n, k)) and torch.allclose(db, want[n * k:])
        dg, dbeta = arena.norm(("norm1", layer))
        assert torch.allclose(dg, norms[layer][0].sum(0)) and torch.allclose(dbeta, norms[layer][1].sum(0))


def test_stack_forward_is_not_taken_off_the_gpu():
    from rl4co_amd import train_ops
    from rl4co_amd.policy import _GraphAttentionNetwork

    net = _GraphAttentionNetwork(8, 128, 2, "instance", 512)
    assert net.fused_stack is True
    assert not train_ops.stack_usable(torch.zeros(2, 10, 128, dtype=torch.bfloat16), net.layers)  # CPU tensor
    out = net(torch.zeros(2, 10, 128))  # CPU, no autocast: the torch modules
    assert out.shape == (2, 10, 128)


def test_compact_bench_line_stays_under_4k_even_for_bloated_details():
    """The driver parses ONE stdout line: whatever the legs / parity block grow to, the line stays <= 4096 bytes
    (r03's 27 KB line came back `parsed: null`)."""
    b = _bench()
    long = "x" * 3000
    roof = {"kernel": "am_decode kernel, STREAM (1 wave / trajectory): " + long, "bound": "hbm", "achieved": 6591.123456, "peak": 8000.0,
            "unit": "GB/s", "frac": 0.8239, "traffic": 1.59e10, "launch_ms_mean": 2.475, "contract_GBs": 12920.0,
            "hbm_read_probe_GBs": 6179.0, "bytes_model": long, "contract_note": long}
    results = {"c2_greedy": {"ms_per_step": 3.3, "value": 1.24e8}}
    for i in range(12):
        results[f"leg_{i}_with_a_long_name"] = {"ms_per_step": 3.3 + i, "value": 1e8, "roofline": dict(roof), "parity": {"blob": long},
                                                  "parity_tours": "4096/4096", "scaling_efficiency": 0.987654321}
    rec = {"identical": 4096, "of": 4096, "flip_regret_max": 0.0, "step_agreement": 0.98, "reward_rel_gap": 4e-4, "blob": long}
    detail = {"metric": "decode_steps_per_sec", "value": 1.24e8, "unit": "instance·step/s", "n_gpus": 1, "steps": 20, "warmup": 5,
              "ms_per_step": 3.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
              "config": {"workload": long, "leg": "c2_greedy", "batch_per_gpu": 4096, "cache_dtype": "bf16", "parallelism": "replicas x1"},
              "launch": "pipeline: " + long, "roofline": roof, "encoder_roofline": dict(roof, bound="mfma"),
              "parity": {"c2_greedy": {"fp32": rec, "bf16_vs_reference_bf16_autocast": rec, "reference_bf16_vs_reference_fp32_identical": 661},
                         "c3_greedy": {"fp32": rec, "bf16_vs_reference_bf16_autocast": rec},
                         "c2_sampling": {"fp32_reference_noise": rec}, "c5_sampling": {"greedy_fp32": rec}, "blob": long},
              "cpu_baseline": {"value": 1.5e5, "unit": "instance·step/s", "cores": 16, "kind": "port", "sample": long, "gpu_over_cpu": 800.0}}
    line = b.compact_line(detail, "c2_greedy", results, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    encoded = json.dumps(line, separators=(",", ":"))
    assert len(encoded.encode()) <= b.MAX_LINE_BYTES
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "legs", "parity", "detail_file"):
        assert k in line, k
    assert len(line["parity"]) <= 10 and line["roofline"]["frac"] == 0.8239 and line["config"]["launch"] == "pipeline"
    assert set(line["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms_mean"}


def test_leg_dtypes_follow_the_leg_name_then_the_flags():
    b = _bench()

    class A:
        cache_dtype, encoder_dtype = "bf16", "bf16"

    assert b.leg_dtypes("c2_greedy", A) == ("bf16", "bf16") and b.leg_dtypes("c2_greedy_fp16", A) == ("f16", "f16")
    assert b.leg_dtypes("c2_greedy_fp32", A) == ("f32", "f32")
    A.cache_dtype = "f32"
    assert b.leg_dtypes("c5_sampling", A) == ("f32", "bf16")


def test_layer_norm_tables_carry_the_bias_in_front_of_the_norm():
    """Normalization("layer") (nn/ops.py:48-51) has no parameters and ONE mean per instance, so the bias of the GEMM in front
    of it does not cancel: the 16-bit kernel reads it from the shift slot (added before the statistics), the fp32 kernel adds
    bo / b2 itself and gets (1, 0) tables; the kernel selector is 2."""
    from rl4co_amd.encoder import PackedEncoder
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", normalization="layer", num_encoder_layers=2).eval()
    with torch.no_grad():
        for layer in pol.encoder.net.layers:
            layer[0].module.out_proj.bias.normal_()
            layer[2].module.lins[1].bias.normal_()
    assert sum(p.numel() for l in pol.encoder.net.layers for p in l[1].parameters()) == 0
    pe = PackedEncoder(pol)
    t = pe.refresh(act_dtype=torch.bfloat16)
    assert pe.norm_kind == 2
    layers = list(pol.encoder.net.layers)
    assert torch.equal(t["n1_shift"], torch.stack([l[0].module.out_proj.bias.detach() for l in layers]))
    assert torch.equal(t["n2_shift"], torch.stack([l[2].module.lins[1].bias.detach() for l in layers]))
    assert bool((t["n1_scale"] == 1).all()) and bool((t["n2_scale"] == 1).all())
    t = pe.refresh(act_dtype=torch.float32)
    assert pe.norm_kind == 2 and not bool(t["n1_shift"].any()) and not bool(t["n2_shift"].any())
    assert torch.equal(t["bo"], torch.stack([l[0].module.out_proj.bias.detach() for l in layers]))
    # the module's own forward is the reference's formula
    x = torch.randn(3, 9, 128)
    want = (x - x.mean((1, 2)).view(-1, 1, 1)) / torch.sqrt(x.var((1, 2)).view(-1, 1, 1) + 1e-05)
    assert torch.equal(layers[0][1](x), want)


def test_layer_norm_training_blocks_are_offered_to_the_kernels():
    from rl4co_amd import train_ops

    assert train_ops._block_norm_args(type("N", (), {"kind": "layer"})()) == (None, None, train_ops.LAYER_NORM_EPS)
    assert train_ops._f32(None) is None
    x = torch.zeros(2, 5, 128)
    assert not train_ops.block_usable(x, "layer", torch.zeros(128, 128))  # CPU tensors: never
    assert not train_ops.block_usable(x, "group", torch.zeros(128, 128))
