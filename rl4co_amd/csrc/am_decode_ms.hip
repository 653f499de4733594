// am_decode_ms.hip — multistart (POMO) decode on the matrix cores.
//
// With multistart decoding (utils/decoding.py:282-330, zoo/pomo/model.py:88-143) the S
// trajectories of one instance read the SAME decoder cache and differ only in their query and
// mask: a decode step for all of them is three small GEMMs — scores^T = K_g . Q^T, glimpse^T =
// V^T . P^T, logits^T = K_l' . heads^T — with the TRAJECTORY as the MFMA column. The streaming
// kernel (am_decode.hip) runs one wave per trajectory and is VALU-bound there (275 M
// trajectory-steps/s at TSP-100 x 4096 x 8); here one 512-thread workgroup owns one COLUMN TILE of an
// instance — 16 of its starts, one per v_mfma_f32_16x16x16_bf16 accumulator column — keeps the instance's
// glimpse keys / values in LDS and its wave's logit-key tile in registers for the whole rollout, and two
// such workgroups share a CU (make_layout).
//
// Layout (the one am_teacher_mma.hip uses): in the 16x16x16 accumulator a lane owns ONE
// trajectory (column lane & 15) and four consecutive rows 4 g .. 4 g + 3, which is also the
// B-operand layout, so scores chain into the glimpse product without a shuffle; the softmax over
// keys, the log-softmax and the arg-max over nodes are in-lane reductions plus two cross-row-group
// exchanges, and every lane carries its trajectory's state (feasibility bit mask, current / first
// node, load) in registers. Wave h owns head h for the glimpse and node tile h for the logits;
// the planes stay in their natural [node][dim] layout and the glimpse product reads V through the
// gfx950 transpose read ds_read_b64_tr_b16. A decode step is a LATENCY chain (dependent MFMAs,
// cross-lane reductions, two workgroup barriers), not an issue-bound stream: the second workgroup on
// the CU (four waves per SIMD) is what fills the gaps.
//
// Environments: all six of the decode kernel. The trajectory state a lane carries is closed-form — bit sets (feasible,
// visited / available, PDP's to-deliver), current node, step counter, and one or two floats (load, tour length,
// collected prize, clock); the instance's demands / prizes, coordinates, entry limits, time windows and service times
// sit in 3.5 KB of LDS. Auto-selected where it was measured faster than one wave per trajectory (am_decode.hip,
// resolve_variant): TSP / PDP / PCTSP from 8 starts, CVRP from 16; OP and CVRP-TW on request.
//
// Numerics: bf16 MFMA inputs (planes, query, softmax numerators, glimpse), fp32 accumulation and
// fp32 softmax / tanh / log-softmax with the hardware's exp2 / log / rcp — the reference's
// mixed-precision regime. Not part of the bit-exact contract of am_decode.hip (a bf16 query cannot
// reproduce the fp32 specified order); pinned instead to the rounding-model oracle oracle_am_decode_ms
// (same bf16 rounding points, fp32 elsewhere): identical trajectories, per-step log-probs within 5e-3
// (tests/test_gpu_decode_ms.py).
#include <hip/hip_runtime.h>

#include "common.h"
#include "elem16.h"
#include "rl4co_math.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kRS = kD + 8;  // LDS row stride in bf16 elements
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kLdsBudget = 160 * 1024;
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;
constexpr float kLog2e = 1.44269504088896341f;

typedef elem_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// C[m = 4 g + r][n = lane & 15] += sum_k A[m = lane & 15][k = 4 g + s] * B[k = 4 g + s][n = lane & 15]
__device__ inline f32x4 mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return rl4co_e16::mfma_16x16x16(a, b, c);
}
__device__ inline f32x4 zero4() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
__device__ inline bf16x4 lds_b64(const elem_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
// the 16 lanes of a row group address a [4 rows][16 columns] block (lane i: row i / 4, columns
// 4 (i % 4) ..); lane c receives column c of it, i.e. four consecutive ROWS
__device__ inline bf16x4 lds_tr(const elem_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }
__device__ inline float rg_max(float v) { return rl4co::bfly_max<16, 64>(v); }

// 128-bit node sets as four NAMED words: an array member invites the optimiser to turn a chain
// of selects into one dynamically indexed load, which sends the whole trajectory state to scratch
// memory (a global-memory round trip per access)
struct Bits128 {
  uint32_t w0, w1, w2, w3;
  __device__ inline uint32_t word(int k) const { return k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3)); }
  __device__ inline void put(int k, uint32_t v) {
    w0 = k == 0 ? v : w0;
    w1 = k == 1 ? v : w1;
    w2 = k == 2 ? v : w2;
    w3 = k == 3 ? v : w3;
  }
  __device__ inline uint32_t any() const { return w0 | w1 | w2 | w3; }
  __device__ inline bool test(int j) const { return (word(j >> 5) >> (j & 31)) & 1u; }
  __device__ inline void set(int j, bool on) {
    const uint32_t bit = 1u << (j & 31);
    const int k = j >> 5;
    const uint32_t m0 = k == 0 ? bit : 0u, m1 = k == 1 ? bit : 0u, m2 = k == 2 ? bit : 0u, m3 = k == 3 ? bit : 0u;
    w0 = on ? (w0 | m0) : (w0 & ~m0);
    w1 = on ? (w1 | m1) : (w1 & ~m1);
    w2 = on ? (w2 | m2) : (w2 & ~m2);
    w3 = on ? (w3 | m3) : (w3 & ~m3);
  }
};

#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 7  // phase clocks (tools/ms_phases.py): shader-clock sums per step segment, per wave class
__device__ unsigned long long g_ms_clk[16];
#define RL4CO_MS_MARK(i)                                          \
  {                                                               \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc_[i] += now_ - clk_;                                       \
    clk_ = now_;                                                  \
  }
#else
#define RL4CO_MS_MARK(i)
#endif

struct __align__(16) Xchg {  // per (node tile, trajectory) pieces of the log-softmax / selection over nodes
  float zmax, se, best_key, best_z;
  int best_idx;
  float forced_z;
  int pad0, pad1;
};

struct Layout {  // byte offsets into dynamic LDS
  int kgs, vs, hs, xs, dems, envf, total;
};
// One workgroup = (instance, column tile of 16 starts). A decode step of 16 trajectories is a latency chain (~8 K cycles of
// dependent MFMAs, transcendentals and two barriers) that leaves every pipe idle most of the time, so the layout is sized
// for TWO workgroups per CU: glimpse keys / values in LDS (61 KB at N = 100), wave w's logit-key tile (16 nodes x 128
// dims, the same at every step) in 16 registers for the whole rollout, the fp32 context rows read from L2 (two rows per
// trajectory-step), one column tile of staging — 70 KB and 128 registers. Measured against the first layout (one
// workgroup per instance advancing two column tiles, all three planes and the context table in LDS, 142 KB, one workgroup
// per CU), TSP-100 x 4096 sampling: 8 starts 7.1 -> 4.9 ms, 32 starts 13.8 -> 9.8, 100 starts 48.1 -> 34.1 (1.19 G
// trajectory-steps/s); the tiles of an instance re-read its planes from the XCD's L2.
// PAIR (r03): with at most 8 starts per instance only half of the 16 accumulator columns carry a trajectory — and the step
// is VALU-bound (r03 counters: 853 VALU instructions per wave and step against 21 MFMAs; four waves per SIMD each issuing
// VALU for 27 % of their residency), so every per-lane instruction is paid for 16 columns whether 8 or 16 are alive. A
// workgroup then takes TWO instances: columns 0-7 the starts of the first, columns 8-15 those of the second. Their planes
// sit side by side in LDS (`copies` = 2), every product is issued twice — once per instance, with the OTHER instance's
// columns of the B operand zeroed, chained through the accumulator (exact zeros: bit-identical results) — and everything
// per-lane (softmax, tanh, noise, selection, transition) serves 16 live trajectories for the same instructions. 131 KB of
// LDS: one workgroup per CU, the matrix pipe (9 % busy before) takes the doubled products without noticing.
__host__ __device__ inline Layout make_layout(int nt, int n, int copies = 1) {
  (void)n;
  Layout L;
  const int plane = nt * 16 * kRS * 2;
  int o = 0;
  L.kgs = o; o += copies * plane;
  L.vs = o; o += copies * plane;
  L.hs = o; o += 16 * kRS * 2;                          // one column tile of glimpses
  L.xs = o; o += kWaves * 16 * (int)sizeof(Xchg);
  L.dems = o; o += copies * 128 * 4;
  L.envf = o; o += copies * 128 * 6 * 4;                 // coordinates | OP entry limits / time windows | service times
  L.total = (o + 15) & ~15;
  return L;
}

struct Traj {  // one trajectory's state, replicated in the four row groups and the eight waves
  Bits128 mw, vw, tw;  // feasible; visited (PDP: available); PDP: to_deliver
  int cur, first, r;
  long long step_i;
  float used, now;     // load / tour length / collected prize; CVRPTW clock
  bool done, ok;
  int nsteps;
  float f4[4];  // context row of the first node (TSP), fetched when it becomes known
  // outputs are parked in registers and written every 32 steps: a global store per step would put
  // its L2 round trip in front of the next barrier 100 times per rollout. Copy 4 w + g of the
  // trajectory parks the step with t % 32 == 4 w + g.
  int park_act;
  float park_logp;
  int64_t park_col;
};

struct Sel {  // log-softmax / selection pieces over a set of nodes
  float zmax, se, key, z, fz;
  int idx;
};
// merge two disjoint node sets; symmetric in its arguments so every replica gets identical bits
// natural log on the hardware log2 (v_log_f32, 1 ulp) and one multiply. The library __logf wraps the same instruction in
// denormal scaling and an extended-precision product — about fifteen issue slots a call, thirteen calls per lane and
// step here (the Gumbel noise is log(-log u) of four uniforms), none of whose arguments can be denormal: uniforms and
// exponential noise are >= 2^-33, -log u >= 2^-25, sums of exponentials >= 1
__device__ inline float ln_fast(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
// BOUNDED (r06): clipped logits live in [-C, C], C = tanh_clipping / temperature; up to C = 60 their exponentials and the
// sum over <= 128 nodes are plain fp32 numbers, so the log-sum-exp needs no running maximum — the pieces are exp(z) sums
// and a merge is ONE addition instead of a maximum, two subtractions, two exponentials and a multiply-add (five merges per
// wave and step: the step is VALU-issue bound, r05 counters: 795 VALU instructions per wave-step against 44 MFMAs). The
// teacher kernel (am_teacher_mma.hip, `bounded`) has summed its log-softmax that way since r03.
__device__ inline Sel merge(const Sel& p, const Sel& q, bool bounded) {
  Sel o;
  if (bounded) {
    o.zmax = 0.0f;
    o.se = p.se + q.se;
  } else {
    o.zmax = fmaxf(p.zmax, q.zmax);
    const float zs = (o.zmax > kNegInf) ? o.zmax : 0.0f;
    o.se = p.se * __expf(p.zmax - zs) + q.se * __expf(q.zmax - zs);  // exp(-inf) = 0 for an empty set
  }
  const bool take_q = (q.idx != 0x7fffffff) & ((p.idx == 0x7fffffff) | (q.key > p.key) | ((q.key == p.key) & (q.idx < p.idx)));
  o.key = take_q ? q.key : p.key;
  o.z = take_q ? q.z : p.z;
  o.idx = take_q ? q.idx : p.idx;
  o.fz = fmaxf(p.fz, q.fz);
  return o;
}
template <int STEP>
__device__ inline Sel partner(const Sel& p) {
  Sel o;
  o.zmax = rl4co::bfly_f<STEP>(p.zmax);
  o.se = rl4co::bfly_f<STEP>(p.se);
  o.key = rl4co::bfly_f<STEP>(p.key);
  o.z = rl4co::bfly_f<STEP>(p.z);
  o.fz = rl4co::bfly_f<STEP>(p.fz);
  o.idx = rl4co::bfly_i<STEP>(p.idx);
  return o;
}

struct Shared {
  const uint16_t* kl_g;  // this instance's logit-key plane in global memory (row stride kl_rs)
  const uint16_t* kl_g2; // PAIR: the second instance's
  int plane2;            // PAIR: elements from the first instance's LDS plane to the second's
  int64_t kl_rs;
  const elem_t *kgs, *vs;
  elem_t* hs;   // [CT][16 trajectories][kRS] glimpses of this step
  Xchg* xs;     // [CT][8 node tiles][16 trajectories]
  const float* dems;  // [128] CVRP / CVRPTW demand (index j - 1 at j), PCTSP real prize at j
  const float* envf;  // [128][2] coordinates | [128] OP entry limits | [128][2] time windows | [128] service times
};

// MODE: 0 greedy, 1 sampling, 2 evaluate (RL4CO_DECODE_*). CT column tiles of 16 trajectories advance together.
template <int ENV, int NT, int MODE, int CT, bool PAIR = false>
__device__ __forceinline__ void rollout_tiles(const rl4co_am_decode_args& a, const Shared& sh, int inst0, int s0,
                                             uint32_t& errbits) {
  static_assert(!PAIR || CT == 1, "two instances share ONE column tile");
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int h = w;
  // PAIR: this lane's column belongs to instance inst0 (columns 0-7) or inst0 + 1 (columns 8-15)
  const int half = PAIR ? (tl >> 3) : 0;
  const bool inst_ok = inst0 + half < a.B_inst;
  const int inst = inst_ok ? inst0 + half : inst0;
  const int N = a.N, S = a.B / a.B_inst;
  const int nao = tl * kRS + 4 * g;                         // natural operand: row lane & 15, columns 4 g ..
  const int tro = (4 * g + (tl >> 2)) * kRS + 4 * (tl & 3);  // transpose read
  const int dcol = 16 * h + 4 * g;                          // the four dims of head h this lane owns
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kClock = ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kScalar = ENV != RL4CO_ENV_TSP && ENV != RL4CO_ENV_PDP;  // one context scalar: cap - used
  constexpr bool kVisited = ENV != RL4CO_ENV_TSP;                         // a visited (PDP: available) set beside the mask
  constexpr bool kStepI = ENV != RL4CO_ENV_CVRP && ENV != RL4CO_ENV_CVRPTW;
  const float* envf_l = sh.envf + half * 768;  // this lane's instance data (PAIR: second copy)
  const float* dems_l = sh.dems + half * 128;
  const float* locs = envf_l;              // [N][2]
  const float* opmax = envf_l + 256;       // [N] OP: longest tour with which node j may be entered
  const float* twin = envf_l + 384;        // [N][2] CVRPTW (start, end)
  const float* dur = envf_l + 640;         // [N] CVRPTW service times
  // context scalar = cap - used: vehicle capacity (CVRP / CVRPTW), prize still required (PCTSP, clamped at 0), longest
  // tour that may still end at the depot minus the tour so far (OP) — env_embeddings/context.py:105-213
  const float cap = (kCvrpLike || ENV == RL4CO_ENV_PCTSP) ? a.vehicle_capacity[inst]
                                                                     : (ENV == RL4CO_ENV_OP ? a.max_length[(int64_t)inst * N] : 0.0f);
  const float thr = cap + 1e-5f;
  // context tables: dense fp32 [B_inst,N,128], or (r06, ctx_dtype) rows in the planes' 16-bit type with the caller's strides
  const bool ctx16 = a.ctx_dtype != RL4CO_DT_F32;
  const int64_t ctx_esz = ctx16 ? 2 : 4;
  const int64_t ctx_rs = (a.ctx_row_stride ? a.ctx_row_stride : (int64_t)kD) * ctx_esz;  // bytes between node rows
  const int64_t ctx_off = ((int64_t)inst * (a.ctx_batch_stride ? a.ctx_batch_stride : (int64_t)N * kD) + dcol) * ctx_esz;
  const char* ctxc = static_cast<const char*>(a.ctx_cur) + ctx_off;
  const char* ctxf = (ENV == RL4CO_ENV_TSP) ? static_cast<const char*>(a.ctx_first) + ctx_off : nullptr;
  auto first_row = [&](int node, float (&f)[4]) {
    const float4 v = rl4co_e16::load_ctx4(ctxf + (int64_t)node * ctx_rs, ctx16);
    f[0] = v.x;
    f[1] = v.y;
    f[2] = v.z;
    f[3] = v.w;
  };
  float qb4[4], qx4[4], qt4[4];  // graph context; placeholder query (TSP) or capacity column; CVRPTW: the time column
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    qb4[e] = a.q_bias ? a.q_bias[(int64_t)inst * kD + dcol + e] : 0.0f;
    qx4[e] = (ENV == RL4CO_ENV_TSP) ? a.q_step0[dcol + e] : (kScalar ? a.w_cap[dcol + e] : 0.0f);
    qt4[e] = kClock ? a.w_time[dcol + e] : 0.0f;
  }
  const bool single = a.max_steps == 1;
  const float inv_temp = 1.0f / a.temperature;
  const float clip_over_temp = a.tanh_clipping * inv_temp;
  const bool clip = a.tanh_clipping > 0.0f;
  const bool bounded = clip && clip_over_temp <= 60.0f;  // (uniform: a scalar branch around the maximum bookkeeping, see merge)
  Bits128 nv;  // nodes that exist (j < N), per 32-node word
#pragma unroll
  for (int k = 0; k < 4; ++k) nv.put(k, (N >= 32 * (k + 1)) ? 0xffffffffu : (N > 32 * k ? ((1u << (N - 32 * k)) - 1u) : 0u));
  const uint32_t inner_sel = a.mask_inner ? 0xffffffffu : 0u, logit_sel = a.mask_logits ? 0xffffffffu : 0u;

  Traj tj[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    Traj& x = tj[c];
    const int sl = PAIR ? (tl & 7) : s0 + 16 * c + tl;
    x.ok = sl < S && inst_ok;
    x.r = (x.ok ? sl : (PAIR ? 0 : s0)) * a.B_inst + inst;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t m = 0, v = 0, d = 0;
      const uint8_t* gm = a.action_mask + (int64_t)x.r * N + 32 * k;
      const uint8_t* gv = kVisited ? a.visited + (int64_t)x.r * N + 32 * k : nullptr;
      const uint8_t* gd = (ENV == RL4CO_ENV_PDP) ? a.to_deliver + (int64_t)x.r * N + 32 * k : nullptr;
      for (int b = 0; b < 32 && 32 * k + b < N; ++b) {
        m |= (x.ok && gm[b]) ? (1u << b) : 0u;
        if (kVisited) v |= (x.ok && gv[b]) ? (1u << b) : 0u;
        if (ENV == RL4CO_ENV_PDP) d |= (x.ok && gd[b]) ? (1u << b) : 0u;
      }
      x.mw.put(k, m);
      x.vw.put(k, v);
      x.tw.put(k, d);
    }
    x.cur = (int)a.current_node[x.r];
    x.first = (ENV == RL4CO_ENV_TSP) ? (int)a.first_node[x.r] : 0;
    x.step_i = kStepI ? a.step_i[x.r] : 0;
    x.used = kScalar ? a.used_capacity[x.r] : 0.0f;
    x.now = kClock ? a.current_time[x.r] : 0.0f;
    x.done = !x.ok || a.done[x.r] != 0;
    x.nsteps = 0;
    x.park_act = 0;
    x.park_logp = 0.0f;
    x.park_col = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) x.f4[e] = 0.0f;
    if (ENV == RL4CO_ENV_TSP && x.step_i > 0) first_row(x.first, x.f4);
  }
  // the logit-key tile of this wave (nodes 16 w .., all 128 dims) is the same at every step — 16 registers for the
  // whole rollout instead of a third LDS plane (rows past the graph: any finite value, their logits are masked)
  bf16x4 lfr[8], lfr2[PAIR ? 8 : 1];
  if (w < NT) {
    const uint16_t* row = sh.kl_g + (int64_t)min(16 * w + tl, N - 1) * sh.kl_rs + 4 * g;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) lfr[ks] = *reinterpret_cast<const bf16x4*>(row + 16 * ks);
    if (PAIR) {
      const uint16_t* row2 = sh.kl_g2 + (int64_t)min(16 * w + tl, N - 1) * sh.kl_rs + 4 * g;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) lfr2[PAIR ? ks : 0] = *reinterpret_cast<const bf16x4*>(row2 + 16 * ks);
    }
  }
  // PAIR (r06): the glimpse operands of head h — key rows and transposed value tiles of BOTH instances, 4 x NT fragments =
  // 56 registers at NT = 7 — are the same at every step too; the paired kernel is LDS-bound at two waves per SIMD with
  // 153 of its 256 registers in use, so they are read ONCE: 4 NT LDS reads per wave and step (of ~45) leave the step's
  // chain and the LDS pipe (conflict ratio 0.46, r05 counters). The unpaired kernel lives in 128 registers and keeps reading.
#ifdef RL4CO_MS_NO_REG_PLANES
  constexpr bool kRegPlanes = false;
#else
  constexpr bool kRegPlanes = PAIR;
#endif
  bf16x4 kfr[kRegPlanes ? NT : 1], kfr2[kRegPlanes ? NT : 1], vfr[kRegPlanes ? NT : 1], vfr2[kRegPlanes ? NT : 1];
  if constexpr (kRegPlanes) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      kfr[jt] = lds_b64(sh.kgs + 16 * jt * kRS + 16 * h + nao);
      kfr2[jt] = lds_b64(sh.kgs + sh.plane2 + 16 * jt * kRS + 16 * h + nao);
      vfr[jt] = lds_tr(sh.vs + 16 * jt * kRS + 16 * h + tro);
      vfr2[jt] = lds_tr(sh.vs + sh.plane2 + 16 * jt * kRS + 16 * h + tro);
    }
  }
  const bf16x4 zero_b = {(elem_t)0.0f, (elem_t)0.0f, (elem_t)0.0f, (elem_t)0.0f};
  // the B operand of a product with the other instance's columns zeroed (PAIR)
  auto only0 = [&](const bf16x4& v) { return half == 0 ? v : zero_b; };
  auto only1 = [&](const bf16x4& v) { return half == 1 ? v : zero_b; };
  int forced[CT];  // evaluate: the given action of the coming step, fetched one step ahead
#pragma unroll
  for (int c = 0; c < CT; ++c)
    forced[c] = (MODE == RL4CO_DECODE_EVALUATE && tj[c].ok) ? (int)a.forced_actions[(int64_t)tj[c].r * a.out_stride + a.t0] : -1;
  __syncthreads();

  // the launch's Philox key, read ONCE: fetched inside the step loop it was a vector-memory load whose wait
  // (vmcnt is one in-order counter) also held the Philox rounds back behind the context row's L2 round trip
  const unsigned long long seed = a.philox_seed ^ (a.philox_seed_dev ? *a.philox_seed_dev : 0ull);
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 7
  unsigned long long acc_[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  int t = 0;
  for (; t < a.max_steps; ++t) {
    bool all_done = true;
#pragma unroll
    for (int c = 0; c < CT; ++c) all_done &= tj[c].done;
    if (!single && __all(all_done)) break;  // identical on every wave
    const int64_t tcol = (int64_t)a.t0 + t;
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 7
    unsigned long long clk_ = __builtin_readcyclecounter();
#endif

    // ---- 1. query of head h (folded context + graph context), x 1/sqrt(16) x log2(e) -----------------
    float4 c4v[CT];
#pragma unroll
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 2  // timing probe: no context-row fetch on the step's chain
    for (int c = 0; c < CT; ++c) c4v[c] = make_float4(0.01f * tj[c].cur, 0.02f, 0.03f, 0.04f);
#else
    for (int c = 0; c < CT; ++c) c4v[c] = rl4co_e16::load_ctx4(ctxc + (int64_t)tj[c].cur * ctx_rs, ctx16);  // L2-resident context row
#endif
    // this step's noise: log(Exp(1) noise) of the four nodes this lane owns in the logits stage (they share
    // one Philox block, rl4co_math.h). Depends on (step, row, node) only, so it is drawn HERE — ten dependent
    // Philox rounds and two logarithms that used to sit between the two barriers, on the step's critical path,
    // now overlap the context row's L2 round trip (just requested) and the glimpse
    float lnz_c[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) lnz_c[c][i] = 0.0f;
      if (MODE == RL4CO_DECODE_SAMPLE && w < NT) {
        const Traj& x = tj[c];
        const int node0 = 16 * w + 4 * g;
        if (a.exp_noise) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            lnz_c[c][i] = (node0 + i < N && x.ok) ? ln_fast(a.exp_noise[((int64_t)t * a.B + x.r) * N + node0 + i]) : 0.0f;
        } else {
          float uu4[4];
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 1  // timing probe: no Philox rounds
          uu4[0] = 0.3f + 1e-3f * (float)(tcol & 7); uu4[1] = 0.5f; uu4[2] = 0.7f; uu4[3] = 0.2f + 1e-3f * (float)node0;
#else
          rl4co_uniform4(seed, a.philox_offset + (uint64_t)tcol,
                         (uint32_t)x.r, (uint32_t)(node0 >> 2), uu4);
#endif
#pragma unroll
          for (int i = 0; i < 4; ++i) lnz_c[c][i] = ln_fast(-ln_fast(uu4[i]));
        }
      }
    }
    RL4CO_MS_MARK(0)  // noise drawn (the context row is in flight)
    bf16x4 qf[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const Traj& x = tj[c];
      const float4 c4 = c4v[c];
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
      float q4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float q;
        if (ENV == RL4CO_ENV_TSP) {
          q = (x.step_i < 1) ? qx4[e] + qb4[e] : (x.f4[e] + cc[e]) + qb4[e];
        } else if (ENV == RL4CO_ENV_PDP) {
          q = cc[e] + qb4[e];  // context.py:232-243: the current node alone
        } else {
          float rem = cap - x.used;
          if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;  // clamp(min=0), context.py:195
          q = fmaf(qx4[e], rem, cc[e]);
          if (kClock) q = fmaf(qt4[e], x.now, q);  // context.py:152-166: second scalar, the current time
          q = q + qb4[e];
        }
        q4[e] = q * (0.25f * kLog2e);
      }
      qf[c] = rl4co_e16::cvt4(q4[0], q4[1], q4[2], q4[3]);  // (two pair conversions: elem16.h)
    }
    // ---- 2. glimpse of head h ---------------------------------------------------------------------------
    RL4CO_MS_MARK(1)  // query ready (includes the wait for the context row)
    {
      f32x4 sc[CT][NT];
      float m[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) m[c] = kNegInf;
#pragma clang loop unroll(full)
      for (int jt = 0; jt < NT; ++jt) {
        const bf16x4 kf = kRegPlanes ? kfr[kRegPlanes ? jt : 0] : lds_b64(sh.kgs + 16 * jt * kRS + 16 * h + nao);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (PAIR) {
            const bf16x4 kf2 = kRegPlanes ? kfr2[kRegPlanes ? jt : 0] : lds_b64(sh.kgs + sh.plane2 + 16 * jt * kRS + 16 * h + nao);
            sc[c][jt] = mfma16(kf2, only1(qf[c]), mfma16(kf, only0(qf[c]), zero4()));
          } else {
            sc[c][jt] = mfma16(kf, qf[c], zero4());
          }
          const uint32_t word = (tj[c].mw.word(jt >> 1) & inner_sel) | (nv.word(jt >> 1) & ~inner_sel);
          const uint32_t bits = word >> (16 * (jt & 1) + 4 * g);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            sc[c][jt][rr] = rl4co::keep_or_neg_inf(bits, rr, sc[c][jt][rr]);
            m[c] = fmaxf(m[c], sc[c][jt][rr]);
          }
        }
      }
      float l[CT];
      f32x4 o0[CT], o1[CT];  // two accumulators per tile: half the dependent-MFMA chain
      // paired: the second instance's products get accumulators of their own and a lane keeps its instance's pair at
      // the end — the same sums, bit for bit, as chaining both through one accumulator with the other instance's
      // columns of the B operand zeroed, without the four selects per node tile that zeroing cost
      f32x4 p0[PAIR ? CT : 1], p1[PAIR ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        m[c] = rg_max(m[c]);
        m[c] = (m[c] > kNegInf) ? m[c] : 0.0f;
        l[c] = 0.0f;
        o0[c] = zero4();
        o1[c] = zero4();
        if (PAIR) {
          p0[c] = zero4();
          p1[c] = zero4();
        }
      }
#pragma clang loop unroll(full)
      for (int jt = 0; jt < NT; ++jt) {
        const bf16x4 vf = kRegPlanes ? vfr[kRegPlanes ? jt : 0] : lds_tr(sh.vs + 16 * jt * kRS + 16 * h + tro);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          float p4[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 5  // timing probe: no exponentials in the glimpse softmax (finite garbage)
            p4[rr] = fmaxf(sc[c][jt][rr] - m[c], -1.0f) + 1.0f;
#else
            p4[rr] = __builtin_amdgcn_exp2f(sc[c][jt][rr] - m[c]);
#endif
            l[c] += p4[rr];
          }
          const bf16x4 pf = rl4co_e16::cvt4(p4[0], p4[1], p4[2], p4[3]);
          if (PAIR) {
            const bf16x4 vf2 = kRegPlanes ? vfr2[kRegPlanes ? jt : 0] : lds_tr(sh.vs + sh.plane2 + 16 * jt * kRS + 16 * h + tro);
            if (jt & 1) {
              o1[c] = mfma16(vf, pf, o1[c]);
              p1[c] = mfma16(vf2, pf, p1[c]);
            } else {
              o0[c] = mfma16(vf, pf, o0[c]);
              p0[c] = mfma16(vf2, pf, p0[c]);
            }
          } else if (jt & 1) {
            o1[c] = mfma16(vf, pf, o1[c]);
          } else {
            o0[c] = mfma16(vf, pf, o0[c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float ls = rg_sum(l[c]);
        const float inv = (ls > 0.0f) ? __builtin_amdgcn_rcpf(ls) : 0.0f;
        float o4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float acc = o0[c][rr] + o1[c][rr];
          if (PAIR) acc = half == 0 ? acc : p0[c][rr] + p1[c][rr];
          o4[rr] = acc * inv;
        }
        *reinterpret_cast<bf16x4*>(sh.hs + (16 * c + tl) * kRS + dcol) = rl4co_e16::cvt4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
    RL4CO_MS_MARK(2)  // glimpse written
    rl4co::lds_barrier();  // B1 (LDS only: parked stores stay in flight)
    RL4CO_MS_MARK(3)  // B1 passed

    // ---- 3. logits of node tile w, local log-softmax / selection pieces ------------------------------------
    if (w < NT) {
      f32x4 u0[CT], u1[CT], v0[PAIR ? CT : 1], v1[PAIR ? CT : 1];  // paired: second instance apart, as in the glimpse
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        u0[c] = zero4();
        u1[c] = zero4();
        if (PAIR) {
          v0[c] = zero4();
          v1[c] = zero4();
        }
      }
#pragma clang loop unroll(full)
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x4 lf = lfr[ks];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const bf16x4 hf = lds_b64(sh.hs + 16 * c * kRS + 16 * ks + nao);
          if (PAIR) {
            const bf16x4 lf2 = lfr2[PAIR ? ks : 0];
            if (ks & 1) {
              u1[c] = mfma16(lf, hf, u1[c]);
              v1[c] = mfma16(lf2, hf, v1[c]);
            } else {
              u0[c] = mfma16(lf, hf, u0[c]);
              v0[c] = mfma16(lf2, hf, v0[c]);
            }
          } else if (ks & 1) {
            u1[c] = mfma16(lf, hf, u1[c]);
          } else {
            u0[c] = mfma16(lf, hf, u0[c]);
          }
        }
      }
      const int node0 = 16 * w + 4 * g;  // this lane's four consecutive nodes
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const Traj& x = tj[c];
        const uint32_t mword = x.mw.word(w >> 1), nword = nv.word(w >> 1);
        const uint32_t lbits = ((mword & logit_sel) | (nword & ~logit_sel)) >> (16 * (w & 1) + 4 * g);
        const float* lnz = lnz_c[c];  // drawn at the top of the step
        Sel p;
        float z[4];
        p.zmax = kNegInf;
        bool nan_seen = false;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          // this variant is tolerance-tested, not bit-exact: hardware reciprocals instead of IEEE division
          float usum = u0[c][rr] + u1[c][rr];
          if (PAIR) usum = half == 0 ? usum : v0[c][rr] + v1[c][rr];
          const float uu = usum * (1.0f / kSqrtD);
          nan_seen |= uu != uu;
#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 6  // timing probe: no tanh
          const float th = fminf(fmaxf(uu, -1.0f), 1.0f) * clip_over_temp;
#else
          const float ex = __expf(-2.0f * fabsf(uu));
          const float th = copysignf((1.0f - ex) * __builtin_amdgcn_rcpf(1.0f + ex), uu) * clip_over_temp;
#endif
          const float zz = clip ? th : uu * inv_temp;
          z[rr] = ((lbits >> rr) & 1u) ? zz : kNegInf;
        }
        if (nan_seen & x.ok & !x.done) errbits |= RL4CO_EBIT_NAN_LOGIT;
        float zs = 0.0f;
        if (bounded) {
          p.zmax = 0.0f;
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) p.zmax = fmaxf(p.zmax, z[rr]);
          zs = (p.zmax > kNegInf) ? p.zmax : 0.0f;
        }
        p.se = 0.0f;
        p.key = kNegInf;
        p.z = kNegInf;
        p.fz = kNegInf;
        p.idx = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int node = node0 + i;
          const float zz = z[i];
          p.se += __expf(zz - zs);
          // multinomial(p,1) == argmax(p / Exp(1)) == argmax(z - log(noise)); greedy: noise = 1
          const float key = (MODE == RL4CO_DECODE_SAMPLE) ? zz - lnz[i] : zz;
          const bool take = (zz > kNegInf) & ((key > p.key) | (p.idx == 0x7fffffff));  // ascending nodes: ties keep the lower
          p.key = take ? key : p.key;
          p.z = take ? zz : p.z;
          p.idx = take ? node : p.idx;
          if (MODE == RL4CO_DECODE_EVALUATE) p.fz = (node == forced[c]) ? zz : p.fz;
        }
        // the four row groups hold different nodes of the same trajectory
        p = merge(p, partner<16>(p), bounded);
        p = merge(p, partner<32>(p), bounded);
        if (g == 0) {
          Xchg e;
          e.zmax = p.zmax;
          e.se = p.se;
          e.best_key = p.key;
          e.best_z = p.z;
          e.best_idx = p.idx;
          e.forced_z = p.fz;
          e.pad0 = 0;
          e.pad1 = 0;
          sh.xs[(c * kWaves + w) * 16 + tl] = e;
        }
      }
    }
    RL4CO_MS_MARK(4)  // logits / selection pieces written
    rl4co::lds_barrier();  // B2
    RL4CO_MS_MARK(5)  // B2 passed

    // ---- 4. every lane: finish the selection of its trajectory, transition ------------------------------
    // row group g folds node tiles g and g + 4, then the row groups meet in two butterfly steps
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      Traj& x = tj[c];
      Sel p;
      {
        const Xchg e = sh.xs[(c * kWaves + g) * 16 + tl];  // g < 4 <= NT... (NT >= 2: tiles 0, 1 exist; others guarded)
        const bool has = g < NT;
        p.zmax = has ? e.zmax : kNegInf;
        p.se = has ? e.se : 0.0f;
        p.key = has ? e.best_key : kNegInf;
        p.z = has ? e.best_z : kNegInf;
        p.idx = has ? e.best_idx : 0x7fffffff;
        p.fz = has ? e.forced_z : kNegInf;
      }
      if (NT > 4) {
        const Xchg e = sh.xs[(c * kWaves + g + 4) * 16 + tl];
        const bool has = g + 4 < NT;
        Sel q;
        q.zmax = has ? e.zmax : kNegInf;
        q.se = has ? e.se : 0.0f;
        q.key = has ? e.best_key : kNegInf;
        q.z = has ? e.best_z : kNegInf;
        q.idx = has ? e.best_idx : 0x7fffffff;
        q.fz = has ? e.forced_z : kNegInf;
        p = merge(p, q, bounded);
      }
      p = merge(p, partner<16>(p), bounded);
      p = merge(p, partner<32>(p), bounded);
      const float lse = (bounded ? 0.0f : p.zmax) + ln_fast(p.se);
      int act;
      float logp;
      if (MODE == RL4CO_DECODE_EVALUATE) {
        logp = p.fz - lse;
        act = x.ok ? forced[c] : 0;
        if (act < 0 || act >= N) {
          if (x.ok && !x.done) errbits |= RL4CO_EBIT_INFEASIBLE;
          act = 0;
        }
      } else {
        act = (p.idx == 0x7fffffff) ? 0 : p.idx;
        logp = p.z - lse;
      }
      if (!x.done) {
        if (!x.mw.test(act)) errbits |= RL4CO_EBIT_INFEASIBLE;
        if (!(logp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
        if ((t & 31) == 4 * w + g) {
          x.park_act = act;
          x.park_logp = logp;
          x.park_col = tcol;
        }
        x.nsteps = t + 1;
        // environment transition on the lane-resident state
        if (ENV == RL4CO_ENV_TSP) {
          if (x.step_i == 0) {
            x.first = act;
            first_row(x.first, x.f4);
          }
          x.cur = act;
          x.step_i += 1;
          x.mw.set(act, false);
          x.done = x.mw.any() == 0u;
        } else if (ENV == RL4CO_ENV_PDP) {
          // pdp/env.py:64-83: the node leaves `available`, its delivery (the depot's: nothing) becomes deliverable
          const int n = N - 1;
          x.vw.set(act, false);
          x.tw.set((act + n / 2) % (n + 1), true);
#pragma unroll
          for (int k = 0; k < 4; ++k) x.mw.put(k, x.vw.word(k) & x.tw.word(k));
          x.done = x.vw.any() == 0u;
          x.step_i += 1;
          x.cur = act;
        } else if (ENV == RL4CO_ENV_PCTSP) {
          // pctsp/env.py:62-75, 141-148: customers while unvisited and the depot not yet closed; the depot opens once a
          // total prize of 1 is collected or no customer is left
          x.used = x.used + dems_l[act];
          x.vw.set(act, true);
          x.done = (x.step_i > 0) && (act == 0);
          x.step_i += 1;
          x.cur = act;
          const bool closed = x.vw.w0 & 1u;
          uint32_t left = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t c = ~x.vw.word(k) & nv.word(k) & (k == 0 ? ~1u : ~0u);
            left |= c;
            x.mw.put(k, closed ? 0u : c);
          }
          if (!((x.used < 1.0f) && left != 0u)) x.mw.w0 |= 1u;
        } else if (ENV == RL4CO_ENV_OP) {
          // op/env.py:67-98, 137-154: unvisited, depot not yet closed, and the node can still be entered
          {
            const float dx = locs[2 * act] - locs[2 * x.cur], dy = locs[2 * act + 1] - locs[2 * x.cur + 1];
            x.used = x.used + sqrtf(fmaf(dy, dy, dx * dx));
          }
          x.vw.set(act, true);
          x.done = (act == 0) && (x.step_i > 0);
          x.step_i += 1;
          x.cur = act;
          const bool closed = x.vw.w0 & 1u;
          const float cx = locs[2 * act], cy = locs[2 * act + 1];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint32_t mbits = 0;
            for (int b = 0; b < 32 && 32 * k + b < N; ++b) {
              const int j = 32 * k + b;
              const float dx = locs[2 * j] - cx, dy = locs[2 * j + 1] - cy;
              const bool exceeds = x.used + sqrtf(fmaf(dy, dy, dx * dx)) > opmax[j];
              const bool v = (x.vw.word(k) >> b) & 1u;
              mbits |= (v || closed || exceeds) ? 0u : (1u << b);
            }
            x.mw.put(k, mbits);
          }
          x.mw.w0 |= 1u;  // the depot can always be visited
        } else {
          const float cx = kClock ? locs[2 * act] : 0.0f, cy = kClock ? locs[2 * act + 1] : 0.0f;
          if (kClock) {  // cvrptw/env.py:97-113: advance by the distance, wait for the window, serve; the depot restarts the clock
            const float dx = cx - locs[2 * x.cur], dy = cy - locs[2 * x.cur + 1];
            x.now = (act != 0 ? 1.0f : 0.0f) * (fmaxf(x.now + sqrtf(fmaf(dy, dy, dx * dx)), twin[2 * act]) + dur[act]);
          }
          const int di = min(max(act - 1, 0), N - 2);
          x.used = (x.used + dems_l[di + 1]) * (act != 0 ? 1.0f : 0.0f);
          x.cur = act;
          x.vw.set(act, true);
          // (every lane of the trajectory's column rebuilds the whole mask: splitting the nodes over the four row groups
          // and exchanging the words was measured and is no faster — the step is a latency chain, not this loop)
          bool all_visited = true, any_feasible = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint32_t mbits = 0;
            for (int b = 0; b < 32 && 32 * k + b < N; ++b) {
              const int j = 32 * k + b;
              const bool v = (x.vw.word(k) >> b) & 1u;
              all_visited &= v;
              if (j >= 1) {
                const bool masked = v || (dems_l[j] + x.used > thr);
                mbits |= masked ? 0u : (1u << b);
                any_feasible |= !masked;
              }
            }
            x.mw.put(k, mbits);
          }
          if (!((x.cur == 0) && any_feasible)) x.mw.w0 |= 1u;
          if (kClock) {  // cvrptw/env.py:91-95: only nodes whose window is still open on arrival (the depot too)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              uint32_t keep = 0;
              for (int b = 0; b < 32 && 32 * k + b < N; ++b) {
                const int j = 32 * k + b;
                const float dx = locs[2 * j] - cx, dy = locs[2 * j + 1] - cy;
                keep |= (x.now + sqrtf(fmaf(dy, dy, dx * dx)) <= twin[2 * j + 1]) ? (1u << b) : 0u;
              }
              x.mw.put(k, x.mw.word(k) & keep);
            }
          }
          x.done = all_visited;
        }
      }
      if (MODE == RL4CO_DECODE_EVALUATE && x.ok && t + 1 < a.max_steps)
        forced[c] = (int)a.forced_actions[(int64_t)x.r * a.out_stride + tcol + 1];
      if (((t & 31) == 31 || single) && x.park_col >= 0 && x.ok) {
        a.actions[(int64_t)x.r * a.out_stride + x.park_col] = x.park_act;
        a.logps[(int64_t)x.r * a.out_stride + x.park_col] = x.park_logp;
        x.park_col = -1;
      }
    }
    RL4CO_MS_MARK(6)  // selection finished, state advanced
    if (single) break;
  }

#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 7
  if (lane == 0 && blockIdx.x < 256)
    for (int i = 0; i < 7; ++i) atomicAdd(&g_ms_clk[(w == 0 ? 0 : 8) + i], acc_[i]);
#endif
  // ---- parked outputs, final state of the column tiles ------------------------------------------------------
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    Traj& x = tj[c];
    if (x.park_col >= 0 && x.ok) {
      a.actions[(int64_t)x.r * a.out_stride + x.park_col] = x.park_act;
      a.logps[(int64_t)x.r * a.out_stride + x.park_col] = x.park_logp;
    }
    if (w == 0 && g == 0 && x.ok) {
      uint8_t* gm = a.action_mask + (int64_t)x.r * N;
      for (int j = 0; j < N; ++j) gm[j] = x.mw.test(j) ? 1 : 0;
      if (kVisited) {
        uint8_t* gv = a.visited + (int64_t)x.r * N;
        for (int j = 0; j < N; ++j) gv[j] = x.vw.test(j) ? 1 : 0;
      }
      if (ENV == RL4CO_ENV_PDP) {
        uint8_t* gd = a.to_deliver + (int64_t)x.r * N;
        for (int j = 0; j < N; ++j) gd[j] = x.tw.test(j) ? 1 : 0;
      }
      if (kScalar) a.used_capacity[x.r] = x.used;
      if (kClock) a.current_time[x.r] = x.now;
      if (ENV == RL4CO_ENV_TSP) a.first_node[x.r] = x.first;
      if (kStepI) a.step_i[x.r] = x.step_i;
      a.current_node[x.r] = x.cur;
      a.done[x.r] = x.done ? 1 : 0;
      if (a.n_steps) a.n_steps[x.r] = x.nsteps;
      if (a.steps_summary) {
        atomicMax(a.steps_summary, x.nsteps);
        atomicAdd(a.steps_summary + 1, x.nsteps);
      }
      if (!single && !x.done) errbits |= RL4CO_EBIT_MAX_STEPS;
    }
  }
  __syncthreads();
}

// plane / instance-data staging of ONE instance into copy `copy` of the workgroup's LDS
template <int ENV, int NT>
__device__ __forceinline__ void stage_instance(const rl4co_am_decode_args& a, const Layout& L, unsigned char* smem, int inst, int copy,
                                               bool present) {
  const int tid = threadIdx.x;
  const int N = a.N;
  elem_t* kgs = reinterpret_cast<elem_t*>(smem + L.kgs) + copy * NT * 16 * kRS;  // [16 NT nodes][kRS] glimpse keys
  elem_t* vs = reinterpret_cast<elem_t*>(smem + L.vs) + copy * NT * 16 * kRS;    // glimpse values
  float* dems = reinterpret_cast<float*>(smem + L.dems) + copy * 128;            // [128] CVRP demands (index j-1 at j), 0 elsewhere
  const uint16_t* gk = static_cast<const uint16_t*>(a.glimpse_key) + (int64_t)inst * a.kvl_batch_stride;
  const uint16_t* gv = static_cast<const uint16_t*>(a.glimpse_val) + (int64_t)inst * a.kvl_batch_stride;
  for (int c = tid; c < NT * 16 * 16; c += kThreads) {  // 16-byte chunks: row = c / 16, col = (c % 16) * 8
    const int row = c >> 4, col = (c & 15) * 8;
    uint4 k4 = make_uint4(0, 0, 0, 0), v4 = k4;
    if (row < N && present) {
      k4 = *reinterpret_cast<const uint4*>(gk + (int64_t)row * a.kvl_row_stride + col);
      v4 = *reinterpret_cast<const uint4*>(gv + (int64_t)row * a.kvl_row_stride + col);
    }
    *reinterpret_cast<uint4*>(kgs + row * kRS + col) = k4;
    *reinterpret_cast<uint4*>(vs + row * kRS + col) = v4;
  }
  constexpr bool kDem = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  for (int j = tid; j < 128; j += kThreads) {
    float d = 0.0f;
    if (kDem && j >= 1 && j < N && present) d = a.demand[(int64_t)inst * (N - 1) + j - 1];
    if (ENV == RL4CO_ENV_PCTSP && j < N && present) d = a.demand[(int64_t)inst * N + j];  // real prize, depot column 0
    dems[j] = d;
  }
  if (ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_CVRPTW) {
    float* envf = reinterpret_cast<float*>(smem + L.envf) + copy * 768;
    for (int j = tid; j < 128; j += kThreads) {
      const bool in = j < N && present;
      envf[2 * j] = in ? a.locs[((int64_t)inst * N + j) * 2] : 0.0f;
      envf[2 * j + 1] = in ? a.locs[((int64_t)inst * N + j) * 2 + 1] : 0.0f;
      envf[256 + j] = (ENV == RL4CO_ENV_OP && in) ? a.max_length[(int64_t)inst * N + j] : 0.0f;
      envf[384 + 2 * j] = (ENV == RL4CO_ENV_CVRPTW && in) ? a.time_windows[((int64_t)inst * N + j) * 2] : 0.0f;
      envf[384 + 2 * j + 1] = (ENV == RL4CO_ENV_CVRPTW && in) ? a.time_windows[((int64_t)inst * N + j) * 2 + 1] : 0.0f;
      envf[640 + j] = (ENV == RL4CO_ENV_CVRPTW && in) ? a.durations[(int64_t)inst * N + j] : 0.0f;
    }
  }
}

// PAIR: at most 8 starts per instance -> one workgroup = two instances in one column tile (see make_layout); two waves per
// SIMD (256 registers: the second logit-key tile), one workgroup per CU
template <int ENV, int NT, int MODE, bool PAIR>
__global__ void __launch_bounds__(kThreads, PAIR ? 2 : 4) am_decode_ms_kernel(const rl4co_am_decode_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int N = a.N;
  const int S = a.B / a.B_inst;
  // one workgroup per (instance, column tile of 16 starts). Workgroup b lands on XCD b % 8 (observed placement, speed
  // only): the tiles of one instance take consecutive slots of one XCD and share its L2 copy of the planes.
  const int ntiles = (S + 15) >> 4, b = blockIdx.x;
  int inst, tile0;
  if (PAIR) {
    inst = 2 * b;
    tile0 = 0;
  } else if ((a.B_inst & 7) == 0) {
    const int xcd = b & 7, k = b >> 3;
    inst = (k / ntiles) * 8 + xcd;
    tile0 = k % ntiles;
  } else {
    inst = b / ntiles;
    tile0 = b % ntiles;
  }
  const Layout L = make_layout(NT, N, PAIR ? 2 : 1);
  if ((tid >> 6) >= 4) __builtin_amdgcn_s_setprio(1);  // even out the younger half of the workgroup (issue arbitration)
  // ---- glimpse planes and instance data HBM / L2 -> LDS, once per workgroup ----------------------------------
  const bool second = PAIR && inst + 1 < a.B_inst;
  stage_instance<ENV, NT>(a, L, smem, inst, 0, true);
  if (PAIR) stage_instance<ENV, NT>(a, L, smem, second ? inst + 1 : inst, 1, second);
  Shared sh;
  sh.kgs = reinterpret_cast<elem_t*>(smem + L.kgs);
  sh.vs = reinterpret_cast<elem_t*>(smem + L.vs);
  sh.plane2 = NT * 16 * kRS;
  sh.kl_g = static_cast<const uint16_t*>(a.logit_key) + (int64_t)inst * a.kvl_batch_stride;
  sh.kl_g2 = static_cast<const uint16_t*>(a.logit_key) + (int64_t)(second ? inst + 1 : inst) * a.kvl_batch_stride;
  sh.kl_rs = a.kvl_row_stride;
  sh.hs = reinterpret_cast<elem_t*>(smem + L.hs);
  sh.xs = reinterpret_cast<Xchg*>(smem + L.xs);
  sh.dems = reinterpret_cast<const float*>(smem + L.dems);
  sh.envf = reinterpret_cast<const float*>(smem + L.envf);
  uint32_t errbits = 0;
  if (tid == 0 && a.steps_summary)  // the planes are read ONCE per workgroup
    atomicAdd(reinterpret_cast<unsigned long long*>(a.steps_summary + 2), (unsigned long long)(N * (second ? 2 : 1)));
  rollout_tiles<ENV, NT, MODE, 1, PAIR>(a, sh, inst, 16 * tile0, errbits);  // this workgroup's column tile
  if (errbits) atomicOr(a.err, (int)errbits);
}

template <int ENV, int NT, int MODE>
int launch_mode(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int starts = a.B / a.B_inst;
  if (starts <= 8 && a.B_inst >= 2) {  // two instances per column tile (make_layout)
    const Layout L = make_layout(NT, a.N, 2);
    RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_ms_kernel<ENV, NT, MODE, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    hipLaunchKernelGGL((am_decode_ms_kernel<ENV, NT, MODE, true>), dim3((a.B_inst + 1) / 2), dim3(kThreads), L.total, stream, a);
    RL4CO_HIP_TRY(hipGetLastError());
    return RL4CO_OK;
  }
  const Layout L = make_layout(NT, a.N);
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_decode_ms_kernel<ENV, NT, MODE, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
  const int ntiles = (starts + 15) / 16;  // one workgroup per (instance, column tile of 16 starts)
  hipLaunchKernelGGL((am_decode_ms_kernel<ENV, NT, MODE, false>), dim3(a.B_inst * ntiles), dim3(kThreads), L.total, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}

template <int ENV, int NT>
int launch_tiles(const rl4co_am_decode_args& a, hipStream_t stream) {
  if (a.mode == RL4CO_DECODE_GREEDY) return launch_mode<ENV, NT, RL4CO_DECODE_GREEDY>(a, stream);
  if (a.mode == RL4CO_DECODE_SAMPLE) return launch_mode<ENV, NT, RL4CO_DECODE_SAMPLE>(a, stream);
  return launch_mode<ENV, NT, RL4CO_DECODE_EVALUATE>(a, stream);
}

template <int ENV>
int dispatch_tiles(const rl4co_am_decode_args& a, hipStream_t stream) {
  const int nt = (a.N + 15) >> 4;
  if (nt <= 2) return launch_tiles<ENV, 2>(a, stream);
  if (nt <= 4) return launch_tiles<ENV, 4>(a, stream);
  if (nt <= 7) return launch_tiles<ENV, 7>(a, stream);
  return launch_tiles<ENV, 8>(a, stream);
}

}  // namespace

#if defined(RL4CO_MS_PROBE) && RL4CO_MS_PROBE == 7
extern "C" int rl4co_ms_probe_read(unsigned long long* out, int reset) {
  RL4CO_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ms_clk), sizeof(g_ms_clk)));
  if (reset) {
    unsigned long long z[16] = {0};
    RL4CO_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ms_clk), z, sizeof(z)));
  }
  return RL4CO_OK;
}
#endif

namespace rl4co {
int RL4CO_CXX(launch_decode_ms)(const rl4co_am_decode_args& a, hipStream_t stream) {
#ifdef RL4CO_MS_PROBE_ONLY  // timing probes (tools/ms_variants.sh): one instantiation instead of 144 — a 10 s build
  return launch_mode<RL4CO_ENV_TSP, 7, RL4CO_DECODE_SAMPLE>(a, stream);
#else
  switch (a.env) {
    case RL4CO_ENV_TSP: return dispatch_tiles<RL4CO_ENV_TSP>(a, stream);
    case RL4CO_ENV_CVRP: return dispatch_tiles<RL4CO_ENV_CVRP>(a, stream);
    case RL4CO_ENV_OP: return dispatch_tiles<RL4CO_ENV_OP>(a, stream);
    case RL4CO_ENV_PCTSP: return dispatch_tiles<RL4CO_ENV_PCTSP>(a, stream);
    case RL4CO_ENV_PDP: return dispatch_tiles<RL4CO_ENV_PDP>(a, stream);
    case RL4CO_ENV_CVRPTW: return dispatch_tiles<RL4CO_ENV_CVRPTW>(a, stream);
    default: return RL4CO_ERR_ARG;
  }
#endif
}
}  // namespace rl4co
