// am_teacher_mma.hip — teacher-forced backward on the matrix cores.
//
// With the actions known, every decode step of a trajectory is known up front
// (decode_type="evaluate", utils/decoding.py:448-461; the two-phase pattern of rl/ppo/ppo.py:128-170):
// the T queries, the T feasibility masks and the T upstream gradients can be laid side by side and
// the per-step GEMVs of am_teacher.hip become small GEMMs with the STEP as the MFMA column:
//
//   scores^T[j,t] = Kg_h[j,:] . Q_h^T[:,t]      glimpse O_h^T[d,t] = sum_j V_h^T[d,j] P^T[j,t]
//   logits^T[j,t] = Kl[j,:] . O^T[:,t]          dO^T[d,t] = sum_j Kl^T[d,j] dU^T[j,t]
//   dA^T[j,t]     = V_h[j,:] . dO_h^T[:,t]      dQ_h^T[d,t] = sum_j Kg_h^T[d,j] dS^T[j,t]
//   dKl^T[d,j] += sum_t O^T[d,t] dU[t,j]        dV_h^T[d,j] += sum_t dO_h^T[d,t] A[t,j]
//   dKg_h^T[d,j] += sum_t Q_h^T[d,t] dS[t,j]
//
// One 512-thread workgroup per INSTANCE (8 waves = 8 heads, two waves per SIMD so the VALU of one
// overlaps the MFMAs of the other); the LIVE steps of its S multistart trajectories are laid side by
// side (make_layout: packed columns, r06) and replayed in blocks of 16 columns, all products on
// v_mfma_f32_16x16x16_bf16. In that instruction's
// accumulator layout a lane owns ONE step (column lane & 15) and four consecutive rows, which is
// also its B-operand layout: softmax / log-softmax over nodes are in-lane reductions plus two
// cross-row-group exchanges, and accumulators chain into the next product without a shuffle.
// The three bf16 planes sit in LDS once, in their natural [node][dim] layout; the products that
// contract over nodes or over steps read them (and the 16-step staging blocks) through the
// gfx950 transpose read ds_read_b64_tr_b16, so no transposed copy exists. The gradients of the
// three planes accumulate in registers over all S x T steps of the instance (84 registers per
// lane) and are written once — no atomics on the planes; only the context-row scatter
// (d ctx_cur[cur_t]) uses fp32 L2 atomics, from the one workgroup that owns the instance.
//
// Numerics: bf16 MFMA operands (planes, queries, softmax numerators, glimpses, dU, dS), fp32
// accumulation and fp32 softmax / tanh / log-softmax — the mixed-precision regime the reference
// trains in (utils/trainer.py:57 precision="16-mixed"). Tested against am_teacher.hip (fp32 replay)
// and torch autograd by tolerance (tests/test_gpu_teacher.py).
#include <hip/hip_runtime.h>

#include "common.h"
#include "elem16.h"

namespace {

constexpr int kD = RL4CO_EMBED_DIM;
constexpr int kRS = kD + 8;  // LDS row stride (bf16 elements): conflict-free 8-byte row reads
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kMaxTiles = 8;  // node tiles of 16: N <= 128
constexpr int kMaxT = 256;    // action columns the step tables hold
constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kSqrtD = 11.3137084989847604f;
constexpr float kLog2e = 1.44269504088896341f;

typedef elem_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));  // elementwise fp32 chains two at a time (v_pk_add / mul / fma_f32)
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// C[m = 4 g + r][n = lane & 15] += sum_k A[m = lane & 15][k = 4 g + s] * B[k = 4 g + s][n = lane & 15]
__device__ inline f32x4 mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return rl4co_e16::mfma_16x16x16(a, b, c);
}
__device__ inline f32x4 zero4() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
__device__ inline bf16x4 lds_b64(const elem_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ inline rl4co_e16::e8 lds_b128(const elem_t* p) { return *reinterpret_cast<const rl4co_e16::e8*>(p); }
// ds_read_b64_tr_b16: the 16 lanes of a row group address a [4 rows][16 columns] block (lane i:
// row i / 4, columns 4 (i % 4) ..) and lane c receives column c of it — four consecutive ROWS
__device__ inline bf16x4 lds_tr(const elem_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ inline bf16x4 pack4(const f32x2& a, const f32x2& b) {  // two v_cvt_pk: (a0, a1), (b0, b1)
  return __builtin_bit_cast(bf16x4, u32x2{rl4co_e16::pack(a[0], a[1]), rl4co_e16::pack(b[0], b[1])});
}
__device__ inline bf16x4 to_bf16(const f32x4& v) { return rl4co_e16::cvt4(v[0], v[1], v[2], v[3]); }  // two pair conversions (elem16.h)
// LDS hand-off inside ONE wave
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// across the four row groups of a wave (lanes differing in bits 4, 5)
__device__ inline float rg_sum(float v) { return rl4co::bfly_sum<16, 64>(v); }
__device__ inline float rg_max(float v) { return rl4co::bfly_max<16, 64>(v); }
// across the sixteen steps of a row group (lane bits 0..3)
__device__ inline float step_sum(float v) { return rl4co::bfly_sum<1, 16>(v); }

struct Layout {  // byte offsets into dynamic LDS
  int kgs, vs, kls, ob, dub, qb, pb, scol, sstep, sg, srem, stime, smask, sraw, spos, xz, xa, sinfo, nact, ng;
  int mc, npre, total;
};
// PACKED COLUMNS (r06). A trajectory used to run in step blocks of its own: at TSP-100 seven blocks, the last one with 4 of
// its 16 columns alive, and column 0 (the imposed multistart node, gradient 0) always dead — 56 blocks per instance for 792
// live steps. Every column of a block is independent of the others (own mask, own query, own upstream gradient), so the LIVE
// columns t0 <= t < t_end of a GROUP of consecutive trajectories are now laid side by side in one column table of `mc`
// entries and walked in blocks of 16: 50 blocks at TSP-100 x 8 starts, one table set-up per group instead of per trajectory.
// What a column needs of its trajectory travels with it: action, current node, first node, step index, trajectory slot.
__host__ __device__ inline Layout make_layout(int nt, int env) {
  Layout L;
  const int plane = nt * 16 * kRS * 2, blk = 16 * kRS * 2;
  L.mc = nt <= 7 ? 416 : kMaxT;   // packed columns per group (a multiple of 16; >= kMaxT so that one trajectory always fits)
  L.npre = nt <= 7 ? 512 : 0;     // raw columns (actions, upstream gradients) fetched one group ahead; 0: read in place
  const int mcp = L.mc + 16;      // + one block: a block's look-ahead reads (next current / first node) stay inside the tables
  const bool scalar = env != RL4CO_ENV_TSP && env != RL4CO_ENV_PDP;
  int o = 0;
  L.kgs = o; o += plane;
  L.vs = o; o += plane;
  L.kls = o; o += plane;
  L.ob = o; o += blk;
  L.dub = o; o += blk;
  L.qb = o; o += blk;
  L.pb = o; o += kWaves * blk;
  // one word per packed column — byte 0 action | byte 1 current node (the previous action of its trajectory; 0 at step 0) |
  // byte 2 first node of its trajectory (TSP: the row of ctx_first) | byte 3 flags: bit 0 valid, bit 1 step 0, bits 2-6
  // trajectory slot inside the group, bit 7 "late" (stage 5). ONE ds_read_b32 per block and lane (and one for the look-ahead);
  // as five byte tables every field cost its own address add, read and mask on a VALU-issue-bound chain
  L.scol = o; o += mcp * 4;
  L.sstep = o; o += mcp;   // step index inside its trajectory (read by the one lane that writes logp_out)
  o = (o + 3) & ~3;
  L.sg = o; o += mcp * 4;
  L.srem = o; o += scalar ? mcp * 4 : 0;
  L.stime = o; o += env == RL4CO_ENV_CVRPTW ? mcp * 4 : 0;
  o = (o + 15) & ~15;
  L.smask = o; o += mcp * 16;
  const int nq = nt <= 7 ? 4 : 1;   // trajectories tabulated at once (eight node tiles leave no LDS for more than one)
  L.sraw = o; o += nq * kMaxT;      // the trajectories under set-up: their T actions
  L.spos = o; o += nq * 128 * 4;    // ... and the first column that visits node j
  L.xz = o; o += kWaves * 16 * 2 * 4;
  L.xa = o; o += 16 * 4;
  L.sinfo = o; o += 16;
  L.nact = o; o += L.npre;      // the NEXT group's actions (bytes; 255 = out of range) and
  L.ng = o; o += L.npre * 4;    // upstream gradients, fetched under this group's set-up
  L.total = (o + 15) & ~15;
  return L;
}

// four consecutive fp32 values through L2 (agent scope: the CU's L1 is bypassed). The context-row sums are accumulated by
// plain read-modify-write AND (blocks that hold two trajectories, the final flush) by L2 atomics: an L1 line filled before
// an atomic landed would serve a stale row.
__device__ inline float4 load4_l2(const float* p) {
  float4 v;
  v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
template <int ENV, int NT, bool CTX16>
__global__ void __launch_bounds__(kThreads, 2) am_teacher_mma_kernel(const rl4co_am_teacher_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int h = w;  // attention stages: wave = head; logits stage: wave = node tile
  const int inst = blockIdx.x;
  const int N = a.N, T = a.T, S = a.B / a.B_inst;
  // the second-dispatched half of an 8-wave workgroup loses the issue arbitration on every segment
  // (older wave first); a static priority for it evens the halves out between the barriers
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  const Layout L = make_layout(NT, ENV);  // NT node tiles of 16 (template): rows N .. 16 NT - 1 are zero
  elem_t* kgs = reinterpret_cast<elem_t*>(smem + L.kgs);
  elem_t* vs = reinterpret_cast<elem_t*>(smem + L.vs);
  elem_t* kls = reinterpret_cast<elem_t*>(smem + L.kls);
  elem_t* ob = reinterpret_cast<elem_t*>(smem + L.ob);    // [16 steps][kRS] glimpses of the block
  elem_t* dub = reinterpret_cast<elem_t*>(smem + L.dub);  // [16 steps][kRS] d logits (pre-clip, raw)
  elem_t* qb = reinterpret_cast<elem_t*>(smem + L.qb);    // [16 steps][kRS] queries (x 0.25 log2 e)
  // d glimpse / softmax denominator of the block takes the glimpses' place: after B3 a wave reads only its own head's
  // columns of `ob` (the O_h^T operand of d Kl) and writes the same columns of `dob` afterwards — eight wave-private
  // column strips, ordered by program order within the wave. 4 KB that let eight node tiles (N <= 128) fit in 160 KB
  elem_t* dob = ob;
  elem_t* pbw = reinterpret_cast<elem_t*>(smem + L.pb) + w * 16 * kRS;  // this wave's [16 steps][kRS] P, then dS
  uint32_t* scol = reinterpret_cast<uint32_t*>(smem + L.scol);
  uint8_t* sstep = smem + L.sstep;
  float* sg = reinterpret_cast<float*>(smem + L.sg);
  float* srem = reinterpret_cast<float*>(smem + L.srem);
  float* stime = reinterpret_cast<float*>(smem + L.stime);  // CVRPTW: the clock before each column
  uint32_t* smask = reinterpret_cast<uint32_t*>(smem + L.smask);
  uint8_t* sraw = smem + L.sraw;
  int* spos = reinterpret_cast<int*>(smem + L.spos);
  uint8_t* snact = smem + L.nact;
  float* sng = reinterpret_cast<float*>(smem + L.ng);
  float* xz = reinterpret_cast<float*>(smem + L.xz);
  float* xa = reinterpret_cast<float*>(smem + L.xa);
  int* sinfo = reinterpret_cast<int*>(smem + L.sinfo);

  // per-lane element offsets: natural operand (row = lane & 15, 4 consecutive columns at 4 g) and
  // transpose read (row 4 g + (lane & 15) / 4, columns 4 (lane & 3))
  const int nao = tl * kRS + 4 * g;
  const int tro = (4 * g + (tl >> 2)) * kRS + 4 * (tl & 3);
  const int dcol = 16 * h + 4 * g;  // the four dims of head h this lane owns in accumulator layout

  // ---- planes HBM -> LDS once per instance (rows >= N zero: they are contracted over) ------------
  {
    const uint16_t* gk = static_cast<const uint16_t*>(a.glimpse_key) + (int64_t)inst * a.kvl_batch_stride;
    const uint16_t* gv = static_cast<const uint16_t*>(a.glimpse_val) + (int64_t)inst * a.kvl_batch_stride;
    const uint16_t* gl = static_cast<const uint16_t*>(a.logit_key) + (int64_t)inst * a.kvl_batch_stride;
    for (int c = tid; c < NT * 16 * 16; c += kThreads) {
      const int row = c >> 4, col = (c & 15) * 8;
      uint4 k4 = make_uint4(0, 0, 0, 0), v4 = k4, l4 = k4;
      if (row < N) {
        k4 = *reinterpret_cast<const uint4*>(gk + (int64_t)row * a.kvl_row_stride + col);
        v4 = *reinterpret_cast<const uint4*>(gv + (int64_t)row * a.kvl_row_stride + col);
        l4 = *reinterpret_cast<const uint4*>(gl + (int64_t)row * a.kvl_row_stride + col);
      }
      *reinterpret_cast<uint4*>(kgs + row * kRS + col) = k4;
      *reinterpret_cast<uint4*>(vs + row * kRS + col) = v4;
      *reinterpret_cast<uint4*>(kls + row * kRS + col) = l4;
    }
  }
  float* dcc = a.d_ctx_cur + (int64_t)inst * N * kD;
  for (int i = tid; i < N * kD / 4; i += kThreads) reinterpret_cast<float4*>(dcc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the zeros are acknowledged by L2 before this workgroup's atomics / loads of the same rows. A counter wait, not
  // __threadfence(): the agent-scope fence also issues buffer_wbl2 + buffer_inv — an L2 write-back and invalidate per
  // workgroup that nothing here needs (every consumer of these rows sits in this workgroup and reads through L2 or after
  // its own CU's write-through stores); at the end of the kernel the same fence cost 1.0 ms of a 5.7 ms launch (r06)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  uint32_t nv[4];  // nodes that exist, per 32-node word
#pragma unroll
  for (int k = 0; k < 4; ++k) nv[k] = (N >= 32 * (k + 1)) ? 0xffffffffu : (N > 32 * k ? ((1u << (N - 32 * k)) - 1u) : 0u);

  // context tables: fp32 [B_inst,N,128] or (CTX16: ctx_dtype = the planes' 16-bit type) with the caller's strides. The row
  // type is a template parameter: chosen at run time each of the two fetches per block paid a dozen selects (elem16.h
  // load_ctx4, which the multistart rollout keeps: its compile time forbids doubling its instantiations)
  constexpr uint32_t ctx_esz = CTX16 ? 2 : 4;
  const uint32_t ctx_rs = (uint32_t)(a.ctx_row_stride ? a.ctx_row_stride : kD) * ctx_esz;  // bytes between node rows (< 2^16)
  const int64_t ctx_bs = a.ctx_batch_stride ? a.ctx_batch_stride : (int64_t)N * kD;
  const char* ctxc = static_cast<const char*>(a.ctx_cur) + ((int64_t)inst * ctx_bs + dcol) * ctx_esz;
  const char* ctxf = (ENV == RL4CO_ENV_TSP) ? static_cast<const char*>(a.ctx_first) + ((int64_t)inst * ctx_bs + dcol) * ctx_esz : nullptr;
  auto ctx_row = [&](const char* table, uint32_t node) -> float4 {  // four dims of a row (node < 128: a 32-bit offset)
    const char* p = table + node * ctx_rs;
    if constexpr (CTX16) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      return make_float4(rl4co_e16::lo(u.x), rl4co_e16::hi(u.x), rl4co_e16::lo(u.y), rl4co_e16::hi(u.y));
    } else {
      return *reinterpret_cast<const float4*>(p);
    }
  };
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kClock = ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kScalar = ENV != RL4CO_ENV_TSP && ENV != RL4CO_ENV_PDP;  // one context scalar: cap - used (PDP: none, context.py:232-243)
  const float* twl = kClock ? a.locs + (int64_t)inst * N * 2 : nullptr;          // coordinates
  const float* tww = kClock ? a.time_windows + (int64_t)inst * N * 2 : nullptr;  // (start, end) per node
  const float* twd = kClock ? a.durations + (int64_t)inst * N : nullptr;         // service times
  const float* dem = kCvrpLike                  ? a.demand + (int64_t)inst * (N - 1)
                     : (ENV == RL4CO_ENV_PCTSP) ? a.demand + (int64_t)inst * N  // real prize, depot column 0
                                                : nullptr;
  const float* oplocs = (ENV == RL4CO_ENV_OP) ? a.locs + (int64_t)inst * N * 2 : nullptr;
  const float* opmax = (ENV == RL4CO_ENV_OP) ? a.max_length + (int64_t)inst * N : nullptr;
  // context scalar = cap - used in both depot environments (OP: longest tour that may still end at
  // the depot minus the tour so far, env_embeddings/context.py:147-149, 211-213)
  // PCTSP: prize_required - prize collected, clamped at 0 (context.py:184-198)
  const float cap = (kCvrpLike || ENV == RL4CO_ENV_PCTSP) ? a.vehicle_capacity[inst]
                                                                      : ((ENV == RL4CO_ENV_OP) ? opmax[0] : 0.0f);
  const float thr = cap + 1e-5f;
  float qb4[4], qx4[4], qt4[4];  // graph context; placeholder query (TSP) or capacity column (CVRP); time column (CVRPTW)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    qb4[e] = a.q_bias ? a.q_bias[(int64_t)inst * kD + dcol + e] : 0.0f;
    qx4[e] = (ENV == RL4CO_ENV_TSP) ? a.q_step0[dcol + e] : ((ENV == RL4CO_ENV_PDP) ? 0.0f : a.w_cap[dcol + e]);
    qt4[e] = kClock ? a.w_time[dcol + e] : 0.0f;
  }
  const float inv_temp = 1.0f / a.temperature;
  const float clip_over_temp = a.tanh_clipping * inv_temp;

  f32x4 dkg[NT], dvg[NT], dkl[NT];  // [d = 4 g + r of head h][node 16 jt + (lane & 15)]
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    dkg[jt] = zero4();
    dvg[jt] = zero4();
    dkl[jt] = zero4();
  }
  float dqb[4] = {0.f, 0.f, 0.f, 0.f}, dqx[4] = {0.f, 0.f, 0.f, 0.f}, dqt[4] = {0.f, 0.f, 0.f, 0.f};
  // d ctx_first (TSP): a lane sums the d query of its columns for as long as they belong to trajectories with the same
  // first node and hands the sum to that row (L2 atomics) when the first node of its column changes, and at the end
  float dqf[4] = {0.f, 0.f, 0.f, 0.f};
  int my_first = -1;
  float* dcf = (ENV == RL4CO_ENV_TSP) ? a.d_ctx_first + (int64_t)inst * N * kD + dcol : nullptr;
  uint32_t errbits = 0;

  // trajectories per group: every live column of the group fits the tables, every raw column the one-group-ahead fetch;
  // a block never holds more than two trajectories (the TSP context-row scatter relies on it: see stage 5)
  const int t0c = min(a.t0, T);
  const int live_max = max(1, T - t0c);
  int G = max(1, L.mc / live_max);
  if (L.npre > 0) G = max(1, min(G, L.npre / T));
  G = min(G, 31);
  if (live_max < 16) G = 1;
#ifdef RL4CO_TEACHER_NOPACK  // timing probe: one trajectory per group, as before r06
  G = 1;
#endif

  // actions / upstream gradients of a group are fetched one group AHEAD (thread i owns raw column i of the group's
  // [trajectory][T] block), so a group's set-up never opens with an HBM round trip
  int an = 0;
  float gn = 0.0f;
  auto fetch_group = [&](int s_first) {
    an = 0;
    gn = 0.0f;
    const int cnt = min(G, S - s_first) * T;
    if (L.npre > 0 && s_first < S && tid < cnt) {
      const int k = tid / T, t = tid - k * T;
      const int64_t idx = (int64_t)((s_first + k) * a.B_inst + inst) * T + t;
      an = (int)a.actions[idx];
      gn = a.grad_logp[idx];
    }
  };
  auto stage_group = [&]() {
    if (L.npre > 0 && tid < L.npre) {
      snact[tid] = (an < 0 || an >= N) ? 255 : an;
      sng[tid] = gn;
    }
  };
  // d ctx_cur[cur] += dq. The table rows of an instance are touched by ITS workgroup only, a wave (head) owns its 16
  // columns and a lane its column. Depot environments (node 0 is revisited): fp32 L2 atomics. TSP: every node is left
  // exactly once per trajectory, so among the columns of ONE trajectory the sum is a plain read-modify-write whose read
  // went out a stage earlier; a block can hold columns of a second trajectory (packed columns), and a column of it that
  // leaves the same node as an earlier column of the block ("late", found at set-up: ~0.5 per two-trajectory block) does
  // its read-modify-write AFTER the others' stores have been acknowledged. (First r06 build: L2 atomics for every column
  // of such a block — a third of all blocks; 2048 lane-atomics per block cost 0.9 ms of the launch, all-atomic 3.6 ms.)
  // Either way the scatter of a block leaves at the top of the NEXT one, behind its context fetch: vmcnt is one in-order
  // counter, issued at the end of their own block the stores stalled each block's first stage (tools/teacher_clock_probe.py).
  auto scatter_pending = [&](const float (&pend)[4], int& pend_cur, bool pend_late) {
    const bool have = pend_cur >= 0;
    if (have && !(ENV == RL4CO_ENV_TSP && pend_late)) {
      float* row = dcc + (int64_t)pend_cur * kD + dcol;
      if (ENV == RL4CO_ENV_TSP) {
        *reinterpret_cast<float4*>(row) = make_float4(pend[0], pend[1], pend[2], pend[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(row + e, pend[e]);
      }
    }
    if (ENV == RL4CO_ENV_TSP && __any(have && pend_late)) {
      // the stores above are acknowledged by L2 before the late lanes read (a plain counter wait: an agent-scope release
      // fence also issues buffer_wbl2, an L2 write-back this exchange inside one workgroup has no use for)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (have && pend_late) {
        float* row = dcc + (int64_t)pend_cur * kD + dcol;
        const float4 old = load4_l2(row);
        *reinterpret_cast<float4*>(row) = make_float4(old.x + pend[0], old.y + pend[1], old.z + pend[2], old.w + pend[3]);
      }
    }
    pend_cur = -1;
  };
  // d ctx_first (TSP): the lanes named by `give` hand their sums to the row of the first node they were summing for — the
  // giving lanes of a row group have the same one (the trajectory their columns just left), so ONE sum over the group's 16
  // lanes and one L2 atomic per dim; per-lane atomics only if that ever does not hold. Called by the whole wave.
  // (First r06 build: four atomics per giving lane, sixteen lanes on the same address — 16 K contended lane-atomics per
  // instance; together with the two-trajectory blocks' atomics the launch took 8.2 ms instead of 5.1.)
  auto flush_first = [&](bool give_in) {
    const bool give = give_in && my_first >= 0;
    const int mine = give ? my_first : -1;
    const int top = rl4co::bfly_i_max<1>(mine);  // (whole wave: every row group flushes the same trajectory)
    const bool uniform_row = __all(!give || my_first == top);
    const unsigned long long bal = __ballot(give);
    const unsigned grp = (unsigned)(bal >> (16 * g)) & 0xffffu;
    if (uniform_row) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = step_sum(give ? dqf[e] : 0.0f);
        if (give && tl == __builtin_ctz(grp | 0x10000u)) unsafeAtomicAdd(dcf + (int64_t)my_first * kD + e, v);
      }
    } else if (give) {
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dcf + (int64_t)my_first * kD + e, dqf[e]);
    }
    if (give_in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dqf[e] = 0.0f;
    }
  };
  fetch_group(0);
  stage_group();

  for (int s0 = 0; s0 < S; s0 += G) {
    const int gsz = min(G, S - s0);
    __syncthreads();  // the previous group's tables are no longer read; the fetched columns are complete
    fetch_group(s0 + G);  // in flight under the set-up below, staged after its last barrier

    // ---- column tables of this group (the environment of each trajectory replayed in closed form) -----------------
    int base = 0;  // packed columns so far (identical in every thread)
    // Up to FOUR trajectories are tabulated at once (N <= 112), one per quarter of the workgroup (128 threads = two waves): one after
    // the other (the first r06 form) the set-up was 7 % of the launch — four barriers and as many dependent LDS round
    // trips per trajectory, 32 of them per instance (profiles/r06_teacher_phase_clocks.json).
    constexpr int kNQ = NT <= 7 ? 4 : 1;  // (make_layout: nq)
    constexpr int kQuarter = kThreads / kNQ;
    const int qd = tid / kQuarter, tq = tid - qd * kQuarter;
    uint8_t* srq = sraw + qd * kMaxT;
    int* spq = spos + qd * 128;
    for (int k0 = 0; k0 < gsz; k0 += kNQ) {
      const int k = k0 + qd;
      const bool active = k < gsz;
      const int r = (s0 + (active ? k : k0)) * a.B_inst + inst;
      // srq[t]: action; spq[j]: first column that visits node j (tsp/env.py:60-86, cvrp/env.py:66-96)
      if (active) {
        for (int tt = tq; tt < kMaxT; tt += kQuarter) {
          int at = 0;
          if (tt < T) {
            if (L.npre > 0) {
              at = snact[k * T + tt];
            } else {
              const int raw = (int)a.actions[(int64_t)r * T + tt];
              at = (raw < 0 || raw >= N) ? 255 : raw;
            }
            if (at == 255) {
              errbits |= RL4CO_EBIT_INFEASIBLE;
              at = 0;
            }
          }
          srq[tt] = at;
        }
        for (int j = tq; j < 128; j += kQuarter) spq[j] = 0x7fffffff;
      }
      __syncthreads();
      if (active)
        for (int t = tq; t < T; t += kQuarter) atomicMin(&spq[srq[t]], t);
      __syncthreads();
      if (active) {
        if (tq == 0) {
          int t_end = T;
          if (kCvrpLike) {  // done once every node (depot included) has been visited (cvrp/env.py:80-83)
            int last = 0;
            for (int j = 0; j < N; ++j) last = max(last, spq[j]);
            if (last != 0x7fffffff) t_end = min(T, last + 1);
          }
          if (ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_PCTSP) {  // done at the first return to the depot after step 0 (op/env.py:84, pctsp/env.py:73)
            for (int t = 1; t < T; ++t)
              if (srq[t] == 0) {
                t_end = t + 1;
                break;
              }
          }
          sinfo[qd] = t_end;
        }
      } else if (tq == 0) {
        sinfo[qd] = t0c;  // (no trajectory in this quarter: no live column)
      }
      __syncthreads();
      // packed place of this quarter's trajectory: behind the live columns of the quarters before it
      int bq = base, live_round = 0;
#pragma unroll
      for (int j = 0; j < kNQ; ++j) {
        const int lj = max(0, sinfo[j] - t0c);
        if (j < qd) bq += lj;
        live_round += lj;
      }
      const int t_end = sinfo[qd];
      const int live = max(0, t_end - t0c);
      if (active && kScalar) {
        // per-column scalars BEFORE column t, written at the column's packed place bq + t - t0 (live columns only: the
        // range behind them belongs to the next quarter's trajectory, which is writing it at the same time)
        if (kClock) {
          // the clock, replayed in visiting order: advance by the distance, wait for the window, serve;
          // back at the depot it restarts (cvrptw/env.py:97-113, same fp32 sequence as the decode kernel)
          for (int t = t0c + tq; t < t_end; t += kQuarter) {
            float now = 0.0f;
            int prev = 0;
            for (int v = 0; v < t; ++v) {
              const int nx = srq[v];
              const float dx = twl[2 * nx] - twl[2 * prev], dy = twl[2 * nx + 1] - twl[2 * prev + 1];
              now = (nx != 0 ? 1.0f : 0.0f) * (fmaxf(now + sqrtf(fmaf(dy, dy, dx * dx)), tww[2 * nx]) + twd[nx]);
              prev = nx;
            }
            stime[bq + t - t0c] = now;
          }
        }
        if (kCvrpLike) {
          // used capacity: the loads since the last depot visit, summed in visiting
          // order from zero — the same fp32 sequence as used = (used + demand) * (action != 0)
          for (int t = t0c + tq; t < t_end; t += kQuarter) {
            int u = t - 1;
            while (u >= 0 && srq[u] != 0) --u;
            float used = 0.0f;
            for (int v = u + 1; v < t; ++v) used = used + dem[min(max((int)srq[v] - 1, 0), N - 2)];
            srem[bq + t - t0c] = used;
          }
        }
        if (ENV == RL4CO_ENV_OP) {
          // tour length, accumulated in visiting order like tour += |loc_a - loc_cur|
          for (int t = t0c + tq; t < t_end; t += kQuarter) {
            float used = 0.0f;
            int prev = 0;
            for (int v = 0; v < t; ++v) {
              const int nx = srq[v];
              const float dx = oplocs[2 * nx] - oplocs[2 * prev], dy = oplocs[2 * nx + 1] - oplocs[2 * prev + 1];
              used = used + sqrtf(fmaf(dy, dy, dx * dx));
              prev = nx;
            }
            srem[bq + t - t0c] = used;
          }
        }
        if (ENV == RL4CO_ENV_PCTSP) {
          // prize collected, accumulated in visiting order like prize += real_prize[a]
          for (int t = t0c + tq; t < t_end; t += kQuarter) {
            float used = 0.0f;
            for (int v = 0; v < t; ++v) used = used + dem[srq[v]];
            srem[bq + t - t0c] = used;
          }
        }
      }
      if (kScalar) __syncthreads();  // the mask words read the scalars of their column
      if (active) {
        if (ENV == RL4CO_ENV_TSP) {
          // feasibility words of a column in two ballots: wave w takes columns w, w + 8, ..; lane = node (and node + 64);
          // node j is feasible at column t until it has been visited, spq[j] >= t
          const int p0 = (lane < N) ? spq[lane] : -1, p1 = (lane + 64 < N) ? spq[lane + 64] : -1;
          for (int t = t0c + (w % (kWaves / kNQ)); t < t_end; t += kWaves / kNQ) {  // the quarter's waves
            const unsigned long long b0 = __ballot(p0 >= t), b1 = __ballot(p1 >= t);
            if (lane == 0)
              *reinterpret_cast<uint4*>(smask + 4 * (bq + t - t0c)) = make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32));
          }
        }
        // feasibility words: thread (t, kw) builds word kw of column t
        for (int idx = tq; ENV != RL4CO_ENV_TSP && idx < live * 4; idx += kQuarter) {
          const int t = t0c + (idx >> 2), kw = idx & 3, col = bq + (idx >> 2);
          uint32_t word = 0;
          if (ENV == RL4CO_ENV_OP) {
            // op/env.py:137-154: unvisited, depot not yet closed, and the node can still be entered
            const float used = srem[col];
            const int cur = (t == 0) ? 0 : srq[t - 1];
            const float cx = oplocs[2 * cur], cy = oplocs[2 * cur + 1];
            const bool depot_visited = spq[0] < t;
            for (int b = 0; b < 32; ++b) {
              const int j = 32 * kw + b;
              if (j < N) {
                const float dx = oplocs[2 * j] - cx, dy = oplocs[2 * j + 1] - cy;
                const bool exceeds = used + sqrtf(fmaf(dy, dy, dx * dx)) > opmax[j];
                if (j == 0 || !(spq[j] < t || depot_visited || exceeds)) word |= 1u << b;
              }
            }
          } else if (ENV == RL4CO_ENV_PDP) {
            // pdp/env.py:64-99: unvisited, a delivery only once its pickup is on the tour; the depot only as the
            // forced first step of force_start_at_depot (recognised by the trajectory starting at node 0)
            const int half = (N - 1) / 2;
            for (int b = 0; b < 32; ++b) {
              const int j = 32 * kw + b;
              if (j >= 1 && j < N && spq[j] >= t && (j <= half || spq[j - half] < t)) word |= 1u << b;
            }
            if (t == 0 && srq[0] == 0) word = (kw == 0) ? 1u : 0u;
          } else if (ENV == RL4CO_ENV_PCTSP) {
            // pctsp/env.py:141-148: customers while unvisited and the depot not yet closed; the depot opens
            // once a total prize of 1 is collected or no customer is left
            const bool depot_visited = spq[0] < t;
            uint32_t left = 0;
            for (int b = 0; b < 32; ++b) {
              const int j = 32 * kw + b;
              if (j >= 1 && j < N && spq[j] >= t) left |= 1u << b;
            }
            word = depot_visited ? 0u : left;
            left |= rl4co::bfly_i<1>((int)left);
            left |= rl4co::bfly_i<2>((int)left);
            if (kw == 0 && !((srem[col] < 1.0f) && left != 0u)) word |= 1u;
          } else {
            const float used = srem[col];
            for (int b = 0; b < 32; ++b) {
              const int j = 32 * kw + b;
              if (j >= 1 && j < N && spq[j] >= t && !(dem[j - 1] + used > thr)) word |= 1u << b;
            }
            // depot: infeasible only while standing on it with a customer still feasible (cvrp/env.py:126-136)
            uint32_t any = word;
            any |= rl4co::bfly_i<1>((int)any);
            any |= rl4co::bfly_i<2>((int)any);
            const int cur = (t == 0) ? 0 : srq[t - 1];
            if (kw == 0 && !((cur == 0) && any != 0u)) word |= 1u;
            if (kClock) {  // cvrptw/env.py:91-95: only nodes whose window is still open on arrival (the depot too)
              const float now = stime[col], cx = twl[2 * cur], cy = twl[2 * cur + 1];
              for (int b = 0; b < 32; ++b) {
                const int j = 32 * kw + b;
                if ((word >> b) & 1u) {
                  const float dx = twl[2 * j] - cx, dy = twl[2 * j + 1] - cy;
                  if (!(now + sqrtf(fmaf(dy, dy, dx * dx)) <= tww[2 * j + 1])) word &= ~(1u << b);
                }
              }
            }
          }
          smask[4 * col + kw] = word;
        }
        for (int i = tq; i < live; i += kQuarter) {
          const int t = t0c + i, col = bq + i;
          scol[col] = (uint32_t)srq[t] | ((t == 0 ? 0u : (uint32_t)srq[t - 1]) << 8) | ((uint32_t)srq[0] << 16) |
                      ((uint32_t)(1 | (t == 0 ? 2 : 0) | (k << 2)) << 24);
          sstep[col] = (uint8_t)t;
          sg[col] = L.npre > 0 ? sng[k * T + t] : a.grad_logp[(int64_t)r * T + t];
        }
      }
      __syncthreads();  // srq / spq / sinfo are rewritten by the next round; the words and scalars are complete
      base += live_round;
    }
    const int ncols = base;
    const int ntb = (ncols + 15) >> 4;
    // dead columns (the last block's tail and the look-ahead block behind it): a finite dummy (node 0 only), gradient 0
    for (int col = ncols + tid; col < 16 * ntb + 16; col += kThreads) {
      scol[col] = 0;
      sstep[col] = 0;
      sg[col] = 0.0f;
      if (kScalar) srem[col] = 0.0f;
      if (kClock) stime[col] = 0.0f;
      *reinterpret_cast<uint4*>(smask + 4 * col) = make_uint4(1u, 0u, 0u, 0u);
    }
    if (kScalar) {  // srem: used -> remaining capacity / length (context.py:147-149, 211-213); the mask words are all built
      for (int col = tid; col < ncols; col += kThreads) {
        float rem = cap - srem[col];
        if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;
        srem[col] = rem;
      }
    }
    if (ENV == RL4CO_ENV_TSP && gsz > 1) {
      // "late" columns (stage 5): a column whose current node is also the current node of an EARLIER column of its block —
      // possible only across the two trajectories a block can hold. Its context-row sum must see the earlier column's.
      for (int col = tid; col < ncols; col += kThreads) {
        const uint32_t f = scol[col];
        bool late = false;
        // (only a column of the block's SECOND trajectory can be late: the scan is skipped for all the others)
        if ((f & 0x03000000u) == 0x01000000u && ((scol[col & ~15] ^ f) & 0x7c000000u) != 0) {  // valid, not step 0: the column scatters
          for (int c2 = col & ~15; c2 < col; ++c2) {
            const uint32_t f2 = scol[c2];  // other scatters, other slot, same current node
            late |= (f2 & 0x03000000u) == 0x01000000u && ((f2 ^ f) & 0x7c000000u) != 0 && ((f2 ^ f) & 0x0000ff00u) == 0;
          }
        }
        if (late) scol[col] = f | 0x80000000u;  // (own word; the scans above ignore bit 31)
      }
    }
    stage_group();
    __syncthreads();

    // context rows of a lane's column are fetched one block ahead (an L2 round trip otherwise opens every block's
    // dependency chain); the first block's leave here
    uint32_t coln = scol[tl];  // this lane's column word, read one block ahead like the rows it names
    float4 c4n = ctx_row(ctxc, (coln >> 8) & 0xffu);
    float4 f4n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ENV == RL4CO_ENV_TSP) f4n = ctx_row(ctxf, (coln >> 16) & 0xffu);
    // the rows just fetched are "used" here, before the loop: otherwise the waitcnt state merged at the loop header still
    // counts them as in flight and every block's first stage waits — vmcnt is one in-order counter — for the fetch it has
    // JUST issued (r06: this line missing cost 0.7 us per block, the whole gain of the packed columns)
    asm volatile("" : "+v"(c4n.x), "+v"(c4n.y), "+v"(c4n.z), "+v"(c4n.w));
    if (ENV == RL4CO_ENV_TSP) asm volatile("" : "+v"(f4n.x), "+v"(f4n.y), "+v"(f4n.z), "+v"(f4n.w));
    float pend[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's deferred context-row scatter
    int pend_cur = -1;
    bool pend_late = false;
    for (int tb = 0; tb < ntb; ++tb) {
      const int t = 16 * tb + tl;  // this lane's packed column (same in the four row groups)
      const uint32_t colw = coln;
      coln = scol[t + 16];
      const int at = colw & 0xffu, cur = (colw >> 8) & 0xffu;
      const float gt = sg[t];
      const uint32_t fl = colw >> 24;
      const bool valid = (fl & 1u) != 0, tzero = (fl & 2u) != 0;
      const uint4 mw4 = *reinterpret_cast<const uint4*>(smask + 4 * t);
      const uint32_t mw[4] = {mw4.x, mw4.y, mw4.z, mw4.w};
      const float rem = kScalar ? srem[t] : 0.0f;
      const float now = kClock ? stime[t] : 0.0f;
      // TSP: an earlier column of this block (the other trajectory) leaves the same node — see stage 5
      const bool late = ENV == RL4CO_ENV_TSP && (fl & 0x80u) != 0;
      const int slot = (int)(fl >> 2) & 31;  // the column's trajectory inside the group

      // ---- 0. query of head h for the block's 16 steps (context.py:105-149, decoder.py:135-136) --
      bf16x4 qf;
      {
        const float4 c4 = c4n;
        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
        const float f4[4] = {f4n.x, f4n.y, f4n.z, f4n.w};
        c4n = ctx_row(ctxc, (coln >> 8) & 0xffu);  // the next block's column of this lane
        if (ENV == RL4CO_ENV_TSP) {
          f4n = ctx_row(ctxf, (coln >> 16) & 0xffu);
          const int fn = (colw >> 16) & 0xffu;
          const bool moved = valid && fn != my_first;  // the lane's column moved on to a trajectory with another first node
          if (__any(moved)) {
            flush_first(moved);
            if (moved) my_first = fn;
          }
        }
        // the PREVIOUS block's context-row scatter leaves here, behind the fetch: vmcnt is one in-order counter on
        // gfx9, so the wait for the row above also waits for every atomic issued before it — issued at the end of
        // their own block they stalled each block's first stage for their whole L2 round trip (tools/teacher_clock_probe.py)
        scatter_pending(pend, pend_cur, pend_late);
        float q4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float q;
          if (ENV == RL4CO_ENV_TSP) q = tzero ? qx4[e] + qb4[e] : (f4[e] + c[e]) + qb4[e];
          else if (kClock) q = fmaf(qt4[e], now, fmaf(qx4[e], rem, c[e])) + qb4[e];  // context.py:152-166
          else q = fmaf(qx4[e], rem, c[e]) + qb4[e];
          q4[e] = q * (0.25f * kLog2e);
        }
        qf = rl4co_e16::cvt4(q4[0], q4[1], q4[2], q4[3]);
        *reinterpret_cast<bf16x4*>(qb + tl * kRS + dcol) = qf;
      }

      // ---- 1. scores^T and softmax numerators over nodes (attention.py:300-314) -----------------------
      bf16x4 pf[NT];
#ifndef RL4CO_TEACHER_NO_BATCH
      bf16x4 vf[NT];
#endif
      float inv_l;
      {
        f32x4 sc[NT];
        float m = kNegInf;
#ifndef RL4CO_TEACHER_NO_BATCH
        // every operand read of a stage goes out BEFORE its first product (r06). Written tile by tile the compiler keeps
        // that order — ds_read, s_waitcnt lgkmcnt(0), v_mfma, per tile — and every one of the ~50 LDS reads of a block
        // exposes its full latency on the wave's chain (r06 ISA; the attention backward had the same shape, r05)
        bf16x4 kf[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) kf[jt] = lds_b64(kgs + 16 * jt * kRS + 16 * h + nao);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          {
#ifndef RL4CO_TEACHER_NO_BATCH
            sc[jt] = mfma16(kf[jt], qf, zero4());
#else
            sc[jt] = mfma16(lds_b64(kgs + 16 * jt * kRS + 16 * h + nao), qf, zero4());
#endif
            const uint32_t bits = (a.mask_inner ? mw[jt >> 1] : nv[jt >> 1]) >> (16 * (jt & 1) + 4 * g);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              sc[jt][rr] = rl4co::keep_or_neg_inf(bits, rr, sc[jt][rr]);
              m = fmaxf(m, sc[jt][rr]);
            }
          }
        }
        m = rg_max(m);
        const f32x2 m2 = {m, m};
        f32x2 l2 = {0.0f, 0.0f};
#ifndef RL4CO_TEACHER_NO_BATCH
        // the glimpse's value tiles travel under the exponentials
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) vf[jt] = lds_tr(vs + 16 * jt * kRS + 16 * h + tro);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          {
            const f32x2 d0 = f32x2{sc[jt][0], sc[jt][1]} - m2, d1 = f32x2{sc[jt][2], sc[jt][3]} - m2;
            const f32x2 p0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
            const f32x2 p1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
            l2 += p0 + p1;
            pf[jt] = pack4(p0, p1);
            *reinterpret_cast<bf16x4*>(pbw + tl * kRS + 16 * jt + 4 * g) = pf[jt];
          }
        }
        const float l = rg_sum(l2[0] + l2[1]);
        inv_l = __builtin_amdgcn_rcpf(l);
      }

      // ---- 2. glimpse O_h^T = V_h^T P^T --------------------------------------------------------------
      f32x4 o = zero4();  // kept: the softmax backward's sum_j a_j dA_j is O_h . dO_h
      {
#ifndef RL4CO_TEACHER_NO_BATCH
        f32x4 o1 = zero4();  // two accumulators: half the dependent-MFMA chain
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (jt & 1) o1 = mfma16(vf[jt], pf[jt], o1);
          else o = mfma16(vf[jt], pf[jt], o);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] += o1[rr];
#else
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          o = mfma16(lds_tr(vs + 16 * jt * kRS + 16 * h + tro), pf[jt], o);
#endif
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] *= inv_l;
        *reinterpret_cast<bf16x4*>(ob + tl * kRS + dcol) = to_bf16(o);
      }
      rl4co::lds_barrier();  // B1: all heads' glimpses

      // ---- 3. logits of node tile w, clip, log-softmax pieces (attention.py:291-293, decoding.py:169-188)
      // clipped logits live in [-C, C], C = tanh_clipping / temperature: up to C = 60 their exponentials and the sum
      // over 128 nodes are plain fp32 numbers, so the log-sum-exp needs no running maximum — one exponential per logit,
      // kept across B2 and scaled by 1 / sum, instead of three and eight more to merge the node tiles' partial sums
      const bool bounded = a.tanh_clipping > 0.0f && clip_over_temp <= 60.0f;
      float z[4], dzdu[4];  // z: the logit, or (bounded) its exponential, 0 for a masked node
      {
        float zmax = kNegInf;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          z[rr] = bounded ? 0.0f : kNegInf;
          dzdu[rr] = 0.0f;
        }
        if (w < NT) {
          // contraction over the 128 dims on v_mfma_f32_16x16x32: 8 consecutive dims per lane from both row-major
          // blocks (one 16-byte read each), two accumulators so the dependent chain is two deep instead of eight
          f32x4 u = zero4(), u1 = zero4();
          const int nao8 = tl * kRS + 8 * g;
#pragma unroll
          for (int ks = 0; ks < 4; ks += 2) {
            u = rl4co_e16::mfma_16x16x32(lds_b128(kls + 16 * w * kRS + 32 * ks + nao8), lds_b128(ob + 32 * ks + nao8), u);
            u1 = rl4co_e16::mfma_16x16x32(lds_b128(kls + 16 * w * kRS + 32 * (ks + 1) + nao8), lds_b128(ob + 32 * (ks + 1) + nao8), u1);
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) u[rr] += u1[rr];
          // w is a runtime value: select, never index (an indexed register array goes to scratch)
          const uint32_t wsel = a.mask_logits ? (w < 2 ? mw4.x : (w < 4 ? mw4.y : (w < 6 ? mw4.z : mw4.w)))
                                              : (w < 2 ? nv[0] : (w < 4 ? nv[1] : (w < 6 ? nv[2] : nv[3])));
          const uint32_t bits = wsel >> (16 * (w & 1) + 4 * g);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float uu = u[rr] * (1.0f / kSqrtD);
            if (uu != uu && valid) errbits |= RL4CO_EBIT_NAN_LOGIT;
            float zz, dd;
            if (a.tanh_clipping > 0.0f) {
              const float ex = __expf(-2.0f * fabsf(uu));
              const float th = copysignf((1.0f - ex) * __builtin_amdgcn_rcpf(1.0f + ex), uu);
              zz = th * clip_over_temp;
              dd = clip_over_temp * (1.0f - th * th);
            } else {
              zz = uu * inv_temp;
              dd = inv_temp;
            }
            const bool f = (bits >> rr) & 1u;
            dzdu[rr] = dd;
            if (16 * w + 4 * g + rr == at) xa[tl] = f ? zz : kNegInf;
            if (bounded) {
              z[rr] = f ? __expf(zz) : 0.0f;
            } else {
              z[rr] = f ? zz : kNegInf;
              zmax = fmaxf(zmax, z[rr]);
            }
          }
        }
        if (bounded) {
          const float se = rg_sum((z[0] + z[1]) + (z[2] + z[3]));
          if (g == 0) xz[(w * 16 + tl) * 2 + 1] = se;
        } else {
          zmax = rg_max(zmax);
          const float zs = (zmax > kNegInf) ? zmax : 0.0f;
          float se = 0.0f;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) se += __expf(z[rr] - zs);
          se = rg_sum(se);
          if (g == 0) {
            xz[(w * 16 + tl) * 2] = zmax;
            xz[(w * 16 + tl) * 2 + 1] = (zmax > kNegInf) ? se : 0.0f;
          }
        }
      }
      rl4co::lds_barrier();  // B2: log-sum-exp pieces of all node tiles
      {
        float lse, inv_tot = 0.0f;
        if (bounded) {
          float tot = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kWaves; ++ww) tot += xz[(ww * 16 + tl) * 2 + 1];
          lse = __logf(tot);
          inv_tot = __builtin_amdgcn_rcpf(tot);
        } else {
          float zm = kNegInf;
#pragma unroll
          for (int ww = 0; ww < kWaves; ++ww) zm = fmaxf(zm, xz[(ww * 16 + tl) * 2]);
          float tot = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kWaves; ++ww) {
            const float zw = xz[(ww * 16 + tl) * 2];
            tot += (zw > kNegInf) ? xz[(ww * 16 + tl) * 2 + 1] * __expf(zw - zm) : 0.0f;
          }
          lse = zm + __logf(tot);
        }
        if (w == 0 && g == 0 && valid) {  // log p(a_t) (decoding.py:381) and the reference's assertions
          const float lp = xa[tl] - lse;
          const int ak = at >> 5;  // static indexing only: a runtime index would spill the words to scratch
          const uint32_t aw = ak == 0 ? mw4.x : (ak == 1 ? mw4.y : (ak == 2 ? mw4.z : mw4.w));
          const bool feasible = (aw >> (at & 31)) & 1u;
          if (!feasible) errbits |= RL4CO_EBIT_INFEASIBLE;
          if (!(lp > -1000.0f)) errbits |= RL4CO_EBIT_NEG_INF_LOGP;
          if (a.logp_out) a.logp_out[(int64_t)((s0 + slot) * a.B_inst + inst) * T + sstep[t]] = lp;
        }
        if (w < NT) {
          f32x4 du;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float prob = bounded ? z[rr] * inv_tot : __expf(z[rr] - lse);  // 0 for masked nodes
            const float dz = gt * (((16 * w + 4 * g + rr) == at ? 1.0f : 0.0f) - prob);
            du[rr] = (bounded ? z[rr] > 0.0f : z[rr] > kNegInf) ? dz * dzdu[rr] * (1.0f / kSqrtD) : 0.0f;
          }
          *reinterpret_cast<bf16x4*>(dub + tl * kRS + 16 * w + 4 * g) = to_bf16(du);
        }
      }
      rl4co::lds_barrier();  // B3: d logits of all node tiles

      // ---- 4. d glimpse of head h, d logit keys ----------------------------------------------------------
      bf16x4 dof;
      {
        f32x4 dO = zero4();
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          dO = mfma16(lds_tr(kls + 16 * jt * kRS + 16 * h + tro), lds_b64(dub + 16 * jt + nao), dO);
        const bf16x4 ot = lds_tr(ob + 16 * h + tro);  // O_h^T[d][steps 4 g ..]
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          dkl[jt] = mfma16(ot, lds_tr(dub + 16 * jt + tro), dkl[jt]);
        dof = to_bf16(dO);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) dO[rr] *= inv_l;
        *reinterpret_cast<bf16x4*>(dob + tl * kRS + dcol) = to_bf16(dO);
      }

      // ---- 5. softmax backward of head h; d values, d keys, d query ------------------------------------
      // sum_j a_j dA_j = sum_j a_j (V_j . dO) = O . dO: four products per lane and one exchange across the row groups
      // instead of a first pass over all node tiles; dA^T itself is produced tile by tile, turned into dS, staged and
      // fed to d query at once
      f32x4 dq = zero4();
      const bool scatter = valid && (ENV != RL4CO_ENV_TSP || !tzero);
      // d ctx_cur[cur] += dq, four dims of one row: see scatter_pending. TSP: the read of the read-modify-write goes out
      // here, a whole stage ahead of its use ("late" columns read at the scatter itself)
      float4 row_old = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ENV == RL4CO_ENV_TSP && scatter && !late) row_old = *reinterpret_cast<const float4*>(dcc + (int64_t)cur * kD + dcol);
      {
        float ada = 0.0f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ada = fmaf(o[rr], (float)dof[rr], ada);
        ada = rg_sum(ada);
        wave_lds_sync();  // this wave's P block, dO and Q columns are in LDS
        {
          const bf16x4 dt = lds_tr(dob + 16 * h + tro);  // (dO_h / l)^T[d][steps]
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
            dvg[jt] = mfma16(dt, lds_tr(pbw + 16 * jt + tro), dvg[jt]);
        }
        wave_lds_sync();  // the transpose reads of P are done: the block is reused for dS
        const f32x2 il2 = {inv_l, inv_l}, nada2 = {-ada * inv_l, -ada * inv_l};
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          {
            const f32x4 da = mfma16(lds_b64(vs + 16 * jt * kRS + 16 * h + nao), dof, zero4());
            // dS = a (dA - sum a dA), a = p / l: one packed fma and one packed multiply per two nodes
            const u32x2 pw = __builtin_bit_cast(u32x2, pf[jt]);
            const f32x2 s0 = __builtin_elementwise_fma(f32x2{da[0], da[1]}, il2, nada2) * f32x2{rl4co_e16::lo(pw[0]), rl4co_e16::hi(pw[0])};
            const f32x2 s1 = __builtin_elementwise_fma(f32x2{da[2], da[3]}, il2, nada2) * f32x2{rl4co_e16::lo(pw[1]), rl4co_e16::hi(pw[1])};
            const bf16x4 dsf = pack4(s0, s1);
            *reinterpret_cast<bf16x4*>(pbw + tl * kRS + 16 * jt + 4 * g) = dsf;
            dq = mfma16(lds_tr(kgs + 16 * jt * kRS + 16 * h + tro), dsf, dq);
          }
        }
        wave_lds_sync();
        {
          const bf16x4 qt = lds_tr(qb + 16 * h + tro);  // Q_h^T[d][steps]
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
            dkg[jt] = mfma16(qt, lds_tr(pbw + 16 * jt + tro), dkg[jt]);
        }
      }

      // ---- 6. d query -> context rows, graph context, placeholder / capacity column ---------------------
      if (valid) {
        const float old4[4] = {row_old.x, row_old.y, row_old.z, row_old.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dqr = 0.25f * dq[e];
          dqb[e] += dqr;
          if (ENV == RL4CO_ENV_TSP) {
            if (tzero) dqx[e] += dqr;
            else dqf[e] += dqr;
          } else {
            dqx[e] = fmaf(dqr, rem, dqx[e]);
            if (kClock) dqt[e] = fmaf(dqr, now, dqt[e]);
          }
          pend[e] = old4[e] + dqr;  // d ctx_cur[cur] += dqr: leaves at the top of the next block (or after the last)
        }
        if (scatter) {
          pend_cur = cur;
          pend_late = late;
        }
      }
      rl4co::lds_barrier();  // B4: the glimpse / d-logit blocks are rewritten by the next step block (LDS only:
                             // the context-row atomics stay in flight)
    }

    scatter_pending(pend, pend_cur, pend_late);  // the last block's scatter
  }
  if (ENV == RL4CO_ENV_TSP) flush_first(my_first >= 0);  // d ctx_first: what the lanes still hold

  // ---- the instance's plane gradients: dims 16 h + 4 g .. + 3 of node 16 jt + (lane & 15) -----------
  const float c = 1.0f / kLog2e;  // the staged queries carried log2(e)
  if (a.d_planes_bf16) {  // bf16 rows in the caller's layout (columns of the fold GEMMs' gradient operand)
    elem_t* dk = static_cast<elem_t*>(a.d_planes_bf16) + (int64_t)inst * a.d_planes_batch_stride + dcol;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int j = 16 * jt + tl;
      if (j < N) {
        elem_t* p0 = dk + (int64_t)j * a.d_planes_row_stride;
        const f32x4 kg = {dkg[jt][0] * c, dkg[jt][1] * c, dkg[jt][2] * c, dkg[jt][3] * c};
        *reinterpret_cast<bf16x4*>(p0) = to_bf16(kg);
        *reinterpret_cast<bf16x4*>(p0 + a.d_planes_plane_stride) = to_bf16(dvg[jt]);
        *reinterpret_cast<bf16x4*>(p0 + 2 * a.d_planes_plane_stride) = to_bf16(dkl[jt]);
      }
    }
  } else {
    float* dk = a.d_kvl + (int64_t)inst * N * kD;
    const int64_t plane = (int64_t)a.B_inst * N * kD;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int j = 16 * jt + tl;
      if (j < N) {
        float* p0 = dk + (int64_t)j * kD + dcol;
        *reinterpret_cast<float4*>(p0) = make_float4(dkg[jt][0] * c, dkg[jt][1] * c, dkg[jt][2] * c, dkg[jt][3] * c);
        *reinterpret_cast<float4*>(p0 + plane) = make_float4(dvg[jt][0], dvg[jt][1], dvg[jt][2], dvg[jt][3]);
        *reinterpret_cast<float4*>(p0 + 2 * plane) = make_float4(dkl[jt][0], dkl[jt][1], dkl[jt][2], dkl[jt][3]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float vb = step_sum(dqb[e]), vx = step_sum(dqx[e]);
    if (tl == 0) {
      if (a.d_q_bias) a.d_q_bias[(int64_t)inst * kD + dcol + e] = vb;
      if (ENV != RL4CO_ENV_PDP) unsafeAtomicAdd((ENV == RL4CO_ENV_TSP ? a.d_q_step0 : a.d_w_cap) + dcol + e, vx);
    }
    if (kClock) {
      const float vt = step_sum(dqt[e]);
      if (tl == 0) unsafeAtomicAdd(a.d_w_time + dcol + e, vt);
    }
  }
  if (errbits) atomicOr(a.err, (int)errbits);
  if (a.d_ctx_in_planes) {
    // the context-table gradients leave as planes 3 / 4 of the caller's 16-bit gradient matrix (the fold GEMMs' operand):
    // every scatter of this instance has been issued by this workgroup — drained (vmcnt(0) inside __syncthreads) and read
    // back through L2 (the sums were built by atomics: load4_l2)
    __syncthreads();  // (s_waitcnt vmcnt(0) of every wave, then the barrier: all stores and atomics are at L2)
    elem_t* dp = static_cast<elem_t*>(a.d_planes_bf16) + (int64_t)inst * a.d_planes_batch_stride;
    const int p_cur = (ENV == RL4CO_ENV_TSP) ? 4 : 3;
    // NT * 16 rows x 32 four-dim pieces over 512 threads = NT pieces per thread: all loads of a table out before the first
    // conversion (a fixed trip count over clamped rows — the rows past N repeat the last one's bytes to the same address)
    float4 v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + kThreads * j, row = min(i >> 5, N - 1), c4 = (i & 31) * 4;
      // (TSP: these sums were built by this CU's own plain read-modify-writes — a plain 16-byte load sees them exactly as
      // the read-modify-write of the next trajectory did; the depot environments add with L2 atomics: read through L2)
      if (ENV == RL4CO_ENV_TSP) v[j] = *reinterpret_cast<const float4*>(dcc + (int64_t)row * kD + c4);
      else v[j] = load4_l2(dcc + (int64_t)row * kD + c4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + kThreads * j, row = min(i >> 5, N - 1), c4 = (i & 31) * 4;
      *reinterpret_cast<bf16x4*>(dp + p_cur * a.d_planes_plane_stride + (int64_t)row * a.d_planes_row_stride + c4) =
          rl4co_e16::cvt4(v[j].x, v[j].y, v[j].z, v[j].w);
    }
    if (ENV == RL4CO_ENV_TSP) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int i = tid + kThreads * j, row = min(i >> 5, N - 1), c4 = (i & 31) * 4;
        v[j] = load4_l2(a.d_ctx_first + ((int64_t)inst * N + row) * kD + c4);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int i = tid + kThreads * j, row = min(i >> 5, N - 1), c4 = (i & 31) * 4;
        *reinterpret_cast<bf16x4*>(dp + 3 * a.d_planes_plane_stride + (int64_t)row * a.d_planes_row_stride + c4) =
            rl4co_e16::cvt4(v[j].x, v[j].y, v[j].z, v[j].w);
      }
    }
  }
}

}  // namespace

namespace rl4co {

#if !RL4CO_ELEM_F16
int teacher_mma_max_nodes() { return 16 * kMaxTiles; }
int teacher_mma_max_steps() { return kMaxT; }
#endif

template <int ENV, int NT, bool CTX16>
static int launch_tiles_ctx(const rl4co_am_teacher_args& a, hipStream_t stream) {
  const Layout L = make_layout(NT, ENV);
  if (L.total > 160 * 1024) return rl4co::record_arg_error("am_teacher_mma: LDS layout exceeds 160 KB");
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_teacher_mma_kernel<ENV, NT, CTX16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
  hipLaunchKernelGGL((am_teacher_mma_kernel<ENV, NT, CTX16>), dim3(a.B_inst), dim3(kThreads), L.total, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
}
template <int ENV, int NT>
static int launch_tiles(const rl4co_am_teacher_args& a, hipStream_t stream) {
  return a.ctx_dtype != RL4CO_DT_F32 ? launch_tiles_ctx<ENV, NT, true>(a, stream) : launch_tiles_ctx<ENV, NT, false>(a, stream);
}

template <int ENV>
static int dispatch_tiles(const rl4co_am_teacher_args& a, hipStream_t stream) {
  const int nt = (a.N + 15) >> 4;
  if (nt <= 2) return launch_tiles<ENV, 2>(a, stream);
  if (nt <= 4) return launch_tiles<ENV, 4>(a, stream);
  if (nt <= 7) return launch_tiles<ENV, 7>(a, stream);
  return launch_tiles<ENV, kMaxTiles>(a, stream);
}

int RL4CO_CXX(launch_teacher_mma)(const rl4co_am_teacher_args& a, hipStream_t stream) {
  if (a.env == RL4CO_ENV_OP) return dispatch_tiles<RL4CO_ENV_OP>(a, stream);
  if (a.env == RL4CO_ENV_PCTSP) return dispatch_tiles<RL4CO_ENV_PCTSP>(a, stream);
  if (a.env == RL4CO_ENV_PDP) return dispatch_tiles<RL4CO_ENV_PDP>(a, stream);
  if (a.env == RL4CO_ENV_CVRPTW) return dispatch_tiles<RL4CO_ENV_CVRPTW>(a, stream);
  return a.env == RL4CO_ENV_TSP ? dispatch_tiles<RL4CO_ENV_TSP>(a, stream) : dispatch_tiles<RL4CO_ENV_CVRP>(a, stream);
}

}  // namespace rl4co
