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
// overlaps the MFMAs of the other); its S multistart trajectories are replayed one after the other
// in blocks of 16 steps, all products on v_mfma_f32_16x16x16_bf16. In that instruction's
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

// d ctx_cur[cur] += dq, four dims of one row. The table rows of an instance are touched by ITS workgroup only, a wave
// (head) owns its 16 columns and a lane its step: the only possible collision is one node being the current node of
// two steps. In the TSP every node is left exactly once per trajectory, so the sum is a plain read-modify-write whose
// read went out a stage earlier (`old` already folded into `v`); the depot environments revisit node 0 and keep the
// fp32 L2 atomics. 2048 scattered lane-atomics per step block held every wave's first stage for ~2 K cycles
// (tools/teacher_clock_probe.py)
template <int ENV>
__device__ inline void scatter_row(float* row, const float (&v)[4]) {
  if (ENV == RL4CO_ENV_TSP) {
    *reinterpret_cast<float4*>(row) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(row + e, v[e]);
  }
}

struct Layout {  // byte offsets into dynamic LDS
  int kgs, vs, kls, ob, dub, qb, pb, sact, srem, stime, sg, smask, spos, sval, xz, xa, sinfo, nact, ng, total;
};
__host__ __device__ inline Layout make_layout(int nt) {
  Layout L;
  const int plane = nt * 16 * kRS * 2, blk = 16 * kRS * 2;
  int o = 0;
  L.kgs = o; o += plane;
  L.vs = o; o += plane;
  L.kls = o; o += plane;
  L.ob = o; o += blk;
  L.dub = o; o += blk;
  L.qb = o; o += blk;
  L.pb = o; o += kWaves * blk;
  L.sact = o; o += kMaxT * 4;
  L.srem = o; o += kMaxT * 4;
  L.stime = o; o += kMaxT * 4;
  L.sg = o; o += kMaxT * 4;
  L.smask = o; o += kMaxT * 16;
  L.spos = o; o += 128 * 4;
  L.xz = o; o += kWaves * 16 * 2 * 4;
  L.xa = o; o += 16 * 4;
  L.sinfo = o; o += 16;
  L.sval = o; o += kMaxT;
  L.nact = o; o += kMaxT;      // the NEXT trajectory's actions (bytes; 255 = out of range) and
  L.ng = o; o += kMaxT * 4;    // upstream gradients, fetched under this trajectory's set-up
  L.total = (o + 15) & ~15;
  return L;
}

template <int ENV, int NT>
__global__ void __launch_bounds__(kThreads, 2) am_teacher_mma_kernel(const rl4co_am_teacher_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, tl = lane & 15, g = lane >> 4;
  const int h = w;  // attention stages: wave = head; logits stage: wave = node tile
  const int inst = blockIdx.x;
  const int N = a.N, T = a.T, S = a.B / a.B_inst;
  const int tpad = min(kMaxT, (T + 15) & ~15);  // columns the step blocks read
  // the second-dispatched half of an 8-wave workgroup loses the issue arbitration on every segment
  // (older wave first); a static priority for it evens the halves out between the barriers
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  const Layout L = make_layout(NT);  // NT node tiles of 16 (template): rows N .. 16 NT - 1 are zero
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
  int* sact = reinterpret_cast<int*>(smem + L.sact);
  float* srem = reinterpret_cast<float*>(smem + L.srem);
  float* stime = reinterpret_cast<float*>(smem + L.stime);  // CVRPTW: the clock before each column
  float* sg = reinterpret_cast<float*>(smem + L.sg);
  uint32_t* smask = reinterpret_cast<uint32_t*>(smem + L.smask);
  int* spos = reinterpret_cast<int*>(smem + L.spos);
  uint8_t* sval = smem + L.sval;
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
  __threadfence();  // the zeros reach L2 before this workgroup's atomics on the same rows

  uint32_t nv[4];  // nodes that exist, per 32-node word
#pragma unroll
  for (int k = 0; k < 4; ++k) nv[k] = (N >= 32 * (k + 1)) ? 0xffffffffu : (N > 32 * k ? ((1u << (N - 32 * k)) - 1u) : 0u);

  const float* ctxc = static_cast<const float*>(a.ctx_cur) + (int64_t)inst * N * kD + dcol;
  const float* ctxf = (ENV == RL4CO_ENV_TSP) ? static_cast<const float*>(a.ctx_first) + (int64_t)inst * N * kD + dcol : nullptr;
  constexpr bool kCvrpLike = ENV == RL4CO_ENV_CVRP || ENV == RL4CO_ENV_CVRPTW;
  constexpr bool kClock = ENV == RL4CO_ENV_CVRPTW;
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
  uint32_t errbits = 0;

  // actions / upstream gradients of a trajectory are fetched one trajectory AHEAD (thread t < kMaxT owns column t), so
  // the set-up of a trajectory never opens with an HBM round trip
  static_assert(kThreads >= kMaxT, "one thread per table column");
  int an = 0;
  float gn = 0.0f;
  if (tid < T) {
    an = (int)a.actions[(int64_t)inst * T + tid];
    gn = a.grad_logp[(int64_t)inst * T + tid];
  }
  if (tid < kMaxT) {
    snact[tid] = (an < 0 || an >= N) ? 255 : an;
    sng[tid] = gn;
  }

  for (int s = 0; s < S; ++s) {
    const int r = s * a.B_inst + inst;
    __syncthreads();  // the previous trajectory's tables are no longer read; the fetched columns are complete

    // ---- step tables of this trajectory (the environment replayed in closed form) -------------------
    // sact[t]: action; spos[j]: first column that visits node j (tsp/env.py:60-86, cvrp/env.py:66-96)
    if (tid < kMaxT) {
      int at = snact[tid];
      if (at == 255) {
        errbits |= RL4CO_EBIT_INFEASIBLE;
        at = 0;
      }
      sact[tid] = at;
    }
    an = 0;
    gn = 0.0f;
    if (s + 1 < S && tid < T) {  // in flight under the set-up below, staged after its last barrier
      an = (int)a.actions[(int64_t)(r + a.B_inst) * T + tid];
      gn = a.grad_logp[(int64_t)(r + a.B_inst) * T + tid];
    }
    for (int j = tid; j < 128; j += kThreads) spos[j] = 0x7fffffff;
    __syncthreads();
    // the two context fetches that open the first step block leave now, under the rest of the set-up
    const int first = sact[0];
    float f4[4] = {0.f, 0.f, 0.f, 0.f}, dqf[4] = {0.f, 0.f, 0.f, 0.f};
    if (ENV == RL4CO_ENV_TSP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) f4[e] = ctxf[(int64_t)first * kD + e];
    }
    // context row of this lane's step, fetched one step block ahead (an L2 round trip otherwise
    // opens every block's dependency chain)
    float4 c4n = *reinterpret_cast<const float4*>(ctxc + (int64_t)(tl == 0 ? 0 : sact[tl - 1]) * kD);
    for (int t = tid; t < T; t += kThreads) atomicMin(&spos[sact[t]], t);
    __syncthreads();
    if (tid == 0) {
      int t_end = T;
      if (kCvrpLike) {  // done once every node (depot included) has been visited (cvrp/env.py:80-83)
        int last = 0;
        for (int j = 0; j < N; ++j) last = max(last, spos[j]);
        if (last != 0x7fffffff) t_end = min(T, last + 1);
      }
      if (ENV == RL4CO_ENV_OP || ENV == RL4CO_ENV_PCTSP) {  // done at the first return to the depot after step 0 (op/env.py:84, pctsp/env.py:73)
        for (int t = 1; t < T; ++t)
          if (sact[t] == 0) {
            t_end = t + 1;
            break;
          }
      }
      sinfo[0] = t_end;
    }
    if (kClock) {
      // the clock BEFORE column t, replayed in visiting order: advance by the distance, wait for the window, serve;
      // back at the depot it restarts (cvrptw/env.py:97-113, same fp32 sequence as the decode kernel)
      for (int t = tid; t < tpad; t += kThreads) {
        float now = 0.0f;
        int prev = 0;
        for (int v = 0; v < min(t, T); ++v) {
          const int nx = sact[v];
          const float dx = twl[2 * nx] - twl[2 * prev], dy = twl[2 * nx + 1] - twl[2 * prev + 1];
          now = (nx != 0 ? 1.0f : 0.0f) * (fmaxf(now + sqrtf(fmaf(dy, dy, dx * dx)), tww[2 * nx]) + twd[nx]);
          prev = nx;
        }
        stime[t] = now;
      }
    }
    if (kCvrpLike) {
      // used capacity BEFORE column t: the loads since the last depot visit, summed in visiting
      // order from zero — the same fp32 sequence as used = (used + demand) * (action != 0)
      for (int t = tid; t < tpad; t += kThreads) {
        int u = min(t, T) - 1;
        while (u >= 0 && sact[u] != 0) --u;
        float used = 0.0f;
        for (int v = u + 1; v < min(t, T); ++v) used = used + dem[min(max(sact[v] - 1, 0), N - 2)];
        srem[t] = used;
      }
    }
    if (ENV == RL4CO_ENV_OP) {
      // tour length BEFORE column t, accumulated in visiting order like tour += |loc_a - loc_cur|
      for (int t = tid; t < tpad; t += kThreads) {
        float used = 0.0f;
        int prev = 0;
        for (int v = 0; v < min(t, T); ++v) {
          const int nx = sact[v];
          const float dx = oplocs[2 * nx] - oplocs[2 * prev], dy = oplocs[2 * nx + 1] - oplocs[2 * prev + 1];
          used = used + sqrtf(fmaf(dy, dy, dx * dx));
          prev = nx;
        }
        srem[t] = used;
      }
    }
    if (ENV == RL4CO_ENV_PDP) {  // no context scalar (context.py:232-243)
      for (int t = tid; t < tpad; t += kThreads) srem[t] = 0.0f;
    }
    if (ENV == RL4CO_ENV_PCTSP) {
      // prize collected BEFORE column t, accumulated in visiting order like prize += real_prize[a]
      for (int t = tid; t < tpad; t += kThreads) {
        float used = 0.0f;
        for (int v = 0; v < min(t, T); ++v) used = used + dem[sact[v]];
        srem[t] = used;
      }
    }
    __syncthreads();
    const int t_end = sinfo[0];
    if (ENV == RL4CO_ENV_TSP) {
      // feasibility words of a column in two ballots: wave w takes columns w, w + 8, ..; lane = node (and node + 64);
      // node j is feasible at column t until it has been visited, spos[j] >= t
      const int p0 = (lane < N) ? spos[lane] : -1, p1 = (lane + 64 < N) ? spos[lane + 64] : -1;
      for (int t = w; t < tpad; t += kWaves) {
        const unsigned long long b0 = __ballot(p0 >= t), b1 = __ballot(p1 >= t);
        if (lane == 0)
          *reinterpret_cast<uint4*>(smask + 4 * t) = (t < t_end) ? make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32))
                                                                 : make_uint4(1u, 0u, 0u, 0u);
      }
    }
    // feasibility words: thread (t, k) builds word k of column t
    for (int idx = tid; ENV != RL4CO_ENV_TSP && idx < tpad * 4; idx += kThreads) {
      const int t = idx >> 2, k = idx & 3;
      const bool live = t < t_end;
      uint32_t word = 0;
      if (ENV == RL4CO_ENV_TSP) {
        for (int b = 0; b < 32; ++b) {
          const int j = 32 * k + b;
          if (j < N && spos[j] >= t) word |= 1u << b;
        }
      } else if (ENV == RL4CO_ENV_OP) {
        // op/env.py:137-154: unvisited, depot not yet closed, and the node can still be entered
        const float used = srem[t];
        const int cur = (t == 0) ? 0 : sact[t - 1];
        const float cx = oplocs[2 * cur], cy = oplocs[2 * cur + 1];
        const bool depot_visited = spos[0] < t;
        for (int b = 0; b < 32; ++b) {
          const int j = 32 * k + b;
          if (j < N) {
            const float dx = oplocs[2 * j] - cx, dy = oplocs[2 * j + 1] - cy;
            const bool exceeds = used + sqrtf(fmaf(dy, dy, dx * dx)) > opmax[j];
            if (j == 0 || !(spos[j] < t || depot_visited || exceeds)) word |= 1u << b;
          }
        }
      } else if (ENV == RL4CO_ENV_PDP) {
        // pdp/env.py:64-99: unvisited, a delivery only once its pickup is on the tour; the depot only as the
        // forced first step of force_start_at_depot (recognised by the trajectory starting at node 0)
        const int half = (N - 1) / 2;
        for (int b = 0; b < 32; ++b) {
          const int j = 32 * k + b;
          if (j >= 1 && j < N && spos[j] >= t && (j <= half || spos[j - half] < t)) word |= 1u << b;
        }
        if (t == 0 && sact[0] == 0) word = (k == 0) ? 1u : 0u;
      } else if (ENV == RL4CO_ENV_PCTSP) {
        // pctsp/env.py:141-148: customers while unvisited and the depot not yet closed; the depot opens
        // once a total prize of 1 is collected or no customer is left
        const bool depot_visited = spos[0] < t;
        uint32_t left = 0;
        for (int b = 0; b < 32; ++b) {
          const int j = 32 * k + b;
          if (j >= 1 && j < N && spos[j] >= t) left |= 1u << b;
        }
        word = depot_visited ? 0u : left;
        left |= rl4co::bfly_i<1>((int)left);
        left |= rl4co::bfly_i<2>((int)left);
        if (k == 0 && !((srem[t] < 1.0f) && left != 0u)) word |= 1u;
      } else {
        const float used = srem[t];
        for (int b = 0; b < 32; ++b) {
          const int j = 32 * k + b;
          if (j >= 1 && j < N && spos[j] >= t && !(dem[j - 1] + used > thr)) word |= 1u << b;
        }
        // depot: infeasible only while standing on it with a customer still feasible (cvrp/env.py:126-136)
        uint32_t any = word;
        any |= rl4co::bfly_i<1>((int)any);
        any |= rl4co::bfly_i<2>((int)any);
        const int cur = (t == 0) ? 0 : sact[t - 1];
        if (k == 0 && !((cur == 0) && any != 0u)) word |= 1u;
        if (kClock) {  // cvrptw/env.py:91-95: only nodes whose window is still open on arrival (the depot too)
          const float now = stime[t], cx = twl[2 * cur], cy = twl[2 * cur + 1];
          for (int b = 0; b < 32; ++b) {
            const int j = 32 * k + b;
            if ((word >> b) & 1u) {
              const float dx = twl[2 * j] - cx, dy = twl[2 * j + 1] - cy;
              if (!(now + sqrtf(fmaf(dy, dy, dx * dx)) <= tww[2 * j + 1])) word &= ~(1u << b);
            }
          }
        }
      }
      smask[idx] = live ? word : (k == 0 ? 1u : 0u);  // dead columns: a finite dummy (node 0 only), gradient 0
    }
    for (int t = tid; t < tpad; t += kThreads) {
      const bool valid = t >= a.t0 && t < t_end;
      sval[t] = valid ? 1 : 0;
      sg[t] = valid ? sng[t] : 0.0f;
    }
    __syncthreads();
    if (tid < kMaxT) {
      snact[tid] = (an < 0 || an >= N) ? 255 : an;
      sng[tid] = gn;
    }
    if (ENV != RL4CO_ENV_TSP) {  // srem: used -> remaining capacity / length (context.py:147-149, 211-213), own entries only
      for (int t = tid; t < tpad; t += kThreads) {
        float rem = cap - srem[t];
        if (ENV == RL4CO_ENV_PCTSP && !(rem > 0.0f)) rem = 0.0f;
        srem[t] = rem;
      }
      __syncthreads();
    }

    const int ntb = (t_end + 15) >> 4;
    // the rows fetched during the set-up are "used" here, before the loop: otherwise the waitcnt state merged at the loop
    // header still counts them as in flight and every block's first stage waits for the fetch it has JUST issued
    asm volatile("" : "+v"(c4n.x), "+v"(c4n.y), "+v"(c4n.z), "+v"(c4n.w));
    if (ENV == RL4CO_ENV_TSP) asm volatile("" : "+v"(f4[0]), "+v"(f4[1]), "+v"(f4[2]), "+v"(f4[3]));
    float pend[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's deferred context-row scatter
    int pend_cur = -1;
    for (int tb = 0; tb < ntb; ++tb) {
      const int t = 16 * tb + tl;  // this lane's column (same in the four row groups)
      const int cur = (t == 0) ? 0 : sact[t - 1];
      const int at = sact[t];
      const float gt = sg[t];
      const bool valid = sval[t] != 0;
      const uint4 mw4 = *reinterpret_cast<const uint4*>(smask + 4 * t);
      const uint32_t mw[4] = {mw4.x, mw4.y, mw4.z, mw4.w};
      const float rem = (ENV != RL4CO_ENV_TSP) ? srem[t] : 0.0f;
      const float now = kClock ? stime[t] : 0.0f;

      // ---- 0. query of head h for the block's 16 steps (context.py:105-149, decoder.py:135-136) --
      bf16x4 qf;
      {
        const float4 c4 = c4n;
        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
        c4n = *reinterpret_cast<const float4*>(ctxc + (int64_t)sact[min(t + 15, kMaxT - 1)] * kD);  // cur of step t + 16
        // the PREVIOUS block's context-row scatter leaves here, behind the fetch: vmcnt is one in-order counter on
        // gfx9, so the wait for the row above also waits for every atomic issued before it — issued at the end of
        // their own block they stalled each block's first stage for their whole L2 round trip (tools/teacher_clock_probe.py)
        if (pend_cur >= 0) {
          scatter_row<ENV>(dcc + (int64_t)pend_cur * kD + dcol, pend);
          pend_cur = -1;
        }
        float q4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float q;
          if (ENV == RL4CO_ENV_TSP) q = (t == 0) ? qx4[e] + qb4[e] : (f4[e] + c[e]) + qb4[e];
          else if (kClock) q = fmaf(qt4[e], now, fmaf(qx4[e], rem, c[e])) + qb4[e];  // context.py:152-166
          else q = fmaf(qx4[e], rem, c[e]) + qb4[e];
          q4[e] = q * (0.25f * kLog2e);
        }
        qf = rl4co_e16::cvt4(q4[0], q4[1], q4[2], q4[3]);
        *reinterpret_cast<bf16x4*>(qb + tl * kRS + dcol) = qf;
      }

      // ---- 1. scores^T and softmax numerators over nodes (attention.py:300-314) -----------------------
      bf16x4 pf[NT];
      float inv_l;
      {
        f32x4 sc[NT];
        float m = kNegInf;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          {
            sc[jt] = mfma16(lds_b64(kgs + 16 * jt * kRS + 16 * h + nao), qf, zero4());
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
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          o = mfma16(lds_tr(vs + 16 * jt * kRS + 16 * h + tro), pf[jt], o);
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
          if (a.logp_out) a.logp_out[(int64_t)r * T + t] = lp;
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
      const bool scatter = valid && (ENV != RL4CO_ENV_TSP || t != 0);
      float4 row_old = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ENV == RL4CO_ENV_TSP && scatter)  // read of the read-modify-write below: a whole stage ahead of its use
        row_old = *reinterpret_cast<const float4*>(dcc + (int64_t)cur * kD + dcol);
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
            if (t == 0) dqx[e] += dqr;
            else dqf[e] += dqr;
          } else {
            dqx[e] = fmaf(dqr, rem, dqx[e]);
            if (kClock) dqt[e] = fmaf(dqr, now, dqt[e]);
          }
          pend[e] = old4[e] + dqr;  // d ctx_cur[cur] += dqr: leaves at the top of the next block (or after the last)
        }
        if (scatter) pend_cur = cur;
      }
      rl4co::lds_barrier();  // B4: the glimpse / d-logit blocks are rewritten by the next step block (LDS only:
                             // the context-row atomics stay in flight)
    }

    if (pend_cur >= 0) {  // the last block's scatter
      scatter_row<ENV>(dcc + (int64_t)pend_cur * kD + dcol, pend);
      pend_cur = -1;
    }
    // d ctx_first: one row per trajectory (every step after the first reads h[first])
    if (ENV == RL4CO_ENV_TSP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = step_sum(dqf[e]);
        if (tl == 0) unsafeAtomicAdd(a.d_ctx_first + ((int64_t)inst * N + first) * kD + dcol + e, v);
      }
    }
  }

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
}

}  // namespace

namespace rl4co {

#if !RL4CO_ELEM_F16
int teacher_mma_max_nodes() { return 16 * kMaxTiles; }
int teacher_mma_max_steps() { return kMaxT; }
#endif

template <int ENV, int NT>
static int launch_tiles(const rl4co_am_teacher_args& a, hipStream_t stream) {
  const Layout L = make_layout(NT);
  RL4CO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(am_teacher_mma_kernel<ENV, NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
  hipLaunchKernelGGL((am_teacher_mma_kernel<ENV, NT>), dim3(a.B_inst), dim3(kThreads), L.total, stream, a);
  RL4CO_HIP_TRY(hipGetLastError());
  return RL4CO_OK;
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
