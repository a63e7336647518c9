// MFMA attention for the text-encoder layers: bf16, head dim 64, up to 256 queries / keys per (sequence, head)
// (128: FCMF-base; 256: FCMF-large, in the <2,2> / 256-key instantiations further down).
//   forward : one workgroup per (sequence, head, 128-query tile); K and V tiles staged once in LDS;
//             each of the 4 waves owns 32 query rows: S^T = K Q^T on v_mfma_f32_16x16x32_bf16 with the
//             key index in the accumulator registers (row max / sum = 32 in-lane values + 2 shuffles);
//             the normalised, dropped-out P^T accumulators ARE the operand of O^T = V^T P^T (keys taken in
//             the order the accumulator registers hold them; V is consumed through ds_read_b64_tr_b16 in
//             that same order straight from its row-major image): 32 KiB of LDS, four workgroups per CU.
//   backward: one workgroup per (sequence, head), Tq <= 128: P is recomputed from the saved logsumexp.
//             The keys go by in chunks of 32: phase 1 (wave = 32 query rows) builds Pdrop^T and dS^T
//             [key][query] of the chunk in LDS, phase 2 adds the chunk to dQ (wave = 32 queries, accumulators
//             live across chunks) and finishes dV / dK of the chunk's keys (wave = 16 head-dim columns, the
//             transposed dO / Q operands stay in registers).  80 KiB of LDS: two workgroups per CU overlap
//             each other's loads, barriers and stores; no atomics.
// One LDS image per tile serves both row reads (ds_read_b128) and transposed reads
// (ds_read_b64_tr_b16); swizzles verified conflict-free with tools/lds_conflicts.py.
#include "common.h"
#include <cstdlib>

constexpr int AD = 64;             // head dim
constexpr int AT = 128;            // tile rows (queries / keys)
constexpr int TILE_B = AT * AD * 2;  // 16 KiB

// [rows][64 bf16] tile, 128-B rows: 32-B pair index XOR ((r>>1)&1 | ((r>>3)&1)<<1)
__device__ __forceinline__ int vkey(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }
__device__ __forceinline__ int off64(int r, int c8) { return r * 128 + ((((c8 >> 1) ^ vkey(r))) << 5) + ((c8 & 1) << 4); }
// [32 keys][128 q bf16] chunk images of the backward (Pdrop^T, dS^T), 256-B rows: 16-B chunk XOR key16(row & 15), a GF(2)-linear key
// (bit columns 2, 4, 8, 9) under which BOTH read patterns are conflict free -- the ds_read_b128 operand reads of phase 2 (row = lane & 15,
// chunk = 4 s + (lane >> 4)) and the ds_read_b64_tr_b16 reads of dS^T (tools/lds_conflicts.py searches the 4 x 4 bit matrices; the
// ds_write_b64 of phase 1 puts 16 rows of one column into a 128-B bank window and is 2-way under any key).  The round-3 key
// ((r & 3) << 2 | (r >> 2) & 3) left every operand read 2-way: 30 % of the kernel's LDS cycles were bank conflicts (r03_attn_pmc.txt).
__device__ __forceinline__ int key16(int r) { return ((r & 7) << 1) ^ (((r >> 3) & 1) * 9); }
__device__ __forceinline__ int off128(int r, int ch) { return r * 256 + ((ch ^ key16(r)) << 4); }
// per-wave P tile of the forward [32][128 bf16]: chunk XOR (row & 15)
__device__ __forceinline__ int offp(int r, int ch) { return r * 256 + ((ch ^ (r & 15)) << 4); }

// forward-only V image: 32-B pair index XOR ((r>>1)&3), conflict free for the KEY-PERMUTED transposed reads
// (k-slot j of k-step s = key 32s + 4*(lane>>4) + (j&3) + 16*(j>>2): the order in which the S^T accumulator
// registers of two adjacent key fragments line up as an MFMA operand, so P never leaves the registers)
__device__ __forceinline__ int off64p(int r, int c8) { return r * 128 + ((((c8 >> 1) ^ ((r >> 1) & 3))) << 5) + ((c8 & 1) << 4); }

typedef bf16x4 __attribute__((address_space(3))) * lds_v4_t;

// stage a [rows<=128][64] bf16 tile (row stride ld elements) into the off64 image, zero-filling
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* __restrict__ src, int64_t ld, int rows, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, r = c >> 3, c8 = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + c8 * 8);
    *reinterpret_cast<uint4*>(lds + off64(r, c8)) = v;
  }
}
__device__ __forceinline__ void stage_tile_p(char* lds, const bf16_t* __restrict__ src, int64_t ld, int rows, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, r = c >> 3, c8 = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + c8 * 8);
    *reinterpret_cast<uint4*>(lds + off64p(r, c8)) = v;
  }
}
// A[row = tile column c0 + (lane&15)][k = keys in the permuted order above] from the off64p image
__device__ __forceinline__ bf16x8 frag_tr64p(const char* lds, int c0, int s, int lane) {
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3;
  const int r = 32 * s + 4 * g4 + q4, c8 = (c0 >> 3) + (p >> 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off64p(r, c8) + (p & 1) * 8));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off64p(r + 16, c8) + (p & 1) * 8));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
// the same staging in two halves, so that a kernel can put ALL its tile loads in flight before the first LDS write
// (the fused form exposes one global round trip per tile)
struct TileRegs { uint4 v[4]; };
__device__ __forceinline__ TileRegs load_tile(const bf16_t* __restrict__ src, int64_t ld, int rows, int tid) {
  TileRegs t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, r = c >> 3, c8 = c & 7;
    t.v[i] = make_uint4(0, 0, 0, 0);
    if (r < rows) t.v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + c8 * 8);
  }
  return t;
}
template <bool PERMUTED>
__device__ __forceinline__ void store_tile(char* lds, const TileRegs& t, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, r = c >> 3, c8 = c & 7;
    *reinterpret_cast<uint4*>(lds + (PERMUTED ? off64p(r, c8) : off64(r, c8))) = t.v[i];
  }
}
// MFMA operand (rows x0..x0+15, k-step s over the 64 columns) by row read
__device__ __forceinline__ bf16x8 frag_row64(const char* lds, int x0, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + off64(x0 + (lane & 15), 4 * s + (lane >> 4)));
}
// MFMA operand whose "row" index is the tile COLUMN (c0..c0+15) and whose k index is the tile ROW
// (32*s .. 32*s+31): transposed read of the row-major image
__device__ __forceinline__ bf16x8 frag_tr64(const char* lds, int c0, int s, int lane) {
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3;
  const int r = 32 * s + 8 * g4 + q4, c8 = (c0 >> 3) + (p >> 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off64(r, c8) + (p & 1) * 8));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off64(r + 4, c8) + (p & 1) * 8));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
__device__ __forceinline__ bf16x8 frag_row128(const char* lds, int x0, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + off128(x0 + (lane & 15), 4 * s + (lane >> 4)));
}
__device__ __forceinline__ bf16x8 frag_tr128(const char* lds, int c0, int s, int lane) {
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3;
  const int r = 32 * s + 8 * g4 + q4, ch = (c0 >> 3) + (p >> 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off128(r, ch) + (p & 1) * 8));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(lds + off128(r + 4, ch) + (p & 1) * 8));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  *reinterpret_cast<bf16x4*>(p) = o;
}

// sum over the 16 lanes of a DPP row (lanes 16i .. 16i+15), result in every lane: four row rotations as DPP operands of
// the adds -- no LDS crossbar traffic (ds_bpermute), which a __shfl_xor butterfly would cost
__device__ __forceinline__ float row16_sum(float x) {
#define FCMF_ROR_ADD(n) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, true))
  FCMF_ROR_ADD(8); FCMF_ROR_ADD(4); FCMF_ROR_ADD(2); FCMF_ROR_ADD(1);
#undef FCMF_ROR_ADD
  return x;
}

struct AttnMfmaParams {
  const bf16_t *q, *k, *v, *o, *dout;
  const float* mask;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;
  float* colsum;       // backward, optional: [G][3 * heads * 64] f32, row g = column sums of dq | dk | dv of sequence g
  int G, heads, Tq, Tk;
  int64_t ldq, ldk, ldo;
  float scale, p;
  uint64_t seed;
};

// =========================================================================================
// NKT = 128-key tiles per (sequence, head): 1 (Tk <= 128: 32 KiB of LDS, three workgroups per CU) or 2 (Tk <= 256: the
// FCMF-large text encoder, 64 KiB, two per CU).  The images of the second tile lie right behind the first (row r of the
// 256-row image = row r - 128 of the second tile: the swizzle keys only use row bits 1..3).
// Q fragments of the wave's 32 query rows straight from global memory in operand layout (16 B per lane)
__device__ __forceinline__ void load_q_frags(const AttnMfmaParams& P, int g, int h, int q0, int lane, bf16x8 (&qf)[2][2]) {
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = q0 + 16 * f + (lane & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < P.Tq) v = *reinterpret_cast<const uint4*>(P.q + ((int64_t)g * P.Tq + row) * P.ldq + h * AD + 32 * s + 8 * (lane >> 4));
      qf[f][s] = *reinterpret_cast<bf16x8*>(&v);
    }
}

// one (sequence, head, 128-query tile) from the staged K / V images: scores, softmax, dropout, P V, store
// (mask_lds != NULL: the additive key mask of the tile was staged in LDS with K / V -- the persistent kernel must not issue a
// global load inside the tile: the counter of vector-memory operations retires in order, so waiting for it would also wait
// for the next tile's operands that are in flight behind it)
template <int NKT>
__device__ __forceinline__ void attn_mfma_fwd_tile(const AttnMfmaParams& P, int g, int h, int q0, int lane, const bf16x8 (&qf)[2][2],
                                                   const char* Ks, const char* Vs, const float* mask_lds = nullptr) {
  constexpr int NKF = 8 * NKT;       // 16-key fragments
  const float* mrow = P.mask ? P.mask + (int64_t)g * P.Tk : nullptr;
  // trailing keys under the "hard" additive mask (<= -1e30: HF's finfo.min padding mask) get probabilities that are EXACTLY
  // zero: their 16-key fragments (scores, softmax terms, P V steps) are skipped with bit-identical results.
  // nvalid = index of the last key that is not hard-masked, + 1 (wave-uniform).
  int nvalid = P.Tk;
  if (mask_lds || mrow) {
    int last = -1;
#pragma unroll
    for (int i = 0; i < 2 * NKT; ++i) {
      const int key = 64 * i + lane;
      const bool live = key < P.Tk && (mask_lds ? mask_lds[key] : mrow[key]) > -1e30f;
      const unsigned long long b = __ballot(live);
      if (b) last = 64 * i + 63 - __builtin_clzll(b);
    }
    nvalid = last >= 0 ? last + 1 : P.Tk;    // no live key at all: nothing may be skipped (softmax over finfo.min scores is uniform over ALL keys)
  }
  // S^T[key][q]: NKF key fragments x 2 query fragments
  f32x4 sc[NKF][2];
#pragma unroll
  for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
    for (int f = 0; f < 2; ++f) sc[kf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (16 * kf >= nvalid) continue;
      const bf16x8 ka = frag_row64(Ks, 16 * kf, s, lane);
#pragma unroll
      for (int f = 0; f < 2; ++f) sc[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[f][s], sc[kf][f], 0, 0, 0);
    }
  // lane holds, for query q0+16f+(lane&15), keys 16kf + 4(lane>>4) + r
  const float inv_keep = P.p > 0.f ? 1.0f / (1.0f - P.p) : 1.0f;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int q = q0 + 16 * f + (lane & 15);
    float m = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (16 * kf >= nvalid) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kf + 4 * (lane >> 4) + r;
        float s = -INFINITY;
        if (key < P.Tk) s = sc[kf][f][r] * P.scale + (mask_lds ? mask_lds[key] : (mrow ? mrow[key] : 0.f));
        sc[kf][f][r] = s;
        m = fmaxf(m, s);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (16 * kf >= nvalid) continue;         // (sc of a skipped fragment stays 0 = its probabilities)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(sc[kf][f][r] - m);
        sc[kf][f][r] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (q < P.Tq && (lane >> 4) == 0 && P.lse) P.lse[((int64_t)g * P.heads + h) * P.Tq + q] = m + __logf(sum);
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (16 * kf >= nvalid) continue;
      float dm[4] = {1.f, 1.f, 1.f, 1.f};
      if (P.p > 0.f)
        dropout_mult4(P.seed, ((uint64_t)g * P.heads + h) * P.Tq * P.Tk + (__umul24((unsigned)q, (unsigned)P.Tk) + (unsigned)(16 * kf + 4 * (lane >> 4))),
                      P.p, inv_keep, dm);
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[kf][f][r] = sc[kf][f][r] * inv * dm[r];
    }
  }
  // O^T[d][q] = sum_key V[key][d] P[q][key]
  f32x4 oc[4][2];
#pragma unroll
  for (int df = 0; df < 4; ++df)
#pragma unroll
    for (int f = 0; f < 2; ++f) oc[df][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4 * NKT; ++s) {
    if (32 * s >= nvalid) continue;
    // P^T operand of k-step s straight from the accumulators of key fragments 2s and 2s+1
    bf16x8 pb[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) { pb[f][r] = (bf16_t)sc[2 * s][f][r]; pb[f][4 + r] = (bf16_t)sc[2 * s + 1][f][r]; }
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const bf16x8 va = frag_tr64p(Vs, 16 * df, s, lane);
#pragma unroll
      for (int f = 0; f < 2; ++f) oc[df][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[f], oc[df][f], 0, 0, 0);
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int q = q0 + 16 * f + (lane & 15);
    if (q < P.Tq) {
      bf16_t* orow = P.out + ((int64_t)g * P.Tq + q) * P.ldo + h * AD + 4 * (lane >> 4);
#pragma unroll
      for (int df = 0; df < 4; ++df) store4(orow + 16 * df, oc[df][f]);
    }
  }
}

template <int NKT>
__device__ __forceinline__ void attn_mfma_fwd_body(const AttnMfmaParams& P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + NKT * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / P.heads, h = blockIdx.x % P.heads;
  const int q0 = blockIdx.y * AT + w * 32;
  const bf16_t* kbase_p = P.k + (int64_t)g * P.Tk * P.ldk + h * AD;
  const bf16_t* vbase_p = P.v + (int64_t)g * P.Tk * P.ldk + h * AD;
  const TileRegs kt = load_tile(kbase_p, P.ldk, P.Tk, tid);
  const TileRegs vt = load_tile(vbase_p, P.ldk, P.Tk, tid);
  bf16x8 qf[2][2];
  load_q_frags(P, g, h, q0, lane, qf);
  store_tile<false>(Ks, kt, tid);      // (all of K, V and Q were requested before the first LDS write)
  store_tile<true>(Vs, vt, tid);
  if constexpr (NKT == 2) {
    const TileRegs kt2 = load_tile(kbase_p + (int64_t)AT * P.ldk, P.ldk, P.Tk - AT, tid);
    const TileRegs vt2 = load_tile(vbase_p + (int64_t)AT * P.ldk, P.ldk, P.Tk - AT, tid);
    store_tile<false>(Ks + TILE_B, kt2, tid);
    store_tile<true>(Vs + TILE_B, vt2, tid);
  }
  __syncthreads();
  attn_mfma_fwd_tile<NKT>(P, g, h, q0, lane, qf, Ks, Vs);
}

// Persistent form for Tk <= 128 (the FCMF-base text encoder: 4608 (sequence, head) tiles per layer): a workgroup walks
// tiles blockIdx.x, + gridDim.x, ...; the K / V / Q loads of the NEXT tile are put in flight (into registers) before the
// current tile is computed and land in the other half of a double-buffered LDS image afterwards, so the global-load
// latency -- 58 % of a wave's lifetime in the one-tile-per-workgroup kernel (SQ_WAIT_ANY) -- hides under the softmax.
// 64 KiB of LDS, two workgroups per CU.
__global__ __launch_bounds__(256, 2) void attn_mfma_fwd_persist_kernel(AttnMfmaParams P, int ntiles, int qtiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  auto where = [&](int tile, int& g, int& h, int& q0) {
    const int gh = tile / qtiles;
    g = gh / P.heads; h = gh - g * P.heads; q0 = (tile - gh * qtiles) * AT + w * 32;
  };
  constexpr int BUF = 2 * TILE_B + 512;      // K image, V image, 128 mask floats
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  int g, h, q0;
  where(tile, g, h, q0);
  bf16x8 qf[2][2];
  auto mask_of = [&](int gg) { return (P.mask && tid < P.Tk) ? P.mask[(int64_t)gg * P.Tk + tid] : 0.f; };
  {
    const TileRegs kt = load_tile(P.k + (int64_t)g * P.Tk * P.ldk + h * AD, P.ldk, P.Tk, tid);
    const TileRegs vt = load_tile(P.v + (int64_t)g * P.Tk * P.ldk + h * AD, P.ldk, P.Tk, tid);
    const float mv = mask_of(g);
    load_q_frags(P, g, h, q0, lane, qf);
    store_tile<false>(smem, kt, tid);
    store_tile<true>(smem + TILE_B, vt, tid);
    if (tid < AT) reinterpret_cast<float*>(smem + 2 * TILE_B)[tid] = mv;
  }
  __syncthreads();
  int cur = 0;
  for (;;) {
    const int nxt = tile + gridDim.x;
    const bool more = nxt < ntiles;
    int g2 = g, h2 = h, q2 = q0;
    TileRegs kt, vt;
    bf16x8 qn[2][2];
    float mv = 0.f;
    if (more) {                              // next tile's operands: in flight under this tile's arithmetic
      where(nxt, g2, h2, q2);
      kt = load_tile(P.k + (int64_t)g2 * P.Tk * P.ldk + h2 * AD, P.ldk, P.Tk, tid);
      vt = load_tile(P.v + (int64_t)g2 * P.Tk * P.ldk + h2 * AD, P.ldk, P.Tk, tid);
      mv = mask_of(g2);
      load_q_frags(P, g2, h2, q2, lane, qn);
    }
    const char* buf = smem + cur * BUF;
    attn_mfma_fwd_tile<1>(P, g, h, q0, lane, qf, buf, buf + TILE_B, reinterpret_cast<const float*>(buf + 2 * TILE_B));
    if (!more) break;
    cur ^= 1;                                // (the other buffer was last read before the previous barrier)
    store_tile<false>(smem + cur * BUF, kt, tid);
    store_tile<true>(smem + cur * BUF + TILE_B, vt, tid);
    if (tid < AT) reinterpret_cast<float*>(smem + cur * BUF + 2 * TILE_B)[tid] = mv;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) qf[f][s2] = qn[f][s2];
    g = g2; h = h2; q0 = q2; tile = nxt;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 3) void attn_mfma_fwd_kernel(AttnMfmaParams P) { attn_mfma_fwd_body<1>(P); }
__global__ __launch_bounds__(256, 2) void attn_mfma_fwd256_kernel(AttnMfmaParams P) { attn_mfma_fwd_body<2>(P); }

// =========================================================================================
// backward.  NQT = 128-query tiles, NKT = 128-key tiles handled by ONE workgroup per (sequence, head):
//   <1,1>  Tq, Tk <= 128 : 80 KiB of LDS, two workgroups per CU (the FCMF-base text encoder);
//   <2,2>  Tq, Tk <= 256 : 144 KiB, one workgroup per CU = one wave per SIMD with the whole register file (the
//          FCMF-large text encoder, S = 256): the fragments of BOTH query tiles stay in registers, the key chunks go by
//          once, and for every chunk the two query tiles take turns in the two chunk images, so that dK / dV of the chunk
//          accumulate over all 256 queries in registers and are stored once -- no partial buffers, no atomics.
template <int NQT, int NKT>
__device__ __forceinline__ void attn_mfma_bwd_body(const AttnMfmaParams& P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                               // NQT tiles
  char* dOs = Qs + NQT * TILE_B;                 // NQT tiles
  char* Ks = dOs + NQT * TILE_B;                 // NKT tiles
  char* Vs = Ks + NKT * TILE_B;                  // NKT tiles
  char* PdT = Vs + NKT * TILE_B;                 // [32 keys][128 q] bf16 of the current (key chunk, query tile), 8 KiB
  char* dST = PdT + TILE_B / 2;                  // 8 KiB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / P.heads, h = blockIdx.x % P.heads;
  const int64_t qbase = (int64_t)g * P.Tq, kbase = (int64_t)g * P.Tk;
  // delta[q] = sum_d dO[q][d] O[q][d] for the wave's 32 query rows of every query tile: two lanes per row, then every
  // lane picks the values of the rows its accumulator registers hold
  float dl4[NQT][2][4], lse4[NQT][2][4];
  {
    TileRegs tq[NQT], td[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      tq[qt] = load_tile(P.q + (qbase + qt * AT) * P.ldq + h * AD, P.ldq, P.Tq - qt * AT, tid);
      td[qt] = load_tile(P.dout + (qbase + qt * AT) * P.ldo + h * AD, P.ldo, P.Tq - qt * AT, tid);
    }
    TileRegs tk = load_tile(P.k + kbase * P.ldk + h * AD, P.ldk, P.Tk, tid);
    TileRegs tv = load_tile(P.v + kbase * P.ldk + h * AD, P.ldk, P.Tk, tid);
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      const int q = qt * AT + 32 * w + (lane >> 1), half = lane & 1;
      float sd = 0.f;
      if (q < P.Tq) {
        const bf16_t* a = P.dout + (qbase + q) * P.ldo + h * AD + 32 * half;
        const bf16_t* b = P.o + (qbase + q) * P.ldo + h * AD + 32 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + 8 * i);
          const bf16x8 y = *reinterpret_cast<const bf16x8*>(b + 8 * i);
#pragma unroll
          for (int j = 0; j < 8; ++j) sd += (float)x[j] * (float)y[j];
        }
      }
      sd += __shfl_xor(sd, 1, 64);
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = 16 * f + 4 * (lane >> 4) + r;          // row inside the wave's 32
          dl4[qt][f][r] = __shfl(sd, 2 * ql, 64);
          const int q2 = qt * AT + 32 * w + ql;
          lse4[qt][f][r] = q2 < P.Tq ? P.lse[((int64_t)g * P.heads + h) * P.Tq + q2] : 0.f;
        }
    }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {     // (every global read of the prologue was requested before the first LDS write)
      store_tile<false>(Qs + qt * TILE_B, tq[qt], tid);
      store_tile<false>(dOs + qt * TILE_B, td[qt], tid);
    }
    store_tile<false>(Ks, tk, tid);
    store_tile<false>(Vs, tv, tid);
    if constexpr (NKT == 2) {
      tk = load_tile(P.k + (kbase + AT) * P.ldk + h * AD, P.ldk, P.Tk - AT, tid);
      tv = load_tile(P.v + (kbase + AT) * P.ldk + h * AD, P.ldk, P.Tk - AT, tid);
      store_tile<false>(Ks + TILE_B, tk, tid);
      store_tile<false>(Vs + TILE_B, tv, tid);
    }
  }
  __syncthreads();

  const float inv_keep = P.p > 0.f ? 1.0f / (1.0f - P.p) : 1.0f;
  const uint64_t drop_base = ((uint64_t)g * P.heads + h) * P.Tq * P.Tk;      // dropout counter of (query 0, key 0)
  const float* mrow = P.mask ? P.mask + kbase : nullptr;
  float score_scale = P.scale;            // (0 for a sequence whose every key is hard-masked: see below)
  // operands that stay in registers, per query tile: the wave's 32 query rows of Q and dO (phase 1) and the transposed
  // 16-column slices dO^T / Q^T [d = 16w ..][q] that dV / dK of every key chunk multiply (phase 2)
  bf16x8 qa[NQT][2][2], da[NQT][2][2], oT[NQT][4], qT[NQT][4];
  f32x4 aQ[NQT][4][2];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        qa[qt][f][s] = frag_row64(Qs + qt * TILE_B, 32 * w + 16 * f, s, lane);
        da[qt][f][s] = frag_row64(dOs + qt * TILE_B, 32 * w + 16 * f, s, lane);
      }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      oT[qt][s] = frag_tr64(dOs + qt * TILE_B, 16 * w, s, lane);     // A[row = d][k = q]
      qT[qt][s] = frag_tr64(Qs + qt * TILE_B, 16 * w, s, lane);
    }
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int f = 0; f < 2; ++f) aQ[qt][df][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- key chunks of 32; per chunk the query tiles take turns: phase 1 (wave = 32 query rows) builds Pdrop^T and dS^T
  // [key][q] of (chunk, query tile) in LDS, phase 2 adds them to dQ of that tile (wave = 32 queries) and to dV / dK of
  // the chunk's 32 keys (wave = 16 columns), which are stored after the last query tile
  const int nchunks_all = (P.Tk + 31) >> 5;
  // trailing keys whose additive mask is the "hard" value (<= -1e30: HF's finfo.min padding mask) have probabilities that
  // are EXACTLY zero (exp underflows), so whole 32-key chunks of them contribute nothing to dQ and get dK = dV = 0: they
  // are skipped, with bit-identical results.  nvalid = index of the last key that is not hard-masked, + 1 (wave-uniform).
  int nvalid = P.Tk;
  if (mrow) {
    int last = -1;
#pragma unroll
    for (int i = 0; i < 2 * NKT; ++i) {
      const int key = 64 * i + lane;
      const bool live = key < P.Tk && mrow[key] > -1e30f;
      const unsigned long long b = __ballot(live);
      if (b) last = 64 * i + 63 - __builtin_clzll(b);
    }
    nvalid = last >= 0 ? last + 1 : P.Tk;    // no live key at all: nothing may be skipped (softmax over finfo.min scores is uniform over ALL keys)
    // ... and the forward's logsumexp of such a sequence is finfo.min itself (log Tk is absorbed), from which exp(s + mask - lse)
    // would recompute probabilities of 1 instead of 1 / Tk.  Wave-uniform substitution, no per-element work: drop the mask and
    // the scores and take lse = log Tk.
    if (last < 0) {
      mrow = nullptr;
      score_scale = 0.f;
      const float ltk = __logf((float)P.Tk);
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) lse4[qt][f][r] = ltk;
    }
  }
  const int nchunks = (nvalid + 31) >> 5;
  // Dropout mask, shared between neighbouring lanes.  One hash decides the element PAIR (q, key & ~1), (q, key | 1); here the two
  // elements sit in lanes l and l ^ 1 (key = .. + (lane & 15)), each holding rows q0 .. q0+3 -- computed per element, both lanes
  // would evaluate the same four hashes (two 32-bit multiplies each, quarter rate: two thirds of this kernel's vector
  // instructions).  Even lanes hash rows q0, q0+1, odd lanes q0+2, q0+3, and a quad-permute DPP move swaps them: two hashes per
  // four elements.  Needs an even Tk (then every row starts on a pair boundary); the high word of the 64-bit pair index is
  // wave-uniform unless the low word wraps inside this (sequence, head) -- checked, else the per-element form below.
  const uint64_t pair_base = drop_base >> 1;
  const uint32_t pb_lo = (uint32_t)pair_base, pb_hi = (uint32_t)(pair_base >> 32);
  const bool share_hash = P.p > 0.f && (P.Tk & 1) == 0 && pb_lo <= 0xFFFF0000u;      // (pair offsets stay below 2^15: Tq, Tk <= 256)
  const uint32_t hash_k0 = (uint32_t)P.seed ^ ((pb_hi << 7) | (pb_hi >> 25)), hash_s1 = (uint32_t)(P.seed >> 32);
  const uint32_t drop_thr = dropout_threshold(P.p);
  const int odd = lane & 1;
  f32x4 cV = f32x4{0.f, 0.f, 0.f, 0.f}, cK = f32x4{0.f, 0.f, 0.f, 0.f};   // column sums of dV / dK over the keys (this lane's keys)
  for (int c = 0; c < nchunks; ++c) {
    f32x4 aV[2], aK[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) { aV[kf] = f32x4{0.f, 0.f, 0.f, 0.f}; aK[kf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      f32x4 sS[2][2], sP[2][2];
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
        for (int f = 0; f < 2; ++f) { sS[k4][f] = f32x4{0.f, 0.f, 0.f, 0.f}; sP[k4][f] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
          const bf16x8 kb = frag_row64(Ks, 32 * c + 16 * k4, s, lane);
          const bf16x8 vb = frag_row64(Vs, 32 * c + 16 * k4, s, lane);
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            sS[k4][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qt][f][s], kb, sS[k4][f], 0, 0, 0);   // D[q][key]
            sP[k4][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[qt][f][s], vb, sP[k4][f], 0, 0, 0);
          }
        }
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
        const int kl = 16 * k4 + (lane & 15), key = 32 * c + kl;
        const float mk = (mrow && key < P.Tk) ? mrow[key] : 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          f32x4 pdv, dsv;
          float mult4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
          if (share_hash) {
            const unsigned q0 = qt * AT + 32 * w + 16 * f + 4 * (lane >> 4);
            const uint32_t o0 = pb_lo + __umul24(q0 + 2 * odd, (unsigned)P.Tk >> 1) + ((unsigned)key >> 1);
            const uint32_t hA = fcmf_hash32_rounds(o0 ^ hash_k0, hash_s1);
            const uint32_t hB = fcmf_hash32_rounds((o0 + ((unsigned)P.Tk >> 1)) ^ hash_k0, hash_s1);
            const uint32_t nA = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hA, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]: lane ^ 1
            const uint32_t nB = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hB, 0xB1, 0xf, 0xf, true);
            const uint32_t hr[4] = {odd ? nA : hA, odd ? nB : hB, odd ? hA : nA, odd ? hB : nB};
#pragma unroll
            for (int r = 0; r < 4; ++r) mult4[r] = ((hr[r] >> (16 * odd)) & 0xFFFFu) >= drop_thr ? inv_keep : 0.f;   // (key & 1 == lane & 1)
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qt * AT + 32 * w + 16 * f + 4 * (lane >> 4) + r;
            float pr = 0.f, mult = mult4[r];
            if (key < P.Tk && q < P.Tq) pr = __expf(sS[k4][f][r] * score_scale + mk - lse4[qt][f][r]);
            // element index ((g heads + h) Tq + q) Tk + key = a wave-uniform 64-bit base + a small 24-bit product: no
            // 64-bit vector multiply per element (integer multiplies are quarter rate; the kernel is VALU-bound)
            if (P.p > 0.f && !share_hash) mult = dropout_mult(P.seed, drop_base + (__umul24((unsigned)q, (unsigned)P.Tk) + (unsigned)key), P.p, inv_keep);
            pdv[r] = pr * mult;
            dsv[r] = pr * (sP[k4][f][r] * mult - dl4[qt][f][r]);
          }
          const int ch = 4 * w + 2 * f + (lane >> 5);
          const int o = off128(kl, ch) + ((lane >> 4) & 1) * 8;
          store4(reinterpret_cast<bf16_t*>(PdT + o), pdv);
          store4(reinterpret_cast<bf16_t*>(dST + o), dsv);
        }
      }
      __syncthreads();
      // dQ^T[d][q] += K^T[d][keys of the chunk] dS^T[keys][q]
      {
        bf16x8 tb[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) tb[f] = frag_tr128(dST, 32 * w + 16 * f, 0, lane);    // B[k = key][col = q]
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          const bf16x8 ka2 = frag_tr64(Ks, 16 * df, c, lane);                              // A[row = d][k = key]
#pragma unroll
          for (int f = 0; f < 2; ++f) aQ[qt][df][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka2, tb[f], aQ[qt][df][f], 0, 0, 0);
        }
      }
      // dV^T / dK^T [d = 16w..][key of the chunk] += dO^T / Q^T [d][q of this tile] x Pdrop / dS [q][key]
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
          const bf16x8 pb = frag_row128(PdT, 16 * kf, s, lane);   // B[k = q][col = key]
          const bf16x8 sb = frag_row128(dST, 16 * kf, s, lane);
          aV[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oT[qt][s], pb, aV[kf], 0, 0, 0);
          aK[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[qt][s], sb, aK[kf], 0, 0, 0);
        }
      __syncthreads();   // the chunk images are rewritten by the next query tile / chunk
    }
    const int dcol = h * AD + 16 * w + 4 * (lane >> 4);
    cV += aV[0] + aV[1];        // (keys past Tk carry zero probabilities: no mask needed)
    cK += aK[0] + aK[1];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      const int key = 32 * c + 16 * kf + (lane & 15);
      if (key < P.Tk) {
        store4(P.dv + (kbase + key) * P.ldk + dcol, aV[kf]);
        f32x4 t = aK[kf];
        t[0] *= P.scale; t[1] *= P.scale; t[2] *= P.scale; t[3] *= P.scale;
        store4(P.dk + (kbase + key) * P.ldk + dcol, t);
      }
    }
  }
  for (int c = nchunks; c < nchunks_all; ++c) {      // the skipped (fully hard-masked) chunks: dK = dV = 0
    const int dcol = h * AD + 16 * w + 4 * (lane >> 4);
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      const int key = 32 * c + 16 * kf + (lane & 15);
      if (key < P.Tk) {
        store4(P.dv + (kbase + key) * P.ldk + dcol, f32x4{0.f, 0.f, 0.f, 0.f});
        store4(P.dk + (kbase + key) * P.ldk + dcol, f32x4{0.f, 0.f, 0.f, 0.f});
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int x = qt * AT + 32 * w + 16 * f + (lane & 15);
      if (x < P.Tq) {
        const int dcol = h * AD + 4 * (lane >> 4);
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          f32x4 t = aQ[qt][df][f];
          t[0] *= P.scale; t[1] *= P.scale; t[2] *= P.scale; t[3] *= P.scale;
          store4(P.dq + (qbase + x) * P.ldq + dcol + 16 * df, t);
        }
      }
    }
  // ---- optional: column sums of this (sequence, head)'s dq | dk | dv (f32 accumulators, before the bf16 rounding): the
  // bias gradient of the fused q|k|v projection is their sum over the sequences -- saves a pass over dqkv
  if (P.colsum) {
    const int HD = P.heads * AD;
    float* row = P.colsum + (int64_t)g * 3 * HD + h * AD;
#pragma unroll
    for (int r = 0; r < 4; ++r) { cV[r] = row16_sum(cV[r]); cK[r] = row16_sum(cK[r]); }
    if ((lane & 15) == 0) {      // d = 16 w + 4 (lane >> 4) + r: the workgroup owns these 64 columns of row g
      const int d0 = 16 * w + 4 * (lane >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { row[2 * HD + d0 + r] = cV[r]; row[HD + d0 + r] = cK[r] * P.scale; }
    }
    // dq: rows q = .. + (lane & 15) over the fragments, query tiles and the four waves (through LDS; PdT is free now)
    float* red = reinterpret_cast<float*>(PdT);    // [4 waves][64 d]
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int f = 0; f < 2; ++f) t += aQ[qt][df][f];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = row16_sum(t[r]);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w * 64 + 16 * df + 4 * (lane >> 4) + r] = t[r];
      }
    }
    __syncthreads();
    if (tid < 64) row[tid] = (red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid]) * P.scale;
  }
}

__global__ __launch_bounds__(256, 2) void attn_mfma_bwd_kernel(AttnMfmaParams P) { attn_mfma_bwd_body<1, 1>(P); }
__global__ __launch_bounds__(256, 1) void attn_mfma_bwd256_kernel(AttnMfmaParams P) { attn_mfma_bwd_body<2, 2>(P); }

// =========================================================================================
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int fcmf_attn_mfma_fwd(const void* q, const void* k, const void* v, const float* mask, void* out, float* lse,
                                  int G, int heads, int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                                  float dropout_p, uint64_t seed, void* stream) {
  if (!q || !k || !v || !out || G <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0) return FCMF_ERR_ARG;
  if (Tk > 2 * AT || ldq % 8 || ldk % 8 || ldo % 4 || !al16(q) || !al16(k) || !al16(v) || !al16(out)) return FCMF_ERR_UNSUPPORTED;
  AttnMfmaParams P{};
  P.q = (const bf16_t*)q; P.k = (const bf16_t*)k; P.v = (const bf16_t*)v; P.mask = mask; P.out = (bf16_t*)out; P.lse = lse;
  P.G = G; P.heads = heads; P.Tq = Tq; P.Tk = Tk; P.ldq = ldq; P.ldk = ldk; P.ldo = ldo;
  P.scale = scale; P.p = dropout_p; P.seed = seed;
  const dim3 grid(G * heads, (Tq + AT - 1) / AT);
  if (Tk > AT) {
    const int smem = 4 * TILE_B;
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_fwd256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr2 = true; }
    hipLaunchKernelGGL(attn_mfma_fwd256_kernel, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  const int ntiles = (int)(grid.x * grid.y);
  static const bool no_persist = getenv("FCMF_ATTN_NO_PERSIST") != nullptr;      // A/B switch for benchmarks
  if (ntiles >= 4 * 512 && !no_persist) {       // enough tiles for every resident workgroup (2 per CU) to walk several
    const int smem = 2 * (2 * TILE_B + 512);
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_fwd_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr3 = true; }
    hipLaunchKernelGGL(attn_mfma_fwd_persist_kernel, dim3(512), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P, ntiles, (int)grid.y);
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  const int smem = 2 * TILE_B;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  hipLaunchKernelGGL(attn_mfma_fwd_kernel, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_attn_mfma_bwd(const void* q, const void* k, const void* v, const float* mask, const void* out,
                                  const void* dout, const float* lse, void* dq, void* dk, void* dv, int G, int heads,
                                  int Tq, int Tk, int64_t ldq, int64_t ldk, int64_t ldo, float scale, float dropout_p,
                                  uint64_t seed, float* colsum, void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || G <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0) return FCMF_ERR_ARG;
  if (Tk > 2 * AT || Tq > 2 * AT || ldq % 8 || ldk % 8 || ldo % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(out) || !al16(dout) ||
      !al16(dq) || !al16(dk) || !al16(dv))
    return FCMF_ERR_UNSUPPORTED;
  AttnMfmaParams P{};
  P.q = (const bf16_t*)q; P.k = (const bf16_t*)k; P.v = (const bf16_t*)v; P.mask = mask; P.o = (const bf16_t*)out;
  P.dout = (const bf16_t*)dout; P.lse = const_cast<float*>(lse); P.dq = (bf16_t*)dq; P.dk = (bf16_t*)dk; P.dv = (bf16_t*)dv;
  P.G = G; P.heads = heads; P.Tq = Tq; P.Tk = Tk; P.ldq = ldq; P.ldk = ldk; P.ldo = ldo;
  P.scale = scale; P.p = dropout_p; P.seed = seed; P.colsum = colsum;
  if (Tk > AT || Tq > AT) {
    const int smem = 9 * TILE_B;   // 2 x (Q, dO) + 2 x (K, V) tiles + the two 8 KiB chunk images = 144 KiB: one workgroup per CU
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_bwd256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr2 = true; }
    hipLaunchKernelGGL(attn_mfma_bwd256_kernel, dim3(G * heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  const int smem = 5 * TILE_B;   // Q, K, V, dO tiles + the two 8 KiB chunk images = 80 KiB: two workgroups per CU
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
  hipLaunchKernelGGL(attn_mfma_bwd_kernel, dim3(G * heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
