// Two-segment multi-head attention, forward and backward, VALU f32 math, any activation dtype.
// One workgroup = one (group g, head h).  The T1 shared keys/values and a block of query rows (with dO in
// the backward) are staged once in LDS; one WAVE owns one query row at a time: its scores over the shared
// keys run one key per lane out of LDS, its T2 private keys are scored one key per lane (the lane walks its key's
// row with 16-byte loads) and their values / keys are accumulated 8 keys x 64 dims per wave instruction (head
// dim 64) or one key per step with the head dimension across the lanes (other head dims).  The
// backward is two passes per row block: pass A (wave = row) builds the dropped probabilities and score
// gradients [row][key] in LDS and finishes dq / dbias (and dk2 / dv2 rows, unless the private keys are shared by a
// group of sequences: then attn_private_grad_kernel sums them over the group), pass B (wave = shared key) reduces
// them over the rows into dk1 / dv1 -- two workgroup barriers per row block, no per-row global round trips.
// Used for every attention on the path in fp32 (parity) mode and, in bf16 mode, for the dead-row-pruned
// fusion layers (1 live query row per image), the 15-token fusion layer, the geometry-biased ROI attention
// and the IAOG decoder; the bf16 text-encoder attention runs on attn_mfma.hip instead.
#include "common.h"

constexpr int AS_MAXT = 512;   // T1 + T2 <= 512 (forward: KPL = 4 or 8 keys per lane; FCMF-large fuses 256 text + 100 ROI keys)
constexpr int AS_MAXD = 128;   // head dim <= 128 (two elements per lane)

struct AttnK {
  fcmf_attn_desc a;
  void* out; float* lse;
  const void* dout; const void* o_in;
  void *dq, *dk1, *dv1, *dk2, *dv2; float* dbias;
  float *pd2, *ds2;            // backward, grouped private keys: [G][heads][R][T2] dropped probabilities / score gradients
  int RB;                      // query rows staged per block (host: what fits the LDS budget)
  int TLP;                     // pitch of the per-row key arrays of the backward (max keys a block sees)
  int KVF;                     // floats occupied by the staged shared K and V images
};

// stage n rows of a (.., t, h, :) tensor into an f32 LDS image with row pitch `pitch`.  Aligned rows go 16 bytes
// per lane, several rows per wave instruction, four instructions in flight per wave; otherwise lanes run across
// the head dimension (coalesced) one row at a time.
template <typename TT, typename TD>   // TD = float (converted) or TT (raw copy: shared keys/values stay in the activation dtype)
__device__ __forceinline__ void stage_rows(TD* dst, int pitch, const TT* src, int64_t row_stride, int n, int d, int w, int lane,
                                           int nw = 4) {
  constexpr int V = 16 / sizeof(TT);
  const bool vec = d % V == 0 && row_stride % V == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  if (vec) {
    const int lpr = d / V, rpi = 64 / lpr;            // lanes per row, rows per wave instruction
    const int rl = lane / lpr, ch = lane - rl * lpr;
    const bool on = rl < rpi;
    for (int t0 = w * rpi; t0 < n; t0 += 4 * nw * rpi) {   // nw waves x 4 instructions per round
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * nw * rpi + rl;
        v[u] = make_uint4(0, 0, 0, 0);
        if (on && t < n) v[u] = *reinterpret_cast<const uint4*>(src + (int64_t)t * row_stride + ch * V);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * nw * rpi + rl;
        if (on && t < n) {
          TD* o = dst + t * pitch + ch * V;
          if constexpr (sizeof(TD) == sizeof(TT)) {
            *reinterpret_cast<uint4*>(o) = v[u];
          } else if constexpr (sizeof(TT) == 2) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(&v[u]);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)x[j];
          } else {
            const float4 x = *reinterpret_cast<const float4*>(&v[u]);
            o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
          }
        }
      }
    }
  } else {
    for (int t = w; t < n; t += nw) {
      const TT* sr = src + (int64_t)t * row_stride;
      for (int c = lane; c < d; c += 64) {
        if constexpr (sizeof(TD) == sizeof(TT)) dst[t * pitch + c] = sr[c];
        else dst[t * pitch + c] = to_f32<TT>(sr[c]);
      }
    }
  }
}

// <x, row> out of LDS (x f32, a broadcast; row in the activation dtype): rows with a 16-byte aligned pitch are
// read 16 bytes at a time
template <typename TT>
__device__ __forceinline__ float lds_dot(const float* __restrict__ x, const TT* __restrict__ row, int d, bool v16) {
  float s = 0.f;
  constexpr int V = 16 / sizeof(TT);
  if (v16) {
    for (int c = 0; c < d; c += V) {
      if constexpr (sizeof(TT) == 2) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(row + c);
        const float4 a0 = *reinterpret_cast<const float4*>(x + c), a1 = *reinterpret_cast<const float4*>(x + c + 4);
        s += a0.x * (float)b[0] + a0.y * (float)b[1] + a0.z * (float)b[2] + a0.w * (float)b[3] +
             a1.x * (float)b[4] + a1.y * (float)b[5] + a1.z * (float)b[6] + a1.w * (float)b[7];
      } else {
        const float4 a = *reinterpret_cast<const float4*>(x + c), b = *reinterpret_cast<const float4*>(row + c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
    }
  } else {
    for (int c = 0; c < d; ++c) s += x[c] * to_f32<TT>(row[c]);
  }
  return s;
}

// <q, row> and <q2, row2> for one private key held by this lane: the lane walks its own global row with 16-byte
// loads (all of them independent, so they are in flight together; q comes from LDS as a broadcast)
template <typename TT>
__device__ __forceinline__ void dot2_row(const float* __restrict__ qa, const TT* __restrict__ ra, const float* __restrict__ qb,
                                         const TT* __restrict__ rb, int d, bool vec, float& sa, float& sb) {
  sa = 0.f; sb = 0.f;
  constexpr int V = 16 / sizeof(TT);
  if (vec) {
    for (int c = 0; c < d; c += V) {
      if constexpr (sizeof(TT) == 2) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(ra + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) sa += qa[c + j] * (float)x[j];
        if (rb) {
          const bf16x8 y = *reinterpret_cast<const bf16x8*>(rb + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) sb += qb[c + j] * (float)y[j];
        }
      } else {
        const float4 x = *reinterpret_cast<const float4*>(ra + c);
        sa += qa[c] * x.x + qa[c + 1] * x.y + qa[c + 2] * x.z + qa[c + 3] * x.w;
        if (rb) {
          const float4 y = *reinterpret_cast<const float4*>(rb + c);
          sb += qb[c] * y.x + qb[c + 1] * y.y + qb[c + 2] * y.z + qb[c + 3] * y.w;
        }
      }
    }
  } else {
    for (int c = 0; c < d; ++c) {
      sa += qa[c] * to_f32<TT>(ra[c]);
      if (rb) sb += qb[c] * to_f32<TT>(rb[c]);
    }
  }
}
template <typename TT>
__device__ __forceinline__ bool private_rows_vectorisable(const fcmf_attn_desc& a) {
  constexpr int V = 16 / sizeof(TT);
  return a.T2 > 0 && a.d % V == 0 && a.k2_sg % V == 0 && a.k2_sr % V == 0 && a.k2_st % V == 0 &&
         (reinterpret_cast<uintptr_t>(a.k2) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.v2) & 15) == 0;
}

// DC = head dimension when it is 64 (the text / fusion layers of FCMF: loops unroll, the 16-byte loads of a private key
// row are all in flight together; 274 -> 120 us on the step's two fusion attentions), 0 = taken from the descriptor.
// (Eight waves per workgroup -- one per query row of the 7-row fusion layers -- were measured slower than four.)
template <typename TT, int KPL, int DC>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnK P) {
  constexpr int NW = 4;
  constexpr int PMAX = 64 * KPL;          // keys a wave's probability row can hold
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = DC ? DC : a.d;
  const int T1 = a.T1, T2 = a.T2, T = T1 + T2, RB = P.RB;
  constexpr int V16 = 16 / sizeof(TT);
  const bool v4 = d % V16 == 0;         // 16-byte LDS accesses throughout
  const int dp = v4 ? d + V16 : d + 1;  // K/V row pitch (elements): +16 B keeps 16-byte reads of 16 lanes on distinct banks; else odd
  TT* K1s = reinterpret_cast<TT*>(sm);
  TT* V1s = K1s + T1 * dp;
  float* Qs = sm + P.KVF;               // [RB][d] (KVF = floats taken by the two K/V images)
  float* ps = Qs + RB * d;              // [NW][PMAX] probability row of each wave
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int hin = a.head_quirk ? (int)(((int64_t)h * a.G + g) % a.heads) : h;
  const int g2 = g / a.group_div;
  const TT* Q = reinterpret_cast<const TT*>(a.q);
  const TT* K2 = reinterpret_cast<const TT*>(a.k2);
  const TT* V2 = reinterpret_cast<const TT*>(a.v2);
  TT* O = reinterpret_cast<TT*>(P.out);
  if (T1 > 0) {
    stage_rows<TT, TT>(K1s, dp, reinterpret_cast<const TT*>(a.k1) + (int64_t)g * a.k1_sg + hin * d, a.k1_st, T1, d, w, lane, NW);
    stage_rows<TT, TT>(V1s, dp, reinterpret_cast<const TT*>(a.v1) + (int64_t)g * a.k1_sg + hin * d, a.k1_st, T1, d, w, lane, NW);
  }
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  float* p = ps + w * PMAX;
  const bool vec2 = private_rows_vectorisable<TT>(a);
  for (int rb0 = 0; rb0 < a.R; rb0 += RB) {
    const int nr = min(RB, a.R - rb0);
    stage_rows<TT, float>(Qs, d, Q + (int64_t)g * a.q_sg + (int64_t)rb0 * a.q_sr + hin * d, a.q_sr, nr, d, w, lane, NW);
    __syncthreads();
    for (int rl = w; rl < nr; rl += NW) {
      const int r = rb0 + rl;
      const float* q = Qs + rl * d;
      float sc[KPL];
#pragma unroll
      for (int n = 0; n < KPL; ++n) {
        const int t = lane + 64 * n;
        float accv = 0.f;
        if (t < T1) {
          accv = lds_dot<TT>(q, K1s + t * dp, d, v4);
        } else if (t < T) {   // private key: this lane walks its own row
          float dummy;
          dot2_row<TT>(q, K2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)(t - T1) * a.k2_st + hin * d, nullptr,
                       (const TT*)nullptr, d, vec2, accv, dummy);
        }
        sc[n] = accv;
      }
      float m = -INFINITY;
#pragma unroll
      for (int n = 0; n < KPL; ++n) {
        const int t = lane + 64 * n;
        float s = -INFINITY;
        if (t < T) {
          s = sc[n] * a.scale;
          if (a.mask) s += a.mask[(int64_t)g * T + t];
          if (a.bias) s += a.bias[(((int64_t)g2 * a.heads + h) * a.R + r) * T + t];
          if (a.causal && t > r) s = -1e4f;
        }
        sc[n] = s;
        m = fmaxf(m, s);
      }
      m = wave_max(m);
      float sum = 0.f;
#pragma unroll
      for (int n = 0; n < KPL; ++n) {
        const int t = lane + 64 * n;
        const float e = t < T ? __expf(sc[n] - m) : 0.f;
        sc[n] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int n = 0; n < KPL; ++n) {
        const int t = lane + 64 * n;
        if (t < T) {
          float pv = sc[n] * inv;
          if (a.dropout_p > 0.f)
            pv *= dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * a.R + r) * T + t, a.dropout_p, inv_keep);
          p[t] = pv;
        }
      }
      if (lane == 0 && P.lse) P.lse[((int64_t)g * a.heads + h) * a.R + r] = m + __logf(sum);
      // out[c] = sum_t p[t] v[t][c]: lanes across the head dimension (p is read back by the wave that wrote it)
      float o0 = 0.f, o1 = 0.f;
      {
        int t = 0;
        for (; v4 && t + 4 <= T1; t += 4) {
          const float4 pv = *reinterpret_cast<const float4*>(p + t);
          const TT* vr = V1s + t * dp;
          auto f = [&](int i) { return to_f32<TT>(vr[i]); };
          if (lane < d) o0 += pv.x * f(lane) + pv.y * f(dp + lane) + pv.z * f(2 * dp + lane) + pv.w * f(3 * dp + lane);
          if (lane + 64 < d) o1 += pv.x * f(lane + 64) + pv.y * f(dp + lane + 64) + pv.z * f(2 * dp + lane + 64) + pv.w * f(3 * dp + lane + 64);
        }
        for (; t < T1; ++t) {
          const float pv = p[t];
          if (lane < d) o0 += pv * to_f32<TT>(V1s[t * dp + lane]);
          if (lane + 64 < d) o1 += pv * to_f32<TT>(V1s[t * dp + lane + 64]);
        }
      }
      if (T2 > 0 && DC == 64 && sizeof(TT) == 2 && vec2) {
        // private values, 16 bytes per lane: lane = (key t2 % 8, 8-dim slice): one wave instruction covers 8 keys x 64
        // dims, every instruction of the row is in flight before the first use; the 8 key groups are then summed across
        // lanes (lanes that share lane & 7 hold the same dims)
        const TT* vb = V2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + hin * d + (lane & 7) * 8;
        const int kg = lane >> 3;
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < T2; t0 += 32) {
          bf16x8 x[4];
          float pw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t2 = t0 + 8 * u + kg;
            pw[u] = 0.f;
            x[u] = bf16x8{};
            if (t2 < T2) { x[u] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)t2 * a.k2_st); pw[u] = p[T1 + t2]; }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] += pw[u] * (float)x[u][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = acc8[j];
          t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
          acc8[j] = t;
        }
        // lane l < 64 owns dim l: it sits in slice l >> 3, element l & 7 -- held by every lane with (lane & 7) == l >> 3
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = __shfl(acc8[j], lane >> 3, 64);     // from lane (l >> 3): its slice is l >> 3
          if ((lane & 7) == j) mine = t;
        }
        o0 += mine;
      } else if (T2 > 0) {
        const TT* vb = V2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + hin * d;
#pragma unroll 8
        for (int t2 = 0; t2 < T2; ++t2) {
          const float pv = p[T1 + t2];
          const TT* vr = vb + (int64_t)t2 * a.k2_st;
          if (lane < d) o0 += pv * to_f32<TT>(vr[lane]);
          if (lane + 64 < d) o1 += pv * to_f32<TT>(vr[lane + 64]);
        }
      }
      TT* orow = O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
      if (lane < d) orow[lane] = from_f32<TT>(o0);
      if (lane + 64 < d) orow[lane + 64] = from_f32<TT>(o1);
    }
    __syncthreads();   // the row block is restaged
  }
}

// Backward.  grid = (G*heads, ceil(T1/128)): block (.., c) owns the shared keys [128c, 128c+128) (their
// dK1/dV1 accumulate in registers across the row blocks, so no atomics are needed) and, for c == 0, the
// private keys.  Each chunk writes its partial dq to dq[c] (dense [chunks,G,R,heads*d]); the host sums the
// chunks.  dk1/dv1 are dense [G,T1,heads*d].
// ONE_BLOCK: all R rows fit one staged block (the usual case): dk1/dv1 are final after pass B and go straight
// to memory; otherwise they accumulate in registers across the row blocks.
// Grouped private keys (P.pd2 != nullptr): rows of `group_div` consecutive groups share their private keys (the six
// aspects of a review), so dk2 / dv2 are sums over the group.  Pass A then only stores the private keys' dropped
// probabilities and score gradients (a few KB per workgroup) and attn_private_grad_kernel forms the sums -- instead of
// one dk2 / dv2 row pair per (group, row, key) written here and summed by another pass (6x the bytes, twice).
// (Four waves and a run-time head dimension: eight waves per workgroup and a compile-time d = 64 were both measured
// SLOWER here -- 180 VGPRs instead of 88 halve the resident waves.)
template <typename TT, bool ONE_BLOCK, bool GROUPED>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(AttnK P) {
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T1 = a.T1, T2 = a.T2, T = T1 + T2, RB = P.RB;
  constexpr int V16 = 16 / sizeof(TT);
  const bool v4 = d % V16 == 0;
  const int dp = v4 ? d + V16 : d + 1;
  const int chunk = blockIdx.y, kc0 = chunk * 128;
  const int T1c = min(128, T1 - kc0);                 // shared keys of this chunk (may be <= 0 if T1 == 0)
  const int T2c = chunk == 0 ? T2 : 0;                // private keys are handled by chunk 0
  const int nsh = T1c > 0 ? T1c : 0;
  const int TL = nsh + T2c;                           // keys this block sees: local index tl
  TT* K1s = reinterpret_cast<TT*>(sm);
  TT* V1s = K1s + nsh * dp;
  float* Qs = sm + P.KVF;            // [RB][d]
  float* dOs = Qs + RB * d;          // [RB][d]
  const int TLP = P.TLP;
  float* PD = dOs + RB * d;          // [RB][TLP] dropped probabilities
  float* DS = PD + RB * TLP;         // [RB][TLP] score gradients
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int hin = a.head_quirk ? (int)(((int64_t)h * a.G + g) % a.heads) : h;
  const int g2 = g / a.group_div;
  const int64_t HD = (int64_t)a.heads * d;
  const TT* Q = reinterpret_cast<const TT*>(a.q);
  const TT* K2 = reinterpret_cast<const TT*>(a.k2);
  const TT* V2 = reinterpret_cast<const TT*>(a.v2);
  const TT* O = reinterpret_cast<const TT*>(P.o_in);
  const TT* dO = reinterpret_cast<const TT*>(P.dout);
  TT* dQ = reinterpret_cast<TT*>(P.dq) + (int64_t)chunk * a.G * a.R * HD;
  TT* dK1 = reinterpret_cast<TT*>(P.dk1);
  TT* dV1 = reinterpret_cast<TT*>(P.dv1);
  TT* dK2 = reinterpret_cast<TT*>(P.dk2);
  TT* dV2 = reinterpret_cast<TT*>(P.dv2);
  if (nsh > 0) {
    stage_rows<TT, TT>(K1s, dp, reinterpret_cast<const TT*>(a.k1) + (int64_t)g * a.k1_sg + (int64_t)kc0 * a.k1_st + hin * d, a.k1_st, nsh, d, w, lane, NW);
    stage_rows<TT, TT>(V1s, dp, reinterpret_cast<const TT*>(a.v1) + (int64_t)g * a.k1_sg + (int64_t)kc0 * a.k1_st + hin * d, a.k1_st, nsh, d, w, lane, NW);
  }
  constexpr int NACC = ONE_BLOCK ? 1 : 128 / NW;      // shared keys of a chunk per wave: w + NW n
  float accK[NACC][2], accV[NACC][2];
#pragma unroll
  for (int n = 0; n < NACC; ++n) { accK[n][0] = accK[n][1] = accV[n][0] = accV[n][1] = 0.f; }
  const bool kv_same = (P.dv1 == nullptr);
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const bool vec2 = private_rows_vectorisable<TT>(a);

  for (int rb0 = 0; rb0 < a.R; rb0 += RB) {
    const int nr = min(RB, a.R - rb0);
    stage_rows<TT, float>(Qs, d, Q + (int64_t)g * a.q_sg + (int64_t)rb0 * a.q_sr + hin * d, a.q_sr, nr, d, w, lane, NW);
    stage_rows<TT, float>(dOs, d, dO + (int64_t)g * a.o_sg + (int64_t)rb0 * a.o_sr + h * d, a.o_sr, nr, d, w, lane, NW);
    __syncthreads();
    // ---- pass A: wave = query row ---------------------------------------------------------------
    for (int rl = w; rl < nr; rl += NW) {
      const int r = rb0 + rl;
      const float* q = Qs + rl * d;
      const float* dov = dOs + rl * d;
      float* pd = PD + rl * TLP;
      float* ds = DS + rl * TLP;
      const float q0 = lane < d ? q[lane] : 0.f, q1 = lane + 64 < d ? q[lane + 64] : 0.f;
      const float o0 = lane < d ? dov[lane] : 0.f, o1 = lane + 64 < d ? dov[lane + 64] : 0.f;
      float delta;
      {
        const TT* orow = O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
        float part = 0.f;
        if (lane < d) part = o0 * to_f32<TT>(orow[lane]);
        if (lane + 64 < d) part += o1 * to_f32<TT>(orow[lane + 64]);
        delta = wave_sum(part);
      }
      // raw scores and dO.v per key: local key tl = lane + 64 n  (shared chunk keys first, then private)
      float sv[4], dv[4];
      const int64_t p2 = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + hin * d;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int tl = lane + 64 * n;
        float s = 0.f, dpd = 0.f;
        if (tl < nsh) {
          s = lds_dot<TT>(q, K1s + tl * dp, d, v4);
          dpd = lds_dot<TT>(dov, V1s + tl * dp, d, v4);
        } else if (tl < TL) {   // private key: this lane walks its own rows of K2 and V2
          const int64_t o2 = p2 + (int64_t)(tl - nsh) * a.k2_st;
          dot2_row<TT>(q, K2 + o2, dov, V2 + o2, d, vec2, s, dpd);
        }
        sv[n] = s; dv[n] = dpd;
      }
      const float lse_r = P.lse[((int64_t)g * a.heads + h) * a.R + r];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int tl = lane + 64 * n;
        if (tl < TL) {
          const int t = tl < nsh ? kc0 + tl : T1 + (tl - nsh);   // global key index
          float s = sv[n] * a.scale;
          if (a.mask) s += a.mask[(int64_t)g * T + t];
          if (a.bias) s += a.bias[(((int64_t)g2 * a.heads + h) * a.R + r) * T + t];
          const bool filled = a.causal && t > r;
          if (filled) s = -1e4f;
          // (a row whose every key carries the hard finfo.min mask: its logsumexp IS finfo.min -- log T is absorbed -- and the
          //  probabilities are uniform, as torch's softmax gives them)
          const float pr = lse_r <= -1e30f ? 1.0f / (float)T : __expf(s - lse_r);
          float mult = 1.0f;
          if (a.dropout_p > 0.f)
            mult = dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * a.R + r) * T + t, a.dropout_p, inv_keep);
          const float dsv = filled ? 0.f : pr * (dv[n] * mult - delta);
          pd[tl] = pr * mult;
          ds[tl] = dsv;
          if (P.dbias) P.dbias[(((int64_t)g * a.heads + h) * a.R + r) * T + t] = dsv;
          if (GROUPED && tl >= nsh) {
            const int64_t i2 = (((int64_t)g * a.heads + h) * a.R + r) * T2 + (tl - nsh);
            P.pd2[i2] = pr * mult;
            P.ds2[i2] = dsv;
          }
        }
      }
      // dq[c] = scale * sum_t ds[t] k[t][c]; private keys also get their dk2 / dv2 rows (outer products)
      float dq0 = 0.f, dq1 = 0.f;
      {
        int tl = 0;
        for (; v4 && tl + 4 <= nsh; tl += 4) {
          const float4 dsv = *reinterpret_cast<const float4*>(ds + tl);
          const TT* kr = K1s + tl * dp;
          auto f = [&](int i) { return to_f32<TT>(kr[i]); };
          if (lane < d) dq0 += dsv.x * f(lane) + dsv.y * f(dp + lane) + dsv.z * f(2 * dp + lane) + dsv.w * f(3 * dp + lane);
          if (lane + 64 < d) dq1 += dsv.x * f(lane + 64) + dsv.y * f(dp + lane + 64) + dsv.z * f(2 * dp + lane + 64) + dsv.w * f(3 * dp + lane + 64);
        }
        for (; tl < nsh; ++tl) {
          const float dsv = ds[tl];
          if (lane < d) dq0 += dsv * to_f32<TT>(K1s[tl * dp + lane]);
          if (lane + 64 < d) dq1 += dsv * to_f32<TT>(K1s[tl * dp + lane + 64]);
        }
      }
      if (GROUPED && T2c > 0 && d == 64 && sizeof(TT) == 2 && vec2) {
        // grouped private keys: only dq needs them here.  16 bytes per lane, lane = (key t2 % 8, 8-dim slice): every load
        // of the row is in flight before the first use; the 8 key groups are summed across lanes afterwards
        const TT* kb = K2 + p2 + (lane & 7) * 8;
        const int kg = lane >> 3;
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < T2c; t0 += 32) {
          bf16x8 x[4];
          float wgt[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t2 = t0 + 8 * u + kg;
            wgt[u] = 0.f;
            x[u] = bf16x8{};
            if (t2 < T2c) { x[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)t2 * a.k2_st); wgt[u] = ds[nsh + t2]; }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] += wgt[u] * (float)x[u][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = acc8[j];
          t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
          acc8[j] = t;
        }
        float mine = 0.f;          // lane l owns dim l = slice l >> 3, element l & 7
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = __shfl(acc8[j], lane >> 3, 64);
          if ((lane & 7) == j) mine = t;
        }
        dq0 += mine;
      } else if (GROUPED) {
#pragma unroll 4
        for (int t2 = 0; t2 < T2c; ++t2) {
          const float dsv = ds[nsh + t2];
          const TT* kr = K2 + p2 + (int64_t)t2 * a.k2_st;
          if (lane < d) dq0 += dsv * to_f32<TT>(kr[lane]);
          if (lane + 64 < d) dq1 += dsv * to_f32<TT>(kr[lane + 64]);
        }
      } else
#pragma unroll 4
      for (int t2 = 0; t2 < T2c; ++t2) {
        const float dsv = ds[nsh + t2], pv = pd[nsh + t2];
        const TT* kr = K2 + p2 + (int64_t)t2 * a.k2_st;
        const int64_t dst = (((int64_t)g * a.R + r) * T2 + t2) * HD + h * d;
        if (lane < d) {
          dq0 += dsv * to_f32<TT>(kr[lane]);
          dK2[dst + lane] = from_f32<TT>(dsv * a.scale * q0);
          dV2[dst + lane] = from_f32<TT>(pv * o0);
        }
        if (lane + 64 < d) {
          dq1 += dsv * to_f32<TT>(kr[lane + 64]);
          dK2[dst + lane + 64] = from_f32<TT>(dsv * a.scale * q1);
          dV2[dst + lane + 64] = from_f32<TT>(pv * o1);
        }
      }
      TT* dqrow = dQ + ((int64_t)g * a.R + r) * HD + h * d;
      if (lane < d) dqrow[lane] = from_f32<TT>(dq0 * a.scale);
      if (lane + 64 < d) dqrow[lane + 64] = from_f32<TT>(dq1 * a.scale);
    }
    __syncthreads();
    // ---- pass B: wave = shared key (w + 4n): reduce over the rows of the block -----------------------
    auto key_sums = [&](int tl, float& k0, float& k1, float& v0, float& v1) {
      k0 = k1 = v0 = v1 = 0.f;
      for (int rl = 0; rl < nr; ++rl) {
        const float dsv = DS[rl * TLP + tl], pv = PD[rl * TLP + tl];
        if (lane < d) { k0 += dsv * Qs[rl * d + lane]; v0 += pv * dOs[rl * d + lane]; }
        if (lane + 64 < d) { k1 += dsv * Qs[rl * d + lane + 64]; v1 += pv * dOs[rl * d + lane + 64]; }
      }
      k0 *= a.scale; k1 *= a.scale;
    };
    auto write_key = [&](int tl, float k0, float k1, float v0, float v1) {
      const int64_t off = ((int64_t)g * T1 + kc0 + tl) * HD + h * d;
      if (lane < d) {
        dK1[off + lane] = from_f32<TT>(kv_same ? k0 + v0 : k0);
        if (!kv_same) dV1[off + lane] = from_f32<TT>(v0);
      }
      if (lane + 64 < d) {
        dK1[off + lane + 64] = from_f32<TT>(kv_same ? k1 + v1 : k1);
        if (!kv_same) dV1[off + lane + 64] = from_f32<TT>(v1);
      }
    };
    if constexpr (ONE_BLOCK) {
      for (int tl = w; tl < nsh; tl += NW) {
        float k0, k1, v0, v1;
        key_sums(tl, k0, k1, v0, v1);
        write_key(tl, k0, k1, v0, v1);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        const int tl = w + NW * n;
        if (tl < nsh) {
          float k0, k1, v0, v1;
          key_sums(tl, k0, k1, v0, v1);
          accK[n][0] += k0; accK[n][1] += k1; accV[n][0] += v0; accV[n][1] += v1;
        }
      }
      __syncthreads();   // the row block is restaged
      if (rb0 + RB >= a.R) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
          const int tl = w + NW * n;
          if (tl < nsh) write_key(tl, accK[n][0], accK[n][1], accV[n][0], accV[n][1]);
        }
      }
    }
  }
}

// dk2 / dv2 of grouped private keys: block = (g2, r) = one set of T2 private keys; thread = output column c (head c / d):
//   dk2[g2][r][t2][c] = scale * sum_a ds2[g2*gd + a][head][r][t2] * q[g2*gd + a][r][c]
//   dv2[g2][r][t2][c] =         sum_a pd2[...]                     * dO[g2*gd + a][r][c]
// (pd2 / ds2 of the group staged in LDS as [a][head][t2]; every output row is one coalesced store)
template <typename TT>
__global__ __launch_bounds__(256) void attn_private_grad_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T2 = a.T2, gd = a.group_div, heads = a.heads, HD = heads * d;
  const int g2 = blockIdx.x / a.R, r = blockIdx.x % a.R;
  const int tb = (T2 + gridDim.y - 1) / gridDim.y, t_lo = blockIdx.y * tb, t_hi = min(T2, t_lo + tb);   // this block's keys
  const int nt = t_hi - t_lo;              // this block's keys only (every block staged all T2 keys of the group: 4x the gather)
  float* PD = sm;                          // [gd][heads][nt]
  float* DS = sm + gd * heads * tb;
  for (int i = threadIdx.x; i < gd * heads * nt; i += 256) {
    const int tl = i % nt, ah = i / nt, hh = ah % heads, aa = ah / heads;
    const int64_t src = ((((int64_t)g2 * gd + aa) * heads + hh) * a.R + r) * T2 + t_lo + tl;
    PD[i] = P.pd2[src];
    DS[i] = P.ds2[src];
  }
  __syncthreads();
  const TT* Q = reinterpret_cast<const TT*>(a.q);
  const TT* dO = reinterpret_cast<const TT*>(P.dout);
  TT* dK2 = reinterpret_cast<TT*>(P.dk2);
  TT* dV2 = reinterpret_cast<TT*>(P.dv2);
  for (int c = threadIdx.x; c < HD; c += 256) {
    const int hh = c / d;
    float qv[8], ov[8];
#pragma unroll
    for (int aa = 0; aa < 8; ++aa) {
      qv[aa] = ov[aa] = 0.f;
      if (aa < gd) {
        const int64_t g = (int64_t)g2 * gd + aa;
        qv[aa] = to_f32<TT>(Q[g * a.q_sg + (int64_t)r * a.q_sr + c]) * a.scale;
        ov[aa] = to_f32<TT>(dO[g * a.o_sg + (int64_t)r * a.o_sr + c]);
      }
    }
    const int64_t out0 = (((int64_t)g2 * a.R + r) * T2) * HD + c;
    for (int t2 = t_lo; t2 < t_hi; ++t2) {
      float kacc = 0.f, vacc = 0.f;
#pragma unroll
      for (int aa = 0; aa < 8; ++aa)
        if (aa < gd) {
          kacc += DS[(aa * heads + hh) * nt + t2 - t_lo] * qv[aa];
          vacc += PD[(aa * heads + hh) * nt + t2 - t_lo] * ov[aa];
        }
      dK2[out0 + (int64_t)t2 * HD] = from_f32<TT>(kacc);
      dV2[out0 + (int64_t)t2 * HD] = from_f32<TT>(vacc);
    }
  }
}

// =========================================================================================================================
// "Tiny dense" attention: one shared key segment of <= 64 keys, <= 64 query rows, bf16, no causal fill (the geometry-biased
// ROI attention: 36 x 36, 8 heads of 96; 4 us of arithmetic per (group, head)).  The general kernels above walk the query
// rows one wave at a time (9 rounds of dependent LDS / reduction latencies here); with everything in LDS this one is a few
// flat, fully parallel loops: thread = (row, key) for the scores, wave = row for the softmax, thread = (row, column) for the
// outputs.  Same arithmetic, same dropout counters, same outputs as the general kernels.
// =========================================================================================================================
constexpr int TINY_MAX = 64;        // rows / keys of the one-block kernels
constexpr int TINY_WIDE_MAX = 128;  // ... of the wide forward / row-blocked backward (LDS permitting)
constexpr size_t TINY_LDS = 160 * 1024;

__device__ __forceinline__ float dot_bf16_lds(const bf16_t* a, const bf16_t* b, int d) {      // d % 8 == 0, 16-byte aligned rows
  // v_dot2c_f32_bf16: two bf16 products added to an f32 accumulator per instruction, no conversions (two interleaved
  // accumulation chains)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  float s0 = 0.f, s1 = 0.f;
  for (int c = 0; c < d; c += 8) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + c), y = *reinterpret_cast<const bf16x8*>(b + c);
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[0], x[1]}, bf16x2_t{y[0], y[1]}, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[2], x[3]}, bf16x2_t{y[2], y[3]}, s1, false);
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[4], x[5]}, bf16x2_t{y[4], y[5]}, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[6], x[7]}, bf16x2_t{y[6], y[7]}, s1, false);
  }
  return s0 + s1;
}

// one 16 x 16 fragment of X Y^T out of two row-major LDS images (rows of d bf16, pitch dp, d % 32 == 0) on the matrix cores:
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane for row (lane & 15) of either operand at k offset 8 (lane >> 4) --
// the same 16-byte read for both (pitch 2 (d + 8) bytes: the 16 rows of a quarter wave start in distinct bank groups).  The lane
// ends up with rows x0 + 4 (lane >> 4) + i, i = 0..3, of column y0 + (lane & 15).  Rows past the end of an operand read whatever
// follows it inside the workgroup's LDS allocation and only reach outputs that are never stored.
__device__ __forceinline__ f32x4 frag_xyT(const bf16_t* X, const bf16_t* Y, int dp, int d, int x0, int y0, int lane) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* xr = X + (x0 + (lane & 15)) * dp + (lane >> 4) * 8;
  const bf16_t* yr = Y + (y0 + (lane & 15)) * dp + (lane >> 4) * 8;
  for (int k0 = 0; k0 < d; k0 += 32)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xr + k0), *reinterpret_cast<const bf16x8*>(yr + k0), acc, 0, 0, 0);
  return acc;
}

template <bool WIDE>      // WIDE: 65..128 keys, two per lane in the softmax
__global__ __launch_bounds__(256) void attn_tiny_fwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T = a.T1, R = a.R, dp = d + 8, TP = T + 1;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);
  bf16_t* Ks = Qs + R * dp;
  bf16_t* Vs = Ks + T * dp;
  float* Ss = reinterpret_cast<float*>(Vs + T * dp);     // [R][TP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads, g2 = g / a.group_div;
  stage_rows<bf16_t, bf16_t>(Qs, dp, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)g * a.q_sg + h * d, a.q_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Ks, dp, reinterpret_cast<const bf16_t*>(a.k1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Vs, dp, reinterpret_cast<const bf16_t*>(a.v1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  // additive terms (bias tile + key mask) staged with the operands: one coalesced pass, all loads in flight together (a
  // global load per score inside the loop below serialises on its latency)
  for (int idx = tid; idx < R * T; idx += 256) {
    const int r = idx / T, t = idx - r * T;
    float add = 0.f;
    if (a.mask) add = a.mask[(int64_t)g * T + t];
    if (a.bias) add += a.bias[(((int64_t)g2 * a.heads + h) * R) * T + idx];
    Ss[r * TP + t] = add;
  }
  __syncthreads();
  if ((d & 31) == 0) {                                   // scores on the matrix cores: wave = 16 x 16 fragments w, w + 4, ...
    const int nfc = (T + 15) >> 4, nf = ((R + 15) >> 4) * nfc;
    for (int f = w; f < nf; f += 4) {
      const int r0 = (f / nfc) * 16, t0 = (f % nfc) * 16;
      const f32x4 acc = frag_xyT(Qs, Ks, dp, d, r0, t0, lane);
      const int t = t0 + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * (lane >> 4) + i;
        if (r < R && t < T) Ss[r * TP + t] += acc[i] * a.scale;
      }
    }
  } else {
    for (int idx = tid; idx < R * T; idx += 256) {
      const int r = idx / T, t = idx - r * T;
      Ss[r * TP + t] += dot_bf16_lds(Qs + r * dp, Ks + t * dp, d) * a.scale;
    }
  }
  __syncthreads();
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  for (int r = w; r < R; r += 4) {
    const float s = lane < T ? Ss[r * TP + lane] : -INFINITY;
    float s1 = -INFINITY;
    if (WIDE) s1 = lane + 64 < T ? Ss[r * TP + lane + 64] : -INFINITY;
    const float m = wave_max(WIDE ? fmaxf(s, s1) : s);
    const float e = lane < T ? __expf(s - m) : 0.f;
    const float e1 = WIDE && lane + 64 < T ? __expf(s1 - m) : 0.f;
    const float sum = wave_sum(WIDE ? e + e1 : e);
    if (lane < T) {
      float pv = e / sum;
      if (a.dropout_p > 0.f) pv *= dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + lane, a.dropout_p, inv_keep);
      Ss[r * TP + lane] = pv;
    }
    if (WIDE && lane + 64 < T) {
      float pv = e1 / sum;
      if (a.dropout_p > 0.f) pv *= dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + lane + 64, a.dropout_p, inv_keep);
      Ss[r * TP + lane + 64] = pv;
    }
    if (lane == 0 && P.lse) P.lse[((int64_t)g * a.heads + h) * R + r] = m + __logf(sum);
  }
  __syncthreads();
  bf16_t* O = reinterpret_cast<bf16_t*>(P.out);
  const int d8 = d >> 3;
  for (int idx = tid; idx < R * d8; idx += 256) {        // thread = (row, 8 columns): one 16-byte V read per key
    const int r = idx / d8, c = (idx - r * d8) * 8;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
      const float pv = Ss[r * TP + t];
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(Vs + t * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += pv * (float)v[j];
    }
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)o[j];
    *reinterpret_cast<bf16x8*>(O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d + c) = ov;
  }
}

__global__ __launch_bounds__(256) void attn_tiny_bwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T = a.T1, R = a.R, dp = d + 8, TP = T + 1;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);
  bf16_t* dOs = Qs + R * dp;
  bf16_t* Ks = dOs + R * dp;
  bf16_t* Vs = Ks + T * dp;
  float* PD = reinterpret_cast<float*>(Vs + T * dp);     // [R][TP] dropped probabilities
  float* DS = PD + R * TP;                               // [R][TP] score gradients
  float* dl = DS + R * TP;                               // [R] delta = <dO, O>
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads, g2 = g / a.group_div;
  const int64_t HD = (int64_t)a.heads * d;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(P.dout);
  const bf16_t* O = reinterpret_cast<const bf16_t*>(P.o_in);
  stage_rows<bf16_t, bf16_t>(Qs, dp, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)g * a.q_sg + h * d, a.q_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(dOs, dp, dO + (int64_t)g * a.o_sg + h * d, a.o_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Ks, dp, reinterpret_cast<const bf16_t*>(a.k1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Vs, dp, reinterpret_cast<const bf16_t*>(a.v1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  for (int idx = tid; idx < R * T; idx += 256) {         // additive terms staged up front (see the forward kernel)
    const int r = idx / T, t = idx - r * T;
    float add = 0.f;
    if (a.mask) add = a.mask[(int64_t)g * T + t];
    if (a.bias) add += a.bias[(((int64_t)g2 * a.heads + h) * R) * T + idx];
    // score offset: + mask + bias - logsumexp (+inf marks a row whose every key carries the hard finfo.min mask: its
    // logsumexp IS finfo.min -- log T is absorbed -- and its probabilities are uniform, as in the general kernel)
    const float lse_r = P.lse[((int64_t)g * a.heads + h) * R + r];
    DS[r * TP + t] = lse_r <= -1e30f ? INFINITY : add - lse_r;
  }
  for (int r = w; r < R; r += 4) {                       // delta[r] straight from global memory (dO, O rows)
    const bf16_t* orow = O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    const bf16_t* drow = dO + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    float part = 0.f;
    for (int c = lane; c < d; c += 64) part += (float)drow[c] * (float)orow[c];
    part = wave_sum(part);
    if (lane == 0) dl[r] = part;
  }
  __syncthreads();
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  auto element = [&](int r, int t, float s, float dpd) {  // probability and score gradient of (row r, key t) from q.k and dO.v
    const float off = DS[r * TP + t];                     // (staged above: + mask + bias - logsumexp)
    const float pr = off == INFINITY ? 1.0f / (float)T : __expf(s * a.scale + off);
    float mult = 1.0f;
    if (a.dropout_p > 0.f) mult = dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + t, a.dropout_p, inv_keep);
    const float dsv = pr * (dpd * mult - dl[r]);
    PD[r * TP + t] = pr * mult;
    DS[r * TP + t] = dsv;
    if (P.dbias) P.dbias[(((int64_t)g * a.heads + h) * R + r) * T + t] = dsv;
  };
  if ((d & 31) == 0) {                                   // Q K^T and dO V^T on the matrix cores (see frag_xyT)
    const int nfc = (T + 15) >> 4, nf = ((R + 15) >> 4) * nfc;
    for (int f = w; f < nf; f += 4) {
      const int r0 = (f / nfc) * 16, t0 = (f % nfc) * 16;
      const f32x4 sv = frag_xyT(Qs, Ks, dp, d, r0, t0, lane), dv = frag_xyT(dOs, Vs, dp, d, r0, t0, lane);
      const int t = t0 + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * (lane >> 4) + i;
        if (r < R && t < T) element(r, t, sv[i], dv[i]);
      }
    }
  } else {
    for (int idx = tid; idx < R * T; idx += 256) {
      const int r = idx / T, t = idx - r * T;
      element(r, t, dot_bf16_lds(Qs + r * dp, Ks + t * dp, d), dot_bf16_lds(dOs + r * dp, Vs + t * dp, d));
    }
  }
  __syncthreads();
  bf16_t* dQ = reinterpret_cast<bf16_t*>(P.dq);
  bf16_t* dK = reinterpret_cast<bf16_t*>(P.dk1);
  bf16_t* dV = reinterpret_cast<bf16_t*>(P.dv1);
  const int d8 = d >> 3;
  for (int idx = tid; idx < R * d8; idx += 256) {        // thread = (row, 8 columns)
    const int r = idx / d8, c = (idx - r * d8) * 8;
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
      const float dsv = DS[r * TP + t];
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(Ks + t * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] += dsv * (float)kk[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(q[j] * a.scale);
    *reinterpret_cast<bf16x8*>(dQ + ((int64_t)g * R + r) * HD + h * d + c) = o;
  }
  for (int idx = tid; idx < T * d8; idx += 256) {        // thread = (key, 8 columns)
    const int t = idx / d8, c = (idx - t * d8) * 8;
    float k[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
      const float dsv = DS[r * TP + t], pv = PD[r * TP + t];
      const bf16x8 qq = *reinterpret_cast<const bf16x8*>(Qs + r * dp + c), gg = *reinterpret_cast<const bf16x8*>(dOs + r * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { k[j] += dsv * (float)qq[j]; v[j] += pv * (float)gg[j]; }
    }
    bf16x8 ok, ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ok[j] = (bf16_t)(k[j] * a.scale); ov[j] = (bf16_t)v[j]; }
    *reinterpret_cast<bf16x8*>(dK + ((int64_t)g * T + t) * HD + h * d + c) = ok;
    *reinterpret_cast<bf16x8*>(dV + ((int64_t)g * T + t) * HD + h * d + c) = ov;
  }
}

// The same backward for geometries whose two [R][T] f32 arrays no longer fit beside the operands (FCMF-large's ROI box
// attention: 100 x 100, heads of 128 -> 190 KB): the query rows go through the score / dQ phases in blocks of RBK rows, and
// each thread keeps its (key, 8 columns) dK / dV sums in registers across the blocks (NI = 8 items of 16 floats: T * d / 8
// <= 2048 items over 256 threads).  Rows in ascending order inside and across blocks: the sums are the ones the
// one-block kernel forms.
__global__ __launch_bounds__(256) void attn_tiny_bwd_blocked_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NI = 8;
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T = a.T1, R = a.R, dp = d + 8, TP = T + 1, RBK = P.RB;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);
  bf16_t* dOs = Qs + R * dp;
  bf16_t* Ks = dOs + R * dp;
  bf16_t* Vs = Ks + T * dp;
  float* PD = reinterpret_cast<float*>(Vs + T * dp);     // [RBK][TP] dropped probabilities of the current row block
  float* DS = PD + RBK * TP;                             // [RBK][TP] score gradients
  float* dl = DS + RBK * TP;                             // [R] delta = <dO, O>
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads, g2 = g / a.group_div;
  const int64_t HD = (int64_t)a.heads * d;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(P.dout);
  const bf16_t* O = reinterpret_cast<const bf16_t*>(P.o_in);
  stage_rows<bf16_t, bf16_t>(Qs, dp, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)g * a.q_sg + h * d, a.q_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(dOs, dp, dO + (int64_t)g * a.o_sg + h * d, a.o_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Ks, dp, reinterpret_cast<const bf16_t*>(a.k1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  stage_rows<bf16_t, bf16_t>(Vs, dp, reinterpret_cast<const bf16_t*>(a.v1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T, d, w, lane);
  for (int r = w; r < R; r += 4) {
    const bf16_t* orow = O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    const bf16_t* drow = dO + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    float part = 0.f;
    for (int c = lane; c < d; c += 64) part += (float)drow[c] * (float)orow[c];
    part = wave_sum(part);
    if (lane == 0) dl[r] = part;
  }
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  bf16_t* dQ = reinterpret_cast<bf16_t*>(P.dq);
  bf16_t* dK = reinterpret_cast<bf16_t*>(P.dk1);
  bf16_t* dV = reinterpret_cast<bf16_t*>(P.dv1);
  const int d8 = d >> 3, nkv = T * d8;
  float kacc[NI][8], vacc[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { kacc[i][j] = 0.f; vacc[i][j] = 0.f; }
  for (int r0 = 0; r0 < R; r0 += RBK) {
    const int rb = R - r0 < RBK ? R - r0 : RBK;
    __syncthreads();                                     // operands + delta staged / the previous block's arrays read
    for (int idx = tid; idx < rb * T; idx += 256) {      // additive terms of the block, all loads in flight together
      const int rl = idx / T, t = idx - rl * T, r = r0 + rl;
      float add = 0.f;
      if (a.mask) add = a.mask[(int64_t)g * T + t];
      if (a.bias) add += a.bias[(((int64_t)g2 * a.heads + h) * R + r) * T + t];
      const float lse_r = P.lse[((int64_t)g * a.heads + h) * R + r];
      DS[rl * TP + t] = lse_r <= -1e30f ? INFINITY : add - lse_r;      // (+inf: fully masked row, see the one-block kernel)
    }
    auto element = [&](int rl, int t, float s, float dpd) {
      const int r = r0 + rl;
      const float off = DS[rl * TP + t];
      const float pr = off == INFINITY ? 1.0f / (float)T : __expf(s * a.scale + off);
      float mult = 1.0f;
      if (a.dropout_p > 0.f) mult = dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + t, a.dropout_p, inv_keep);
      const float dsv = pr * (dpd * mult - dl[r]);
      PD[rl * TP + t] = pr * mult;
      DS[rl * TP + t] = dsv;
      if (P.dbias) P.dbias[(((int64_t)g * a.heads + h) * R + r) * T + t] = dsv;
    };
    if ((d & 31) == 0) {                                 // Q K^T and dO V^T of the row block on the matrix cores (see frag_xyT)
      __syncthreads();                                   // (another thread staged the element's offset)
      const int nfc = (T + 15) >> 4, nf = ((rb + 15) >> 4) * nfc;
      for (int f = w; f < nf; f += 4) {
        const int rl0 = (f / nfc) * 16, t0 = (f % nfc) * 16;
        const f32x4 sv = frag_xyT(Qs, Ks, dp, d, r0 + rl0, t0, lane), dv = frag_xyT(dOs, Vs, dp, d, r0 + rl0, t0, lane);
        const int t = t0 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rl = rl0 + 4 * (lane >> 4) + i;
          if (rl < rb && t < T) element(rl, t, sv[i], dv[i]);
        }
      }
    } else {
      for (int idx = tid; idx < rb * T; idx += 256) {    // (same thread per element as the staging loop: no barrier)
        const int rl = idx / T, t = idx - rl * T;
        element(rl, t, dot_bf16_lds(Qs + (r0 + rl) * dp, Ks + t * dp, d), dot_bf16_lds(dOs + (r0 + rl) * dp, Vs + t * dp, d));
      }
    }
    __syncthreads();
    for (int idx = tid; idx < rb * d8; idx += 256) {     // dQ rows of the block: thread = (row, 8 columns)
      const int rl = idx / d8, c = (idx - rl * d8) * 8;
      float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < T; ++t) {
        const float dsv = DS[rl * TP + t];
        const bf16x8 kk = *reinterpret_cast<const bf16x8*>(Ks + t * dp + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] += dsv * (float)kk[j];
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(q[j] * a.scale);
      *reinterpret_cast<bf16x8*>(dQ + ((int64_t)g * R + r0 + rl) * HD + h * d + c) = o;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {                       // dK / dV partial sums: thread = (key, 8 columns), in registers
      const int idx = tid + 256 * i;
      if (idx < nkv) {
        const int t = idx / d8, c = (idx - t * d8) * 8;
        for (int rl = 0; rl < rb; ++rl) {
          const float dsv = DS[rl * TP + t], pv = PD[rl * TP + t];
          const bf16x8 qq = *reinterpret_cast<const bf16x8*>(Qs + (r0 + rl) * dp + c);
          const bf16x8 gg = *reinterpret_cast<const bf16x8*>(dOs + (r0 + rl) * dp + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) { kacc[i][j] += dsv * (float)qq[j]; vacc[i][j] += pv * (float)gg[j]; }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 256 * i;
    if (idx < nkv) {
      const int t = idx / d8, c = (idx - t * d8) * 8;
      bf16x8 ok, ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ok[j] = (bf16_t)(kacc[i][j] * a.scale); ov[j] = (bf16_t)vacc[i][j]; }
      *reinterpret_cast<bf16x8*>(dK + ((int64_t)g * T + t) * HD + h * d + c) = ok;
      *reinterpret_cast<bf16x8*>(dV + ((int64_t)g * T + t) * HD + h * d + c) = ov;
    }
  }
}

// =========================================================================================================================
// "Few query rows" backward: the two fusion attentions of the step (R = 7 [CLS] rows, one per image, against <= 128 shared
// text keys and / or each row's own private keys; bf16, no bias, no causal fill).  The general kernel gives a wave a row (7
// rows on 4 waves: two rounds, the second three quarters full) and then a shared key (32 keys per wave one after the other,
// 7-long dependent sums, 128-byte stores).  Nothing here needs a reduction -- the probabilities come from the saved
// logsumexp -- so the work is flat: thread = (row, key) for probabilities / score gradients, thread = (row, 8 columns,
// quarter of the keys) for dq, thread = (shared key, 8 columns) for dk1 / dv1 with 16-byte stores.  Same dropout counters;
// the private keys of grouped sequences leave through the pd2 / ds2 scratch exactly as in the general kernel.
// =========================================================================================================================
__device__ __forceinline__ float dot_bf16_lds_gl(const bf16_t* a_lds, const bf16_t* __restrict__ b_gl, int d) {   // d % 8 == 0
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  float s0 = 0.f, s1 = 0.f;
  for (int c = 0; c < d; c += 8) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a_lds + c), y = *reinterpret_cast<const bf16x8*>(b_gl + c);
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[0], x[1]}, bf16x2_t{y[0], y[1]}, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[2], x[3]}, bf16x2_t{y[2], y[3]}, s1, false);
    s0 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[4], x[5]}, bf16x2_t{y[4], y[5]}, s0, false);
    s1 = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x[6], x[7]}, bf16x2_t{y[6], y[7]}, s1, false);
  }
  return s0 + s1;
}

__global__ __launch_bounds__(256) void attn_rows_flat_bwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T1 = a.T1, T2 = a.T2, T = T1 + T2, R = a.R, dp = d + 8, TP = T + 1, d8 = d >> 3;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);            // [R][dp]
  bf16_t* dOs = Qs + R * dp;                             // [R][dp]
  bf16_t* Ks = dOs + R * dp;                             // [T1][dp]
  bf16_t* Vs = Ks + T1 * dp;                             // [T1][dp]
  float* PD = reinterpret_cast<float*>(Vs + T1 * dp);    // [R][TP] dropped probabilities
  float* DS = PD + R * TP;                               // [R][TP] score gradients
  float* dl = DS + R * TP;                               // [R] delta = <dO, O>
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads, g2 = g / a.group_div;
  const int64_t HD = (int64_t)a.heads * d;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(P.dout);
  const bf16_t* O = reinterpret_cast<const bf16_t*>(P.o_in);
  const bf16_t* K2 = reinterpret_cast<const bf16_t*>(a.k2);
  const bf16_t* V2 = reinterpret_cast<const bf16_t*>(a.v2);
  stage_rows<bf16_t, bf16_t>(Qs, dp, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)g * a.q_sg + h * d, a.q_sr, R, d, w, lane);
  stage_rows<bf16_t, bf16_t>(dOs, dp, dO + (int64_t)g * a.o_sg + h * d, a.o_sr, R, d, w, lane);
  if (T1 > 0) {
    stage_rows<bf16_t, bf16_t>(Ks, dp, reinterpret_cast<const bf16_t*>(a.k1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T1, d, w, lane);
    stage_rows<bf16_t, bf16_t>(Vs, dp, reinterpret_cast<const bf16_t*>(a.v1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T1, d, w, lane);
  }
  for (int idx = tid; idx < R * T; idx += 256) {         // score offsets: + mask - logsumexp (+inf: fully hard-masked row)
    const int r = idx / T, t = idx - r * T;
    const float add = a.mask ? a.mask[(int64_t)g * T + t] : 0.f;
    const float lse_r = P.lse[((int64_t)g * a.heads + h) * R + r];
    DS[r * TP + t] = lse_r <= -1e30f ? INFINITY : add - lse_r;
  }
  for (int r = w; r < R; r += 4) {
    const bf16_t* orow = O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    const bf16_t* drow = dO + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    float part = 0.f;
    for (int c = lane; c < d; c += 64) part += (float)drow[c] * (float)orow[c];
    part = wave_sum(part);
    if (lane == 0) dl[r] = part;
  }
  __syncthreads();
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  for (int idx = tid; idx < R * T; idx += 256) {         // thread = (row, key)
    const int r = idx / T, t = idx - r * T;
    float s, dpd;
    if (t < T1) {
      s = dot_bf16_lds(Qs + r * dp, Ks + t * dp, d);
      dpd = dot_bf16_lds(dOs + r * dp, Vs + t * dp, d);
    } else {                                             // private key: its K2 / V2 rows straight from global memory
      const int64_t o2 = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)(t - T1) * a.k2_st + h * d;
      s = dot_bf16_lds_gl(Qs + r * dp, K2 + o2, d);
      dpd = dot_bf16_lds_gl(dOs + r * dp, V2 + o2, d);
    }
    const float off = DS[r * TP + t];                    // (staged by this same thread)
    const float pr = off == INFINITY ? 1.0f / (float)T : __expf(s * a.scale + off);
    float mult = 1.0f;
    if (a.dropout_p > 0.f) mult = dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + t, a.dropout_p, inv_keep);
    const float dsv = pr * (dpd * mult - dl[r]);
    PD[r * TP + t] = pr * mult;
    DS[r * TP + t] = dsv;
    if (t >= T1 && P.pd2) {
      const int64_t i2 = (((int64_t)g * a.heads + h) * R + r) * T2 + (t - T1);
      P.pd2[i2] = pr * mult;
      P.ds2[i2] = dsv;
    }
  }
  __syncthreads();
  // dq[r][c] = scale * sum_t ds[r][t] k[t][c]: thread = (row, 8 columns, quarter of the keys), the four quarters in
  // neighbouring lanes and summed across them
  bf16_t* dQ = reinterpret_cast<bf16_t*>(P.dq);
  for (int item = tid >> 2; item < R * d8; item += 64) {
    const int r = item / d8, c = (item - r * d8) * 8, qt = tid & 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = qt; t < T1; t += 4) {
      const float dsv = DS[r * TP + t];
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(Ks + t * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += dsv * (float)kk[j];
    }
    const int64_t p2 = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + h * d + c;
    for (int t2 = qt; t2 < T2; t2 += 4) {
      const float dsv = DS[r * TP + T1 + t2];
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(K2 + p2 + (int64_t)t2 * a.k2_st);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += dsv * (float)kk[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[j];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      o[j] = (bf16_t)(v * a.scale);
    }
    if (qt == 0) *reinterpret_cast<bf16x8*>(dQ + ((int64_t)g * R + r) * HD + h * d + c) = o;
  }
  // dk1 / dv1 [t][c] = sum_r ds / pd [r][t] * q / dO [r][c]: thread = (shared key, 8 columns)
  bf16_t* dK = reinterpret_cast<bf16_t*>(P.dk1);
  bf16_t* dV = reinterpret_cast<bf16_t*>(P.dv1);
  for (int idx = tid; idx < T1 * d8; idx += 256) {
    const int t = idx / d8, c = (idx - t * d8) * 8;
    float k[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
      const float dsv = DS[r * TP + t], pv = PD[r * TP + t];
      const bf16x8 qq = *reinterpret_cast<const bf16x8*>(Qs + r * dp + c), gg = *reinterpret_cast<const bf16x8*>(dOs + r * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { k[j] += dsv * (float)qq[j]; v[j] += pv * (float)gg[j]; }
    }
    bf16x8 ok, ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ok[j] = (bf16_t)(k[j] * a.scale); ov[j] = (bf16_t)v[j]; }
    *reinterpret_cast<bf16x8*>(dK + ((int64_t)g * T1 + t) * HD + h * d + c) = ok;
    *reinterpret_cast<bf16x8*>(dV + ((int64_t)g * T1 + t) * HD + h * d + c) = ov;
  }
}

// the forward of the same geometry, flat as well: thread = (row, key) scores, wave = row softmax (a few elements per lane),
// thread = (row, 8 columns, quarter of the keys) outputs
__global__ __launch_bounds__(256) void attn_rows_flat_fwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T1 = a.T1, T2 = a.T2, T = T1 + T2, R = a.R, dp = d + 8, TP = T + 1, d8 = d >> 3;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);            // [R][dp]
  bf16_t* Ks = Qs + R * dp;                              // [T1][dp]
  bf16_t* Vs = Ks + T1 * dp;                             // [T1][dp]
  float* Ss = reinterpret_cast<float*>(Vs + T1 * dp);    // [R][TP] scores, then dropped probabilities
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads, g2 = g / a.group_div;
  const bf16_t* K2 = reinterpret_cast<const bf16_t*>(a.k2);
  const bf16_t* V2 = reinterpret_cast<const bf16_t*>(a.v2);
  stage_rows<bf16_t, bf16_t>(Qs, dp, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)g * a.q_sg + h * d, a.q_sr, R, d, w, lane);
  if (T1 > 0) {
    stage_rows<bf16_t, bf16_t>(Ks, dp, reinterpret_cast<const bf16_t*>(a.k1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T1, d, w, lane);
    stage_rows<bf16_t, bf16_t>(Vs, dp, reinterpret_cast<const bf16_t*>(a.v1) + (int64_t)g * a.k1_sg + h * d, a.k1_st, T1, d, w, lane);
  }
  __syncthreads();
  for (int idx = tid; idx < R * T; idx += 256) {         // thread = (row, key)
    const int r = idx / T, t = idx - r * T;
    float s;
    if (t < T1) s = dot_bf16_lds(Qs + r * dp, Ks + t * dp, d);
    else s = dot_bf16_lds_gl(Qs + r * dp, K2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)(t - T1) * a.k2_st + h * d, d);
    s *= a.scale;
    if (a.mask) s += a.mask[(int64_t)g * T + t];
    Ss[r * TP + t] = s;
  }
  __syncthreads();
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  for (int r = w; r < R; r += 4) {                       // wave = row: T <= 384 keys, up to 6 per lane
    float sv[6], m = -INFINITY;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int t = lane + 64 * n;
      sv[n] = t < T ? Ss[r * TP + t] : -INFINITY;
      m = fmaxf(m, sv[n]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int t = lane + 64 * n;
      sv[n] = t < T ? __expf(sv[n] - m) : 0.f;
      sum += sv[n];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int t = lane + 64 * n;
      if (t < T) {
        float pv = sv[n] * inv;
        if (a.dropout_p > 0.f) pv *= dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * R + r) * T + t, a.dropout_p, inv_keep);
        Ss[r * TP + t] = pv;
      }
    }
    if (lane == 0 && P.lse) P.lse[((int64_t)g * a.heads + h) * R + r] = m + __logf(sum);
  }
  __syncthreads();
  bf16_t* O = reinterpret_cast<bf16_t*>(P.out);
  for (int item = tid >> 2; item < R * d8; item += 64) {  // thread = (row, 8 columns, quarter of the keys)
    const int r = item / d8, c = (item - r * d8) * 8, qt = tid & 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = qt; t < T1; t += 4) {
      const float pv = Ss[r * TP + t];
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(Vs + t * dp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pv * (float)vv[j];
    }
    const int64_t p2 = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + h * d + c;
    for (int t2 = qt; t2 < T2; t2 += 4) {
      const float pv = Ss[r * TP + T1 + t2];
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(V2 + p2 + (int64_t)t2 * a.k2_st);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pv * (float)vv[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[j];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      o[j] = (bf16_t)v;
    }
    if (qt == 0) *reinterpret_cast<bf16x8*>(O + (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d + c) = o;
  }
}

// the flat few-rows backward's preconditions (host)
static bool rows_flat_ok(const fcmf_attn_desc* a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!(a->dtype == FCMF_BF16 && !a->causal && !a->head_quirk && !a->bias && a->R <= 16 && a->T1 <= 128 && a->d % 8 == 0 &&
        a->q_sr % 8 == 0 && a->q_sg % 8 == 0 && a->o_sr % 8 == 0 && a->o_sg % 8 == 0 && al16(a->q)))
    return false;
  if (a->T1 > 0 && !(a->k1_st % 8 == 0 && a->k1_sg % 8 == 0 && al16(a->k1) && al16(a->v1))) return false;
  if (a->T2 > 0 && !(a->k2_sg % 8 == 0 && a->k2_sr % 8 == 0 && a->k2_st % 8 == 0 && al16(a->k2) && al16(a->v2))) return false;
  return true;
}

// the tiny kernels' preconditions (host)
static bool tiny_ok(const fcmf_attn_desc* a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return a->dtype == FCMF_BF16 && a->T2 == 0 && a->T1 > 0 && a->T1 <= TINY_WIDE_MAX && a->R <= TINY_WIDE_MAX && !a->causal && !a->head_quirk &&
         a->d % 8 == 0 && a->q_sr % 8 == 0 && a->q_sg % 8 == 0 && a->k1_st % 8 == 0 && a->k1_sg % 8 == 0 && a->o_sr % 8 == 0 &&
         a->o_sg % 8 == 0 && al16(a->q) && al16(a->k1) && al16(a->v1);      // (+ 16-byte aligned outputs: checked by the callers)
}

static int check_desc(const fcmf_attn_desc* a) {
  if (!a || !a->q) return FCMF_ERR_ARG;
  if (a->dtype != FCMF_F32 && a->dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  if (a->G <= 0 || a->heads <= 0 || a->R <= 0 || a->d <= 0 || a->d > AS_MAXD) return FCMF_ERR_ARG;
  if (a->T1 < 0 || a->T2 < 0 || a->T1 + a->T2 <= 0 || a->T1 + a->T2 > AS_MAXT || a->T2 > 128) return FCMF_ERR_ARG;
  if (a->T1 > 0 && (!a->k1 || !a->v1)) return FCMF_ERR_ARG;
  if (a->T2 > 0 && (!a->k2 || !a->v2)) return FCMF_ERR_ARG;
  if (a->group_div <= 0) return FCMF_ERR_ARG;
  return FCMF_OK;
}

// query rows per staged block: all R if the workgroup then stays within 80 KiB of LDS (two or more workgroups
// per CU); else blocks of >= 8 rows within 80 KiB if they fit; else as many rows as fit in 160 KiB
static int rows_per_block(int R, size_t base_bytes, size_t per_row_bytes) {
  const size_t soft = 80 * 1024, hard = 160 * 1024;
  if (base_bytes + (size_t)R * per_row_bytes <= soft) return R;
  if (base_bytes + 8 * per_row_bytes <= soft) return (int)((soft - base_bytes) / per_row_bytes);
  if (base_bytes + per_row_bytes > hard) return 0;
  const size_t rb = (hard - base_bytes) / per_row_bytes;
  return (int)(rb > (size_t)R ? R : rb);
}

extern "C" int fcmf_attn_small_fwd(const fcmf_attn_desc* desc, void* out, float* lse, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!out) return FCMF_ERR_ARG;
  AttnK P{};
  P.a = *desc; P.out = out; P.lse = lse;
  if (tiny_ok(desc) && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int dp = desc->d + 8;
    // (the matrix-core score fragments read up to 15 rows past the end of the K image: into V and the score array, or the pad)
    size_t smem = (size_t)(desc->R + 2 * desc->T1) * dp * 2 + sizeof(float) * (size_t)desc->R * (desc->T1 + 1);
    const size_t after_k = (size_t)desc->T1 * dp * 2 + sizeof(float) * (size_t)desc->R * (desc->T1 + 1), over = (size_t)15 * dp * 2;
    if (after_k < over) smem += over - after_k;
    if (smem <= TINY_LDS) {
      auto k = desc->T1 > 64 ? attn_tiny_fwd_kernel<true> : attn_tiny_fwd_kernel<false>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(k, dim3(desc->G * desc->heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
  }
  if (rows_flat_ok(desc) && desc->T1 + desc->T2 <= 384 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int dp = desc->d + 8, TP = desc->T1 + desc->T2 + 1;
    const size_t smem = (size_t)(desc->R + 2 * desc->T1) * dp * 2 + sizeof(float) * (size_t)desc->R * TP;
    if (smem <= 64 * 1024) {
      hipLaunchKernelGGL(attn_rows_flat_fwd_kernel, dim3(desc->G * desc->heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
  }
  const size_t esz = desc->dtype == FCMF_F32 ? 4 : 2;
  P.KVF = (int)((2 * (size_t)desc->T1 * (desc->d + 16 / esz) * esz + 15) / 16 * 4);   // K and V images, rounded to 16 B, in floats
  const int kpl = desc->T1 + desc->T2 <= 256 ? 4 : 8;     // keys per lane of the forward
  const int nw = 4;
  const size_t base = sizeof(float) * ((size_t)P.KVF + (size_t)nw * 64 * kpl);
  const size_t per_row = sizeof(float) * (size_t)desc->d;
  P.RB = rows_per_block(desc->R, base, per_row);
  if (P.RB <= 0) return FCMF_ERR_UNSUPPORTED;
  const size_t smem = base + (size_t)P.RB * per_row;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(desc->G * desc->heads);
#define FCMF_ATTN_FWD(TT, KPL, DC)                                                                                         \
  do {                                                                                                                     \
    auto k = attn_small_fwd_kernel<TT, KPL, DC>;                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);    \
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);                                                                   \
  } while (0)
#define FCMF_ATTN_FWD_T(TT)                                                                                                \
  do {                                                                                                                     \
    if (desc->d == 64) { if (kpl == 4) FCMF_ATTN_FWD(TT, 4, 64); else FCMF_ATTN_FWD(TT, 8, 64); }                          \
    else               { if (kpl == 4) FCMF_ATTN_FWD(TT, 4, 0); else FCMF_ATTN_FWD(TT, 8, 0); }                            \
  } while (0)
  if (desc->dtype == FCMF_F32) FCMF_ATTN_FWD_T(float); else FCMF_ATTN_FWD_T(bf16_t);
#undef FCMF_ATTN_FWD_T
#undef FCMF_ATTN_FWD
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

static int attn_small_bwd_impl(const fcmf_attn_desc* desc, const void* out, const void* dout, const float* lse,
                               void* dq, void* dk1, void* dv1, void* dk2, void* dv2, float* dbias,
                               float* scratch, int64_t scratch_bytes, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!out || !dout || !lse || !dq) return FCMF_ERR_ARG;
  if (desc->T1 > 0 && !dk1) return FCMF_ERR_ARG;
  if (desc->T1 > 0 && !dv1 && desc->v1 != desc->k1) return FCMF_ERR_ARG;
  if (desc->T2 > 0 && (!dk2 || !dv2)) return FCMF_ERR_ARG;
  AttnK P{};
  P.a = *desc; P.o_in = out; P.dout = dout; P.lse = const_cast<float*>(lse);
  P.dq = dq; P.dk1 = dk1; P.dv1 = dv1; P.dk2 = dk2; P.dv2 = dv2; P.dbias = dbias;
  const bool grouped = scratch != nullptr;
  const int64_t n2 = (int64_t)desc->G * desc->heads * desc->R * desc->T2;
  const size_t smem2 = sizeof(float) * 2 * (size_t)desc->group_div * desc->heads * desc->T2;
  if (grouped) {
    if (desc->T2 <= 0 || desc->group_div > 8 || desc->G % desc->group_div || desc->head_quirk || smem2 > 150 * 1024)
      return FCMF_ERR_UNSUPPORTED;
    if (scratch_bytes < 2 * n2 * (int64_t)sizeof(float)) return FCMF_ERR_ARG;
    P.pd2 = scratch; P.ds2 = scratch + n2;
  }
  if (!grouped && dv1 && tiny_ok(desc) &&
      ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) |
        reinterpret_cast<uintptr_t>(dk1) | reinterpret_cast<uintptr_t>(dv1)) & 15) == 0) {
    const int dp = desc->d + 8;
    size_t opnd = (size_t)(2 * desc->R + 2 * desc->T1) * dp * 2 + sizeof(float) * (size_t)desc->R;
    {   // the score fragments read up to 15 rows past the end of the V image: into the f32 arrays behind it, or a pad
      const size_t tail = sizeof(float) * ((size_t)2 * (desc->R < 16 ? desc->R : 16) * (desc->T1 + 1) + desc->R), over = (size_t)15 * dp * 2;
      if (tail < over) opnd += over - tail;
    }
    const size_t row = sizeof(float) * 2 * (size_t)(desc->T1 + 1);      // one row of the two [rows][T + 1] f32 arrays
    if (desc->R <= TINY_MAX && desc->T1 <= TINY_MAX && opnd + desc->R * row <= TINY_LDS) {
      const size_t smem = opnd + desc->R * row;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_tiny_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(attn_tiny_bwd_kernel, dim3(desc->G * desc->heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
    if (opnd + 16 * row <= TINY_LDS) {                   // row blocks of >= 16 rows, as even as the LDS allows
      const int fit = (int)((TINY_LDS - opnd) / row);
      const int nblk = (desc->R + fit - 1) / fit;
      P.RB = (desc->R + nblk - 1) / nblk;
      const size_t smem = opnd + P.RB * row;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_tiny_bwd_blocked_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(attn_tiny_bwd_blocked_kernel, dim3(desc->G * desc->heads), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), P);
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
  }
  if (rows_flat_ok(desc) && !dbias && (desc->T2 == 0 || grouped) && (desc->T1 == 0 || dv1) &&
      ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) |
        reinterpret_cast<uintptr_t>(dk1) | reinterpret_cast<uintptr_t>(dv1)) & 15) == 0) {
    const int dp = desc->d + 8, TP = desc->T1 + desc->T2 + 1;
    const size_t smem = (size_t)(2 * desc->R + 2 * desc->T1) * dp * 2 + sizeof(float) * ((size_t)2 * desc->R * TP + desc->R);
    if (smem <= 64 * 1024) {
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
      hipLaunchKernelGGL(attn_rows_flat_bwd_kernel, dim3(desc->G * desc->heads), dim3(256), smem, st, P);
      FCMF_CHECK_LAUNCH();
      if (grouped) {
        dim3 grid2((desc->G / desc->group_div) * desc->R, 4);
        auto k = attn_private_grad_kernel<bf16_t>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        hipLaunchKernelGGL(k, grid2, dim3(256), smem2, st, P);
        FCMF_CHECK_LAUNCH();
      }
      return FCMF_OK;
    }
  }
  const int nsh = desc->T1 < 128 ? desc->T1 : 128;
  const size_t esz = desc->dtype == FCMF_F32 ? 4 : 2;
  P.KVF = (int)((2 * (size_t)nsh * (desc->d + 16 / esz) * esz + 15) / 16 * 4);
  const size_t base = sizeof(float) * (size_t)P.KVF;
  P.TLP = (nsh + desc->T2 + 3) & ~3;   // multiple of 4: rows of the key arrays stay 16-byte aligned
  const size_t per_row = sizeof(float) * (2 * (size_t)desc->d + 2 * (size_t)P.TLP);
  P.RB = rows_per_block(desc->R, base, per_row);
  if (P.RB <= 0) return FCMF_ERR_UNSUPPORTED;
  const size_t smem = base + (size_t)P.RB * per_row;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nchunks = desc->T1 > 0 ? (desc->T1 + 127) / 128 : 1;
  dim3 grid(desc->G * desc->heads, nchunks);
  const bool one = P.RB >= desc->R;
#define LAUNCH_(T, ONE, GR)                                                                                                \
  do {                                                                                                                     \
    auto k = attn_small_bwd_kernel<T, ONE, GR>;                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);    \
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);                                                                   \
  } while (0)
#define LAUNCH_T(T)                                                                                                        \
  do {                                                                                                                     \
    if (grouped) { if (one) LAUNCH_(T, true, true); else LAUNCH_(T, false, true); }                                        \
    else         { if (one) LAUNCH_(T, true, false); else LAUNCH_(T, false, false); }                                      \
  } while (0)
  if (desc->dtype == FCMF_F32) LAUNCH_T(float); else LAUNCH_T(bf16_t);
#undef LAUNCH_T
#undef LAUNCH_
  FCMF_CHECK_LAUNCH();
  if (grouped) {
    dim3 grid2((desc->G / desc->group_div) * desc->R, 4);    // 4 key ranges per (review, row): ~1800 workgroups on the step
    if (desc->dtype == FCMF_F32) {
      auto k = attn_private_grad_kernel<float>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      hipLaunchKernelGGL(k, grid2, dim3(256), smem2, st, P);
    } else {
      auto k = attn_private_grad_kernel<bf16_t>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      hipLaunchKernelGGL(k, grid2, dim3(256), smem2, st, P);
    }
    FCMF_CHECK_LAUNCH();
  }
  return FCMF_OK;
}

extern "C" int fcmf_attn_small_bwd(const fcmf_attn_desc* desc, const void* out, const void* dout, const float* lse,
                                   void* dq, void* dk1, void* dv1, void* dk2, void* dv2, float* dbias,
                                   void* stream) {
  return attn_small_bwd_impl(desc, out, dout, lse, dq, dk1, dv1, dk2, dv2, dbias, nullptr, 0, stream);
}

extern "C" int fcmf_attn_small_bwd_grouped(const fcmf_attn_desc* desc, const void* out, const void* dout, const float* lse,
                                           void* dq, void* dk1, void* dv1, void* dk2, void* dv2, float* dbias,
                                           float* scratch, int64_t scratch_bytes, void* stream) {
  if (!scratch) return FCMF_ERR_ARG;
  return attn_small_bwd_impl(desc, out, dout, lse, dq, dk1, dv1, dk2, dv2, dbias, scratch, scratch_bytes, stream);
}
