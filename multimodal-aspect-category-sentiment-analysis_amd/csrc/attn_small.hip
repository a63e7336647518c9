// Two-segment multi-head attention, forward and backward, VALU f32 math, any activation dtype.
// One workgroup = one (group g, head h).  The T1 shared keys/values are staged once in LDS and
// reused by all R query rows of the group; the T2 private keys/values of each row stream from
// global memory.  Used for every attention on the path in fp32 (parity) mode and, in bf16 mode,
// for the dead-row-pruned fusion layers (1 live query row per image), the 15-token fusion layer,
// the geometry-biased ROI attention and the IAOG decoder; the bf16 text-encoder attention runs
// on attn_mfma.hip instead.
#include "common.h"

constexpr int AS_MAXT = 256;   // T1 + T2 <= 256 (one key per thread in the backward)
constexpr int AS_MAXD = 128;   // head dim <= 128 (two elements per lane)
constexpr int AS_NT1 = 32;     // T1 <= 128: per-thread accumulators cover keys w + 4n, n < 32

struct AttnK {
  fcmf_attn_desc a;
  void* out; float* lse;
  const void* dout; const void* o_in;
  void *dq, *dk1, *dv1, *dk2, *dv2; float* dbias;
};

template <typename TT>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T1 = a.T1, T2 = a.T2, T = T1 + T2, dp = d + 1;
  float* K1s = sm;
  float* V1s = K1s + T1 * dp;
  float* qs = V1s + T1 * dp;            // [4][AS_MAXD]
  float* ps = qs + 4 * AS_MAXD;         // [4][AS_MAXT]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int hin = a.head_quirk ? (int)(((int64_t)h * a.G + g) % a.heads) : h;
  const int g2 = g / a.group_div;
  const TT* Q = reinterpret_cast<const TT*>(a.q);
  const TT* K1 = reinterpret_cast<const TT*>(a.k1);
  const TT* V1 = reinterpret_cast<const TT*>(a.v1);
  const TT* K2 = reinterpret_cast<const TT*>(a.k2);
  const TT* V2 = reinterpret_cast<const TT*>(a.v2);
  TT* O = reinterpret_cast<TT*>(P.out);

  for (int e = tid; e < T1 * d; e += 256) {
    int t = e / d, c = e - t * d;
    int64_t off = (int64_t)g * a.k1_sg + (int64_t)t * a.k1_st + hin * d + c;
    K1s[t * dp + c] = to_f32<TT>(K1[off]);
    V1s[t * dp + c] = to_f32<TT>(V1[off]);
  }
  __syncthreads();

  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const int iters = (a.R + 3) / 4;
  for (int it = 0; it < iters; ++it) {
    const int r = it * 4 + w;
    const bool act = r < a.R;
    float* q = qs + w * AS_MAXD;
    float* p = ps + w * AS_MAXT;
    if (act)
      for (int c = lane; c < d; c += 64) q[c] = to_f32<TT>(Q[(int64_t)g * a.q_sg + (int64_t)r * a.q_sr + hin * d + c]);
    __syncthreads();
    float sc[4];
    float m = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int t = lane + 64 * n;
      float s = -INFINITY;
      if (act && t < T) {
        float accv = 0.f;
        if (t < T1) {
          const float* kr = K1s + t * dp;
          for (int c = 0; c < d; ++c) accv += q[c] * kr[c];
        } else {
          const TT* kr = K2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)(t - T1) * a.k2_st + hin * d;
          for (int c = 0; c < d; ++c) accv += q[c] * to_f32<TT>(kr[c]);
        }
        s = accv * a.scale;
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
    for (int n = 0; n < 4; ++n) {
      const int t = lane + 64 * n;
      float e = (act && t < T) ? __expf(sc[n] - m) : 0.f;
      sc[n] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = act ? 1.0f / sum : 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int t = lane + 64 * n;
      if (act && t < T) {
        float pv = sc[n] * inv;
        if (a.dropout_p > 0.f)
          pv *= dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * a.R + r) * T + t, a.dropout_p, inv_keep);
        p[t] = pv;
      }
    }
    if (act && lane == 0 && P.lse) P.lse[((int64_t)g * a.heads + h) * a.R + r] = m + __logf(sum);
    __syncthreads();
    if (act) {
      for (int c = lane; c < d; c += 64) {
        float o = 0.f;
        for (int t = 0; t < T1; ++t) o += p[t] * V1s[t * dp + c];
        const TT* vb = V2 + (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + hin * d + c;
        for (int t = 0; t < T2; ++t) o += p[T1 + t] * to_f32<TT>(vb[(int64_t)t * a.k2_st]);
        O[(int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d + c] = from_f32<TT>(o);
      }
    }
    __syncthreads();
  }
}

// Backward.  grid = (G*heads, ceil(T1/128)): block (.., c) owns the shared keys [128c, 128c+128)
// (their dK1/dV1 live in registers across the R query rows, so no atomics are needed) and, for
// c == 0, the private keys.  Each chunk writes its partial dq to dq[c] (dense [chunks,G,R,heads*d]);
// the host sums the chunks.  dk1/dv1 are dense [G,T1,heads*d].
template <typename TT>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(AttnK P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const fcmf_attn_desc& a = P.a;
  const int d = a.d, T1 = a.T1, T2 = a.T2, T = T1 + T2, dp = d + 1;
  const int chunk = blockIdx.y, kc0 = chunk * 128;
  const int T1c = min(128, T1 - kc0);                 // shared keys of this chunk (may be <= 0 if T1 == 0)
  const int T2c = chunk == 0 ? T2 : 0;                // private keys are handled by chunk 0
  const int nsh = T1c > 0 ? T1c : 0;
  float* K1s = sm;
  float* V1s = K1s + nsh * dp;
  float* qs = V1s + nsh * dp;       // [AS_MAXD]
  float* dos = qs + AS_MAXD;        // [AS_MAXD]
  float* pd = dos + AS_MAXD;        // [256] dropped probabilities: [0,128) shared chunk, [128,256) private
  float* ds = pd + 256;             // [256] score gradients
  float* red = ds + 256;            // [4][AS_MAXD] dq partials
  float* misc = red + 4 * AS_MAXD;  // [4] wave partials of delta
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int hin = a.head_quirk ? (int)(((int64_t)h * a.G + g) % a.heads) : h;
  const int g2 = g / a.group_div;
  const int64_t HD = (int64_t)a.heads * d;
  const TT* Q = reinterpret_cast<const TT*>(a.q);
  const TT* K1 = reinterpret_cast<const TT*>(a.k1);
  const TT* V1 = reinterpret_cast<const TT*>(a.v1);
  const TT* K2 = reinterpret_cast<const TT*>(a.k2);
  const TT* V2 = reinterpret_cast<const TT*>(a.v2);
  const TT* O = reinterpret_cast<const TT*>(P.o_in);
  const TT* dO = reinterpret_cast<const TT*>(P.dout);
  TT* dQ = reinterpret_cast<TT*>(P.dq) + (int64_t)chunk * a.G * a.R * HD;
  TT* dK1 = reinterpret_cast<TT*>(P.dk1);
  TT* dV1 = reinterpret_cast<TT*>(P.dv1);
  TT* dK2 = reinterpret_cast<TT*>(P.dk2);
  TT* dV2 = reinterpret_cast<TT*>(P.dv2);

  for (int e = tid; e < nsh * d; e += 256) {
    int t = e / d, c = e - t * d;
    int64_t off = (int64_t)g * a.k1_sg + (int64_t)(kc0 + t) * a.k1_st + hin * d + c;
    K1s[t * dp + c] = to_f32<TT>(K1[off]);
    V1s[t * dp + c] = to_f32<TT>(V1[off]);
  }
  float accK[AS_NT1][2], accV[AS_NT1][2];
#pragma unroll
  for (int n = 0; n < AS_NT1; ++n) { accK[n][0] = accK[n][1] = accV[n][0] = accV[n][1] = 0.f; }
  const float inv_keep = a.dropout_p > 0.f ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  __syncthreads();

  for (int r = 0; r < a.R; ++r) {
    const int64_t orow = (int64_t)g * a.o_sg + (int64_t)r * a.o_sr + h * d;
    float part = 0.f;
    if (tid < d) {
      float qv = to_f32<TT>(Q[(int64_t)g * a.q_sg + (int64_t)r * a.q_sr + hin * d + tid]);
      float dv = to_f32<TT>(dO[orow + tid]);
      qs[tid] = qv;
      dos[tid] = dv;
      part = dv * to_f32<TT>(O[orow + tid]);
    }
    part = wave_sum(part);
    if (lane == 0) misc[w] = part;
    __syncthreads();
    const float delta = misc[0] + misc[1] + misc[2] + misc[3];
    // ---- one key per thread: threads [0,128) shared chunk keys, [128,256) private keys ------
    {
      const bool shared_key = tid < 128;
      const int tl = shared_key ? tid : tid - 128;
      const bool valid = shared_key ? (tl < nsh) : (tl < T2c);
      float pdv = 0.f, dsv = 0.f;
      if (valid) {
        const int t = shared_key ? kc0 + tl : T1 + tl;  // global key index
        float s = 0.f, dpd = 0.f;
        if (shared_key) {
          const float* kr = K1s + tl * dp;
          const float* vr = V1s + tl * dp;
          for (int c = 0; c < d; ++c) { s += qs[c] * kr[c]; dpd += dos[c] * vr[c]; }
        } else {
          const int64_t o2 = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)tl * a.k2_st + hin * d;
          for (int c = 0; c < d; ++c) { s += qs[c] * to_f32<TT>(K2[o2 + c]); dpd += dos[c] * to_f32<TT>(V2[o2 + c]); }
        }
        s *= a.scale;
        if (a.mask) s += a.mask[(int64_t)g * T + t];
        if (a.bias) s += a.bias[(((int64_t)g2 * a.heads + h) * a.R + r) * T + t];
        const bool filled = a.causal && t > r;
        if (filled) s = -1e4f;
        const float pr = __expf(s - P.lse[((int64_t)g * a.heads + h) * a.R + r]);
        float mult = 1.0f;
        if (a.dropout_p > 0.f)
          mult = dropout_mult(a.seed, (((uint64_t)g * a.heads + h) * a.R + r) * T + t, a.dropout_p, inv_keep);
        dsv = filled ? 0.f : pr * (dpd * mult - delta);
        pdv = pr * mult;
        if (P.dbias) P.dbias[(((int64_t)g * a.heads + h) * a.R + r) * T + t] = dsv;
      }
      pd[tid] = pdv;
      ds[tid] = dsv;
    }
    __syncthreads();
    // ---- shared-segment accumulators (thread owns chunk keys w+4n, dims lane, lane+64) ------
    float dq0 = 0.f, dq1 = 0.f;
    const float q0 = lane < d ? qs[lane] * a.scale : 0.f, q1 = lane + 64 < d ? qs[lane + 64] * a.scale : 0.f;
    const float o0 = lane < d ? dos[lane] : 0.f, o1 = lane + 64 < d ? dos[lane + 64] : 0.f;
#pragma unroll
    for (int n = 0; n < AS_NT1; ++n) {
      const int t = w + 4 * n;
      if (t < nsh) {
        const float dsv = ds[t], pv = pd[t];
        accK[n][0] += dsv * q0; accK[n][1] += dsv * q1;
        accV[n][0] += pv * o0;  accV[n][1] += pv * o1;
        if (lane < d) dq0 += dsv * K1s[t * dp + lane];
        if (lane + 64 < d) dq1 += dsv * K1s[t * dp + lane + 64];
      }
    }
    // ---- private segment: direct dK2/dV2 writes and the K2 part of dq ------------------------
    for (int t2 = w; t2 < T2c; t2 += 4) {
      const float dsv = ds[128 + t2], pv = pd[128 + t2];
      const int64_t src = (int64_t)g2 * a.k2_sg + (int64_t)r * a.k2_sr + (int64_t)t2 * a.k2_st + hin * d;
      const int64_t dst = (((int64_t)g * a.R + r) * T2 + t2) * HD + h * d;
      for (int c = lane; c < d; c += 64) {
        const float kv = to_f32<TT>(K2[src + c]);
        if (c < 64) dq0 += dsv * kv; else dq1 += dsv * kv;
        dK2[dst + c] = from_f32<TT>(dsv * a.scale * qs[c]);
        dV2[dst + c] = from_f32<TT>(pv * dos[c]);
      }
    }
    if (lane < d) red[w * AS_MAXD + lane] = dq0;
    if (lane + 64 < d) red[w * AS_MAXD + lane + 64] = dq1;
    __syncthreads();
    if (tid < d) {
      float v = red[tid] + red[AS_MAXD + tid] + red[2 * AS_MAXD + tid] + red[3 * AS_MAXD + tid];
      dQ[((int64_t)g * a.R + r) * HD + h * d + tid] = from_f32<TT>(v * a.scale);
    }
    __syncthreads();
  }
  const bool kv_same = (P.dv1 == nullptr);
#pragma unroll
  for (int n = 0; n < AS_NT1; ++n) {
    const int t = w + 4 * n;
    if (t < nsh) {
      const int64_t off = ((int64_t)g * T1 + kc0 + t) * HD + h * d;
      if (lane < d) {
        dK1[off + lane] = from_f32<TT>(kv_same ? accK[n][0] + accV[n][0] : accK[n][0]);
        if (!kv_same) dV1[off + lane] = from_f32<TT>(accV[n][0]);
      }
      if (lane + 64 < d) {
        dK1[off + lane + 64] = from_f32<TT>(kv_same ? accK[n][1] + accV[n][1] : accK[n][1]);
        if (!kv_same) dV1[off + lane + 64] = from_f32<TT>(accV[n][1]);
      }
    }
  }
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

extern "C" int fcmf_attn_small_fwd(const fcmf_attn_desc* desc, void* out, float* lse, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!out) return FCMF_ERR_ARG;
  AttnK P{};
  P.a = *desc; P.out = out; P.lse = lse;
  size_t smem = sizeof(float) * (2 * (size_t)desc->T1 * (desc->d + 1) + 4 * AS_MAXD + 4 * AS_MAXT);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(desc->G * desc->heads);
  if (desc->dtype == FCMF_F32) {
    auto k = attn_small_fwd_kernel<float>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);
  } else {
    auto k = attn_small_fwd_kernel<bf16_t>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);
  }
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_attn_small_bwd(const fcmf_attn_desc* desc, const void* out, const void* dout, const float* lse,
                                   void* dq, void* dk1, void* dv1, void* dk2, void* dv2, float* dbias,
                                   void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!out || !dout || !lse || !dq) return FCMF_ERR_ARG;
  if (desc->T1 > 0 && !dk1) return FCMF_ERR_ARG;
  if (desc->T1 > 0 && !dv1 && desc->v1 != desc->k1) return FCMF_ERR_ARG;
  if (desc->T2 > 0 && (!dk2 || !dv2)) return FCMF_ERR_ARG;
  AttnK P{};
  P.a = *desc; P.o_in = out; P.dout = dout; P.lse = const_cast<float*>(lse);
  P.dq = dq; P.dk1 = dk1; P.dv1 = dv1; P.dk2 = dk2; P.dv2 = dv2; P.dbias = dbias;
  const int nsh = desc->T1 < 128 ? desc->T1 : 128;
  size_t smem = sizeof(float) * (2 * (size_t)nsh * (desc->d + 1) + 2 * AS_MAXD + 2 * 256 + 4 * AS_MAXD + 4);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nchunks = desc->T1 > 0 ? (desc->T1 + 127) / 128 : 1;
  dim3 grid(desc->G * desc->heads, nchunks);
  if (desc->dtype == FCMF_F32) {
    auto k = attn_small_bwd_kernel<float>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);
  } else {
    auto k = attn_small_bwd_kernel<bf16_t>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, P);
  }
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
