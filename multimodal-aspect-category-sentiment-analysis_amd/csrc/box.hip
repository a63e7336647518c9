// Geometry bias of the ROI relation attention, fused so that the [G,N,N,64] relational embedding
// is never materialised (roi_modeling.py:79-138 BoxRelationalEmbedding, :161-163 eight Linear(64,1)
// + ReLU, :40 log(clamp(.,1e-6))).  The embedding is evaluated in the dtype of the box
// coordinates (float64 in the fine-tune loop, float32 in IAOG), then rounded to float32 before the
// WG dot products, exactly as the reference does (roi_modeling.py:150).
#include "common.h"

template <typename CT> struct Trig;
template <> struct Trig<double> {
  static __device__ __forceinline__ void sc(double x, double* s, double* c) { sincos(x, s, c); }
  static __device__ __forceinline__ double lg(double x) { return log(x); }
  static __device__ __forceinline__ double ab(double x) { return fabs(x); }
  static __device__ __forceinline__ double mx(double a, double b) { return fmax(a, b); }
};
template <> struct Trig<float> {
  static __device__ __forceinline__ void sc(float x, float* s, float* c) { sincosf(x, s, c); }
  static __device__ __forceinline__ float lg(float x) { return logf(x); }
  static __device__ __forceinline__ float ab(float x) { return fabsf(x); }
  static __device__ __forceinline__ float mx(float a, float b) { return fmaxf(a, b); }
};

// emb[0..31] = sin(100 * delta_c * dim_mat[k]), emb[32..63] = cos(...), index c*8+k (:128-135)
// FAST (float coordinates only, FCMF_BOX_FAST_TRIG): the hardware's v_sin_f32 / v_cos_f32 (input in revolutions) instead of sincosf
template <typename CT, bool FAST = false>
__device__ __forceinline__ void box_embed(const CT* __restrict__ coords, const float* __restrict__ dim_mat, int64_t g,
                                          int N, int i, int j, float (&emb)[64]) {
  const CT* bi = coords + (g * N + i) * 4;
  const CT* bj = coords + (g * N + j) * 4;
  const CT cxi = (bi[0] + bi[1]) * (CT)0.5, cyi = (bi[2] + bi[3]) * (CT)0.5;
  const CT wi = (bi[1] - bi[0]) + (CT)1, hi = (bi[3] - bi[2]) + (CT)1;
  const CT cxj = (bj[0] + bj[1]) * (CT)0.5, cyj = (bj[2] + bj[3]) * (CT)0.5;
  const CT wj = (bj[1] - bj[0]) + (CT)1, hj = (bj[3] - bj[2]) + (CT)1;
  CT dl[4];
  dl[0] = Trig<CT>::lg(Trig<CT>::mx(Trig<CT>::ab((cxi - cxj) / wi), (CT)1e-3));
  dl[1] = Trig<CT>::lg(Trig<CT>::mx(Trig<CT>::ab((cyi - cyj) / hi), (CT)1e-3));
  dl[2] = Trig<CT>::lg(wi / wj);
  dl[3] = Trig<CT>::lg(hi / hj);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const CT base = (CT)100 * dl[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      CT s, co;
      if constexpr (FAST) {
        const float rev = (float)(base * (CT)dim_mat[k]) * 0.15915494309189535f;
        s = __builtin_amdgcn_sinf(rev);
        co = __builtin_amdgcn_cosf(rev);
      } else {
        Trig<CT>::sc(base * (CT)dim_mat[k], &s, &co);
      }
      emb[c * 8 + k] = (float)s;
      emb[32 + c * 8 + k] = (float)co;
    }
  }
}

template <typename CT, bool FAST = false>
__global__ __launch_bounds__(256) void box_bias_fwd_kernel(const CT* __restrict__ coords, const float* __restrict__ dim_mat,
                                                           const float* __restrict__ wg_w, const float* __restrict__ wg_b,
                                                           float* __restrict__ bias, float* __restrict__ emb_out,
                                                           int G, int N, int heads) {
  __shared__ float w[16 * 64 + 16];
  for (int e = threadIdx.x; e < heads * 64 + heads; e += 256)
    w[e] = wg_w ? (e < heads * 64 ? wg_w[e] : wg_b[e - heads * 64]) : 0.f;
  __syncthreads();
  const int64_t total = (int64_t)G * N * N;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t g = idx / (N * N);
    const int rem = (int)(idx - g * N * N);
    const int i = rem / N, j = rem - i * N;
    float emb[64];
    box_embed<CT, FAST>(coords, dim_mat, g, N, i, j, emb);
    if (emb_out) {
#pragma unroll
      for (int e = 0; e < 64; ++e) emb_out[idx * 64 + e] = emb[e];
    }
    if (bias) {
      for (int h = 0; h < heads; ++h) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) acc += emb[e] * w[h * 64 + e];
        acc += w[heads * 64 + h];
        const float v = fmaxf(acc, 0.f);
        bias[((g * heads + h) * N + i) * N + j] = logf(fmaxf(v, 1e-6f));
      }
    }
  }
}

template <typename CT, bool FAST = false>
__global__ __launch_bounds__(256) void box_bias_bwd_kernel(const CT* __restrict__ coords, const float* __restrict__ dim_mat,
                                                           const float* __restrict__ wg_w, const float* __restrict__ wg_b,
                                                           const float* __restrict__ dbias, float* __restrict__ dwg_w,
                                                           float* __restrict__ dwg_b, int G, int N, int heads) {
  // per iteration the block stages 256 embeddings E[256][65] and pre-activation gradients D[256][8]; then lane e of wave v adds
  // the rows r = v, v + 4, ... into ITS partial sums of column e for all 8 heads (one E read and two broadcast 16-byte D reads per
  // 8 FMAs: the round-3 form -- two (head, e) outputs per thread over all 256 rows -- spent three LDS reads on every two FMAs and
  // two thirds of the kernel's time in that loop).  The four waves' partial sums meet in LDS once, after the last iteration.
  __shared__ float w[8 * 64 + 8];
  __shared__ float E[256][65];
  __shared__ __attribute__((aligned(16))) float D[256][8];
  for (int e = threadIdx.x; e < heads * 64 + heads; e += 256) w[e] = e < heads * 64 ? wg_w[e] : wg_b[e - heads * 64];
  __syncthreads();
  const int64_t total = (int64_t)G * N * N;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // d WG[h][lane] over this wave's rows
  float ab[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // d b[h] over this wave's rows (lane-strided, summed at the end)
  for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {
    const int64_t idx = base + tid;
    float dh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx < total) {
      const int64_t g = idx / (N * N);
      const int rem = (int)(idx - g * N * N);
      const int i = rem / N, j = rem - i * N;
      float emb[64];
      box_embed<CT, FAST>(coords, dim_mat, g, N, i, j, emb);
#pragma unroll
      for (int e = 0; e < 64; ++e) E[tid][e] = emb[e];
      for (int h = 0; h < heads; ++h) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) a += emb[e] * w[h * 64 + e];
        a += w[heads * 64 + h];
        const float db = dbias[((g * heads + h) * N + i) * N + j];
        // d log(max(relu(x),1e-6)) / dx = 1/x where x > 1e-6, else 0
        dh[h] = a > 1e-6f ? db / a : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 64; ++e) E[tid][e] = 0.f;
    }
    *reinterpret_cast<float4*>(&D[tid][0]) = make_float4(dh[0], dh[1], dh[2], dh[3]);
    *reinterpret_cast<float4*>(&D[tid][4]) = make_float4(dh[4], dh[5], dh[6], dh[7]);
#pragma unroll
    for (int h = 0; h < 8; ++h) ab[h] += dh[h];
    __syncthreads();
#pragma unroll 4
    for (int r = wv; r < 256; r += 4) {
      const float er = E[r][lane];
      const float4 d0 = *reinterpret_cast<const float4*>(&D[r][0]), d1 = *reinterpret_cast<const float4*>(&D[r][4]);
      acc[0] += d0.x * er; acc[1] += d0.y * er; acc[2] += d0.z * er; acc[3] += d0.w * er;
      acc[4] += d1.x * er; acc[5] += d1.y * er; acc[6] += d1.z * er; acc[7] += d1.w * er;
    }
    __syncthreads();
  }
  // cross-wave sums through LDS (E is free now): P[wave][h][e], then one atomic per output and block
  float* Pw = &E[0][0];
#pragma unroll
  for (int h = 0; h < 8; ++h) Pw[(wv * 8 + h) * 64 + lane] = acc[h];
  float* Pb = Pw + 4 * 8 * 64;                           // [wave][h]
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const float sgm = wave_sum(ab[h]);
    if (lane == 0) Pb[wv * 8 + h] = sgm;
  }
  __syncthreads();
  for (int o = tid; o < heads * 64; o += 256) {
    const int h = o >> 6, e = o & 63;
    atomicAdd(dwg_w + o, Pw[(0 * 8 + h) * 64 + e] + Pw[(1 * 8 + h) * 64 + e] + Pw[(2 * 8 + h) * 64 + e] + Pw[(3 * 8 + h) * 64 + e]);
  }
  if (tid < heads) atomicAdd(dwg_b + tid, Pb[tid] + Pb[8 + tid] + Pb[16 + tid] + Pb[24 + tid]);
}

static int box_grid(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int fcmf_box_bias_fwd(const void* coords, int coord_dtype, const float* dim_mat, const float* wg_w,
                                 const float* wg_b, float* bias, int G, int N, int heads, void* stream) {
  if (!coords || !dim_mat || !wg_w || !wg_b || !bias || G <= 0 || N <= 0 || heads <= 0 || heads > 8) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(box_grid((int64_t)G * N * N));
  if (coord_dtype == FCMF_F64)
    hipLaunchKernelGGL((box_bias_fwd_kernel<double>), grid, dim3(256), 0, st, (const double*)coords, dim_mat, wg_w, wg_b, bias, (float*)nullptr, G, N, heads);
  else if (coord_dtype == FCMF_F32)
    hipLaunchKernelGGL((box_bias_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)coords, dim_mat, wg_w, wg_b, bias, (float*)nullptr, G, N, heads);
  else if (coord_dtype == (FCMF_F32 | FCMF_BOX_FAST_TRIG))
    hipLaunchKernelGGL((box_bias_fwd_kernel<float, true>), grid, dim3(256), 0, st, (const float*)coords, dim_mat, wg_w, wg_b, bias, (float*)nullptr, G, N, heads);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_box_embedding(const void* coords, int coord_dtype, const float* dim_mat, float* emb, int G, int N,
                                  void* stream) {
  if (!coords || !dim_mat || !emb || G <= 0 || N <= 0) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(box_grid((int64_t)G * N * N));
  if (coord_dtype == FCMF_F64)
    hipLaunchKernelGGL((box_bias_fwd_kernel<double>), grid, dim3(256), 0, st, (const double*)coords, dim_mat, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, emb, G, N, 0);
  else if (coord_dtype == FCMF_F32)
    hipLaunchKernelGGL((box_bias_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)coords, dim_mat, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, emb, G, N, 0);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_box_bias_bwd(const void* coords, int coord_dtype, const float* dim_mat, const float* wg_w,
                                 const float* wg_b, const float* dbias, float* dwg_w, float* dwg_b, int G, int N,
                                 int heads, void* stream) {
  if (!coords || !dim_mat || !wg_w || !wg_b || !dbias || !dwg_w || !dwg_b || G <= 0 || N <= 0 || heads <= 0 || heads > 8)
    return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t b = ((int64_t)G * N * N + 255) / 256;
  dim3 grid((int)(b > 512 ? 512 : b));
  if (coord_dtype == FCMF_F64)
    hipLaunchKernelGGL((box_bias_bwd_kernel<double>), grid, dim3(256), 0, st, (const double*)coords, dim_mat, wg_w, wg_b, dbias, dwg_w, dwg_b, G, N, heads);
  else if (coord_dtype == FCMF_F32)
    hipLaunchKernelGGL((box_bias_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)coords, dim_mat, wg_w, wg_b, dbias, dwg_w, dwg_b, G, N, heads);
  else if (coord_dtype == (FCMF_F32 | FCMF_BOX_FAST_TRIG))
    hipLaunchKernelGGL((box_bias_bwd_kernel<float, true>), grid, dim3(256), 0, st, (const float*)coords, dim_mat, wg_w, wg_b, dbias, dwg_w, dwg_b, G, N, heads);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
