// Cross entropy, casts, dropout, axis sums and the fused clip + AdamW / BertAdam updates.
// All of these are HBM-bound streaming kernels: 16-byte accesses, grid-stride, no re-reads.
#include "common.h"

// ---- cross entropy: one wave per row, classes strided over lanes --------------------------
template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, float* __restrict__ loss_rows,
                                                       float* __restrict__ nvalid, int n, int C, int64_t ignore_index) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t lab = labels[row];
  if (lab == ignore_index) { if (lane == 0) loss_rows[row] = 0.f; return; }
  const T* lr = logits + (int64_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, to_f32<T>(lr[c]));
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(to_f32<T>(lr[c]) - m);
  s = wave_sum(s);
  if (lane == 0) {
    loss_rows[row] = (m + __logf(s)) - to_f32<T>(lr[lab]);
    if (nvalid) atomicAdd(nvalid, 1.0f);
  }
}

// mean over the non-ignored rows (x mult) and the backward's per-row scale, by ONE workgroup in a fixed order (f64 partial sums):
// out2[0] = mult * sum(loss_rows) / nvalid, out2[1] = mult / nvalid  (0 / 0 = NaN when every row is ignored, as torch)
__global__ __launch_bounds__(256) void xent_mean_kernel(const float* __restrict__ loss_rows, const int64_t* __restrict__ labels,
                                                        int n, int64_t ignore_index, float mult, float* __restrict__ out2) {
  __shared__ double ssum[256];
  __shared__ int scnt[256];
  double s = 0.0;
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256)
    if (labels[i] != ignore_index) { s += (double)loss_rows[i]; ++c; }
  ssum[threadIdx.x] = s; scnt[threadIdx.x] = c;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) { ssum[threadIdx.x] += ssum[threadIdx.x + h]; scnt[threadIdx.x] += scnt[threadIdx.x + h]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nv = (float)scnt[0];
    out2[0] = mult * ((float)ssum[0] / nv);
    out2[1] = mult / nv;
  }
}

// additive attention mask from a 0 / 1 integer mask: out[r][c] = (mask[r][c] - 1) * (-value)  (= (1 - mask) * value)
__global__ __launch_bounds__(256) void additive_mask_kernel(const int64_t* __restrict__ mask, int64_t ld, float* __restrict__ out,
                                                            int rows, int cols, float value) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  out[i] = (float)(mask[(int64_t)r * ld + c] - 1) * (-value);
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const T* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, T* __restrict__ dlogits,
                                                       int64_t ldd, const float* __restrict__ scale_ptr, float extra,
                                                       int n, int C, int64_t ignore_index) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t lab = labels[row];
  T* dr = dlogits + (int64_t)row * ldd;
  if (lab == ignore_index) { for (int c = lane; c < C; c += 64) dr[c] = from_f32<T>(0.f); return; }
  const T* lr = logits + (int64_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, to_f32<T>(lr[c]));
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(to_f32<T>(lr[c]) - m);
  s = wave_sum(s);
  const float sc = (scale_ptr ? *scale_ptr : 1.0f) * extra;
  for (int c = lane; c < C; c += 64) {
    float pr = __expf(to_f32<T>(lr[c]) - m) / s;
    dr[c] = from_f32<T>((pr - (c == lab ? 1.f : 0.f)) * sc);
  }
}

// ---- cross entropy over a WIDE class axis (IAOG: the 64001-entry vocabulary, run_pretraining_fcmf.py:322-324):
// one workgroup per row, 16-byte loads, one online (max, sum-exp) pass; the backward may run IN PLACE
// (dlogits == logits: every element is read and written by the same thread).
template <typename T> struct XVec;
template <> struct XVec<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct XVec<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
  m = mn;
}

// block-wide (max, sum exp(x - max)) of lr[0..C): every thread returns the same pair
template <typename T>
__device__ __forceinline__ void row_softmax_stats(const T* __restrict__ lr, int C, bool vec, float& m_out, float& s_out) {
  constexpr int VN = XVec<T>::N;
  typedef typename XVec<T>::type vec_t;
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float m = -INFINITY, s = 0.f;
  const int Cv = vec ? (C / VN) * VN : 0;
  for (int c = tid * VN; c < Cv; c += 256 * VN) {
    const vec_t v = *reinterpret_cast<const vec_t*>(lr + c);
    float x[VN], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < VN; ++e) { x[e] = (float)v[e]; mx = fmaxf(mx, x[e]); }
    const float mn = fmaxf(m, mx);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < VN; ++e) acc += __expf(x[e] - mn);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + acc;
    m = mn;
  }
  for (int c = Cv + tid; c < C; c += 256) online_merge(m, s, to_f32<T>(lr[c]), 1.f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) online_merge(m, s, __shfl_xor(m, o, 64), __shfl_xor(s, o, 64));
  if (lane == 0) { red[0][w] = m; red[1][w] = s; }
  __syncthreads();
  m = red[0][0]; s = red[1][0];
#pragma unroll
  for (int k = 1; k < 4; ++k) online_merge(m, s, red[0][k], red[1][k]);
  m_out = m; s_out = s;
}

template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_wide_kernel(const T* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ labels, float* __restrict__ loss_rows,
                                                            float* __restrict__ nvalid, int C, int64_t ignore_index, int vec) {
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  if (lab == ignore_index) { if (threadIdx.x == 0) loss_rows[row] = 0.f; return; }
  const T* lr = logits + (int64_t)row * ld;
  float m, s;
  row_softmax_stats<T>(lr, C, vec != 0, m, s);
  if (threadIdx.x == 0) {
    loss_rows[row] = (m + __logf(s)) - to_f32<T>(lr[lab]);
    if (nvalid) atomicAdd(nvalid, 1.0f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_wide_kernel(const T* logits, int64_t ld, const int64_t* __restrict__ labels,
                                                            T* dlogits, int64_t ldd, const float* __restrict__ scale_ptr,
                                                            float extra, int C, int64_t ignore_index, int vec) {
  constexpr int VN = XVec<T>::N;
  typedef typename XVec<T>::type vec_t;
  const int row = blockIdx.x, tid = threadIdx.x;
  const int64_t lab = labels[row];
  T* dr = dlogits + (int64_t)row * ldd;
  const int Cv = vec ? (C / VN) * VN : 0;
  if (lab == ignore_index) {
    vec_t z;
#pragma unroll
    for (int e = 0; e < VN; ++e) z[e] = from_f32<T>(0.f);
    for (int c = tid * VN; c < Cv; c += 256 * VN) *reinterpret_cast<vec_t*>(dr + c) = z;
    for (int c = Cv + tid; c < C; c += 256) dr[c] = from_f32<T>(0.f);
    return;
  }
  const T* lr = logits + (int64_t)row * ld;
  float m, s;
  row_softmax_stats<T>(lr, C, vec != 0, m, s);
  const float sc = (scale_ptr ? *scale_ptr : 1.0f) * extra, inv = sc / s;
  for (int c = tid * VN; c < Cv; c += 256 * VN) {
    const vec_t v = *reinterpret_cast<const vec_t*>(lr + c);
    vec_t o;
#pragma unroll
    for (int e = 0; e < VN; ++e) o[e] = from_f32<T>(__expf((float)v[e] - m) * inv - ((c + e) == lab ? sc : 0.f));
    *reinterpret_cast<vec_t*>(dr + c) = o;
  }
  for (int c = Cv + tid; c < C; c += 256) dr[c] = from_f32<T>(__expf(to_f32<T>(lr[c]) - m) * inv - (c == lab ? sc : 0.f));
}

// ---- elementwise ---------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    Vec4<TD>::store(d + i * 4, Vec4<TS>::load(s + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) d[n4 * 4 + threadIdx.x] = from_f32<TD>(to_f32<TS>(s[n4 * 4 + threadIdx.x]));
}

// dst[c][r] = bf16(src[r][c]): 64x64 tiles through LDS, coalesced on both sides (transposed bf16 weight copies, so
// that dX = dY * W reads W as a K-contiguous operand)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    tile[ty * 16 + i][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
  if ((R & 3) == 0) {        // 8-byte stores: thread = (output row, 4 consecutive elements of it); the column reads of the 65-float pitch
#pragma unroll             // are conflict free (2 bytes per lane made every wave store a 128-byte sliver: 2.8 TB/s over the step's weights)
    for (int i = 0; i < 4; ++i) {
      const int cc = i * 16 + (threadIdx.x >> 4), rr = 4 * (threadIdx.x & 15);
      const int c = c0 + cc, r = r0 + rr;
      if (c < C && r < R) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)tile[rr + j][cc];
        *reinterpret_cast<bf16x4*>(dst + (int64_t)c * R + r) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * R + r] = (bf16_t)tile[tx][ty * 16 + i];
  }
}

// the same for MANY weights in one launch (after an optimizer step every transposed copy is stale at once: 69 launches
// of ~10 us each per step otherwise).  desc[b] = {tensor, tile row, tile column} of workgroup b.
__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const int64_t* __restrict__ src_ptrs, const int64_t* __restrict__ dst_ptrs,
                                                                   const int32_t* __restrict__ dims, const int32_t* __restrict__ desc) {
  __shared__ float tile[64][65];
  const int t = desc[3 * blockIdx.x], r0 = desc[3 * blockIdx.x + 1] * 64, c0 = desc[3 * blockIdx.x + 2] * 64;
  const float* __restrict__ src = reinterpret_cast<const float*>(src_ptrs[t]);
  bf16_t* __restrict__ dst = reinterpret_cast<bf16_t*>(dst_ptrs[t]);
  const int R = dims[2 * t], C = dims[2 * t + 1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    tile[ty * 16 + i][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * R + r] = (bf16_t)tile[tx][ty * 16 + i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p,
                                                      float inv_keep, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = from_f32<T>(to_f32<T>(x[i]) * dropout_mult(seed, (uint64_t)i, p, inv_keep));
}

// out = dy * act'(aux): kind 0 = tanh (aux = tanh output), kind 1 = gelu (aux = pre-activation)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux, T* __restrict__ out,
                                                      int64_t n, int kind) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = to_f32<T>(aux[i]), d = to_f32<T>(dy[i]);
    out[i] = from_f32<T>(kind == 0 ? d * (1.0f - a * a) : d * dgelu_f(a));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_axis_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t outer,
                                                       int reps, int64_t inner) {
  const int64_t total = outer * inner;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i / inner, c = i - o * inner;
    float s = 0.f;
    for (int r = 0; r < reps; ++r) s += to_f32<T>(in[(o * reps + r) * inner + c]);
    out[i] = from_f32<T>(s);
  }
}

// ---- optimizer ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multi_sumsq_kernel(const int64_t* __restrict__ g_ptrs, const int64_t* __restrict__ sizes,
                                                          const int32_t* __restrict__ chunk_tensor,
                                                          const int64_t* __restrict__ chunk_offset, int chunk_size,
                                                          double* __restrict__ sumsq, double* __restrict__ per_tensor) {
  __shared__ double red[4];
  const int t = chunk_tensor[blockIdx.x];
  const int64_t off = chunk_offset[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(g_ptrs[t]) + off;
  int64_t n = sizes[t] - off;
  if (n > chunk_size) n = chunk_size;
  double s = 0.0;
  const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
      s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = red[0] + red[1] + red[2] + red[3];
    if (sumsq) atomicAdd(sumsq, tot);
    if (per_tensor) atomicAdd(per_tensor + t, tot);
  }
}

struct AdamArgs {
  const int64_t *p_ptrs, *g_ptrs, *m_ptrs, *v_ptrs, *bf16_ptrs, *sizes;
  const int32_t *group_of, *chunk_tensor;
  const int64_t* chunk_offset;
  int chunk_size;
  float lr[8], wd[8];
  float beta1, beta2, eps, bc1, bc2_sqrt;
  const double* sumsq;
  float max_norm;
};

__global__ __launch_bounds__(256) void multi_adamw_kernel(AdamArgs a) {
  const int t = a.chunk_tensor[blockIdx.x];
  const int64_t off = a.chunk_offset[blockIdx.x];
  float* p = reinterpret_cast<float*>(a.p_ptrs[t]) + off;
  const float* g = reinterpret_cast<const float*>(a.g_ptrs[t]) + off;
  float* m = reinterpret_cast<float*>(a.m_ptrs[t]) + off;
  float* v = reinterpret_cast<float*>(a.v_ptrs[t]) + off;
  bf16_t* sh = a.bf16_ptrs && a.bf16_ptrs[t] ? reinterpret_cast<bf16_t*>(a.bf16_ptrs[t]) + off : nullptr;
  int64_t n = a.sizes[t] - off;
  if (n > a.chunk_size) n = a.chunk_size;
  const int grp = a.group_of[t];
  const float lr = a.lr[grp], wd = a.wd[grp];
  float coef = 1.0f;
  if (a.max_norm > 0.f && a.sumsq) {
    const float tn = (float)sqrt(*a.sumsq);
    coef = fminf(1.0f, a.max_norm / (tn + 1e-6f));
  }
  const float step_size = lr / a.bc1;
  // 16 bytes per lane where the chunk allows; the moments and the gradient are touched once per step: nontemporal
  typedef float f4 __attribute__((ext_vector_type(4)));
  int64_t i0 = 0;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
      (!sh || (reinterpret_cast<uintptr_t>(sh) & 7) == 0)) {
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const f4 g4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i) * coef;
      f4 p4 = reinterpret_cast<const f4*>(p)[i] * (1.0f - lr * wd);
      const f4 m4 = a.beta1 * __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i) + (1.0f - a.beta1) * g4;
      const f4 v4 = a.beta2 * __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i) + (1.0f - a.beta2) * g4 * g4;
#pragma unroll
      for (int e = 0; e < 4; ++e) p4[e] -= step_size * (m4[e] / (sqrtf(v4[e]) / a.bc2_sqrt + a.eps));
      reinterpret_cast<f4*>(p)[i] = p4;
      __builtin_nontemporal_store(m4, reinterpret_cast<f4*>(m) + i);
      __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(v) + i);
      if (sh) {
        bf16x4 o;
        o[0] = (bf16_t)p4[0]; o[1] = (bf16_t)p4[1]; o[2] = (bf16_t)p4[2]; o[3] = (bf16_t)p4[3];
        reinterpret_cast<bf16x4*>(sh)[i] = o;
      }
    }
    i0 = n4 << 2;
  }
  for (int64_t i = i0 + threadIdx.x; i < n; i += 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = a.beta1 * m[i] + (1.0f - a.beta1) * gi;
    const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * gi * gi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (sh) sh[i] = (bf16_t)pi;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)g[i] * g[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bertadam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, float lr, float b1, float b2, float e,
                                                       float wd, float max_norm, const double* __restrict__ sumsq) {
  float coef = 1.0f;
  if (max_norm > 0.f) coef = fminf(1.0f, max_norm / ((float)sqrt(*sumsq) + 1e-6f));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    float upd = mi / (sqrtf(vi) + e);
    if (wd > 0.f) upd += wd * p[i];
    p[i] -= lr * upd;
    m[i] = mi; v[i] = vi;
  }
}

// ---- host ----------------------------------------------------------------------------------
static int grid_for(int64_t n, int per) {
  int64_t b = (n + per - 1) / per;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// 2: fcmf_attn_mfma_bwd gained `colsum`; 3: explicit GEMM context (fcmf_gemm takes a ctx), fcmf_dp_*; 4: fcmf_embed_scale_* take the table's row count
extern "C" int fcmf_abi_version(void) { return 4; }
extern "C" const char* fcmf_build_info(void) { return "libfcmf_hip gfx950 (CDNA4, wave64) abi 4"; }

extern "C" int fcmf_xent_fwd(const void* logits, int64_t ld, const int64_t* labels, float* loss_rows, float* nvalid, int n,
                             int C, int64_t ignore_index, int dtype, void* stream) {
  if (!logits || !labels || !loss_rows || n < 0 || C <= 0) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C >= 1024) {     // wide class axis: one workgroup per row
    const int es = dtype == FCMF_F32 ? 4 : 2;
    const int vec = ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) && ((ld * es) % 16 == 0);
    if (dtype == FCMF_F32) hipLaunchKernelGGL((xent_fwd_wide_kernel<float>), dim3(n), dim3(256), 0, st, (const float*)logits, ld, labels, loss_rows, nvalid, C, ignore_index, vec);
    else if (dtype == FCMF_BF16) hipLaunchKernelGGL((xent_fwd_wide_kernel<bf16_t>), dim3(n), dim3(256), 0, st, (const bf16_t*)logits, ld, labels, loss_rows, nvalid, C, ignore_index, vec);
    else return FCMF_ERR_UNSUPPORTED;
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  dim3 grid((n + 3) / 4);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((xent_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)logits, ld, labels, loss_rows, nvalid, n, C, ignore_index);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((xent_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)logits, ld, labels, loss_rows, nvalid, n, C, ignore_index);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_xent_mean(const float* loss_rows, const int64_t* labels, int n, int64_t ignore_index, float mult,
                              float* out2, void* stream) {
  if (!loss_rows || !labels || !out2 || n < 0) return FCMF_ERR_ARG;
  hipLaunchKernelGGL(xent_mean_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), loss_rows, labels, n,
                     ignore_index, mult, out2);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_additive_mask(const int64_t* mask, int64_t ld, float* out, int rows, int cols, float value, void* stream) {
  if (!mask || !out || rows < 0 || cols < 0 || ld < cols) return FCMF_ERR_ARG;
  if (rows == 0 || cols == 0) return FCMF_OK;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(additive_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     mask, ld, out, rows, cols, value);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_xent_bwd(const void* logits, int64_t ld, const int64_t* labels, void* dlogits, int64_t ldd,
                             const float* scale_ptr, float extra_scale, int n, int C, int64_t ignore_index, int dtype,
                             void* stream) {
  if (!logits || !labels || !dlogits || n < 0 || C <= 0) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C >= 1024) {
    const int es = dtype == FCMF_F32 ? 4 : 2;
    const int vec = (((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0) &&
                    ((ld * es) % 16 == 0) && ((ldd * es) % 16 == 0);
    if (dtype == FCMF_F32) hipLaunchKernelGGL((xent_bwd_wide_kernel<float>), dim3(n), dim3(256), 0, st, (const float*)logits, ld, labels, (float*)dlogits, ldd, scale_ptr, extra_scale, C, ignore_index, vec);
    else if (dtype == FCMF_BF16) hipLaunchKernelGGL((xent_bwd_wide_kernel<bf16_t>), dim3(n), dim3(256), 0, st, (const bf16_t*)logits, ld, labels, (bf16_t*)dlogits, ldd, scale_ptr, extra_scale, C, ignore_index, vec);
    else return FCMF_ERR_UNSUPPORTED;
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  dim3 grid((n + 3) / 4);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((xent_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)logits, ld, labels, (float*)dlogits, ldd, scale_ptr, extra_scale, n, C, ignore_index);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((xent_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)logits, ld, labels, (bf16_t*)dlogits, ldd, scale_ptr, extra_scale, n, C, ignore_index);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
  if (!src || !dst || n < 0) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(n, 1024));
  if (src_dtype == FCMF_F32 && dst_dtype == FCMF_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == FCMF_BF16 && dst_dtype == FCMF_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == FCMF_F32 && dst_dtype == FCMF_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == FCMF_BF16 && dst_dtype == FCMF_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_cast_transpose(const float* src, void* dst, int rows, int cols, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0) return FCMF_ERR_ARG;
  if (rows == 0 || cols == 0) return FCMF_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (bf16_t*)dst, rows, cols);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_multi_cast_transpose(const int64_t* src_ptrs, const int64_t* dst_ptrs, const int32_t* dims,
                                         const int32_t* tile_desc, int ntiles, void* stream) {
  if (!src_ptrs || !dst_ptrs || !dims || !tile_desc || ntiles < 0) return FCMF_ERR_ARG;
  if (ntiles == 0) return FCMF_OK;
  hipLaunchKernelGGL(multi_cast_transpose_kernel, dim3(ntiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src_ptrs,
                     dst_ptrs, dims, tile_desc);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(n, 256));
  const float inv_keep = 1.0f / (1.0f - p);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((dropout_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (float*)y, n, p, inv_keep, seed);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, p, inv_keep, seed);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_act_bwd(const void* dy, const void* aux, void* out, int64_t n, int kind, int dtype, void* stream) {
  if (!dy || !aux || !out || n < 0 || (kind != 0 && kind != 1)) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(n, 256));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((act_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dy, (const float*)aux, (float*)out, n, kind);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)aux, (bf16_t*)out, n, kind);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

// ---- IAOG decoder glue (reference mm_modeling.py:79-92 head-tiled projections, :650 scaled embedding, :619-633 positional
// encoding): the three pieces that round 2 left to torch (scatter_add_, weight[ids] * scale + P, index_add_) --------------------
// out[row] = weight[ids[row]] * scale + P[row % S]   (P may be NULL)
template <typename T>
__global__ __launch_bounds__(256) void embed_scale_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ w,
                                                              const float* __restrict__ P, T* __restrict__ out, int n, int H,
                                                              int S, float scale, int64_t V) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t id = ids[row];
  if (id < 0 || id >= V) {      // a token id outside the table: the row is NaN (loud in the loss), nothing is read
    const float q = __builtin_nanf("");
    for (int c = lane * 4; c < H; c += 256) Vec4<T>::store(out + (int64_t)row * H + c, make_float4(q, q, q, q));
    return;
  }
  const float* wr = w + id * (int64_t)H;
  const float* pr = P ? P + (int64_t)(row % S) * H : nullptr;
  for (int c = lane * 4; c < H; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(wr + c);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    if (pr) {
      const float4 q = *reinterpret_cast<const float4*>(pr + c);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    Vec4<T>::store(out + (int64_t)row * H + c, v);
  }
}
// dweight[ids[row]] += dy[row] * scale  (rows of a token that occurs several times add up: float atomics)
template <typename T>
__global__ __launch_bounds__(256) void embed_scale_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ ids,
                                                              float* __restrict__ dw, int n, int H, float scale, int64_t V,
                                                              int32_t* __restrict__ oob) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t id = ids[row];
  if (id < 0 || id >= V) {      // never write outside the gradient slice (it lies in the flat arena: the neighbour is another gradient)
    if (oob && lane == 0) atomicAdd(oob, 1);
    return;
  }
  float* wr = dw + id * (int64_t)H;
  for (int c = lane; c < H; c += 64)      // (256 contiguous bytes per atomic wave-instruction: see embed_bwd_kernel)
    atomicAdd(wr + c, to_f32<T>(dy[(int64_t)row * H + c]) * scale);
}
// the decoder Attention pairs output slot s of batch element g with head (s * G + g) % heads (mm_modeling.py:79-85: inputs are
// tiled head-major, weights batch-major).  The attention backward produces gradients per SLOT; the projections want them per
// HEAD: out[g, t, h, :] = sum over the slots s that read head h of slot[g, t, s, :]  (several slots can share a head).
template <typename T>
__global__ __launch_bounds__(256) void head_gather_kernel(const T* __restrict__ slot, T* __restrict__ out, int64_t ldo, int G, int Tn,
                                                          int heads, int d) {
  const int d4 = d >> 2;
  const int64_t total = (int64_t)G * Tn * heads * d4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % d4) * 4;
    const int64_t r = i / d4;
    const int h = (int)(r % heads);
    const int64_t gt = r / heads;
    const int g = (int)(gt / Tn);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < heads; ++sl) {
      if ((int)(((int64_t)sl * G + g) % heads) != h) continue;
      const float4 v = Vec4<T>::load(slot + (gt * heads + sl) * d + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    Vec4<T>::store(out + gt * ldo + (int64_t)h * d + c, acc);
  }
}

extern "C" int fcmf_embed_scale_fwd(const int64_t* ids, const float* weight, const float* pos_table, void* out, int n, int H, int S,
                                    int64_t V, float scale, int dtype, void* stream) {
  if (!ids || !weight || !out || n < 0 || H <= 0 || H % 4 || S <= 0 || V <= 0) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((n + 3) / 4);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((embed_scale_fwd_kernel<float>), grid, dim3(256), 0, st, ids, weight, pos_table, (float*)out, n, H, S, scale, V);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((embed_scale_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, ids, weight, pos_table, (bf16_t*)out, n, H, S, scale, V);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_embed_scale_bwd(const void* dy, const int64_t* ids, float* dweight, int n, int H, int64_t V, int32_t* oob_count,
                                    float scale, int dtype, void* stream) {
  if (!dy || !ids || !dweight || n < 0 || H <= 0 || H % 4 || V <= 0) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((n + 3) / 4);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((embed_scale_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dy, ids, dweight, n, H, scale, V, oob_count);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((embed_scale_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dy, ids, dweight, n, H, scale, V, oob_count);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_head_gather(const void* slot, void* out, int64_t ldo, int G, int T, int heads, int d, int dtype, void* stream) {
  if (!slot || !out || G < 0 || T <= 0 || heads <= 0 || d <= 0 || d % 4 || ldo < (int64_t)heads * d) return FCMF_ERR_ARG;
  if (G == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for((int64_t)G * T * heads * (d / 4), 256));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((head_gather_kernel<float>), grid, dim3(256), 0, st, (const float*)slot, (float*)out, ldo, G, T, heads, d);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((head_gather_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)slot, (bf16_t*)out, ldo, G, T, heads, d);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_sum_axis(const void* in, void* out, int64_t outer, int reps, int64_t inner, int dtype, void* stream) {
  if (!in || !out || outer < 0 || reps <= 0 || inner <= 0) return FCMF_ERR_ARG;
  if (outer == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(outer * inner, 256));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((sum_axis_kernel<float>), grid, dim3(256), 0, st, (const float*)in, (float*)out, outer, reps, inner);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((sum_axis_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, outer, reps, inner);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_multi_sumsq(const int64_t* g_ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                                const int64_t* chunk_offset, int nchunks, int chunk_size, double* sumsq,
                                double* per_tensor, void* stream) {
  if (!g_ptrs || !sizes || !chunk_tensor || !chunk_offset || nchunks < 0 || chunk_size <= 0) return FCMF_ERR_ARG;
  if (nchunks == 0) return FCMF_OK;
  hipLaunchKernelGGL(multi_sumsq_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_ptrs, sizes,
                     chunk_tensor, chunk_offset, chunk_size, sumsq, per_tensor);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_multi_adamw(const int64_t* p_ptrs, const int64_t* g_ptrs, const int64_t* m_ptrs, const int64_t* v_ptrs,
                                const int64_t* bf16_ptrs, const int64_t* sizes, const int32_t* group_of,
                                const int32_t* chunk_tensor, const int64_t* chunk_offset, int nchunks, int chunk_size,
                                const float* lr, const float* wd, int ngroups, float beta1, float beta2, float eps, int step,
                                const double* sumsq, float max_norm, void* stream) {
  if (!p_ptrs || !g_ptrs || !m_ptrs || !v_ptrs || !sizes || !group_of || !chunk_tensor || !chunk_offset || !lr || !wd)
    return FCMF_ERR_ARG;
  if (ngroups <= 0 || ngroups > 8 || step < 1 || chunk_size <= 0 || nchunks < 0) return FCMF_ERR_ARG;
  if (nchunks == 0) return FCMF_OK;
  AdamArgs a{};
  a.p_ptrs = p_ptrs; a.g_ptrs = g_ptrs; a.m_ptrs = m_ptrs; a.v_ptrs = v_ptrs; a.bf16_ptrs = bf16_ptrs; a.sizes = sizes;
  a.group_of = group_of; a.chunk_tensor = chunk_tensor; a.chunk_offset = chunk_offset; a.chunk_size = chunk_size;
  for (int i = 0; i < ngroups; ++i) { a.lr[i] = lr[i]; a.wd[i] = wd[i]; }
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.sumsq = sumsq; a.max_norm = max_norm;
  hipLaunchKernelGGL(multi_adamw_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_bertadam(float* p, const float* g, float* m, float* v, int64_t n, float lr_scheduled, float beta1,
                             float beta2, float eps, float weight_decay, float max_grad_norm, double* scratch,
                             void* stream) {
  if (!p || !g || !m || !v || n < 0) return FCMF_ERR_ARG;
  if (max_grad_norm > 0.f && !scratch) return FCMF_ERR_ARG;
  if (n == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(n, 1024));
  if (max_grad_norm > 0.f) {
    if (hipMemsetAsync(scratch, 0, sizeof(double), st) != hipSuccess) return FCMF_ERR_LAUNCH;
    hipLaunchKernelGGL(sumsq_kernel, grid, dim3(256), 0, st, g, n, scratch);
  }
  hipLaunchKernelGGL(bertadam_kernel, grid, dim3(256), 0, st, p, g, m, v, n, lr_scheduled, beta1, beta2, eps, weight_decay,
                     max_grad_norm, scratch);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
