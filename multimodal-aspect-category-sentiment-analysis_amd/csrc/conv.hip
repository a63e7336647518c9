// ResNet-152 trunk helpers for gfx950 (reference: fcmf_framework/resnet_utils.py:13-56 driving torchvision's
// resnet152).  Activations are NHWC ([N*H*W, C] row-major), so every convolution is a GEMM of the step's MFMA kernels
// (gemm.hip): 1x1 / stride 1 directly on the activation matrix, everything else on the patch matrix built here.
//   im2col_kernel        patch matrix [N*Ho*Wo, Kpad] (k = (r, s, c), zero padding and zero tail columns)
//   bn_stats_kernel      GROUPED BatchNorm batch statistics: per (call group, channel) sum / sum of squares (double)
//   bn_finalize_kernel   statistics -> per-(group, channel) scale/shift + the running-statistics EMA in call order
//   bn_apply_kernel      y = relu?(x * scale + shift (+ residual)), in place
//   maxpool3x3s2_kernel  3x3 / stride 2 / pad 1
//   avgpool_kernel       adaptive average pool to [N, C, oh, ow] (the reference's output layout) or [N, oh*ow, C]
// All of them are HBM-bound streaming kernels: 8/16-byte accesses, a wave reads whole rows.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// patch matrix.  src element (n, h, w, c) at n*sn + h*sh + w*sw + c*sc (any layout: NCHW float32 crops, NHWC
// activations); dst[row, (r*kw + s)*C + c], row = (n*Ho + ho)*Wo + wo; columns >= kh*kw*C are zero.
// ---------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void im2col_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int N, int H, int W,
                                                     int C, int64_t sn, int64_t sh, int64_t sw, int64_t sc, int kh, int kw,
                                                     int stride, int pad, int Ho, int Wo, int Kpad) {
  // one wave per patch row: (n, ho, wo) are decomposed once per row, the lanes run over the row's columns with 32-bit
  // index arithmetic (a flat index with 64-bit divisions per element made the 7x7 stem VALU-bound: 1.6 ms per call)
  const int64_t rows = (int64_t)N * Ho * Wo;
  const int K = kh * kw * C;
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const int wo = (int)(row % Wo);
    const int64_t t = row / Wo;
    const int ho = (int)(t % Ho);
    const int64_t n = t / Ho;
    const TS* img = src + n * sn;
    TD* out = dst + row * Kpad;
    const int h0 = ho * stride - pad, w0 = wo * stride - pad;
    for (int col = lane; col < Kpad; col += 64) {
      float v = 0.f;
      if (col < K) {
        const int tap = col / C, c = col - tap * C, r = tap / kw, sx = tap - r * kw;
        const int hi = h0 + r, wi = w0 + sx;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = to_f32<TS>(img[hi * sh + wi * sw + c * sc]);
      }
      out[col] = from_f32<TD>(v);
    }
  }
}

// NHWC source with C % 8 == 0 (every convolution but the stem), same dtype both sides: one 16-byte (bf16) or two
// 16-byte (f32) accesses per 8 channels; a wave writes 64 consecutive 8-channel pieces of the patch matrix.
template <typename T>
__global__ __launch_bounds__(256) void im2col_nhwc8_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H,
                                                           int W, int C, int kh, int kw, int stride, int pad, int Ho, int Wo,
                                                           int Kpad) {
  const int64_t rows = (int64_t)N * Ho * Wo;
  const int K8 = Kpad >> 3, C8 = C >> 3, Kv = kh * kw * C8;
  typedef T vec8 __attribute__((ext_vector_type(8)));
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < rows * K8; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / K8;
    const int cv = (int)(idx - row * K8);
    vec8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.f);
    if (cv < Kv) {
      const int c8 = cv % C8, tap = cv / C8, s = tap % kw, r = tap / kw;
      const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
      const int64_t n = row / ((int64_t)Wo * Ho);
      const int hi = ho * stride + r - pad, wi = wo * stride + s - pad;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W)
        v = *reinterpret_cast<const vec8*>(src + ((n * H + hi) * W + wi) * C + c8 * 8);
    }
    *reinterpret_cast<vec8*>(dst + idx * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// RGB crops (any layout: element strides sn, sh, sw, sc; float32 / float64 / bf16) -> the zero-bordered NHWC input of
// fcmf_conv_gemm_runs: dst [N, H + 2 pad, Wp >= W + 2 pad, 4] in bf16, channel 3 = 0; only the interior is written (the border
// was zeroed once by the buffer's owner).  One thread per pixel, 8-byte stores; reads are coalesced along w.
// ---------------------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void pack_rgb0_kernel(const TS* __restrict__ src, bf16_t* __restrict__ dst, int N, int H, int W,
                                                        int64_t sn, int64_t sh, int64_t sw, int64_t sc, int pad, int Wp) {
  const int64_t total = (int64_t)N * H * W;
  const int Hp = H + 2 * pad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % W);
    const int64_t t = idx / W;
    const int h = (int)(t % H);
    const int64_t n = t / H;
    const TS* q = src + n * sn + h * sh + w * sw;
    bf16x4 o;
    o[0] = (bf16_t)(float)q[0]; o[1] = (bf16_t)(float)q[sc]; o[2] = (bf16_t)(float)q[2 * sc]; o[3] = (bf16_t)0.f;
    *reinterpret_cast<bf16x4*>(dst + ((n * Hp + h + pad) * Wp + w + pad) * 4) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// grouped batch statistics.  grid (channel slabs of 256, row chunks, groups); a thread owns 4 adjacent channels and
// every (256 / vecs-per-row)-th row of its chunk; waves are reduced through LDS; one double atomic per channel and
// workgroup.  sums [G, C, 2] (sum, sum of squares) must be zero on entry.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, double* __restrict__ sums, int C,
                                                       int64_t rows_per_group, int chunk) {
  __shared__ float4 red[2][256];
  const int slab_c = C < 256 ? C : 256, vpr = slab_c >> 2, rpp = 256 / vpr;      // vectors per row, rows per pass
  const int cv = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int c0 = blockIdx.x * 256 + cv * 4;
  const int g = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(rows_per_group, r0 + chunk);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  if (c0 < C) {
    const T* base = x + ((int64_t)g * rows_per_group) * C + c0;
    int64_t r = r0 + rsub;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {        // four rows in flight per thread (one load per iteration is latency-bound)
      const float4 v0 = Vec4<T>::load(base + r * C), v1 = Vec4<T>::load(base + (r + rpp) * C);
      const float4 v2 = Vec4<T>::load(base + (r + 2 * rpp) * C), v3 = Vec4<T>::load(base + (r + 3 * rpp) * C);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
      q.x += (v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x); q.y += (v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y);
      q.z += (v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z); q.w += (v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w);
    }
    for (; r < r1; r += rpp) {
      const float4 v = Vec4<T>::load(base + r * C);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rsub == 0 && c0 < C) {
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    for (int k = 0; k < rpp; ++k) {
      const float4 a = red[0][k * vpr + cv], b = red[1][k * vpr + cv];
      ds[0] += a.x; ds[1] += a.y; ds[2] += a.z; ds[3] += a.w;
      dq[0] += b.x; dq[1] += b.y; dq[2] += b.z; dq[3] += b.w;
    }
    // partial of this (group, row chunk): plain stores (hundreds of workgroups adding into the same few cache lines
    // with double atomics made this kernel 6x slower than its HBM traffic); the finalize kernels sum the chunks
    double* o = sums + (((int64_t)g * gridDim.y + blockIdx.y) * C + c0) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = ds[e]; o[2 * e + 1] = dq[e]; }
  }
}

// the row-chunk partials of fcmf_bn_stats / the backward statistics -> totals tot[G][C][2].  grid (C/64, G), 16 waves:
// lane = channel of a 64-channel slab, the waves stride the chunks (16-byte coalesced loads), LDS tree over the waves.
__global__ __launch_bounds__(1024) void bn_reduce_chunks_kernel(const double* __restrict__ sums, double* __restrict__ tot, int C,
                                                                int nchunks) {
  __shared__ double red[2][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, g = blockIdx.y;
  double s = 0, q = 0;
  if (c < C) {
    const double* p = sums + (((int64_t)g * nchunks + w) * C + c) * 2;
    for (int k = w; k < nchunks; k += 16) { s += p[0]; q += p[1]; p += (int64_t)16 * C * 2; }
  }
  red[0][w][lane] = s;
  red[1][w][lane] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    for (int k = 1; k < 16; ++k) { s += red[0][k][lane]; q += red[1][k][lane]; }
    tot[((int64_t)g * C + c) * 2] = s;
    tot[((int64_t)g * C + c) * 2 + 1] = q;
  }
}

// the same reduction over the float32 block statistics a colstats GEMM left behind (fcmf_gemm_colstats: [row block of 128][C][2]):
// group g = blocks g * nblocks .. (g + 1) * nblocks - 1.  Summed in double, in a fixed order.
__global__ __launch_bounds__(1024) void bn_reduce_blocks_kernel(const float* __restrict__ stats, double* __restrict__ tot, int C,
                                                                int nblocks) {
  __shared__ double red[2][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, g = blockIdx.y;
  double s = 0, q = 0;
  if (c < C) {
    const float2* p = reinterpret_cast<const float2*>(stats) + ((int64_t)g * nblocks + w) * C + c;
    int k = w;
    for (; k + 48 < nblocks; k += 64) {       // four block rows in flight per lane
      const float2 a = p[0], b = p[(int64_t)16 * C], d = p[(int64_t)32 * C], e = p[(int64_t)48 * C];
      s += ((double)a.x + (double)b.x) + ((double)d.x + (double)e.x);
      q += ((double)a.y + (double)b.y) + ((double)d.y + (double)e.y);
      p += (int64_t)64 * C;
    }
    for (; k < nblocks; k += 16) { const float2 a = p[0]; s += a.x; q += a.y; p += (int64_t)16 * C; }
  }
  red[0][w][lane] = s;
  red[1][w][lane] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    for (int k = 1; k < 16; ++k) { s += red[0][k][lane]; q += red[1][k][lane]; }
    tot[((int64_t)g * C + c) * 2] = s;
    tot[((int64_t)g * C + c) * 2 + 1] = q;
  }
}

// one thread per channel.  training: group g's batch mean / biased variance -> scale/shift[g]; running statistics
// take one EMA update per group IN GROUP ORDER with the unbiased variance (nn.BatchNorm2d, one reference call per
// group).  eval (sums == NULL): scale/shift[0] from the running statistics.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, int C, int G, double count,
                                                          double inv_count, float momentum, float eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float w = gamma[c], b = beta[c];
  if (!sums) {
    const float rs = 1.0f / sqrtf(rvar[c] + eps), sc = w * rs;
    scale[c] = sc;
    shift[c] = b - rmean[c] * sc;
    if (mean_out) { mean_out[c] = rmean[c]; rstd_out[c] = rs; }
    return;
  }
  float rm = rmean[c], rv = rvar[c];
  for (int g = 0; g < G; ++g) {
    const double s = sums[((int64_t)g * C + c) * 2], q = sums[((int64_t)g * C + c) * 2 + 1];      // totals
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const float rs = 1.0f / sqrtf((float)var + eps), sc = w * rs;
    scale[(int64_t)g * C + c] = sc;
    shift[(int64_t)g * C + c] = b - (float)mean * sc;
    if (mean_out) { mean_out[(int64_t)g * C + c] = (float)mean; rstd_out[(int64_t)g * C + c] = rs; }
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    rm = (1.f - momentum) * rm + momentum * (float)mean;
    rv = (1.f - momentum) * rv + momentum * (float)unbiased;
  }
  rmean[c] = rm;
  rvar[c] = rv;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int64_t rows, int C, int64_t rows_per_group, int relu, int H, int W, int pad) {
  // pad > 0: y is an NHWC tensor WITH a zero border of `pad` pixels ([n, H + 2 pad, W + 2 pad, C]; the border was zeroed once
  // by its owner): the input of an implicit-GEMM 3x3 convolution (fcmf_conv_gemm), written here without a padding pass
  const int C4 = C >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < rows * C4; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / C4;
    const int c = (int)(idx - row * C4) * 4;
    int64_t orow = row;
    if (pad > 0) {
      const int w = (int)(row % W);
      const int64_t t = row / W;
      const int h = (int)(t % H);
      orow = ((t / H) * (H + 2 * pad) + h + pad) * (W + 2 * pad) + w + pad;
    }
    const int64_t g = row / rows_per_group;
    const float4 sc = *reinterpret_cast<const float4*>(scale + g * C + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + g * C + c);
    float4 v = Vec4<T>::load(x + row * C + c);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (res) {
      const float4 r = Vec4<T>::load(res + row * C + c);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    Vec4<T>::store(y + orow * C + c, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fcmf_bn_finalize + fcmf_bn_apply in one launch (the trunk's forward ran 155 BatchNorms per pass as statistics / reduce /
// finalize / apply: the one-thread-per-channel finalize kernel was 5 us of launch latency each).  A workgroup streams a
// contiguous row range of ONE group; a thread owns E = 16 / sizeof(T) fixed channels (C / E lanes per row, a power of two
// <= 256, so a lane's channels do not change as it strides the rows): it derives their scale / shift from the group's totals
// itself (same arithmetic as bn_finalize_kernel), then moves 16 B per access, four rows in flight.  The first workgroup of
// every group leaves mean / rstd for the backward; workgroup 0 applies the `groups` running-statistics updates in group order.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                                const double* __restrict__ tot, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int C, int logCE, int64_t rows_per_group,
                                                                int groups, int rows_per_block, int blocks_per_group, double count,
                                                                double inv_count, float momentum, float eps, int relu, int H, int W, int pad) {
  constexpr int E = 16 / (int)sizeof(T);
  typedef T VecT __attribute__((ext_vector_type(E)));
  const int tid = threadIdx.x, CE = 1 << logCE;
  const int g = blockIdx.x / blocks_per_group, bg = blockIdx.x - g * blocks_per_group;
  const int c0 = (tid & (CE - 1)) * E;
  // the workgroups of a group sweep it TOGETHER (vector i of the group belongs to workgroup (i / 256) % blocks_per_group): the
  // chip works inside one moving window per group instead of thousands of distant streams
  const int64_t r0 = (int64_t)g * rows_per_group;
  const int64_t n = rows_per_group << logCE;           // 16-byte vectors of the group: contiguous in x / res
  const int64_t S = (int64_t)blocks_per_group * 256;
  const VecT* xv = reinterpret_cast<const VecT*>(x + r0 * C);
  const VecT* rv4 = res ? reinterpret_cast<const VecT*>(res + r0 * C) : nullptr;
  // software pipeline, four vectors per thread and stage; the FIRST stage is requested before the scale / shift derivation
  // below, whose dependent chain (totals -> double arithmetic -> rsqrt) would otherwise sit in front of every workgroup's
  // first memory round trip (the tensors of layer3 are 4096 vectors per workgroup: the chain was a fifth of their time)
  VecT na[4], nb[4];
  auto load_stage = [&](int64_t i) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * S < n) { na[k] = xv[i + k * S]; if (rv4) nb[k] = rv4[i + k * S]; }
  };
  int64_t i = (int64_t)bg * 256 + tid;
  load_stage(i);

  float sc[E], sh[E];
  auto stats_of = [&](int gg, int c, float& mean, float& rs, double& var) {
    const double s = tot[((int64_t)gg * C + c) * 2], q = tot[((int64_t)gg * C + c) * 2 + 1];
    const double m = s * inv_count;        // (as bn_finalize_kernel: a double division per channel and thread would cost more than the rows)
    var = q * inv_count - m * m;
    var = var > 0 ? var : 0;
    mean = (float)m;
    rs = 1.0f / sqrtf((float)var + eps);
  };
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = c0 + e;
    float mean, rs;
    double var;
    if (tot) stats_of(g, c, mean, rs, var);
    else { mean = rmean[c]; rs = 1.0f / sqrtf(rvar[c] + eps); }
    sc[e] = gamma[c] * rs;
    sh[e] = beta[c] - mean * sc[e];
    if (bg == 0 && tid < CE && mean_out) { mean_out[(int64_t)g * C + c] = mean; rstd_out[(int64_t)g * C + c] = rs; }
  }
  if (tot && blockIdx.x == 0 && tid < CE) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = c0 + e;
      float rm = rmean[c], rv = rvar[c];
      for (int gg = 0; gg < groups; ++gg) {
        float mean, rs;
        double var;
        stats_of(gg, c, mean, rs, var);
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        rm = (1.f - momentum) * rm + momentum * mean;
        rv = (1.f - momentum) * rv + momentum * (float)unbiased;
      }
      rmean[c] = rm;
      rvar[c] = rv;
    }
  }
  auto out_ptr = [&](int64_t j) -> VecT* {
    int64_t row = r0 + (j >> logCE);
    if (pad > 0) {
      const unsigned rw = (unsigned)row;               // (rows < 2^31: checked by the host)
      const unsigned t = rw / (unsigned)W, w = rw - t * (unsigned)W;
      const unsigned nimg = t / (unsigned)H, h = t - nimg * (unsigned)H;
      row = ((int64_t)nimg * (H + 2 * pad) + h + pad) * (W + 2 * pad) + w + pad;
    }
    return reinterpret_cast<VecT*>(y + row * C + c0);
  };
  auto norm = [&](VecT v, VecT r, bool has_r) -> VecT {
    VecT o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float f = (float)v[e] * sc[e] + sh[e];
      if (has_r) f += (float)r[e];
      if (relu) f = fmaxf(f, 0.f);
      o[e] = (T)f;
    }
    return o;
  };
  for (; i < n; i += 4 * S) {
    VecT ca[4], cb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ca[k] = na[k]; cb[k] = nb[k]; }
    if (i + 4 * S < n) load_stage(i + 4 * S);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * S < n) *out_ptr(i + k * S) = norm(ca[k], cb[k], rv4 != nullptr);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                           int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const int64_t p = idx / C4;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho);
    const int64_t n = p / ((int64_t)Wo * Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 + r - 1;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 + s - 1;
        if (wi < 0 || wi >= W) continue;
        const float4 v = Vec4<T>::load(x + ((n * H + hi) * W + wi) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    Vec4<T>::store(y + p * C + c, m);
  }
}

// adaptive average pool (F.adaptive_avg_pool2d: window [floor(i*H/oh), ceil((i+1)*H/oh)) ); float32 output in
// NCHW ([N, C, oh, ow], layout = 0: what myResNetImg returns) or token-major ([N, oh*ow, C], layout = 1)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                                      int oh, int ow, int layout) {
  const int64_t total = (int64_t)N * oh * ow * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const int64_t p = idx / C;
    const int j = (int)(p % ow), i = (int)((p / ow) % oh);
    const int64_t n = p / ((int64_t)ow * oh);
    const int h0 = (i * H) / oh, h1 = ((i + 1) * H + oh - 1) / oh;
    const int w0 = (j * W) / ow, w1 = ((j + 1) * W + ow - 1) / ow;
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) s += to_f32<T>(x[((n * H + h) * W + w) * C + c]);
    s /= (float)((h1 - h0) * (w1 - w0));
    if (layout == 0) y[((n * C + c) * oh + i) * ow + j] = s;
    else y[idx] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (fine-tuning the CNN, resnet_utils.py if_fine_tune=True / --fine_tune_cnn)
// ---------------------------------------------------------------------------------------------------------------
// BatchNorm backward, pass 1: per (group, channel) s1 = sum gm, s2 = sum gm * xhat with gm = g * [z > 0] (the ReLU
// that follows the normalisation; z == NULL: no ReLU) and xhat = (y - mean) * rstd.  Same decomposition as bn_stats.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ g, const T* __restrict__ z, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           double* __restrict__ sums, int C, int64_t rows_per_group, int chunk) {
  __shared__ float4 red[2][256];
  const int slab_c = C < 256 ? C : 256, vpr = slab_c >> 2, rpp = 256 / vpr;
  const int cv = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int c0 = blockIdx.x * 256 + cv * 4;
  const int gi = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(rows_per_group, r0 + chunk);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  if (c0 < C) {
    const int64_t base = ((int64_t)gi * rows_per_group) * C + c0;
    const float4 mu = *reinterpret_cast<const float4*>(mean + (int64_t)gi * C + c0);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + (int64_t)gi * C + c0);
    for (int64_t r = r0 + rsub; r < r1; r += rpp) {
      float4 gv = Vec4<T>::load(g + base + r * C);
      const float4 yv = Vec4<T>::load(y + base + r * C);
      if (z) {
        const float4 zv = Vec4<T>::load(z + base + r * C);
        gv.x = zv.x > 0.f ? gv.x : 0.f; gv.y = zv.y > 0.f ? gv.y : 0.f; gv.z = zv.z > 0.f ? gv.z : 0.f; gv.w = zv.w > 0.f ? gv.w : 0.f;
      }
      s.x += gv.x; s.y += gv.y; s.z += gv.z; s.w += gv.w;
      q.x += gv.x * (yv.x - mu.x) * rs.x; q.y += gv.y * (yv.y - mu.y) * rs.y;
      q.z += gv.z * (yv.z - mu.z) * rs.z; q.w += gv.w * (yv.w - mu.w) * rs.w;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rsub == 0 && c0 < C) {
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    for (int k = 0; k < rpp; ++k) {
      const float4 a = red[0][k * vpr + cv], b = red[1][k * vpr + cv];
      ds[0] += a.x; ds[1] += a.y; ds[2] += a.z; ds[3] += a.w;
      dq[0] += b.x; dq[1] += b.y; dq[2] += b.z; dq[3] += b.w;
    }
    double* o = sums + (((int64_t)gi * gridDim.y + blockIdx.y) * C + c0) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = ds[e]; o[2 * e + 1] = dq[e]; }
  }
}

// dgamma[c] += sum_g s2[g,c], dbeta[c] += sum_g s1[g,c] from the totals
__global__ __launch_bounds__(256) void bn_bwd_params_kernel(const double* __restrict__ tot, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int C, int G) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0, b = 0;
  for (int g = 0; g < G; ++g) { b += tot[((int64_t)g * C + c) * 2]; a += tot[((int64_t)g * C + c) * 2 + 1]; }
  dgamma[c] += (float)a;
  dbeta[c] += (float)b;
}

// pass 2: dy = gamma * rstd * (gm - s1/n - xhat * s2/n) (training) or gamma * rstd * gm (eval: statistics are constants);
// gres (optional) <- gm, the gradient that flows on into the identity branch of a bottleneck.  dy / gres may alias g.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* g, const T* __restrict__ z, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           T* dy, T* gres, int64_t rows, int C, int64_t rows_per_group) {
  const int C4 = C >> 2;
  const double inv_n = 1.0 / (double)rows_per_group;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < rows * C4; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / C4;
    const int c = (int)(idx - row * C4) * 4;
    const int64_t gi = row / rows_per_group;
    float4 gv = Vec4<T>::load(g + row * C + c);
    if (z) {
      const float4 zv = Vec4<T>::load(z + row * C + c);
      gv.x = zv.x > 0.f ? gv.x : 0.f; gv.y = zv.y > 0.f ? gv.y : 0.f; gv.z = zv.z > 0.f ? gv.z : 0.f; gv.w = zv.w > 0.f ? gv.w : 0.f;
    }
    if (gres) Vec4<T>::store(gres + row * C + c, gv);
    const float4 yv = Vec4<T>::load(y + row * C + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + gi * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + gi * C + c);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    float4 o;
    if (sums) {
      const double* sp = sums + (gi * C + c) * 2;
      const float m1[4] = {(float)(sp[0] * inv_n), (float)(sp[2] * inv_n), (float)(sp[4] * inv_n), (float)(sp[6] * inv_n)};
      const float m2[4] = {(float)(sp[1] * inv_n), (float)(sp[3] * inv_n), (float)(sp[5] * inv_n), (float)(sp[7] * inv_n)};
      o.x = gm.x * rs.x * (gv.x - m1[0] - (yv.x - mu.x) * rs.x * m2[0]);
      o.y = gm.y * rs.y * (gv.y - m1[1] - (yv.y - mu.y) * rs.y * m2[1]);
      o.z = gm.z * rs.z * (gv.z - m1[2] - (yv.z - mu.z) * rs.z * m2[2]);
      o.w = gm.w * rs.w * (gv.w - m1[3] - (yv.w - mu.w) * rs.w * m2[3]);
    } else {
      o.x = gm.x * rs.x * gv.x; o.y = gm.y * rs.y * gv.y; o.z = gm.z * rs.z * gv.z; o.w = gm.w * rs.w * gv.w;
    }
    Vec4<T>::store(dy + row * C + c, o);
  }
}

// transpose of im2col without atomics: every INPUT pixel gathers the patch-matrix gradients of the windows that
// cover it.  dA [N*Ho*Wo, Kpad] (k = (r, s, c)) -> dX [N, H, W, C]; 4 channels per thread, f32 accumulation.
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dA, T* __restrict__ dX, int N, int H, int W, int C,
                                                     int kh, int kw, int stride, int pad, int Ho, int Wo, int Kpad) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const int64_t p = idx / C4;
    const int wi = (int)(p % W), hi = (int)((p / W) % H);
    const int64_t n = p / ((int64_t)W * H);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < kh; ++r) {
      const int th = hi + pad - r;
      if (th < 0 || th % stride != 0) continue;
      const int ho = th / stride;
      if (ho >= Ho) continue;
      for (int s2 = 0; s2 < kw; ++s2) {
        const int tw = wi + pad - s2;
        if (tw < 0 || tw % stride != 0) continue;
        const int wo = tw / stride;
        if (wo >= Wo) continue;
        const float4 v = Vec4<T>::load(dA + ((n * Ho + ho) * Wo + wo) * Kpad + (r * kw + s2) * C + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    Vec4<T>::store(dX + p * C + c, acc);
  }
}

// 3x3 / stride 2 / pad 1 max-pool backward, gather form: an input pixel receives dY of every window whose FIRST maximum
// (scan order r, then s: torch's tie rule) it is.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dY, T* __restrict__ dX,
                                                               int N, int H, int W, int C, int Ho, int Wo) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const int64_t p = idx / C;
    const int wi = (int)(p % W), hi = (int)((p / W) % H);
    const int64_t n = p / ((int64_t)W * H);
    float acc = 0.f;
    for (int ho = max(0, (hi - 1 + 1) / 2); ho <= min(Ho - 1, (hi + 1) / 2); ++ho)
      for (int wo = max(0, (wi - 1 + 1) / 2); wo <= min(Wo - 1, (wi + 1) / 2); ++wo) {
        // first maximum of window (ho, wo)
        float best = -INFINITY; int bh = -1, bw = -1;
        for (int r = 0; r < 3; ++r) {
          const int h2 = ho * 2 + r - 1;
          if (h2 < 0 || h2 >= H) continue;
          for (int s2 = 0; s2 < 3; ++s2) {
            const int w2 = wo * 2 + s2 - 1;
            if (w2 < 0 || w2 >= W) continue;
            const float v = to_f32<T>(x[((n * H + h2) * W + w2) * C + c]);
            if (v > best || bh < 0) { best = v; bh = h2; bw = w2; }
          }
        }
        if (bh == hi && bw == wi) acc += to_f32<T>(dY[((n * Ho + ho) * Wo + wo) * C + c]);
      }
    dX[idx] = from_f32<T>(acc);
  }
}

// adaptive average pool backward: dY float32 in the forward's output layout -> dX [N,H,W,C]
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dY, T* __restrict__ dX, int N, int H, int W, int C,
                                                          int oh, int ow, int layout) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const int64_t p = idx / C;
    const int w = (int)(p % W), h = (int)((p / W) % H);
    const int64_t n = p / ((int64_t)W * H);
    float acc = 0.f;
    for (int i = 0; i < oh; ++i) {
      const int h0 = (i * H) / oh, h1 = ((i + 1) * H + oh - 1) / oh;
      if (h < h0 || h >= h1) continue;
      for (int j = 0; j < ow; ++j) {
        const int w0 = (j * W) / ow, w1 = ((j + 1) * W + ow - 1) / ow;
        if (w < w0 || w >= w1) continue;
        const float d = layout == 0 ? dY[((n * C + c) * oh + i) * ow + j] : dY[((n * oh + i) * ow + j) * C + c];
        acc += d / (float)((h1 - h0) * (w1 - w0));
      }
    }
    dX[idx] = from_f32<T>(acc);
  }
}

static inline int grid_for(int64_t work) {
  const int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

extern "C" int fcmf_conv_im2col(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int H, int W, int C,
                                int64_t sn, int64_t sh, int64_t sw, int64_t sc, int kh, int kw, int stride, int pad,
                                int Kpad, void* stream) {
  if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
      Kpad < kh * kw * C)
    return FCMF_ERR_ARG;
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * Ho * Wo;
  const bool nhwc = sc == 1 && sw == C && sh == (int64_t)W * C && sn == (int64_t)H * W * C;
  const bool al16 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (nhwc && C % 8 == 0 && Kpad % 8 == 0 && src_dtype == dst_dtype && al16) {
    const int g = grid_for(rows * (Kpad / 8));
    if (src_dtype == FCMF_BF16)
      hipLaunchKernelGGL((im2col_nhwc8_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad);
    else if (src_dtype == FCMF_F32)
      hipLaunchKernelGGL((im2col_nhwc8_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, N, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad);
    else return FCMF_ERR_UNSUPPORTED;
    FCMF_CHECK_LAUNCH();
    return FCMF_OK;
  }
  const int g = grid_for(rows * Kpad);
#define FCMF_IM2COL(TS, TD) \
  hipLaunchKernelGGL((im2col_kernel<TS, TD>), dim3(g), dim3(256), 0, st, (const TS*)src, (TD*)dst, N, H, W, C, sn, sh, sw, sc, kh, kw, stride, pad, Ho, Wo, Kpad)
  if (src_dtype == FCMF_F32 && dst_dtype == FCMF_F32) FCMF_IM2COL(float, float);
  else if (src_dtype == FCMF_F32 && dst_dtype == FCMF_BF16) FCMF_IM2COL(float, bf16_t);
  else if (src_dtype == FCMF_BF16 && dst_dtype == FCMF_BF16) FCMF_IM2COL(bf16_t, bf16_t);
  else if (src_dtype == FCMF_BF16 && dst_dtype == FCMF_F32) FCMF_IM2COL(bf16_t, float);
  else return FCMF_ERR_UNSUPPORTED;
#undef FCMF_IM2COL
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_pack_rgb0(const void* src, int src_dtype, void* dst, int N, int H, int W, int64_t sn, int64_t sh, int64_t sw,
                              int64_t sc, int pad, int Wp, void* stream) {
  if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || pad < 0 || Wp < W + 2 * pad) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * H * W);
  if (src_dtype == FCMF_F32) hipLaunchKernelGGL((pack_rgb0_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, N, H, W, sn, sh, sw, sc, pad, Wp);
  else if (src_dtype == FCMF_F64) hipLaunchKernelGGL((pack_rgb0_kernel<double>), dim3(g), dim3(256), 0, st, (const double*)src, (bf16_t*)dst, N, H, W, sn, sh, sw, sc, pad, Wp);
  else if (src_dtype == FCMF_BF16) hipLaunchKernelGGL((pack_rgb0_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, H, W, sn, sh, sw, sc, pad, Wp);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

// row chunking of the statistics kernels: enough workgroups to fill the chip, at most 4096 rows per workgroup (a thread's
// float partial sum covers <= 1024 values; everything above is added in double).  Returns the number of chunks per group (= partial rows per (group, channel) in `sums`).
static int bn_chunking(int64_t rows_per_group, int groups, int C, int* chunk_rows) {
  const int slabs = (C + 255) / 256;
  int64_t chunks = (1024 + (int64_t)slabs * groups - 1) / ((int64_t)slabs * groups);
  int64_t chunk = (rows_per_group + chunks - 1) / chunks;
  if (chunk > 4096) chunk = 4096;
  if (chunk < 16) chunk = 16;
  chunks = (rows_per_group + chunk - 1) / chunk;
  if (chunk_rows) *chunk_rows = (int)chunk;
  return (int)chunks;
}
static bool bn_shape_ok(int C) { return C > 0 && C % 4 == 0 && (C >= 256 ? C % 256 == 0 : 256 % (C / 4) == 0); }

extern "C" int64_t fcmf_bn_stats_workspace(int64_t rows_per_group, int groups, int C) {
  if (rows_per_group <= 0 || groups <= 0 || !bn_shape_ok(C)) return 0;
  return ((int64_t)groups * bn_chunking(rows_per_group, groups, C, nullptr) + groups) * C * 2;      // doubles: partials + totals
}
// where the totals [G][C][2] live inside the workspace
static double* bn_totals(double* sums, int64_t rows_per_group, int groups, int C) {
  return sums + (int64_t)groups * bn_chunking(rows_per_group, groups, C, nullptr) * C * 2;
}

extern "C" int fcmf_bn_stats(const void* x, double* sums, int64_t rows_per_group, int groups, int C, int dtype,
                             void* stream) {
  if (!x || !sums || rows_per_group <= 0 || groups <= 0 || !bn_shape_ok(C)) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = (C + 255) / 256;
  int chunk;
  const int chunks = bn_chunking(rows_per_group, groups, C, &chunk);
  if (chunks > 65535 || groups > 65535) return FCMF_ERR_UNSUPPORTED;
  dim3 grid(slabs, (unsigned)chunks, groups);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((bn_stats_kernel<float>), grid, dim3(256), 0, st, (const float*)x, sums, C, rows_per_group, (int)chunk);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((bn_stats_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, sums, C, rows_per_group, (int)chunk);
  else return FCMF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(bn_reduce_chunks_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, sums,
                     bn_totals(sums, rows_per_group, groups, C), C, chunks);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

// block statistics of a colstats GEMM -> the per-group totals, where fcmf_bn_finalize expects them inside `sums` (a workspace of
// fcmf_bn_stats_workspace(rows_per_group, groups, C) doubles, as for fcmf_bn_stats).  Groups are whole multiples of block_rows
// (fcmf_gemm_colstats_block_rows of the producing GEMM).
extern "C" int fcmf_bn_stats_blocks(const float* blockstats, double* sums, int64_t rows_per_group, int groups, int C, int block_rows,
                                    void* stream) {
  if (!blockstats || !sums || rows_per_group <= 0 || groups <= 0 || block_rows <= 0 || !bn_shape_ok(C)) return FCMF_ERR_ARG;
  if (rows_per_group % block_rows != 0 || groups > 65535) return FCMF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(bn_reduce_blocks_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     blockstats, bn_totals(sums, rows_per_group, groups, C), C, (int)(rows_per_group / block_rows));
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                                int groups, int64_t count, float momentum, float eps, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0 || groups <= 0) return FCMF_ERR_ARG;
  if ((sums && count <= 0) || ((mean_out == nullptr) != (rstd_out == nullptr))) return FCMF_ERR_ARG;
  const double* tot = sums ? bn_totals(const_cast<double*>(sums), count, groups, C) : nullptr;   // where fcmf_bn_stats left them
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tot,
                     gamma, beta, running_mean, running_var, scale, shift, mean_out, rstd_out, C, groups, (double)count,
                     count > 0 ? 1.0 / (double)count : 0.0, momentum, eps);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

// fcmf_bn_finalize + fcmf_bn_apply(_pad) in ONE launch (see bn_finalize_apply_kernel).  sums as for fcmf_bn_finalize (NULL = eval:
// running statistics); groups * rows_per_group rows; pad = 0: y [rows, C] (may alias x), else y the zero-bordered NHWC buffer of
// fcmf_bn_apply_pad.  FCMF_ERR_UNSUPPORTED unless C / (16 / sizeof(T)) is a power of two <= 256 and the tensors are 16-byte
// aligned: the caller then uses the two separate entry points.
extern "C" int fcmf_bn_finalize_apply(const void* x, const void* res, void* y, const double* sums, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float* mean_out, float* rstd_out, int C, int groups,
                                      int64_t rows_per_group, float momentum, float eps, int relu, int H, int W, int pad, int dtype,
                                      void* stream) {
  if (!x || !y || !gamma || !beta || !running_mean || !running_var || C <= 0 || groups <= 0 || rows_per_group <= 0) return FCMF_ERR_ARG;
  if ((mean_out == nullptr) != (rstd_out == nullptr) || pad < 0) return FCMF_ERR_ARG;
  if (pad > 0 && (H <= 0 || W <= 0 || rows_per_group * groups % ((int64_t)H * W) != 0)) return FCMF_ERR_ARG;
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  const int E = dtype == FCMF_F32 ? 4 : 8;
  if (C % E != 0 || groups > 65535) return FCMF_ERR_UNSUPPORTED;
  const int CE = C / E;
  int logCE = 0;
  while ((1 << logCE) < CE) ++logCE;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((1 << logCE) != CE || CE > 256 || !al16(x) || !al16(y) || (res && !al16(res))) return FCMF_ERR_UNSUPPORTED;
  const double* tot = sums ? bn_totals(const_cast<double*>(sums), rows_per_group, groups, C) : nullptr;
  // >= 4096 vectors per workgroup (16 per thread) unless that leaves the chip empty; at most ~4096 workgroups
  const int64_t vec_per_group = rows_per_group * CE;
  int64_t bpg = (vec_per_group + 4095) / 4096;
  const int64_t cap = (4096 + groups - 1) / groups;
  if (bpg > cap) bpg = cap;
  if (bpg < 1) bpg = 1;
  const int64_t rpb = 0;      // (kept in the kernel's signature; the sweep is interleaved, not by row ranges)
  if (rows_per_group * groups >= (1ll << 31)) return FCMF_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(bpg * groups));
  if (dtype == FCMF_F32)
    hipLaunchKernelGGL((bn_finalize_apply_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (const float*)res, (float*)y, tot, gamma, beta,
                       running_mean, running_var, mean_out, rstd_out, C, logCE, rows_per_group, groups, (int)rpb, (int)bpg,
                       (double)rows_per_group, 1.0 / (double)rows_per_group, momentum, eps, relu, H, W, pad);
  else
    hipLaunchKernelGGL((bn_finalize_apply_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y, tot, gamma,
                       beta, running_mean, running_var, mean_out, rstd_out, C, logCE, rows_per_group, groups, (int)rpb, (int)bpg,
                       (double)rows_per_group, 1.0 / (double)rows_per_group, momentum, eps, relu, H, W, pad);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

static int bn_apply_impl(const void* x, const void* res, void* y, const float* scale, const float* shift, int64_t rows, int C,
                         int64_t rows_per_group, int relu, int H, int W, int pad, int dtype, void* stream);
extern "C" int fcmf_bn_apply(const void* x, const void* res, void* y, const float* scale, const float* shift,
                             int64_t rows, int C, int64_t rows_per_group, int relu, int dtype, void* stream) {
  return bn_apply_impl(x, res, y, scale, shift, rows, C, rows_per_group, relu, 1, 1, 0, dtype, stream);
}
extern "C" int fcmf_bn_apply_pad(const void* x, const void* res, void* y_padded, const float* scale, const float* shift,
                                 int64_t rows, int C, int64_t rows_per_group, int relu, int H, int W, int pad, int dtype,
                                 void* stream) {
  if (H <= 0 || W <= 0 || pad < 0 || rows % ((int64_t)H * W) != 0) return FCMF_ERR_ARG;
  return bn_apply_impl(x, res, y_padded, scale, shift, rows, C, rows_per_group, relu, H, W, pad, dtype, stream);
}
static int bn_apply_impl(const void* x, const void* res, void* y, const float* scale, const float* shift, int64_t rows, int C,
                         int64_t rows_per_group, int relu, int H, int W, int pad, int dtype, void* stream) {
  if (!x || !y || !scale || !shift || rows < 0 || C <= 0 || C % 4 != 0 || rows_per_group <= 0) return FCMF_ERR_ARG;
  if (rows == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(rows * (C / 4));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)res, (float*)y, scale, shift, rows, C, rows_per_group, relu, H, W, pad);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y, scale, shift, rows, C, rows_per_group, relu, H, W, pad);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0) return FCMF_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * Ho * Wo * (C / 4));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((maxpool3x3s2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((maxpool3x3s2_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_adaptive_avgpool(const void* x, float* y, int N, int H, int W, int C, int oh, int ow, int layout,
                                     int dtype, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || oh <= 0 || ow <= 0 || (layout != 0 && layout != 1)) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * oh * ow * C);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((avgpool_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, y, N, H, W, C, oh, ow, layout);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((avgpool_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, y, N, H, W, C, oh, ow, layout);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_bn_bwd(const void* g, const void* z, const void* y, const float* mean, const float* rstd, const float* gamma,
                           double* sums, void* dy, void* gres, float* dgamma, float* dbeta, int64_t rows_per_group, int groups,
                           int C, int training, int dtype, void* stream) {
  if (!g || !y || !mean || !rstd || !gamma || !dy || !dgamma || !dbeta || !sums || rows_per_group <= 0 || groups <= 0 ||
      !bn_shape_ok(C))
    return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = (C + 255) / 256;
  int chunk;
  const int chunks = bn_chunking(rows_per_group, groups, C, &chunk);
  if (chunks > 65535 || groups > 65535) return FCMF_ERR_UNSUPPORTED;
  dim3 grid(slabs, (unsigned)chunks, groups);
  double* tot = bn_totals(sums, rows_per_group, groups, C);     // totals [G][C][2] behind the chunk partials
  const int64_t rows = rows_per_group * groups;
  const int ga = grid_for(rows * (C / 4));
#define FCMF_BNB(T)                                                                                                              \
  do {                                                                                                                          \
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T>), grid, dim3(256), 0, st, (const T*)g, (const T*)z, (const T*)y, mean, rstd, sums, C, \
                       rows_per_group, chunk);                                                                                  \
    hipLaunchKernelGGL(bn_reduce_chunks_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, sums, tot, C, chunks);         \
    hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, st, tot, dgamma, dbeta, C, groups);          \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ga), dim3(256), 0, st, (const T*)g, (const T*)z, (const T*)y, mean, rstd,  \
                       gamma, training ? (const double*)tot : (const double*)nullptr, (T*)dy, (T*)gres, rows, C, rows_per_group); \
  } while (0)
  if (dtype == FCMF_F32) FCMF_BNB(float);
  else if (dtype == FCMF_BF16) FCMF_BNB(bf16_t);
  else return FCMF_ERR_UNSUPPORTED;
#undef FCMF_BNB
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_conv_col2im(const void* dA, void* dX, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                                int Kpad, int dtype, void* stream) {
  if (!dA || !dX || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
      Kpad < kh * kw * C)
    return FCMF_ERR_ARG;
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * H * W * (C / 4));
  if (dtype == FCMF_F32) hipLaunchKernelGGL((col2im_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dA, (float*)dX, N, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((col2im_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)dA, (bf16_t*)dX, N, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
  if (!x || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FCMF_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((maxpool3x3s2_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((maxpool3x3s2_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_adaptive_avgpool_bwd(const float* dy, void* dx, int N, int H, int W, int C, int oh, int ow, int layout,
                                         int dtype, void* stream) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || oh <= 0 || ow <= 0 || (layout != 0 && layout != 1)) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)N * H * W * C);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((avgpool_bwd_kernel<float>), dim3(g), dim3(256), 0, st, dy, (float*)dx, N, H, W, C, oh, ow, layout);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, dy, (bf16_t*)dx, N, H, W, C, oh, ow, layout);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
