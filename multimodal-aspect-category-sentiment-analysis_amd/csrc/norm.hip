// Fused dropout + residual + LayerNorm (TF style: biased variance, eps inside the sqrt) and the
// RoBERTa embedding gather + LayerNorm.  One wave per row, 4 elements per lane per pass
// (8-byte bf16 / 16-byte f32 accesses), statistics in f32.  HBM-bound: per row the forward
// reads x (+res) and writes y (+z); nothing else is materialised.
#include "common.h"

constexpr int LN_MAXP = 8;  // up to 8 passes of 256 columns: H <= 2048 (NP template = passes)

template <typename T, int NP>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         int64_t res_stride, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         T* __restrict__ z, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int rows, int H, float eps,
                                                         float p, uint64_t seed, unsigned char* __restrict__ q8 = nullptr,
                                                         float* __restrict__ qscale = nullptr) {
  const int lane = threadIdx.x & 63;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  // grid-stride over rows with a one-row software pipeline: the next row's x (+res) is requested before the current
  // row's two wave reductions and stores
  typedef typename Vec4<T>::raw Raw;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  Raw nx[NP], nr[NP];
  auto fetch = [&](int r) {
    if (r < rows) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < H) {
          nx[i] = Vec4<T>::load_raw(x + (int64_t)r * H + c);
          if (res) nr[i] = Vec4<T>::load_raw(res + (int64_t)r * res_stride + c);
        }
      }
    }
  };
  fetch(row);
  for (; row < rows; row += stride) {
  Raw cx[NP], cr[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) { cx[i] = nx[i]; cr[i] = nr[i]; }
  fetch(row + stride);
  float4 v[NP];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float4 a = Vec4<T>::cvt(cx[i]);
      if (p > 0.f) {
        const uint64_t base = (uint64_t)row * H + c;
        float dm[4];
        dropout_mult4(seed, base, p, inv_keep, dm);
        a.x *= dm[0]; a.y *= dm[1]; a.z *= dm[2]; a.w *= dm[3];
      }
      if (res) {
        float4 r = Vec4<T>::cvt(cr[i]);
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
      }
      v[i] = a;
      s += a.x + a.y + a.z + a.w;
    }
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      if (z) Vec4<T>::store(z + (int64_t)row * H + c, v[i]);
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
      Vec4<T>::store(y + (int64_t)row * H + c, o);
      if (q8) {          // the values AS STORED (rounded to T): what fcmf_quant_fp8_rows would read back
        o.x = to_f32<T>(from_f32<T>(o.x)); o.y = to_f32<T>(from_f32<T>(o.y)); o.z = to_f32<T>(from_f32<T>(o.z)); o.w = to_f32<T>(from_f32<T>(o.w));
        v[i] = o;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      }
    }
  }
  if (q8) {              // e4m3 copy of the output row + its scale (the fp8 GEMM that consumes y: no separate quantisation pass)
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
    if (lane == 0) qscale[row] = sc;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        int w8 = 0;
        w8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, w8, false);
        w8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, w8, true);
        *reinterpret_cast<int*>(q8 + (int64_t)row * H + c) = w8;
      }
    }
  }
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, T* __restrict__ dz,
                                                         T* __restrict__ dx, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                         float* __restrict__ partial, int rows, int H, float p,
                                                         uint64_t seed, unsigned char* __restrict__ q8 = nullptr,
                                                         float* __restrict__ qscale = nullptr) {
  __shared__ float4 red[2][3][NP][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float4 ag[NP], ab[NP], gm[NP], ax[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); ax[i] = make_float4(0, 0, 0, 0);
    const int c = (i * 64 + lane) * 4;
    gm[i] = c < H ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0, 0, 0, 0);
  }
  // software pipeline over the wave's rows: the NEXT row's dy / z (and statistics) are requested before the current row is
  // reduced, so the two dependent wave reductions and the stores of a row overlap the next row's memory latency
  typedef typename Vec4<T>::raw Raw;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + w;
  Raw nd[NP], nz[NP];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int r) {
    if (r < rows) {
      nmu = mean[r]; nrs = rstd[r];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < H) { nd[i] = Vec4<T>::load_raw(dy + (int64_t)r * H + c); nz[i] = Vec4<T>::load_raw(z + (int64_t)r * H + c); }
      }
    }
  };
  fetch(row);
  for (; row < rows; row += stride) {
    const float mu = nmu, rs = nrs;
    Raw cd[NP], cz[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { cd[i] = nd[i]; cz[i] = nz[i]; }
    fetch(row + stride);
    float4 gy[NP], xh[NP];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const float4 d = Vec4<T>::cvt(cd[i]);
        const float4 zz = Vec4<T>::cvt(cz[i]);
        float4 h; h.x = (zz.x - mu) * rs; h.y = (zz.y - mu) * rs; h.z = (zz.z - mu) * rs; h.w = (zz.w - mu) * rs;
        ag[i].x += d.x * h.x; ag[i].y += d.y * h.y; ag[i].z += d.z * h.z; ag[i].w += d.w * h.w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        float4 g; g.x = d.x * gm[i].x; g.y = d.y * gm[i].y; g.z = d.z * gm[i].z; g.w = d.w * gm[i].w;
        gy[i] = g; xh[i] = h;
        s1 += g.x + g.y + g.z + g.w;
        s2 += g.x * h.x + g.y * h.y + g.z * h.z + g.w * h.w;
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        float4 o;
        o.x = rs * (gy[i].x - s1 - xh[i].x * s2); o.y = rs * (gy[i].y - s1 - xh[i].y * s2);
        o.z = rs * (gy[i].z - s1 - xh[i].z * s2); o.w = rs * (gy[i].w - s1 - xh[i].w * s2);
        Vec4<T>::store(dz + (int64_t)row * H + c, o);
        if (dx) {
          const uint64_t base = (uint64_t)row * H + c;
          float dm[4];
          dropout_mult4(seed, base, p, inv_keep, dm);
          o.x *= dm[0]; o.y *= dm[1]; o.z *= dm[2]; o.w *= dm[3];
          Vec4<T>::store(dx + (int64_t)row * H + c, o);
        }
        // column sums of the gradient that flows into the producing Linear = its bias gradient
        ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
        if (q8) {        // e4m3 copy of the gradient that flows into the producing Linear (dx with dropout, else dz), as stored
          o.x = to_f32<T>(from_f32<T>(o.x)); o.y = to_f32<T>(from_f32<T>(o.y)); o.z = to_f32<T>(from_f32<T>(o.z)); o.w = to_f32<T>(from_f32<T>(o.w));
          gy[i] = o;
        }
      }
    }
    if (q8) {
      float amax = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if ((i * 64 + lane) * 4 < H) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(gy[i].x), fabsf(gy[i].y)), fmaxf(fabsf(gy[i].z), fabsf(gy[i].w))));
      amax = wave_max(amax);
      const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
      if (lane == 0) qscale[row] = sc;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < H) {
          int w8 = 0;
          w8 = __builtin_amdgcn_cvt_pk_fp8_f32(gy[i].x * inv, gy[i].y * inv, w8, false);
          w8 = __builtin_amdgcn_cvt_pk_fp8_f32(gy[i].z * inv, gy[i].w * inv, w8, true);
          *reinterpret_cast<int*>(q8 + (int64_t)row * H + c) = w8;
        }
      }
    }
  }
  // cross-wave reduction: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds; atomics
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const int half = step == 0 ? 2 : 1;
    if (w >= half && w < 2 * half) {
#pragma unroll
      for (int i = 0; i < NP; ++i) { red[w - half][0][i][lane] = ag[i]; red[w - half][1][i][lane] = ab[i]; red[w - half][2][i][lane] = ax[i]; }
    }
    __syncthreads();
    if (w < half) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        float4 a = red[w][0][i][lane], b = red[w][1][i][lane], c2 = red[w][2][i][lane];
        ag[i].x += a.x; ag[i].y += a.y; ag[i].z += a.z; ag[i].w += a.w;
        ab[i].x += b.x; ab[i].y += b.y; ab[i].z += b.z; ab[i].w += b.w;
        ax[i].x += c2.x; ax[i].y += c2.y; ax[i].z += c2.z; ax[i].w += c2.w;
      }
    }
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        if (partial) {
          // per-block partial sums [block][3][H]: no atomics; ln_partial_reduce_kernel finishes the sum
          float* pb = partial + (int64_t)blockIdx.x * 3 * H + c;
          *reinterpret_cast<float4*>(pb) = ag[i];
          *reinterpret_cast<float4*>(pb + H) = ab[i];
          *reinterpret_cast<float4*>(pb + 2 * H) = ax[i];
        } else {
          atomicAdd(dgamma + c + 0, ag[i].x); atomicAdd(dgamma + c + 1, ag[i].y);
          atomicAdd(dgamma + c + 2, ag[i].z); atomicAdd(dgamma + c + 3, ag[i].w);
          atomicAdd(dbeta + c + 0, ab[i].x); atomicAdd(dbeta + c + 1, ab[i].y);
          atomicAdd(dbeta + c + 2, ab[i].z); atomicAdd(dbeta + c + 3, ab[i].w);
          if (dxsum) {
            atomicAdd(dxsum + c + 0, ax[i].x); atomicAdd(dxsum + c + 1, ax[i].y);
            atomicAdd(dxsum + c + 2, ax[i].z); atomicAdd(dxsum + c + 3, ax[i].w);
          }
        }
      }
    }
  }
}

// out[which][c] += sum_b partial[b][which][c].  grid (H/64, 3, 8): each workgroup sums one eighth of the
// per-workgroup partial rows for 64 columns (4 waves stride the rows), then one float atomic per column.
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partial, int nblocks, int H,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dxsum) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane, which = blockIdx.y;
  float* out = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
  if (!out) return;
  const int per = (nblocks + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  if (col < H)
    for (int b = b0 + w; b < b1; b += 4) s += partial[((int64_t)b * 3 + which) * H + col];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < H) atomicAdd(out + col, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// ---- RoBERTa embeddings -----------------------------------------------------------------
__global__ __launch_bounds__(64) void position_ids_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ pos,
                                                          int S, int pad_id) {
  // one wave per sequence: running count of non-pad tokens via ballot prefix
  const int lane = threadIdx.x;
  const int64_t* row = ids + (int64_t)blockIdx.x * S;
  int64_t* out = pos + (int64_t)blockIdx.x * S;
  int running = 0;
  for (int t0 = 0; t0 < S; t0 += 64) {
    const int t = t0 + lane;
    const bool nz = t < S && row[t] != pad_id;
    const unsigned long long b = __ballot(nz);
    const int incl = __popcll(b & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1)));
    if (t < S) out[t] = nz ? (int64_t)(running + incl + pad_id) : (int64_t)pad_id;
    running += __popcll(b);
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos,
                                                           const int64_t* __restrict__ tt, const float* __restrict__ word,
                                                           const float* __restrict__ ptab, const float* __restrict__ ttab,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ y, T* __restrict__ z, float* __restrict__ mean,
                                                           float* __restrict__ rstd, int ntok, int H, float eps, float p,
                                                           uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ntok) return;
  const float* wr = word + ids[row] * H;
  const float* pr = ptab + pos[row] * H;
  const float* tr = ttab + (tt ? tt[row] : 0) * H;
  float4 v[NP];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const float4 a = *reinterpret_cast<const float4*>(wr + c);
      const float4 b = *reinterpret_cast<const float4*>(tr + c);
      const float4 d = *reinterpret_cast<const float4*>(pr + c);
      // same association as HF: (word + type) + position
      float4 o; o.x = (a.x + b.x) + d.x; o.y = (a.y + b.y) + d.y; o.z = (a.z + b.z) + d.z; o.w = (a.w + b.w) + d.w;
      v[i] = o;
      s += o.x + o.y + o.z + o.w;
    }
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      if (z) Vec4<T>::store(z + (int64_t)row * H + c, v[i]);
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
      if (p > 0.f) {  // HF applies dropout AFTER the embedding LayerNorm
        const uint64_t base = (uint64_t)row * H + c;
        float dm[4];
        dropout_mult4(seed, base, p, inv_keep, dm);
        o.x *= dm[0]; o.y *= dm[1]; o.z *= dm[2]; o.w *= dm[3];
      }
      Vec4<T>::store(y + (int64_t)row * H + c, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dz, const int64_t* __restrict__ ids,
                                                        const int64_t* __restrict__ pos, const int64_t* __restrict__ tt,
                                                        float* __restrict__ dword, float* __restrict__ dpos,
                                                        float* __restrict__ dtt, int ntok, int H, int pad_id) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ntok) return;
  const int64_t id = ids[row], ps = pos[row], ty = tt ? tt[row] : 0;
  // one column per lane and trip: every atomic wave-instruction covers 256 CONTIGUOUS bytes of the destination row -- the shape the
  // memory-side float atomics run at full rate for (MI355X_MICROARCH.md, Global float atomics).  Four consecutive columns per lane
  // (round 3) spread each instruction over a 1-KiB span: 327 us for the 49152 x 768 word-table gradient against ~120 us of atomic bandwidth.
  for (int c = lane; c < H; c += 64) {
    const float d = to_f32<T>(dz[(int64_t)row * H + c]);
    if (id != pad_id) atomicAdd(dword + id * H + c, d);
    if (dpos && ps != pad_id) atomicAdd(dpos + ps * H + c, d);
    if (dtt && ty != 0) atomicAdd(dtt + ty * H + c, d);  // type row 0 is summed by a column reduction (every token hits it)
  }
}

// type-row-0 gradient: column sum over the tokens whose type id is 0
template <typename T>
__global__ __launch_bounds__(256) void embed_type0_kernel(const T* __restrict__ dz, const int64_t* __restrict__ tt,
                                                          float* __restrict__ dtt, int ntok, int H, int rows_per_block) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(ntok, r0 + rows_per_block);
  float s = 0.f;
  if (col < H)
    for (int r = r0 + w; r < r1; r += 4)
      if (!tt || tt[r] == 0) s += to_f32<T>(dz[(int64_t)r * H + col]);
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < H) atomicAdd(dtt + col, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// position-table gradient, one workgroup per (sequence offset s, 256 columns): the tokens at offset s of all
// sequences nearly always share one position id (RoBERTa ids count the non-pad prefix), so their rows are summed in
// registers (64 sequences per workgroup) and leave as ONE atomic per column; tokens with another id fall back to
// their own atomics.  (The
// per-token scatter of embed_bwd_kernel makes every sequence hit the same <= S rows: 384-way contention at B=64.)
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const T* __restrict__ dz, const int64_t* __restrict__ pos,
                                                            float* __restrict__ dpos, int nseq, int S, int H, int pad_id) {
  const int s = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= H) return;
  int64_t p0 = -1;
  float acc = 0.f;
  const int g0 = blockIdx.z * 64, g1 = min(nseq, g0 + 64);      // 64 sequences per workgroup
#pragma unroll 8
  for (int g = g0; g < g1; ++g) {
    const int64_t row = (int64_t)g * S + s, id = pos[row];
    if (id == pad_id) continue;
    const float d = to_f32<T>(dz[row * H + c]);
    if (p0 < 0) p0 = id;
    if (id == p0) acc += d;
    else atomicAdd(dpos + id * H + c, d);
  }
  if (p0 >= 0) atomicAdd(dpos + p0 * H + c, acc);
}

// position-table AND type-table gradients of tokens laid out [nseq, S], in ONE pass over dz with 16-byte loads (the two kernels
// above read it a second and a third time, two bytes per lane: 48 + 77 us against 15 us of traffic on the step).
// block = (sequence offset s, 64 sequences); thread = (16 bytes of columns, slot of sequences).  Per thread: the rows of its
// sequences that share the first position id seen are summed in registers (as in embed_pos_bwd_kernel), type-0 rows likewise;
// anything else leaves through its own atomics.  The slots' sums meet in LDS, and one column per lane goes out per atomic
// instruction (256 contiguous bytes).
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_type_bwd_kernel(const T* __restrict__ dz, const int64_t* __restrict__ pos,
                                                                 const int64_t* __restrict__ tt, float* __restrict__ dpos,
                                                                 float* __restrict__ dtt, int nseq, int S, int H, int pad_id) {
  constexpr int E = 16 / (int)sizeof(T);
  typedef T VecT __attribute__((ext_vector_type(E)));
  extern __shared__ __attribute__((aligned(16))) float red[];      // [2][slots][H] + slots position ids
  const int CE = H / E, slots = 256 / CE;                  // (host: H % E == 0, CE <= 256)
  const int tid = threadIdx.x, cg = tid % CE, slot = tid / CE, c0 = cg * E;
  const int s = blockIdx.x, g0 = blockIdx.y * 64, g1 = min(nseq, g0 + 64);
  float* rp = red;                                         // position sums [slots][H]
  float* rt = red + slots * H;                             // type-0 sums   [slots][H]
  int* p0s = reinterpret_cast<int*>(red + 2 * slots * H);
  if (slot < slots) {
    float acc[E], tsum[E];
#pragma unroll
    for (int j = 0; j < E; ++j) acc[j] = tsum[j] = 0.f;
    int64_t p0 = -1;
#pragma unroll 4
    for (int g = g0 + slot; g < g1; g += slots) {
      const int64_t row = (int64_t)g * S + s, id = pos[row], ty = tt ? tt[row] : 0;
      const VecT v = *reinterpret_cast<const VecT*>(dz + row * H + c0);
      if (dtt) {
        if (ty == 0) {
#pragma unroll
          for (int j = 0; j < E; ++j) tsum[j] += (float)v[j];
        } else {
#pragma unroll
          for (int j = 0; j < E; ++j) atomicAdd(dtt + ty * H + c0 + j, (float)v[j]);
        }
      }
      if (id == pad_id) continue;
      if (p0 < 0) p0 = id;
      if (id == p0) {
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] += (float)v[j];
      } else {
#pragma unroll
        for (int j = 0; j < E; ++j) atomicAdd(dpos + id * H + c0 + j, (float)v[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) { rp[slot * H + c0 + j] = acc[j]; rt[slot * H + c0 + j] = tsum[j]; }
    if (cg == 0) p0s[slot] = (int)p0;
  }
  __syncthreads();
  for (int c = tid; c < H; c += 256) {
    float ts = 0.f;
    for (int k = 0; k < slots; ++k) ts += rt[k * H + c];
    if (dtt) atomicAdd(dtt + c, ts);
    for (int k = 0; k < slots; ++k) {                      // slots that saw the same first id go out together
      const int pk = p0s[k];
      if (pk < 0) continue;
      bool first = true;
      for (int m = 0; m < k; ++m) first = first && p0s[m] != pk;
      if (!first) continue;
      float ps = 0.f;
      for (int m = k; m < slots; ++m)
        if (p0s[m] == pk) ps += rp[m * H + c];
      atomicAdd(dpos + (int64_t)pk * H + c, ps);
    }
  }
}

// pick the smallest pass count NP (256 columns per pass) that covers H
#define LN_DISPATCH_T(T, H)                         \
  do {                                              \
    const int np__ = ((H) + 255) / 256;             \
    if (np__ <= 1) LAUNCH_(T, 1);                   \
    else if (np__ <= 2) LAUNCH_(T, 2);              \
    else if (np__ <= 3) LAUNCH_(T, 3);              \
    else if (np__ <= 4) LAUNCH_(T, 4);              \
    else LAUNCH_(T, 8);                             \
  } while (0)
#define LN_DISPATCH(dtype, H)                                     \
  do {                                                            \
    if ((dtype) == FCMF_F32) LN_DISPATCH_T(float, H);             \
    else LN_DISPATCH_T(bf16_t, H);                                \
  } while (0)

// ---- host ---------------------------------------------------------------------------------
static int add_ln_fwd_impl(const void* x, const void* res, int64_t res_stride, const float* gamma, const float* beta, void* y,
                           void* z, float* mean, float* rstd, int rows, int H, float eps, float dropout_p, uint64_t seed,
                           int dtype, void* stream, void* q8, float* qscale);
extern "C" int fcmf_add_ln_fwd(const void* x, const void* res, int64_t res_stride, const float* gamma,
                               const float* beta, void* y, void* z, float* mean, float* rstd, int rows, int H,
                               float eps, float dropout_p, uint64_t seed, int dtype, void* stream) {
  return add_ln_fwd_impl(x, res, res_stride, gamma, beta, y, z, mean, rstd, rows, H, eps, dropout_p, seed, dtype, stream, nullptr, nullptr);
}
extern "C" int fcmf_add_ln_fwd_fp8(const void* x, const void* res, int64_t res_stride, const float* gamma,
                                   const float* beta, void* y, void* z, float* mean, float* rstd, int rows, int H,
                                   float eps, float dropout_p, uint64_t seed, int dtype, void* q8, float* qscale, void* stream) {
  if (!q8 || !qscale) return FCMF_ERR_ARG;
  return add_ln_fwd_impl(x, res, res_stride, gamma, beta, y, z, mean, rstd, rows, H, eps, dropout_p, seed, dtype, stream, q8, qscale);
}
static int add_ln_fwd_impl(const void* x, const void* res, int64_t res_stride, const float* gamma, const float* beta, void* y,
                           void* z, float* mean, float* rstd, int rows, int H, float eps, float dropout_p, uint64_t seed,
                           int dtype, void* stream, void* q8, float* qscale) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0 || H <= 0) return FCMF_ERR_ARG;
  if (H % 4 != 0 || H > LN_MAXP * 256 || (res && res_stride % 4 != 0)) return FCMF_ERR_UNSUPPORTED;
  if (rows == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = (rows + 3) / 4;
  dim3 grid(nblk > 2048 ? 2048 : nblk);      // 8 workgroups per CU; every wave walks its rows with the next one in flight
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
#define LAUNCH_(T, NP) hipLaunchKernelGGL((add_ln_fwd_kernel<T, NP>), grid, dim3(256), 0, st, (const T*)x, (const T*)res, \
    res_stride, gamma, beta, (T*)y, (T*)z, mean, rstd, rows, H, eps, dropout_p, seed, (unsigned char*)q8, qscale)
  LN_DISPATCH(dtype, H);
#undef LAUNCH_
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

static int ln_bwd_blocks(int rows) {
  int blocks = (rows + 3) / 4;
  return blocks > 1024 ? 1024 : blocks;   // 4 workgroups per CU keep enough rows in flight; fewer partial rows for the reduce pass
}

extern "C" int64_t fcmf_add_ln_bwd_workspace(int rows, int H) { return (int64_t)ln_bwd_blocks(rows) * 3 * H; }

static int add_ln_bwd_impl(const void* dy, const void* z, const float* gamma, const float* mean, const float* rstd, void* dz,
                           void* dx, float* dgamma, float* dbeta, float* dxsum, float* workspace, int rows, int H,
                           float dropout_p, uint64_t seed, int dtype, void* stream, void* q8, float* qscale);
extern "C" int fcmf_add_ln_bwd(const void* dy, const void* z, const float* gamma, const float* mean, const float* rstd,
                               void* dz, void* dx, float* dgamma, float* dbeta, float* dxsum, float* workspace,
                               int rows, int H, float dropout_p, uint64_t seed, int dtype, void* stream) {
  return add_ln_bwd_impl(dy, z, gamma, mean, rstd, dz, dx, dgamma, dbeta, dxsum, workspace, rows, H, dropout_p, seed, dtype, stream, nullptr, nullptr);
}
extern "C" int fcmf_add_ln_bwd_fp8(const void* dy, const void* z, const float* gamma, const float* mean, const float* rstd,
                                   void* dz, void* dx, float* dgamma, float* dbeta, float* dxsum, float* workspace,
                                   int rows, int H, float dropout_p, uint64_t seed, int dtype, void* q8, float* qscale, void* stream) {
  if (!q8 || !qscale) return FCMF_ERR_ARG;
  return add_ln_bwd_impl(dy, z, gamma, mean, rstd, dz, dx, dgamma, dbeta, dxsum, workspace, rows, H, dropout_p, seed, dtype, stream, q8, qscale);
}
static int add_ln_bwd_impl(const void* dy, const void* z, const float* gamma, const float* mean, const float* rstd, void* dz,
                           void* dx, float* dgamma, float* dbeta, float* dxsum, float* workspace, int rows, int H,
                           float dropout_p, uint64_t seed, int dtype, void* stream, void* q8, float* qscale) {
  if (!dy || !z || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta || rows < 0 || H <= 0) return FCMF_ERR_ARG;
  if (H % 4 != 0 || H > LN_MAXP * 256) return FCMF_ERR_UNSUPPORTED;
  if (dropout_p > 0.f && !dx) return FCMF_ERR_ARG;
  if (rows == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int blocks = ln_bwd_blocks(rows);
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
#define LAUNCH_(T, NP) hipLaunchKernelGGL((add_ln_bwd_kernel<T, NP>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)z, \
    gamma, mean, rstd, (T*)dz, (T*)(dropout_p > 0.f ? dx : nullptr), dgamma, dbeta, dxsum, workspace, rows, H, dropout_p, seed, \
    (unsigned char*)q8, qscale)
  LN_DISPATCH(dtype, H);
#undef LAUNCH_
  if (workspace)
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((H + 63) / 64, 3, 8), dim3(256), 0, st, workspace, blocks, H, dgamma, dbeta, dxsum);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_position_ids(const int64_t* ids, int64_t* pos, int nseq, int S, int pad_id, void* stream) {
  if (!ids || !pos || nseq < 0 || S <= 0) return FCMF_ERR_ARG;
  if (nseq == 0) return FCMF_OK;
  hipLaunchKernelGGL(position_ids_kernel, dim3(nseq), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ids, pos, S, pad_id);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_embed_ln_fwd(const int64_t* ids, const int64_t* pos, const int64_t* type_ids, const float* word,
                                 const float* pos_table, const float* type_table, const float* gamma, const float* beta,
                                 void* y, void* z, float* mean, float* rstd, int ntok, int H, float eps, float dropout_p,
                                 uint64_t seed, int dtype, void* stream) {
  if (!ids || !pos || !word || !pos_table || !type_table || !gamma || !beta || !y || !mean || !rstd) return FCMF_ERR_ARG;
  if (H % 4 != 0 || H > LN_MAXP * 256) return FCMF_ERR_UNSUPPORTED;
  if (ntok <= 0) return ntok == 0 ? FCMF_OK : FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((ntok + 3) / 4);
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
#define LAUNCH_(T, NP) hipLaunchKernelGGL((embed_ln_fwd_kernel<T, NP>), grid, dim3(256), 0, st, ids, pos, type_ids, word, \
    pos_table, type_table, gamma, beta, (T*)y, (T*)z, mean, rstd, ntok, H, eps, dropout_p, seed)
  LN_DISPATCH(dtype, H);
#undef LAUNCH_
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_embed_bwd(const void* dz, const int64_t* ids, const int64_t* pos, const int64_t* type_ids, float* dword,
                              float* dpos, float* dtype_table, int ntok, int H, int pad_id, int dtype, void* stream) {
  if (!dz || !ids || !pos || !dword) return FCMF_ERR_ARG;     // dpos may be NULL: see fcmf_embed_pos_bwd
  if (H % 4 != 0) return FCMF_ERR_UNSUPPORTED;
  if (ntok <= 0) return ntok == 0 ? FCMF_OK : FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((ntok + 3) / 4);
  const int rpb = 512;
  dim3 g2((H + 63) / 64, (ntok + rpb - 1) / rpb);
  if (dtype == FCMF_F32) {
    hipLaunchKernelGGL((embed_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dz, ids, pos, type_ids, dword, dpos, dtype_table, ntok, H, pad_id);
    if (dtype_table) hipLaunchKernelGGL((embed_type0_kernel<float>), g2, dim3(256), 0, st, (const float*)dz, type_ids, dtype_table, ntok, H, rpb);
  } else if (dtype == FCMF_BF16) {
    hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dz, ids, pos, type_ids, dword, dpos, dtype_table, ntok, H, pad_id);
    if (dtype_table) hipLaunchKernelGGL((embed_type0_kernel<bf16_t>), g2, dim3(256), 0, st, (const bf16_t*)dz, type_ids, dtype_table, ntok, H, rpb);
  } else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_embed_pos_type_bwd(const void* dz, const int64_t* pos, const int64_t* type_ids, float* dpos, float* dtype_table,
                                       int nseq, int S, int H, int pad_id, int dtype, void* stream) {
  if (!dz || !pos || !dpos || nseq < 0 || S <= 0 || H <= 0) return FCMF_ERR_ARG;
  if (nseq == 0) return FCMF_OK;
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  const int E = dtype == FCMF_F32 ? 4 : 8;
  if (H % E != 0 || H / E > 256 || (reinterpret_cast<uintptr_t>(dz) & 15)) return FCMF_ERR_UNSUPPORTED;   // callers fall back to the two separate passes
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slots = 256 / (H / E);
  const size_t smem = sizeof(float) * 2 * (size_t)slots * H + sizeof(int) * slots;
  dim3 grid(S, (nseq + 63) / 64);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((embed_pos_type_bwd_kernel<float>), grid, dim3(256), smem, st, (const float*)dz, pos, type_ids, dpos, dtype_table, nseq, S, H, pad_id);
  else hipLaunchKernelGGL((embed_pos_type_bwd_kernel<bf16_t>), grid, dim3(256), smem, st, (const bf16_t*)dz, pos, type_ids, dpos, dtype_table, nseq, S, H, pad_id);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_embed_pos_bwd(const void* dz, const int64_t* pos, float* dpos, int nseq, int S, int H, int pad_id,
                                  int dtype, void* stream) {
  if (!dz || !pos || !dpos || nseq < 0 || S <= 0 || H <= 0) return FCMF_ERR_ARG;
  if (nseq == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(S, (H + 255) / 256, (nseq + 63) / 64);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((embed_pos_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dz, pos, dpos, nseq, S, H, pad_id);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((embed_pos_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dz, pos, dpos, nseq, S, H, pad_id);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
