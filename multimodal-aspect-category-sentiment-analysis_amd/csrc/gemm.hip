// GEMM kernels for gfx950.
//   gemm_bf16_kernel   : bf16 x bf16 -> f32 accumulate on v_mfma_f32_16x16x32_bf16, 128x128x64
//                        block tile, 4 waves (2x2), LDS double buffer, XOR-swizzled images,
//                        transposed operands consumed through ds_read_b64_tr_b16 (so dX = dY*W and
//                        dW = dY^T*X need no transposed copies), split-K with f32 atomics.
//   gemm_generic_kernel: any dtype / any stride, exact-f32 v_mfma_f32_16x16x4_f32.  Parity path
//                        (fp32 mode) and odd shapes (classifier N=4, box WG 64->8 ...).
#include "common.h"

struct GemmParams {
  const void* A; const void* B; void* C; const float* bias; void* aux;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int epilogue, accumulate, ksplit, ktiles_per_split;
};

__device__ __forceinline__ float apply_epilogue(float v, int epi, float auxv) {
  switch (epi) {
    case FCMF_EPI_GELU: return gelu_f(v);
    case FCMF_EPI_TANH: return tanhf(v);
    case FCMF_EPI_DGELU: return v * dgelu_f(auxv);
    case FCMF_EPI_DTANH: return v * (1.0f - auxv * auxv);
    default: return v;
  }
}

// =========================================================================================
// bf16 MFMA kernel
// =========================================================================================
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile (16 KiB) in either layout

// LDS image of a K-contiguous operand tile [128 rows][64 k]: 128-B rows, 16-B chunk index XORed
// with (row & 7) -> ds_read_b128 fragment reads are conflict free (checked per 16-lane group).
__device__ __forceinline__ int lds_off_rowmajor(int r, int kc) { return r * 128 + ((kc ^ (r & 7)) << 4); }
// LDS image of a transposed operand tile [64 k][128 x] (x contiguous, 256-B rows): the 32-B
// column pair index is XORed with key(k) = (k&3) | ((k>>3)&1)<<2 so that the 8 rows one
// 32-lane half touches in a ds_read_b64_tr_b16 land in 8 distinct 32-B slots.
__device__ __forceinline__ int tr_key(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }
__device__ __forceinline__ int lds_off_tr(int kk, int c16) {
  return kk * 256 + ((((c16 >> 1) ^ tr_key(kk))) << 5) + ((c16 & 1) << 4);
}

template <bool TR>
__device__ __forceinline__ void load_tile_global(const bf16_t* __restrict__ X, int64_t ld, int x0, int xdim,
                                                 int k0, int K, int tid, uint4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!TR) {
      int r = (tid >> 3) + 32 * i, kc = tid & 7;
      int gx = x0 + r, gk = k0 + kc * 8;
      if (gx < xdim && gk < K) v = *reinterpret_cast<const uint4*>(X + (int64_t)gx * ld + gk);
    } else {
      int kk = (tid >> 4) + 16 * i, xc = tid & 15;
      int gk = k0 + kk, gx = x0 + xc * 8;
      if (gk < K && gx < xdim) v = *reinterpret_cast<const uint4*>(X + (int64_t)gk * ld + gx);
    }
    regs[i] = v;
  }
}

template <bool TR>
__device__ __forceinline__ void store_tile_lds(char* lds, int tid, const uint4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int off;
    if (!TR) { int r = (tid >> 3) + 32 * i, kc = tid & 7; off = lds_off_rowmajor(r, kc); }
    else { int kk = (tid >> 4) + 16 * i, xc = tid & 15; off = lds_off_tr(kk, xc); }
    *reinterpret_cast<uint4*>(lds + off) = regs[i];
  }
}

// fragment for the 16 rows/cols [x0, x0+16) and k-step s (32 k) of the tile: 8 bf16 per lane,
// lane l holds x = x0 + (l&15), k = 32*s + 8*(l>>4) + j.
template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int x0, int s, int lane) {
  if (!TR) {
    int r = x0 + (lane & 15);
    int kc = s * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + lds_off_rowmajor(r, kc));
  } else {
    int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    int kk = s * 32 + 8 * g + q;
    int f = x0 >> 4;
    int off = kk * 256 + ((f ^ tr_key(kk)) << 5) + ((p >> 1) << 4) + ((p & 1) << 3);
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds + off));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds + off + 4 * 256));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

template <bool A_TR, bool B_TR, typename TC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile + B tile) = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks that share an XCD (bid % 8) walk consecutive logical ids, and
  // consecutive ids share the A row-panel (all N tiles of one M tile) -> panel re-reads hit L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int i0 = tile_m * BM, j0 = tile_n * BN;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * p.ktiles_per_split;
  const int kt_end = min(nk_total, kt_begin + p.ktiles_per_split);

  f32x4 acc[4][4];  // [j frag][i frag]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (kt_begin < kt_end) {
    load_tile_global<A_TR>(A, p.lda, i0, p.M, kt_begin * BK, p.K, tid, ra);
    load_tile_global<B_TR>(B, p.ldb, j0, p.N, kt_begin * BK, p.K, tid, rb);
    store_tile_lds<A_TR>(smem, tid, ra);
    store_tile_lds<B_TR>(smem + TILE_BYTES, tid, rb);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) {
      load_tile_global<A_TR>(A, p.lda, i0, p.M, (kt + 1) * BK, p.K, tid, ra);
      load_tile_global<B_TR>(B, p.ldb, j0, p.N, (kt + 1) * BK, p.K, tid, rb);
    }
    const char* la = smem + cur * 2 * TILE_BYTES;
    const char* lb = la + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fa[f] = read_frag<A_TR>(la, wm * 64 + f * 16, s, lane);
        fb[f] = read_frag<B_TR>(lb, wn * 64 + f * 16, s, lane);
      }
#pragma unroll
      for (int fj = 0; fj < 4; ++fj)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
          // D rows <- B operand (j), D cols <- A operand (i): each lane ends up with 4
          // consecutive j of one row i, i.e. a contiguous 8/16-byte piece of C.
          acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[fj], fa[fi], acc[fj][fi], 0, 0, 0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
      store_tile_lds<A_TR>(na, tid, ra);
      store_tile_lds<B_TR>(na + TILE_BYTES, tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue -----------------------------------------------------------------------
  TC* C = reinterpret_cast<TC*>(p.C);
  TC* AUX = reinterpret_cast<TC*>(p.aux);
  const bool atomic = (p.ksplit > 1);
  const bool lead = (blockIdx.z == 0);
#pragma unroll
  for (int fi = 0; fi < 4; ++fi) {
    const int i = i0 + wm * 64 + fi * 16 + (lane & 15);
    if (i >= p.M) continue;
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      const int j = j0 + wn * 64 + fj * 16 + (lane >> 4) * 4;
      if (j >= p.N) continue;
      float4 v = make_float4(acc[fj][fi][0], acc[fj][fi][1], acc[fj][fi][2], acc[fj][fi][3]);
      if (p.bias && lead) {
        float4 b = *reinterpret_cast<const float4*>(p.bias + j);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      const int64_t off = (int64_t)i * p.ldc + j;
      if (p.epilogue != FCMF_EPI_NONE) {
        float4 a = make_float4(0, 0, 0, 0);
        if (p.epilogue == FCMF_EPI_GELU) { if (AUX) Vec4<TC>::store(AUX + off, v); }
        else if (p.epilogue != FCMF_EPI_TANH) a = Vec4<TC>::load(AUX + off);
        v.x = apply_epilogue(v.x, p.epilogue, a.x); v.y = apply_epilogue(v.y, p.epilogue, a.y);
        v.z = apply_epilogue(v.z, p.epilogue, a.z); v.w = apply_epilogue(v.w, p.epilogue, a.w);
      }
      if constexpr (sizeof(TC) == 4) {
        float* cf = reinterpret_cast<float*>(C) + off;
        if (atomic) {
          atomicAdd(cf + 0, v.x); atomicAdd(cf + 1, v.y); atomicAdd(cf + 2, v.z); atomicAdd(cf + 3, v.w);
        } else if (p.accumulate) {
          float4 o = *reinterpret_cast<float4*>(cf);
          o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
          *reinterpret_cast<float4*>(cf) = o;
        } else {
          *reinterpret_cast<float4*>(cf) = v;
        }
      } else {
        Vec4<TC>::store(C + off, v);
      }
    }
  }
}

// =========================================================================================
// generic kernel: C = op(A) op(B) with arbitrary element strides, f32 MFMA (exact fmaf chains)
// =========================================================================================
struct GenericParams {
  const void* A; const void* B; void* C; const float* bias; void* aux;
  int M, N, K;
  int64_t a_si, a_sk, b_sj, b_sk, ldc;  // element strides: A(i,k) = A[i*a_si + k*a_sk]
  int epilogue, accumulate;
};

template <typename TI, typename TC>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GenericParams p) {
  constexpr int GT = 64, GK = 16, LDT = GT + 4;
  __shared__ float As[GK][LDT];
  __shared__ float Bs[GK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const TI* A = reinterpret_cast<const TI*>(p.A);
  const TI* B = reinterpret_cast<const TI*>(p.B);
  // thread -> (row, k) mapping follows the contiguous axis of each operand so that the global
  // reads of a wave are coalesced for both layouts.
  const bool a_kfast = (p.a_sk == 1), b_kfast = (p.b_sk == 1);

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += GK) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int e = tid + 256 * it;  // 0..1023
      int ia, ka, jb, kb;
      if (a_kfast) { ka = e & 15; ia = e >> 4; } else { ia = e & 63; ka = e >> 6; }
      if (b_kfast) { kb = e & 15; jb = e >> 4; } else { jb = e & 63; kb = e >> 6; }
      float va = 0.f, vb = 0.f;
      if (i0 + ia < p.M && k0 + ka < p.K) va = to_f32<TI>(A[(int64_t)(i0 + ia) * p.a_si + (int64_t)(k0 + ka) * p.a_sk]);
      if (j0 + jb < p.N && k0 + kb < p.K) vb = to_f32<TI>(B[(int64_t)(j0 + jb) * p.b_sj + (int64_t)(k0 + kb) * p.b_sk]);
      As[ka][ia] = va;
      Bs[kb][jb] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < GK / 4; ++k4) {
      float a[2], b[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        a[f] = As[k4 * 4 + (lane >> 4)][wm * 32 + f * 16 + (lane & 15)];
        b[f] = Bs[k4 * 4 + (lane >> 4)][wn * 32 + f * 16 + (lane & 15)];
      }
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
          acc[fi][fj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[fi], b[fj], acc[fi][fj], 0, 0, 0);
    }
    __syncthreads();
  }
  TC* C = reinterpret_cast<TC*>(p.C);
  TC* AUX = reinterpret_cast<TC*>(p.aux);
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = i0 + wm * 32 + fi * 16 + (lane >> 4) * 4 + r;
        int j = j0 + wn * 32 + fj * 16 + (lane & 15);
        if (i >= p.M || j >= p.N) continue;
        float v = acc[fi][fj][r];
        if (p.bias) v += p.bias[j];
        int64_t off = (int64_t)i * p.ldc + j;
        if (p.epilogue != FCMF_EPI_NONE) {
          float a = 0.f;
          if (p.epilogue == FCMF_EPI_GELU) { if (AUX) AUX[off] = from_f32<TC>(v); }
          else if (p.epilogue != FCMF_EPI_TANH) a = to_f32<TC>(AUX[off]);
          v = apply_epilogue(v, p.epilogue, a);
        }
        if (p.accumulate) v += to_f32<TC>(C[off]);
        C[off] = from_f32<TC>(v);
      }
}

// column sums ----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ out, int M, int N,
                                                     int64_t ldx, int rows_per_block) {
  // block (bx, by): columns [bx*64, +64), rows [by*rows_per_block, ...); 4 waves split the rows
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + wave; r < r1; r += 4) s += to_f32<T>(X[(int64_t)r * ldx + col]);
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) atomicAdd(out + col, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// =========================================================================================
// host dispatch
// =========================================================================================
template <bool A_TR, bool B_TR>
static int launch_bf16(const GemmParams& p, int out_dtype, dim3 grid, hipStream_t st) {
  size_t smem = 4 * TILE_BYTES;
  if (out_dtype == FCMF_F32) {
    auto k = gemm_bf16_kernel<A_TR, B_TR, float>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
  } else {
    auto k = gemm_bf16_kernel<A_TR, B_TR, bf16_t>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
  }
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_gemm(const void* A, const void* B, void* C, const float* bias, void* aux, int M, int N, int K,
                         int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b, int in_dtype,
                         int out_dtype, int epilogue, int accumulate, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return FCMF_ERR_ARG;
  if (M == 0 || N == 0) return FCMF_OK;
  if (accumulate && out_dtype != FCMF_F32) return FCMF_ERR_ARG;
  if ((epilogue == FCMF_EPI_DGELU || epilogue == FCMF_EPI_DTANH) && !aux) return FCMF_ERR_ARG;
  if (in_dtype != FCMF_F32 && in_dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  if (out_dtype != FCMF_F32 && out_dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // contiguous extents that the 16-byte loaders walk must be multiples of 8 elements
  const int a_contig = trans_a ? M : K, b_contig = trans_b ? N : K;
  const bool fast = in_dtype == FCMF_BF16 && al16(A) && al16(B) && al16(C) && (!aux || al16(aux)) &&
                    (!bias || al16(bias)) && (lda % 8 == 0) && (ldb % 8 == 0) && (a_contig % 8 == 0) &&
                    (b_contig % 8 == 0) && (N % 4 == 0) && (ldc % 4 == 0) && K > 0;
  if (fast) {
    GemmParams p{A, B, C, bias, aux, M, N, K, lda, ldb, ldc, epilogue, accumulate, 1, 0};
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int nk = (K + BK - 1) / BK;
    int ksplit = 1;
    // split K only where the output grid cannot fill the chip and C is an f32 accumulator
    // (weight gradients: K = number of tokens).
    if (accumulate && epilogue == FCMF_EPI_NONE && tiles < 512) {
      ksplit = (768 + tiles - 1) / tiles;
      if (ksplit > nk / 4) ksplit = nk / 4 > 0 ? nk / 4 : 1;
      if (ksplit > 32) ksplit = 32;
    }
    p.ksplit = ksplit;
    p.ktiles_per_split = (nk + ksplit - 1) / ksplit;
    p.ksplit = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    dim3 grid(tiles, 1, p.ksplit);
    if (!trans_a && !trans_b) return launch_bf16<false, false>(p, out_dtype, grid, st);
    if (!trans_a && trans_b) return launch_bf16<false, true>(p, out_dtype, grid, st);
    if (trans_a && !trans_b) return launch_bf16<true, false>(p, out_dtype, grid, st);
    return launch_bf16<true, true>(p, out_dtype, grid, st);
  }
  GenericParams g{A, B, C, bias, aux, M, N, K,
                  trans_a ? 1 : lda, trans_a ? lda : 1, trans_b ? 1 : ldb, trans_b ? ldb : 1, ldc,
                  epilogue, accumulate};
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  if (in_dtype == FCMF_F32 && out_dtype == FCMF_F32)
    hipLaunchKernelGGL((gemm_generic_kernel<float, float>), grid, dim3(256), 0, st, g);
  else if (in_dtype == FCMF_BF16 && out_dtype == FCMF_BF16)
    hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, g);
  else if (in_dtype == FCMF_BF16 && out_dtype == FCMF_F32)
    hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, float>), grid, dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_generic_kernel<float, bf16_t>), grid, dim3(256), 0, st, g);
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_colsum(const void* X, float* out, int M, int N, int64_t ldx, int dtype, int accumulate,
                           void* stream) {
  if (!X || !out || M < 0 || N <= 0) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!accumulate) { if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) return FCMF_ERR_LAUNCH; }
  if (M == 0) return FCMF_OK;
  int rpb = 512;
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, st, (const float*)X, out, M, N, ldx, rpb);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)X, out, M, N, ldx, rpb);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
