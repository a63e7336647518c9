// GEMM kernels for gfx950: bf16 x bf16 -> f32 accumulate on v_mfma_f32_16x16x32_bf16, operands row-major in either
// orientation (transposed operands are consumed through ds_read_b64_tr_b16, so dX = dY*W and dW = dY^T*X need no
// transposed copies).
//   gemm_bf16_tile256_kernel / gemm_bf16_tile192_kernel: persistent 256x256 / 192x256 ping-pong kernels for the big
//                        GEMMs of the step (description further down);
//   gemm_bf16_kernel   : 128x128x32 block tile, 4 waves (2x2), 4-stage LDS-DMA ring, two workgroups per CU, split-K
//                        with f32 atomics -- small / ragged / narrow outputs and tanh epilogues;
//   gemm_generic_kernel: any dtype / any stride, exact-f32 v_mfma_f32_16x16x4_f32.  Parity path (fp32 mode) and odd
//                        shapes (classifier N=4, box WG 64->8 ...).
#include "common.h"
#include <cstdio>

struct GemmParams {
  const void* A; const void* B; void* C; const float* bias; void* aux;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int epilogue, accumulate, ksplit, ktiles_per_split;
  unsigned a_bytes, b_bytes;   // extents of A / B for the buffer range check
  unsigned c_bytes;            // extent of C (and aux) in bytes, tile kernels with 2-byte outputs
  float* colsum;               // optional: colsum[n] += sum_m C[m,n] (bias gradient of the producing layer)
  int tiles, total_items;      // persistent big kernel: output tiles, tiles x k-splits
  float* ws;                   // split-K partial tiles [ksplit][M][N] (plain stores + a reduce pass) or null (float atomics)
  int nt_out;                  // bf16 epilogue: nontemporal stores (streamed outputs must not evict the operands from L2)
  const float* sa; const float* sb;   // fp8 kernels: per-row dequantisation scales of A [M] and of B [N]
  // implicit-GEMM convolution (cv_C > 0): A is an NHWC activation [n, Hp, Wp, C] (zero border included where the convolution
  // pads), row m = output pixel (n, oy, ox), k = (ky, kx, c).  C is a power of two and a multiple of the k-tile depth.
  int cv_C, cv_logC, cv_Hp, cv_Wp, cv_Ho, cv_Wo, cv_kw, cv_inv_kw, cv_stride;
  // optional (256-row bf16 kernels, no aux operand): colstats[b][n] = (sum, sum of squares) over the rows 128 b .. 128 b + 127 of
  // column n of C as stored -- the BatchNorm statistics of a convolution's output without a pass over it (plain stores)
  float* colstats;
  // log2 of the element distance between neighbouring pixels of the convolution's input (= cv_logC for an NHWC activation
  // whose k-tiles walk the channels of one tap; smaller for fcmf_conv_gemm_runs, whose "tap" is a run of several whole pixels)
  int cv_logP;
  // batched weight gradients (gemm_bf16_dw_batched_kernel): the `tiles` output tiles cover `tiles / tiles_per_mat` same-shape
  // matrices, tile t belongs to matrix t / tiles_per_mat whose operand / output pointers come from the BatchPtrs argument
  int tiles_per_mat;
};
constexpr int BATCH_MAX = 32;      // matrices per launch (the three pointer tables travel as a kernel argument: 768 bytes)
struct BatchPtrs { const void* A[BATCH_MAX]; const void* B[BATCH_MAX]; void* C[BATCH_MAX]; };

// element offset of the receptive-field origin of output row `row` in the (padded) input
__device__ __forceinline__ int64_t conv_row_base(const GemmParams& p, int row) {
  const int hw = p.cv_Ho * p.cv_Wo, n = row / hw, rem = row - n * hw, oy = rem / p.cv_Wo, ox = rem - oy * p.cv_Wo;
  return (((int64_t)n * p.cv_Hp + oy * p.cv_stride) * p.cv_Wp + ox * p.cv_stride) << p.cv_logP;
}
// byte offset (wave-uniform: the DMA's scalar offset) of the k-tile that starts at contraction index kk: tap (ky, kx), channel c0
__device__ __forceinline__ unsigned conv_k_offset(const GemmParams& p, int kk) {
  const int tap = kk >> p.cv_logC, c0 = kk & (p.cv_C - 1);
  const int ky = (tap * p.cv_inv_kw) >> 16, kx = tap - ky * p.cv_kw;
  return (unsigned)((((ky * p.cv_Wp + kx) << p.cv_logP) + c0) * 2);
}

// bf16-output epilogues use odd polynomials instead of erf/exp (no transcendental issue slots, no
// selects): with xc = clamp(x, -X, X),
//   Phi(x)   ~= 0.5 + xc * P7(xc^2), X = 4.0  (|err| <= 3.4e-5; gelu = x * Phi: |err| <= 1.4e-4)
//   gelu'(x) ~= 0.5 + xc * Q9(xc^2), X = 4.5  (|err| <= 2e-4)
// both constrained to hit exactly 1 (0) at +X (-X), so the clamp alone saturates them; all far below
// bf16 resolution.  Evaluated two elements at a time on the packed-f32 pipe (v_pk_fma_f32).  The fits are
// reproduced by the snippet in DESIGN.md.  The f32 parity path keeps the exact erff forms.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 clamp2(f32x2 x, float lim) {
  return f32x2{__builtin_amdgcn_fmed3f(x[0], -lim, lim), __builtin_amdgcn_fmed3f(x[1], -lim, lim)};
}
// (written on 4-wide vectors: the compiler splits every step into two independent v_pk_fma_f32, so the
// two Horner chains interleave and hide the packed-math dependency stall)
__device__ __forceinline__ f32x4 clamp4(f32x4 x, float lim) {
  return f32x4{__builtin_amdgcn_fmed3f(x[0], -lim, lim), __builtin_amdgcn_fmed3f(x[1], -lim, lim),
               __builtin_amdgcn_fmed3f(x[2], -lim, lim), __builtin_amdgcn_fmed3f(x[3], -lim, lim)};
}
__device__ __forceinline__ f32x4 splat4(float c) { return f32x4{c, c, c, c}; }
__device__ __forceinline__ f32x4 phi_poly4(f32x4 x) {      // Phi(x)
  const f32x4 xc = clamp4(x, 4.0f), t = xc * xc;
  f32x4 p = splat4(-1.304578543e-09f);
  p = p * t + splat4(1.060087872e-07f); p = p * t + splat4(-3.746289010e-06f); p = p * t + splat4(7.662426288e-05f);
  p = p * t + splat4(-1.023770734e-03f); p = p * t + splat4(9.590060957e-03f); p = p * t + splat4(-6.607606400e-02f);
  p = p * t + splat4(3.988102128e-01f);
  return xc * p + splat4(0.5f);
}
__device__ __forceinline__ f32x4 dgelu_poly4(f32x4 x) {    // gelu'(x)
  const f32x4 xc = clamp4(x, 4.5f), t = xc * xc;
  f32x4 p = splat4(-2.396099311e-11f);
  p = p * t + splat4(2.702686828e-09f); p = p * t + splat4(-1.343048027e-07f); p = p * t + splat4(3.892256086e-06f);
  p = p * t + splat4(-7.353425424e-05f); p = p * t + splat4(9.596712397e-04f); p = p * t + splat4(-8.909952021e-03f);
  p = p * t + splat4(5.886488750e-02f); p = p * t + splat4(-2.652524630e-01f); p = p * t + splat4(7.977590902e-01f);
  return xc * p + splat4(0.5f);
}
__device__ __forceinline__ f32x2 phi_poly2(f32x2 x) { const f32x4 r = phi_poly4(f32x4{x[0], x[1], x[0], x[1]}); return f32x2{r[0], r[1]}; }
__device__ __forceinline__ f32x2 dgelu_poly2(f32x2 x) { const f32x4 r = dgelu_poly4(f32x4{x[0], x[1], x[0], x[1]}); return f32x2{r[0], r[1]}; }
__device__ __forceinline__ float gelu_poly(float x) { const f32x2 v{x, x}; return (v * phi_poly2(v))[0]; }
__device__ __forceinline__ float dgelu_poly(float x) { return dgelu_poly2(f32x2{x, x})[0]; }
__device__ __forceinline__ float apply_epilogue_fast(float v, int epi, float auxv) {
  switch (epi) {
    case FCMF_EPI_GELU: return gelu_poly(v);
    case FCMF_EPI_TANH: return tanhf(v);
    case FCMF_EPI_DGELU: return v * dgelu_poly(auxv);
    case FCMF_EPI_DTANH: return v * (1.0f - auxv * auxv);
    case FCMF_EPI_ADD: return v + auxv;
    default: return v;
  }
}

__device__ __forceinline__ float apply_epilogue(float v, int epi, float auxv) {
  switch (epi) {
    case FCMF_EPI_GELU: return gelu_f(v);
    case FCMF_EPI_TANH: return tanhf(v);
    case FCMF_EPI_DGELU: return v * dgelu_f(auxv);
    case FCMF_EPI_DTANH: return v * (1.0f - auxv * auxv);
    case FCMF_EPI_ADD: return v + auxv;
    default: return v;
  }
}

// =========================================================================================
// bf16 MFMA kernel: 128x128 block tile, BK = 32, 4 waves (2x2, 64x64 each), 4-stage LDS ring
// filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write), three k-tiles
// in flight behind a counted s_waitcnt vmcnt and ONE raw s_barrier per k-tile; 64 KiB of LDS so
// that two workgroups share a CU (one's epilogue / prologue hides under the other's MFMAs).
// =========================================================================================
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NSTAGE = 4;
constexpr int TILE_BYTES = 128 * BK * 2;   // one operand tile: 8 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

typedef __attribute__((address_space(3))) void* lds_void_t;

// Image of a K-contiguous operand tile [128 rows][32 k] (64-B rows): 16-B chunk index c (0..3)
// stored at c ^ (bit3(row) << 1): the ds_read_b128 fragment reads are conflict free.
__device__ __forceinline__ int swz_row(int r) { return ((r >> 3) & 1) << 1; }
// Image of a transposed operand tile [32 k][128 x] (x contiguous, 256-B rows): 32-B pair index
// XOR key(k) = (k&3) | ((k>>3)&1)<<2: the ds_read_b64_tr_b16 reads are conflict free.
__device__ __forceinline__ int tr_key(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }
// (both verified with tools/lds_conflicts.py)

// LDS-DMA writes LDS linearly (wave-uniform base + lane*16), so the swizzle is applied to the
// per-lane SOURCE address: lane -> linear 16-B slot -> (row, physical chunk) -> logical chunk.
// Returns the byte offset of the lane's 16 B inside the operand for k-tile 0 (0x80000000 = out
// of range: the buffer range check then returns zeros).
template <bool TR>
__device__ __forceinline__ unsigned dma_voffset(int wave, int j, int lane, int64_t ld, int x0, int xdim) {
  const int q = (wave * 2 + j) * 64 + lane;   // 16-B slot inside the 8 KiB tile
  if (!TR) {
    const int row = q >> 2, c = (q & 3) ^ swz_row(row);
    if (x0 + row >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)(x0 + row) * ld + c * 8) * 2);
  } else {
    const int kk = q >> 4, cp = q & 15;
    const int c16 = ((((cp >> 1) ^ tr_key(kk))) << 1) | (cp & 1);
    if (x0 + c16 * 8 >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)kk * ld + x0 + c16 * 8) * 2);
  }
}


// ---- LDS-DMA issued from inline asm (kernels with a TRANSPOSED operand) ---------------------------------------------------
// hipcc (ROCm 7.2) treats a compiler-visible LDS-DMA (`__builtin_amdgcn_raw_ptr_buffer_load_lds`) as a pending LDS store and
// puts `s_waitcnt vmcnt(0)` in front of every `__builtin_amdgcn_ds_read_tr16_b64` that follows it -- not in front of plain
// `ds_read_b128` fragment loads.  In the main loop that wait sat between the DMA of k-tile t+3 and the transposed fragment
// reads of k-tile t: the whole prefetch pipeline drained once per k-tile, and the weight-gradient kernel (dW = dY^T X, both
// operands transposed) ran one DMA round trip per k-tile (~1.0 us, 1700 cycles against 1024 cycles of MFMA; round-2
// profiles/r02_gemm_k64.txt "TN").  An asm statement without outputs is register-safe (cdna_hip_programming.md section 5.7,
// item 1); ordering is the kernel's own counted vmcnt + barrier, exactly as for the builtin.  M0 (the LDS destination) is
// written in the statement that reads it, one wait state ahead of the load.
__device__ __forceinline__ u32x4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)reinterpret_cast<unsigned long long>((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ void dma16_asm(const u32x4& rsrc, unsigned lds_dst, unsigned voffset, unsigned soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ void dma16_asm0(const u32x4& rsrc, unsigned lds_dst, unsigned voffset) {   // soffset = 0
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_dst), "v"(voffset), "s"(rsrc) : "memory");
}

template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int x0, int lane) {
  if (!TR) {
    const int r = x0 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(lds + r * 64 + ((((lane >> 4)) ^ swz_row(r)) << 4));
  } else {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int kk = 8 * g + q, f = x0 >> 4;
    const int off = kk * 256 + ((f ^ tr_key(kk)) << 5) + ((p >> 1) << 4) + ((p & 1) << 3);
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds + off));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(lds + off + 4 * 256));   // key(kk+4) == key(kk)
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

template <bool A_TR, bool B_TR, typename TC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  // (an 8-stage ring -- seven k-tiles in flight -- was tried for the grids that leave most CUs empty, the IAOG decoder's 768-row
  //  GEMMs: 36 workgroups, 24 k-tiles, 20 us.  No gain: those launches are a serial chain of per-k-tile fixed costs -- wait, barrier,
  //  DMA issue, fragment reads -- not of DMA round trips.)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NSTAGE x (A tile + B tile) = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * p.ktiles_per_split;
  const int kt_end = min(nk_total, kt_begin + p.ktiles_per_split);
  const int nkt = kt_end - kt_begin;

  // buffer descriptors: the hardware range check zero-fills rows past M/N and, for transposed
  // operands, k rows past K
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);
  unsigned va[2], vb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    va[j] = dma_voffset<A_TR>(wave, j, lane, p.lda, i0, p.M);
    vb[j] = dma_voffset<B_TR>(wave, j, lane, p.ldb, j0, p.N);
    if constexpr (!A_TR) {
      if (p.cv_C) {              // implicit-GEMM convolution: the row's offset is that of its receptive field
        const int q = (wave * 2 + j) * 64 + lane, row = q >> 2, c = (q & 3) ^ swz_row(row);
        va[j] = i0 + row < p.M ? (unsigned)((conv_row_base(p, i0 + row) + c * 8) * 2) : 0x80000000u;
      }
    }
  }
  // per-k-tile advance: K-contiguous operands move by BK elements (scalar offset), transposed
  // operands by BK rows (added to the per-lane offset so that the range check sees it)
  const unsigned a_step = A_TR ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
  const unsigned b_step = B_TR ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);

  constexpr bool ASM_DMA = A_TR || B_TR;    // (see dma16_asm: transposed fragment reads must not see a compiler-visible LDS-DMA)
  [[maybe_unused]] const u32x4 wA = rsrc_words(p.A, p.a_bytes), wB = rsrc_words(p.B, p.b_bytes);
  auto issue = [&](int t) {   // k-tile index relative to kt_begin -> ring stage t & 3
    char* st = smem + (t & (NSTAGE - 1)) * STAGE_BYTES + (wave * 2) * 1024;
    const unsigned ka = (!A_TR && p.cv_C) ? conv_k_offset(p, (kt_begin + t) * BK) : (unsigned)(kt_begin + t) * a_step;
    const unsigned kb = (unsigned)(kt_begin + t) * b_step;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (ASM_DMA) {
        if (A_TR) dma16_asm0(wA, lds_addr_of(st + j * 1024), va[j] + ka);
        else      dma16_asm(wA, lds_addr_of(st + j * 1024), va[j], ka);
        if (B_TR) dma16_asm0(wB, lds_addr_of(st + TILE_BYTES + j * 1024), vb[j] + kb);
        else      dma16_asm(wB, lds_addr_of(st + TILE_BYTES + j * 1024), vb[j], kb);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t)(st + j * 1024), 16, va[j], ka, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t)(st + TILE_BYTES + j * 1024), 16, vb[j], kb, 0, 0);
      }
    }
  };

  f32x4 acc[4][4];  // [j frag][i frag]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nkt) issue(t);

  for (int t = 0; t < nkt; ++t) {
    // tile t has landed once at most the DMAs of the (<= 2) younger tiles are outstanding
    const int younger = nkt - 1 - t;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave's DMAs for tile t are in LDS; stage (t-1)&3 is free
    if (t + NSTAGE - 1 < nkt) issue(t + NSTAGE - 1);
    const char* la = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const char* lb = la + TILE_BYTES;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      fa[f] = read_frag<A_TR>(la, wm * 64 + f * 16, lane);
      fb[f] = read_frag<B_TR>(lb, wn * 64 + f * 16, lane);
    }
#pragma unroll
    for (int fj = 0; fj < 4; ++fj)
#pragma unroll
      for (int fi = 0; fi < 4; ++fi)
        // D rows <- B operand (j), D cols <- A operand (i): each lane ends up with 4 consecutive
        // j of one row i, i.e. a contiguous 8/16-byte piece of C.
        acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[fj], fa[fi], acc[fj][fi], 0, 0, 0);
  }

  // ---- epilogue -----------------------------------------------------------------------
  TC* C = reinterpret_cast<TC*>(p.C);
  TC* AUX = reinterpret_cast<TC*>(p.aux);
  const bool atomic = (p.ksplit > 1);
  const bool lead = (blockIdx.z == 0);
  float4 cs[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
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
        if constexpr (sizeof(TC) == 2) {
          v.x = apply_epilogue_fast(v.x, p.epilogue, a.x); v.y = apply_epilogue_fast(v.y, p.epilogue, a.y);
          v.z = apply_epilogue_fast(v.z, p.epilogue, a.z); v.w = apply_epilogue_fast(v.w, p.epilogue, a.w);
        } else {
          v.x = apply_epilogue(v.x, p.epilogue, a.x); v.y = apply_epilogue(v.y, p.epilogue, a.y);
          v.z = apply_epilogue(v.z, p.epilogue, a.z); v.w = apply_epilogue(v.w, p.epilogue, a.w);
        }
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
      if (p.colsum) { cs[fj].x += v.x; cs[fj].y += v.y; cs[fj].z += v.z; cs[fj].w += v.w; }
    }
  }
  if (p.colsum) {
    // lanes that share (lane >> 4) hold the same 4 columns for 16 different rows: butterfly over lane & 15
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      float4 t = cs[fj];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
      }
      const int j = j0 + wn * 64 + fj * 16 + (lane >> 4) * 4;
      if ((lane & 15) == 0 && j < p.N) {
        atomicAdd(p.colsum + j, t.x); atomicAdd(p.colsum + j + 1, t.y); atomicAdd(p.colsum + j + 2, t.z); atomicAdd(p.colsum + j + 3, t.w);
      }
    }
  }
}

// =========================================================================================
// 256x256 kernel for the big GEMMs of the step.  8 waves (2 x 4), every wave owns a 128x64 output tile
// (32 accumulator fragments of v_mfma_f32_16x16x32_bf16: 12 ds_read_b128 feed 32 MFMAs per k-tile).
//  * persistent: grid = resident workgroups (one per CU, 160 KiB LDS), each walks the work items
//    (output tile x k-split) in an XCD-aware order;
//  * 4-stage LDS ring filled by buffer_load_dwordx4 ... lds (swizzle on the source address, range check =
//    zero fill), counted vmcnt + ONE raw s_barrier per k-tile; the ring position RUNS ON across work items;
//  * PING-PONG between the two waves that share a SIMD (see the main loop);
//  * bf16 outputs: the epilogue is wave-local and barrier-free.  After its last MFMA a wave (1) puts the first
//    three k-tiles of the NEXT work item in flight (into the three ring stages that are already drained),
//    (2) turns its accumulators into finished bf16 values in the fragment layout (bias, GELU, gelu', residual
//    add as packed-f32 math), (3) transposes them through a private 4 KiB LDS slice above the ring and
//    writes whole 128-byte row pieces with buffer stores.  Waves drift apart
//    freely; the first barrier of the next item re-aligns them.
//  * f32 outputs (weight gradients: accumulate / split-K atomics) stage through the whole ring between
//    barriers: accumulators -> f32 image -> whole-row stores or 256-byte float atomics.
// The epilogue kind is a template parameter: one straight-line epilogue per kernel keeps the register
// allocation of the main loop clean (2 waves/SIMD -> 256 VGPRs, 128 of them accumulators).
// =========================================================================================
constexpr int GB = 256;                       // block tile rows and columns
constexpr int RING_BYTES = 128 * 1024;        // operand ring of the 32-deep k-tile kernels: 4 stages of (A tile + B tile); the epilogue's
                                              // transposition slices lie above it.  The 64-deep kernels use all 160 KiB as five operand slots.

// Image of a K-contiguous operand tile with 64-deep k-tiles [256 rows][64 k] (128-B rows = whole cache lines):
// 16-B chunk c (0..7) stored at c ^ ((row >> 1) & 7): the ds_read_b128 fragment reads of either k-half are
// conflict free (tools/lds_conflicts.py).
__device__ __forceinline__ int swz_row64(int r) { return (r >> 1) & 7; }

// per-lane source offset of DMA piece `piece` (1 KiB of the operand tile's LDS image) for k-tile 0.
//  KB = 32: K-contiguous tile = 16 rows x 64 B per piece; transposed tile = 2 k-rows x 512 B per piece.
//  KB = 64: K-contiguous tile = 8 rows x 128 B per piece (every 128-B line is fetched by ONE wave instruction:
//           with 64-B row pieces the CU's L2->L1 path moves each line twice, measured 1.5x slower);
//           transposed tile as for KB = 32 with 64 k-rows.
//  FP8 (KB = 64 geometry, 128 e4m3 per 128-B row): lane group g of v_mfma_scale_f32_16x16x128_f8f6f4 consumes 32 CONSECUTIVE
//           k of a row = the two 16-B chunks 2g, 2g+1; they are stored where the bf16 kernel's fragment reads look
//           (physical chunk (g ^ swizzle) and the same + 4: conflict-free ds_read_b128 pairs), i.e. logical chunk
//           c = 2g + h lives at physical chunk (g | h << 2) ^ swizzle.
//  PAD (transposed tiles of the weight-gradient kernel): piece i holds k-rows r0(i) = 8 (i >> 2) + (i & 3) and r0(i) + 4 in natural
//           column order and lies at LDS offset 1056 i (1 KiB + 32 B): the eight pieces a transposed fragment read touches start
//           32 B apart modulo the 256-B bank row, so ds_read_b64_tr_b16 is conflict free WITHOUT an XOR on the column index --
//           fragment f is a compile-time immediate offset (32 f) from one per-lane base (tools/lds_conflicts.py).
template <bool TR, int KB, bool FP8 = false, bool PAD = false>
__device__ __forceinline__ unsigned dma_voffset_t(int piece, int lane, int64_t ld, int x0, int xdim, const GemmParams* cv = nullptr) {
  const int q = piece * 64 + lane;             // 16-B slot inside the operand tile
  if constexpr (PAD) {
    static_assert(TR && KB == 32 && !FP8, "padded image: transposed 32-deep tiles");
    const int kk = ((piece >> 2) << 3) + (piece & 3) + ((lane >> 5) << 2), c16 = lane & 31;
    if (x0 + c16 * 8 >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)kk * ld + x0 + c16 * 8) * 2);
  }
  if (!TR && !FP8 && cv) {                     // implicit-GEMM convolution (A operand): row -> receptive-field origin
    const int row = KB == 32 ? q >> 2 : q >> 3, c = KB == 32 ? (q & 3) ^ swz_row(row) : (q & 7) ^ swz_row64(row);
    if (x0 + row >= xdim) return 0x80000000u;
    return (unsigned)((conv_row_base(*cv, x0 + row) + c * 8) * 2);
  }
  if constexpr (FP8) {
    static_assert(!TR && KB == 64, "fp8 tiles: K-contiguous operands, 128-byte rows");
    const int row = q >> 3, x = (q & 7) ^ swz_row64(row), c = ((x & 3) << 1) | (x >> 2);
    if (x0 + row >= xdim) return 0x80000000u;
    return (unsigned)((int64_t)(x0 + row) * ld + c * 16);
  }
  if (!TR && KB == 32) {
    const int row = q >> 2, c = (q & 3) ^ swz_row(row);
    if (x0 + row >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)(x0 + row) * ld + c * 8) * 2);
  } else if (!TR) {
    const int row = q >> 3, c = (q & 7) ^ swz_row64(row);
    if (x0 + row >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)(x0 + row) * ld + c * 8) * 2);
  } else {
    const int kk = q >> 5, cp = q & 31;        // 512-B rows: 32 chunks of 16 B
    const int c16 = ((((cp >> 1) ^ tr_key(kk))) << 1) | (cp & 1);
    if (x0 + c16 * 8 >= xdim) return 0x80000000u;
    return (unsigned)(((int64_t)kk * ld + x0 + c16 * 8) * 2);
  }
}

template <int V> struct IntTag { static constexpr int value = V; };

// 8 consecutive outputs of one row <-> four packed-f32 pairs
__device__ __forceinline__ void load8f(const float* q, f32x2 (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
  v[0] = f32x2{a[0], a[1]}; v[1] = f32x2{a[2], a[3]}; v[2] = f32x2{b[0], b[1]}; v[3] = f32x2{b[2], b[3]};
}
__device__ __forceinline__ void store8f(float* q, const f32x2 (&v)[4]) {
  *reinterpret_cast<f32x4*>(q) = f32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
  *reinterpret_cast<f32x4*>(q + 4) = f32x4{v[2][0], v[2][1], v[3][0], v[3][1]};
}

// MI = 16-row fragments per wave along M: 8 -> 256x256 block tile, 6 -> 192x256 (row-major A only).  The
// 192-row variant exists for tile-count quantisation: M = 49152, N = 768 gives 576 tiles of 256 rows (2.25
// rounds on 256 CUs, the last one a quarter full) but 768 tiles of 192 rows = exactly 3 rounds of 3/4 the work.
// KB = depth of a k-tile: 32 (4-stage ring, three k-tiles in flight) or 64 (2-stage ring, ONE k-tile in flight, half
// the barriers; K-contiguous operands then arrive as whole 128-B lines -- see dma_voffset_t).
// FP8: e4m3 operands (K-contiguous, KB = 64 geometry = 128 elements per 128-byte row), v_mfma_scale_f32_16x16x128_f8f6f4 with
// unit block scales (2x the bf16 MFMA rate per clock), per-row dequantisation scales applied to the f32 accumulators in the
// epilogue (acc[i][j] * sa[i] * sb[j]).  Ring, DMA pieces, swizzled LDS image and fragment READ addresses are those of the
// bf16 64-deep kernel; a k-tile is consumed in two M-halves (B fragments + A fragments 0-3, then A fragments 4-7) instead of
// two K-halves, so the fragment registers stay at 64.
// wait until at most N vector-memory operations are outstanding; the four registers of the awaited aux round are operands of the
// statement, so no use of them is scheduled above it (cdna_hip_programming.md section 5.7, item 1, form (ii))
template <int N>
__device__ __forceinline__ void wait_vm(u32x4 (&r)[4]) {
  static_assert(N >= 0 && N < 64, "vmcnt field");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}
typedef int i32x8 __attribute__((ext_vector_type(8)));
// WNW = wave columns (64 output columns each): 4 -> the 256-column tiles above; 2 / 1 -> NARROW outputs (N <= 128 / 64: the
// 64- and 128-channel convolutions of the ResNet trunk): the same 256-row tile, ring and DMA, but the eight waves stand 4 x 2 or
// 8 x 1 over 256 x 128 / 256 x 64 outputs (MI = 4 / 2 fragments of 16 rows per wave), so no matrix instruction multiplies the
// zero-filled columns a 256-column tile would carry (those layers ran on the 128 x 128 kernel at ~370 TFLOP/s, half of it on
// zeros for N = 64).  K-contiguous operands, 64-deep k-tiles, bf16 output, no epilogue operand.
template <bool A_TR, bool B_TR, typename TC, int EPI, int MI, int KB, bool FP8 = false, bool BATCH = false, int WNW = 4>
__device__ __forceinline__ void gemm_bf16_tile256_body(const GemmParams& p, const BatchPtrs* bp = nullptr) {
  static_assert(!BATCH || (A_TR && B_TR && sizeof(TC) == 4 && MI == 8 && KB == 32), "batched: the weight-gradient kernel");
  static_assert(WNW == 4 ? (MI == 8 || (MI == 6 && !A_TR && sizeof(TC) == 2))
                         : ((WNW == 2 && MI == 4) || (WNW == 1 && MI == 2)) && !A_TR && !B_TR && sizeof(TC) == 2 && KB == 64 && !FP8 &&
                               EPI == FCMF_EPI_NONE,
                "256-row tiles (192 rows: row-major A, bf16 output); narrow tiles: 4 x 2 / 8 x 1 waves, plain bf16 NT GEMM");
  static_assert(KB == 32 || KB == 64, "k-tile depth");
  static_assert(!FP8 || (KB == 64 && !A_TR && !B_TR && sizeof(TC) == 2), "fp8: K-contiguous operands, bf16 output");
  constexpr int KE = FP8 ? 2 * KB : KB;       // contraction elements per k-tile
  constexpr int NW = 8;
  constexpr int WM = 16 * MI;                  // rows per wave
  constexpr int TM = (8 / WNW) * WM;           // block tile rows (WNW = 4: 32 * MI)
  constexpr bool FULL_A = TM == 256;           // every wave loads A_PIECES pieces of the A tile
  // dW = dY^T X (both operands transposed, f32 output, no bf16 epilogue slices above the ring): padded transposed image
  constexpr bool PAD_TR = A_TR && B_TR && sizeof(TC) == 4 && KB == 32;
  constexpr int PIECE_STRIDE = PAD_TR ? 1056 : 1024;
  constexpr int A_TILE_BYTES = PAD_TR ? 16 * PIECE_STRIDE : GB * KB * 2;    // LDS bytes per operand tile (the 192-row A tile leaves a quarter unused)
  constexpr int TSTAGE_BYTES = 2 * A_TILE_BYTES;
  constexpr int TNST = 4;                      // KB = 32: stages of the ring (A tile + B tile each)
  constexpr bool RING5 = KB == 64;             // KB = 64: five 32-KiB slots, one OPERAND tile each (see `base` below)
  constexpr int NH = KB / 32;                  // 32-deep halves of a k-tile (MFMA k = 32)
  constexpr int A_PIECES = KB / 16, B_PIECES = KB / 16;   // 1 KiB DMA pieces per wave per operand tile (A: see na_pieces)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;  // rows wm*WM, cols wn*64 (WNW = 4: wave >> 2, wave & 3)
  // the A tile of the 192-row variant: 12 pieces at KB = 32 (two for waves 0-3, one -- pieces 8..11 -- for waves 4-7),
  // 24 pieces at KB = 64 (three per wave)
  const int na_pieces = FULL_A ? A_PIECES : (KB == 64 ? 3 : (wave < 4 ? 2 : 1));
  const int a_piece0 = FULL_A ? wave * A_PIECES : (KB == 64 ? wave * 3 : (wave < 4 ? wave * 2 : 4 + wave));

  // XCD-aware order inside a round: the workgroups of one XCD (blockIdx % 8) take consecutive logical
  // items, and consecutive items share the A row-panel (all N tiles of one M tile) -> L2 hits.
  const int tiles_n = (p.N + GB - 1) / GB;
  const int nblk = gridDim.x;
  int slot = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = slot & 7, local = slot >> 3;
    slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int nk_total = (p.K + KE - 1) / KE;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);
  constexpr bool ASM_DMA = A_TR || B_TR;    // (see dma16_asm: transposed fragment reads must not see a compiler-visible LDS-DMA)
  [[maybe_unused]] u32x4 wA = rsrc_words(p.A, p.a_bytes), wB = rsrc_words(p.B, p.b_bytes);   // (BATCH: re-pointed per work item)
  const unsigned a_step = A_TR ? (unsigned)(KB * p.lda * 2) : (unsigned)(KB * 2);
  const unsigned b_step = B_TR ? (unsigned)(KB * p.ldb * 2) : (unsigned)(KB * 2);
  // Per-lane LDS offsets are computed ONCE; fragment f of a K-contiguous operand is a constant 1 KiB
  // step away (row bit 3, which drives the swizzle, does not depend on f).
  typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
  const int rowl = lane & 15, g4 = lane >> 4, q4 = rowl >> 2, p4 = rowl & 3;
  // K-contiguous image: rows of 2*KB bytes; at KB = 64 the k-half h flips bit 2 of the (swizzled) chunk index = bit 6 of
  // the byte offset.  Transposed image: 512-B k-rows, k-half h is 32 rows = 16 KiB further (tr_key(k + 32) == tr_key(k)).
  const int row_base = KB == 32 ? rowl * 64 + ((g4 ^ swz_row(rowl)) << 4) : rowl * 128 + ((g4 ^ swz_row64(rowl)) << 4);
  const int trk = tr_key(8 * g4 + q4);
  const int tr_col = ((p4 >> 1) << 4) + ((p4 & 1) << 3);
  // (padded image: k-row 8 g + q is the first half of piece 4 g + q, k-row + 4 its second half)
  const int pad_lane = (4 * g4 + q4) * PIECE_STRIDE + p4 * 8;
  const int a_lane = PAD_TR ? pad_lane + wm * 256 : A_TR ? (8 * g4 + q4) * 512 + tr_col : row_base + wm * WM * (2 * KB);
  const int b_lane = PAD_TR ? pad_lane + wn * 128 : B_TR ? (8 * g4 + q4) * 512 + tr_col : row_base + wn * 64 * (2 * KB);
  auto frag_a = [&](const char* st, int f, int h) -> bf16x8 {     // f = 0..MI-1: 16-row fragment of this wave's rows
    if (!A_TR) return *reinterpret_cast<const bf16x8*>(st + (a_lane ^ (h << 6)) + f * (32 * KB));
    const char* q = PAD_TR ? st + a_lane + f * 32 : st + a_lane + (((wm * 8 + f) ^ trk) << 5) + h * 16384;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)q);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(q + (PAD_TR ? 512 : 4 * 512)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  };
  auto frag_b = [&](const char* st, int f, int h) -> bf16x8 {     // f = 0..3
    if (!B_TR) return *reinterpret_cast<const bf16x8*>(st + (b_lane ^ (h << 6)) + f * (32 * KB));
    const char* q = PAD_TR ? st + b_lane + f * 32 : st + b_lane + (((wn * 4 + f) ^ trk) << 5) + h * 16384;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)q);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(q + (PAD_TR ? 512 : 4 * 512)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  };

  struct Item { int i0, j0, kt_begin, nkt, zsplit, batch; };
  auto decode = [&](int item) -> Item {
    const int zsplit = item / p.tiles, tile = item - zsplit * p.tiles;
    int local = tile, batch = 0;
    if constexpr (BATCH) { batch = tile / p.tiles_per_mat; local = tile - batch * p.tiles_per_mat; }
    const int tile_m = local / tiles_n, tile_n = local % tiles_n;
    const int kb = zsplit * p.ktiles_per_split;
    return Item{tile_m * TM, tile_n * GB, kb, min(nk_total, kb + p.ktiles_per_split) - kb, zsplit, batch};
  };
  // KB = 32 ring: k-tile t of the current item lives in stage (base + t) % 4 (A tile, then B tile); `base` runs on across
  // items.  Three k-tiles in flight.
  // KB = 64 ring: the whole 160 KiB as five slots of one operand tile; with n = 2t for A_t and n = 2t + 1 for B_t, tile n
  // lives in slot (base + n) % 5.  While tile t is multiplied, A_{t+1} (issued one barrier earlier), B_{t+1} and A_{t+2}
  // are in flight: the DMA stream never drains at a barrier, which a 2 x 64 KiB double buffer (one whole k-tile issued
  // and awaited per barrier interval: latency-bound) cannot offer.  The bf16 epilogue's transposition slices take the
  // slot of the NEXT item's A_1, which is therefore issued after that item's first barrier.
  int base = 0;                              // KB = 32: stage of the current item's tile 0; KB = 64: slot of A_t (runs with t)
  auto stage_at = [&](int rel) -> char* {    // KB = 32; rel = k-tile index relative to the current item's tile 0 (may run into the next item)
    return smem + ((base + rel) & (TNST - 1)) * TSTAGE_BYTES;
  };
  auto slot_at = [&](int d) -> char* {       // KB = 64: the slot d (0..4) positions after that of the current A tile
    int x = base + d;
    x = x >= 5 ? x - 5 : x;
    return smem + x * 32768;
  };
  // per-lane DMA source offsets of an item's operand tiles (k-tile 0): computed ONCE per item
  struct Src { unsigned a[A_PIECES], b[B_PIECES]; };
  auto sources = [&](const Item& w) -> Src {
    Src r;
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) r.a[j] = dma_voffset_t<A_TR, KB, FP8, PAD_TR>(a_piece0 + j, lane, p.lda, w.i0, p.M, p.cv_C ? &p : nullptr);
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) r.b[j] = dma_voffset_t<B_TR, KB, FP8, PAD_TR>(wave * B_PIECES + j, lane, p.ldb, w.j0, p.N);
    return r;
  };
  // DMA of k-tile t of item w: A tile to sa and / or B tile to sb (nullptr = skip)
  auto issue_ab = [&](const Item& w, const Src& src, int t, char* sa, char* sb) {
    const unsigned ka = (!A_TR && !FP8 && p.cv_C) ? conv_k_offset(p, (w.kt_begin + t) * KB) : (unsigned)(w.kt_begin + t) * a_step;
    const unsigned kb = (unsigned)(w.kt_begin + t) * b_step;
    if (sa) {
#pragma unroll
      for (int j = 0; j < A_PIECES; ++j) {
        if (!FULL_A && j >= na_pieces) break;
        char* d = sa + (a_piece0 + j) * PIECE_STRIDE;
        if constexpr (ASM_DMA) {
          if (A_TR) dma16_asm0(wA, lds_addr_of(d), src.a[j] + ka);
          else      dma16_asm(wA, lds_addr_of(d), src.a[j], ka);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t)d, 16, src.a[j], ka, 0, 0);
        }
      }
    }
    if (sb) {
#pragma unroll
      for (int j = 0; j < B_PIECES; ++j) {
        char* d = sb + (wave * B_PIECES + j) * PIECE_STRIDE;
        if constexpr (ASM_DMA) {
          if (B_TR) dma16_asm0(wB, lds_addr_of(d), src.b[j] + kb);
          else      dma16_asm(wB, lds_addr_of(d), src.b[j], kb);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t)d, 16, src.b[j], kb, 0, 0);
        }
      }
    }
  };
  auto issue = [&](const Item& w, const Src& src, int t, int rel) {     // KB = 32: both tiles of k-tile t into stage `rel`
    char* st = stage_at(rel);
    issue_ab(w, src, t, st, st + A_TILE_BYTES);
  };
  auto lds_barrier = [&]() {   // LDS traffic of this wave retired, then the workgroup barrier; global stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  constexpr bool PREFETCH = sizeof(TC) == 2;   // bf16 epilogue leaves the ring alone -> next item's tiles 0, 1 fly under it
  bool pre = false;

  for (int item = slot; item < p.total_items; item += nblk) {
  if constexpr (ASM_DMA && sizeof(TC) == 4) {
    // The DMA of these kernels is invisible to hipcc (dma16_asm), but the previous item's output stores are not: left pending
    // over the loop back-edge they make hipcc guard the first overwrite of their data registers with `s_waitcnt vmcnt(0)` -- and
    // where that lands is scheduling luck (it has landed INSIDE the k-loop, draining the DMA ring once per k-tile).  A wait the
    // compiler can see, here, where nothing of this item is in flight yet, empties its scoreboard for the whole main loop.
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched
  }
  const Item w = decode(item);
  if constexpr (BATCH) { wA = rsrc_words(bp->A[w.batch], p.a_bytes); wB = rsrc_words(bp->B[w.batch], p.b_bytes); }
  Src src = sources(w);
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j) asm volatile("" : "+v"(src.a[j]), "+v"(src.b[j]));   // materialised here, not re-derived per k-tile
  const int i0 = w.i0, j0 = w.j0, nkt = w.nkt;
  // Issue schedule: tiles 0, 1, 2 before the loop (or already in flight from the previous epilogue), tile t + 3
  // after barrier t.  Own DMAs of tile t have landed once at most the two younger tiles (4 DMAs each) are
  // outstanding: loads retire in order, and stores of the previous epilogue that are still in flight only
  // make the wait longer.
  // KB = 64 (five operand slots): A_0, B_0 before the loop (or from the previous epilogue); after every barrier t:
  // B_{t+1}, then the A tiles up to A_{t+2} that are not yet in flight (A_1 and A_2 after barrier 0, one afterwards).
  // Before barrier t only A_{t+1} (this wave's youngest DMAs), if already issued, may be outstanding.
  [[maybe_unused]] int a_next = 1;            // first A tile of this item that has not been issued
  auto wait_landed = [&](int t, auto per_tile) {
    constexpr int PT = decltype(per_tile)::value;   // DMAs per k-tile of this wave: 4, or 3 for waves 4-7 of the 192-row tile
    const int younger = nkt - 1 - t;
    if constexpr (RING5) {
      if (a_next > t + 1) { if (FULL_A) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    if (younger >= 2) { if (PT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else if (younger == 1) { if (PT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto issue_after_barrier = [&](int t) {
    if constexpr (RING5) {
      if (t + 1 < nkt) issue_ab(w, src, t + 1, nullptr, slot_at(3));
      const int a_end = min(nkt, t + 3);
      for (; a_next < a_end; ++a_next)         // (two trips after barrier 0, one afterwards)
        issue_ab(w, src, a_next, slot_at(2 * (a_next - t)), nullptr);
      return;
    }
    if (t + 3 < nkt) issue(w, src, t + 3, t + 3);
  };

  f32x4 acc[4][MI];  // [j frag][i frag]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[MI], fb[4];
  auto load_frags = [&](int t, int h) {     // h: 32-deep half of the k-tile (always 0 at KB = 32)
    const char* sa = RING5 ? slot_at(0) : stage_at(t);
    const char* sb = RING5 ? slot_at(1) : stage_at(t) + A_TILE_BYTES;
#pragma unroll
    for (int f = 0; f < 4; ++f) fb[f] = frag_b(sb, f, h);
#pragma unroll
    for (int f = 0; f < MI; ++f) fa[f] = frag_a(sa, f, h);
  };
  // fp8: `h` counts the M-halves of the k-tile (h = 0: the B fragments and A fragments 0-3, h = 1: A fragments 4..MI-1)
  [[maybe_unused]] i32x8 qa[FP8 ? 4 : 1], qb[FP8 ? 4 : 1];
  auto frag32 = [&](const char* st, int lane_off, int f) -> i32x8 {   // 32 consecutive k of row (lane & 15) of fragment f
    const u32x4 lo = *reinterpret_cast<const u32x4*>(st + lane_off + f * (32 * KB));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(st + (lane_off ^ 64) + f * (32 * KB));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto load_frags8 = [&](auto half) {
    constexpr int h = decltype(half)::value;
    const char* sa = slot_at(0);
    if constexpr (h == 0) {
      const char* sb = slot_at(1);
#pragma unroll
      for (int f = 0; f < 4; ++f) qb[f] = frag32(sb, b_lane, f);
    }
#pragma unroll
    for (int f = 0; f < (h == 0 ? 4 : MI - 4); ++f) qa[f] = frag32(sa, a_lane, 4 * h + f);
  };
  auto mma8 = [&](auto half) {
    constexpr int h = decltype(half)::value;
#pragma unroll
    for (int fi = 0; fi < (h == 0 ? 4 : MI - 4); ++fi)
#pragma unroll
      for (int fj = 0; fj < 4; ++fj)
        // e4m3 x e4m3 (cbsz = blgp = 0), E8M0 block scales 127 = 2^0 for both operands
        acc[fj][4 * h + fi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qb[fj], qa[fi], acc[fj][4 * h + fi], 0, 0, 0, 127, 0, 127);
  };
  auto mma = [&]() {
#pragma unroll
    for (int fi = 0; fi < MI; ++fi)
#pragma unroll
      for (int fj = 0; fj < 4; ++fj)
        // D rows <- B operand (j), D cols <- A operand (i)
        acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[fj], fa[fi], acc[fj][fi], 0, 0, 0);
  };

  if (!pre) {
    if constexpr (RING5) issue_ab(w, src, 0, slot_at(0), slot_at(1));
    else issue(w, src, 0, 0);
    if constexpr (KB == 32) {
      if (1 < nkt) issue(w, src, 1, 1);
      if (2 < nkt) issue(w, src, 2, 2);
    }
  }
  // aux operand of the epilogue (gelu' argument / residual), row layout: 16 B per lane, 8 rows of the wave's
  // 128x64 sub-tile per instruction, 4 instructions per 32-row round.  Rounds 0 and 1 are requested before the
  // LAST block of MFMAs of the main loop, the rest right after it (into the dead fragment registers), so the
  // whole operand is in flight before the epilogue starts.
  constexpr bool HAS_AUX = sizeof(TC) == 2 && (EPI == FCMF_EPI_DGELU || EPI == FCMF_EPI_ADD);
  constexpr int NRND = MI / 2;
  constexpr bool EARLY_AUX = HAS_AUX && MI == 6;   // the 256-row tile has no registers to spare inside the main loop
  constexpr int NAX = !HAS_AUX ? 1 : (EARLY_AUX ? NRND : 2);   // 256-row tile: a ring of two rounds, refilled as consumed
  [[maybe_unused]] u32x4 ax[NAX][4];
  [[maybe_unused]] unsigned aux_base = 0;
  if constexpr (HAS_AUX) {
    const int gj_r = j0 + wn * 64 + (lane & 7) * 8;
    aux_base = gj_r < p.N ? ((unsigned)(i0 + wm * WM + (lane >> 3)) * (unsigned)p.ldc + (unsigned)gj_r) * 2u : 0x80000000u;
  }
  // 64-deep kernels: the aux rows are loaded from inline asm and awaited with HAND-COUNTED vmcnt.  hipcc does not trust the
  // issue order between loads and stores on gfx9 (one vmcnt counter for both): with compiler-visible aux loads it closed every
  // round of the epilogue with a vmcnt ladder down to 0, i.e. each round waited for the PREVIOUS round's output stores to be
  // acknowledged by memory (~2 us each, three to four times per work item: the gelu' GEMM ran 720 TFLOP/s against 1025 of the
  // plain kernel).  vmcnt retires in issue order on gfx950 (MI355X_MICROARCH.md, s_waitcnt paragraph), so "all but the N
  // youngest" with N = the operations issued after the awaited loads is exact; a wait that names FEWER younger operations than
  // really exist only waits longer, never too little (the optional bias loads and the next item's DMA prefetch are handled so).
  constexpr bool ASM_AUX = HAS_AUX && RING5 && MI == 8;   // (192-row tile: its aux loads sit INSIDE the main loop; asm there cost 136 B of scratch)
  [[maybe_unused]] const u32x4 wX = rsrc_words(p.aux, p.c_bytes);
  auto load_aux = [&](int rnd) __attribute__((always_inline)) {
    if constexpr (HAS_AUX) {
      const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, p.c_bytes, 0x00020000);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const unsigned off = aux_base + (unsigned)(rnd * 32 + it * 8) * (unsigned)p.ldc * 2u;
        if constexpr (ASM_AUX) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ax[rnd % NAX][it]) : "v"(off), "s"(wX) : "memory");
        else ax[rnd % NAX][it] = __builtin_amdgcn_raw_buffer_load_b128(rX, off, 0, 0);
      }
    }
  };
  // (waits: wait_vm<N> below names the four registers of the awaited round as operands, so no use of them is scheduled above it)
  // PING-PONG: waves w and w+4 share a SIMD.  Between barrier t and barrier t+1 group A (waves 0-3) feeds
  // (DMA, fragment reads of tile t) THEN multiplies tile t, while group B (waves 4-7) multiplies tile t-1
  // FIRST (fragments read in the previous interval) and feeds tile t afterwards: the SIMD's matrix pipe
  // sees A's 32 MFMAs while B feeds and vice versa.
  const bool group_b = wave >= 4;              // wave is an SGPR: a uniform branch
  if (!group_b) {
    for (int t = 0; t < nkt; ++t) {
      wait_landed(t, IntTag<4>{});
      __builtin_amdgcn_s_barrier();            // tile t visible; the stage of tile t-1 is no longer read
      issue_after_barrier(t);
      if constexpr (EARLY_AUX) {
        if (t == nkt - 1) { load_aux(0); load_aux(1); }
      }
      if constexpr (FP8) {
        load_frags8(IntTag<0>{}); mma8(IntTag<0>{});
        load_frags8(IntTag<1>{}); mma8(IntTag<1>{});
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          load_frags(t, h);
          mma();
        }
      }
      if constexpr (RING5) { base += 2; base = base >= 5 ? base - 5 : base; }
    }
  } else {
    for (int t = 0; t < nkt; ++t) {
      wait_landed(t, IntTag<(FULL_A ? 4 : 3)>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads of tile t-1 have left LDS
      __builtin_amdgcn_s_barrier();
      if constexpr (FP8) {
        if (t > 0) mma8(IntTag<1>{});                       // second M-half of tile t-1
        issue_after_barrier(t);
        load_frags8(IntTag<0>{}); mma8(IntTag<0>{});
        load_frags8(IntTag<1>{});
      } else {
        if (t > 0) mma();                                   // last half of tile t-1
        issue_after_barrier(t);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          load_frags(t, h);
          if (h + 1 < NH) mma();
        }
      }
      if constexpr (RING5) { base += 2; base = base >= 5 ? base - 5 : base; }
    }
    if constexpr (EARLY_AUX) { load_aux(0); load_aux(1); }
    if constexpr (FP8) mma8(IntTag<1>{}); else mma();       // last half of the last tile (nkt >= 1 always)
  }
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int rnd = EARLY_AUX ? 2 : 0; rnd < (EARLY_AUX ? NRND : 2); ++rnd) load_aux(rnd);
  }

  TC* C = reinterpret_cast<TC*>(BATCH ? bp->C[w.batch] : p.C);
  if constexpr (sizeof(TC) == 2) {
    // ---- wave-local bf16 epilogue ------------------------------------------------------------------
    // Every wave of the workgroup has passed barrier nkt-1, so every stage except that of tile nkt-1
    // is drained: the positions nkt + 0 / 1 / 2 take the next item's tiles 0 / 1 / 2 now (tile 3 follows
    // after ITS barrier 0, the regular schedule).  The 32 KiB above the ring hold a private 4 KiB
    // transposition slice per wave ([32 rows][64 columns] bf16, 16-B chunk ^ ((row >> 1) & 7): conflict-free
    // both ways), so the epilogue never touches the ring.
    int lane_e = lane;                          // (laundered: keeps the epilogue's per-lane constants from being
    asm volatile("" : "+v"(lane_e));            //  hoisted above the main loop, where every VGPR is spoken for)
    const int er = lane_e & 15, eg = lane_e >> 4;
    const unsigned ldc2 = (unsigned)p.ldc * 2u;
    const unsigned tile_off = (unsigned)(i0 + wm * WM) * ldc2 + (unsigned)(j0 + wn * 64) * 2u;   // wave's sub-tile
    // transposition slice: fragment-layout accesses are 8 B per lane, row-layout accesses 16 B per lane (whole
    // 128-B row pieces, 8 rows per instruction); LDS executes a wave's accesses in order
    char* slice = (RING5 ? slot_at(2) : smem + RING_BYTES) + wave * 4096;   // (KB = 64: `base` is the next item's A_0 slot by now; its A_1 slot)
    char* wbase = slice + er * 128 + (eg & 1) * 8;
    auto frag_addr = [&](int h, int fj) __attribute__((always_inline)) -> char* {   // fragment (row block h of the round, fj)
      const int row = h * 16 + er, chunk = fj * 2 + (eg >> 1);
      return wbase + h * 16 * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    };
    const int rrow = lane_e >> 3, rchunk = lane_e & 7;          // row-layout lane mapping
    auto row_addr = [&](int it) __attribute__((always_inline)) -> char* {
      const int row = it * 8 + rrow;
      return slice + row * 128 + ((rchunk ^ ((row >> 1) & 7)) << 4);
    };
    // global side of the row layout through buffer descriptors: 32-bit offsets, rows past M are dropped / read
    // as zeros by the range check, lanes past N carry the out-of-range sentinel -- no exec-mask branches
    const int gj_r = j0 + wn * 64 + rchunk * 8;
    const unsigned sbase = gj_r < p.N ? tile_off + (unsigned)rrow * ldc2 + (unsigned)rchunk * 16u : 0x80000000u;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);
    if constexpr (PREFETCH) {
      const int nxt = item + nblk;
      pre = nxt < p.total_items;
      if (pre) {
        const Item wnx = decode(nxt);
        const Src snx = sources(wnx);
        if constexpr (RING5) issue_ab(wnx, snx, 0, slot_at(0), slot_at(1));
        else issue(wnx, snx, 0, nkt);
        if constexpr (KB == 32) {
          if (1 < wnx.nkt) issue(wnx, snx, 1, nkt + 1);
          if (2 < wnx.nkt) issue(wnx, snx, 2, nkt + 2);
        }
      }
    }
    if constexpr (FP8) {
      // acc[i][j] of quantised operands -> x sa[i] x sb[j] (row scales of A and of B), before the bias
      f32x4 sbq[4];
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) {
        const int gj = j0 + wn * 64 + eg * 4 + fj * 16;
        sbq[fj] = gj < p.N ? *reinterpret_cast<const f32x4*>(p.sb + gj) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int fi = 0; fi < MI; ++fi) {
        const int gi = i0 + wm * WM + fi * 16 + er;
        const float sai = gi < p.M ? p.sa[gi] : 0.f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) acc[fj][fi] = acc[fj][fi] * (sbq[fj] * sai);
      }
    }
    if (p.bias) {
#pragma unroll
      for (int fj = 0; fj < 4; ++fj) {
        const int gj = j0 + wn * 64 + eg * 4 + fj * 16;
        const f32x4 bq = gj < p.N ? *reinterpret_cast<const f32x4*>(p.bias + gj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int fi = 0; fi < MI; ++fi) acc[fj][fi] = acc[fj][fi] + bq;
      }
    }
    auto pack = [&](f32x4 v) __attribute__((always_inline)) -> u32x2 {
      bf16x4 o;
      o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
      u32x2 r = __builtin_bit_cast(u32x2, o);
      asm volatile("" : "+v"(r));            // pins the conversion here (no sinking into the loops below)
      return r;
    };
    float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // column sums of the values as stored (bf16-rounded)
    [[maybe_unused]] float csq[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // ... and of their squares (colstats)
    // rows of the slice -> global (16 B per lane), optionally summing the columns
    auto store_round = [&](int rnd, const __amdgpu_buffer_rsrc_t& rD, bool sums) __attribute__((always_inline)) {
      bf16x8 x[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) x[it] = *reinterpret_cast<const bf16x8*>(row_addr(it));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        // (aux 2 = nt: the outputs stream to memory without displacing the A / W lines the other workgroups of the XCD
        //  are about to re-read from L2 -- 43.3 -> 42.2 ms per step)
        if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD, sbase + (unsigned)(rnd * 32 + it * 8) * ldc2, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD, sbase + (unsigned)(rnd * 32 + it * 8) * ldc2, 0, 0);
        if (sums) {
          const int gi = i0 + wm * WM + rnd * 32 + it * 8 + rrow;
          if (gi < p.M) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float xv = (float)x[it][e];
              csum[e] += xv;
              if constexpr ((MI == 8 || WNW != 4) && !HAS_AUX) csq[e] = __builtin_fmaf(xv, xv, csq[e]);   // (colstats; the aux epilogues have no register to spare)
            }
          }
        }
      }
    };
    if constexpr (HAS_AUX) {
      // round by round (32 rows): aux rows (already in registers, full cache lines: a fragment-layout global read
      // would touch 16 lines per instruction) -> slice -> fragment layout, math, results -> slice -> rows -> C
#pragma unroll
      for (int rnd = 0; rnd < MI / 2; ++rnd) {
        if constexpr (ASM_AUX) {
          // operations issued after the loads of round `rnd` (see load_aux): P = this wave's DMA pieces of the next item's first
          // tiles (issued above, only if there is a next item), 4 loads per later aux round, 4 stores per finished round
          constexpr int P = MI == 8 ? 8 : 7;
          if constexpr (MI == 8) {            // ring of two: L0 L1 [P] | r0: L2, S0 | r1: L3, S1 | r2: S2 | r3
            if (rnd == 0) { if (pre) wait_vm<4 + P>(ax[0]); else wait_vm<4>(ax[0]); }
            else if (rnd == 1) { if (pre) wait_vm<8 + P>(ax[1 % NAX]); else wait_vm<8>(ax[1 % NAX]); }
            else if (rnd == 2) wait_vm<12>(ax[2 % NAX]);
            else wait_vm<8>(ax[3 % NAX]);
          } else {                            // all three rounds loaded up front: L0 L1 L2 [P] | r0: S0 | r1: S1 | r2
            if (rnd == 0) { if (pre) wait_vm<8 + P>(ax[0]); else wait_vm<8>(ax[0]); }
            else if (rnd == 1) { if (pre) wait_vm<8 + P>(ax[1 % NAX]); else wait_vm<8>(ax[1 % NAX]); }
            else { if (pre) wait_vm<8 + P>(ax[2 % NAX]); else wait_vm<8>(ax[2 % NAX]); }
          }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(row_addr(it)) = ax[rnd % NAX][it];
        if constexpr (!EARLY_AUX) { if (rnd + 2 < NRND) load_aux(rnd + 2); }
        u32x2 o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int fj = 0; fj < 4; ++fj) {
            const bf16x4 a4 = __builtin_bit_cast(bf16x4, *reinterpret_cast<const u32x2*>(frag_addr(h, fj)));
            const f32x4 a{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
            f32x4 v = acc[fj][rnd * 2 + h];
            if constexpr (EPI == FCMF_EPI_ADD) v = v + a;
            else v = v * dgelu_poly4(a);
            o[h * 4 + fj] = pack(v);
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int fj = 0; fj < 4; ++fj) *reinterpret_cast<u32x2*>(frag_addr(h, fj)) = o[h * 4 + fj];
        store_round(rnd, rC, p.colsum != nullptr);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const bool two_pass = (EPI == FCMF_EPI_GELU) && p.aux != nullptr;
      // accumulators -> finished bf16 values (straight-line code; the accumulator registers die as it goes).
      // o1 = the C values, o0 = the pre-activations that FCMF_EPI_GELU also writes to aux.
      u32x2 o0[EPI == FCMF_EPI_GELU ? 4 * MI : 1], o1[4 * MI];
#pragma unroll
      for (int fi = 0; fi < MI; ++fi) {
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
          f32x4 v = acc[fj][fi];
          if constexpr (EPI == FCMF_EPI_GELU) {
            o0[fi * 4 + fj] = pack(v);
            v = v * phi_poly4(v);
          }
          o1[fi * 4 + fj] = pack(v);
        }
        __builtin_amdgcn_sched_barrier(0);   // one fragment row at a time: short live ranges
      }
      auto write_out = [&](const auto& o, const __amdgpu_buffer_rsrc_t& rD, bool sums) __attribute__((always_inline)) {
#pragma unroll
        for (int rnd = 0; rnd < MI / 2; ++rnd) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) *reinterpret_cast<u32x2*>(frag_addr(h, fj)) = o[(rnd * 2 + h) * 4 + fj];
          store_round(rnd, rD, sums);
        }
      };
      if constexpr (EPI == FCMF_EPI_GELU) {
        if (two_pass) write_out(o0, __builtin_amdgcn_make_buffer_rsrc(p.aux, 0, p.c_bytes, 0x00020000), false);
      }
      write_out(o1, rC, p.colsum != nullptr || p.colstats != nullptr);
    }
    if (p.colsum) {
      // a lane owns 8 fixed columns for the rows it visited; lanes that share (lane & 7) share the columns
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = csum[e];
        t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
        if (lane_e < 8 && gj_r < p.N) atomicAdd(p.colsum + gj_r + e, t);
      }
    }
    if constexpr ((MI == 8 || WNW != 4) && !HAS_AUX) {
      if (p.colstats && (WNW != 4 || i0 + wm * WM < p.M)) {
        // the wave's WM rows x 64 columns: lanes that share (lane & 7) share 8 columns; lanes 0-7 hold (sum, sum of squares) of
        // their 8 columns = 64 contiguous bytes of a block row
        float st[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = csum[e], u = csq[e];
          t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
          u += __shfl_xor(u, 8, 64); u += __shfl_xor(u, 16, 64); u += __shfl_xor(u, 32, 64);
          st[2 * e] = t; st[2 * e + 1] = u;
        }
        if constexpr (WNW == 4) {        // 256-column tiles: a wave is a whole 128-row block -- no workgroup traffic
          if (lane_e < 8 && gj_r < p.N) {
            float* o = p.colstats + ((int64_t)((i0 + wm * WM) >> 7) * p.N + gj_r) * 2;
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(o + 4 * v) = f32x4{st[4 * v], st[4 * v + 1], st[4 * v + 2], st[4 * v + 3]};
          }
        } else {
          // narrow layouts: the 8 / WNW wave rows of a column group meet in LDS (each wave's own transposition slice, idle now;
          // the next item's DMA takes that slot only after ITS first barrier) and wave row 0 stores ONE block per 256-row tile
          // (32- / 64-row blocks made the reduce pass read 4-8x the partial rows from 7 workgroups)
          float* sl = reinterpret_cast<float*>(slice);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane_e < 8) {
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(sl + lane_e * 16 + 4 * v) = f32x4{st[4 * v], st[4 * v + 1], st[4 * v + 2], st[4 * v + 3]};
          }
          lds_barrier();
          if (wm == 0 && lane_e < 8 && gj_r < p.N) {
            f32x4 a[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const float* s0 = reinterpret_cast<const float*>(slice - wave * 4096);       // slice of wave 0
#pragma unroll
            for (int r = 0; r < 8 / WNW; ++r) {
              const float* q = s0 + (r * WNW + wn) * 1024 + lane_e * 16;                 // (4096 bytes = 1024 floats per wave)
#pragma unroll
              for (int v = 0; v < 4; ++v) a[v] += *reinterpret_cast<const f32x4*>(q + 4 * v);
            }
            float* o = p.colstats + ((int64_t)(i0 >> 8) * p.N + gj_r) * 2;
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(o + 4 * v) = a[v];
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice reads have returned before the wave moves on
  } else if constexpr (MI == 8) {
    // ---- f32 epilogue (weight gradients: accumulate / split-K; epilogue kind is always NONE here):
    // accumulators -> LDS (f32, one 128-row half at a time, the whole ring) -> row-wise output with
    // whole-row stores or 256-byte float atomics for split-K
    constexpr int LPR = GB / 8, RPI = 64 / LPR, RPW = 128 / NW;
    const bool to_ws = p.ws != nullptr && p.ksplit > 1;     // k-split partials go to the workspace with plain stores
    if (to_ws && !p.bias && !p.colsum) {
      // The workspace layout is ours to choose: the partial tile goes out in the FRAGMENT layout, straight from the accumulator
      // registers -- every store instruction is one contiguous KiB (64 lanes x 16 B), no LDS staging, no barrier.  The reduce pass
      // (splitk_reduce_frag_kernel) reads the partials back in the same order and only there maps (fragment, lane) to (row, column).
      // (The row-layout path below staged 2 x 128 KiB through LDS between four barriers.)  One VGPR of addressing: lane * 16;
      // everything else of the address is wave-uniform (scalar offset of the buffer store).
      int lane_w = lane;
      asm volatile("" : "+v"(lane_w));            // (laundered: not hoisted above the main loop, where every VGPR is spoken for)
      const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, 0x7FFFFFFF, 0x00020000);
      const unsigned wbase = (unsigned)item * (unsigned)(GB * GB * 4) + (unsigned)wave * (32u * 1024u);
#pragma unroll
      for (int fj = 0; fj < 4; ++fj)
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[fj][fi]), rW, (unsigned)lane_w * 16u,
                                                 wbase + (unsigned)(fj * 8 + fi) * 1024u, 0);
      base = (base + nkt) & (TNST - 1);
      continue;                                   // (barrier-free, like the bf16 epilogue: the next item's tiles 0-2 go to the three
                                                  //  stages every wave has finished reading; the fourth waits for ITS barrier 0)
    }
    const bool atomic = (p.ksplit > 1) && !to_ws;
    const bool lead = (w.zsplit == 0);
    float* Cout = to_ws ? p.ws + (int64_t)w.zsplit * p.M * p.N : reinterpret_cast<float*>(C);
    const int64_t ldo = to_ws ? (int64_t)p.N : p.ldc;
    const bool rmw = p.accumulate && !to_ws;                // (the reduce pass adds the old C once)
    float* Ct = reinterpret_cast<float*>(smem);   // [128][256] f32; 16-B chunk index XOR (row & 7)
    int lane_e = lane;                          // (laundered, as in the bf16 epilogue)
    asm volatile("" : "+v"(lane_e));
    const int c0 = (lane_e % LPR) * 2, gj = j0 + c0 * 4, rsub = lane_e / LPR;
    f32x2 csum[4] = {splat2(0.f), splat2(0.f), splat2(0.f), splat2(0.f)};
    lds_barrier();                                // every wave is done with the operand ring
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      if (wm == half) {
#pragma unroll
        for (int fi = 0; fi < 8; ++fi) {
          const int row = fi * 16 + (lane_e & 15);
#pragma unroll
          for (int fj = 0; fj < 4; ++fj) {
            const int chunk = wn * 16 + fj * 4 + (lane_e >> 4);
            *reinterpret_cast<f32x4*>(Ct + row * GB + ((chunk ^ (row & 7)) << 2)) = acc[fj][fi];
          }
        }
      }
      lds_barrier();
      if (atomic) {
        // the k-splits of one output tile finish together and add into the same lines: every split starts at
        // a different row of the tile so that concurrent splits touch different lines
        float* Cf = reinterpret_cast<float*>(C);
        const int rot = (w.zsplit * 5) & (RPW - 1);
        for (int it0 = 0; it0 < RPW; ++it0) {
          const int it = (it0 + rot) & (RPW - 1);
          const int row = ((wave + w.zsplit) & (NW - 1)) * RPW + it, gi = i0 + half * 128 + row;
          if (gi >= p.M) continue;
#pragma unroll
          for (int k = 0; k < GB / 64; ++k) {
            const int col = lane_e + 64 * k, gjc = j0 + col;
            if (gjc < p.N) atomicAdd(Cf + (int64_t)gi * p.ldc + gjc, Ct[row * GB + ((((col >> 2)) ^ (row & 7)) << 2) + (col & 3)]);
          }
        }
      } else {
        f32x2 bv[4] = {splat2(0.f), splat2(0.f), splat2(0.f), splat2(0.f)};
        if (p.bias && lead && gj < p.N) load8f(p.bias + gj, bv);
#pragma unroll 2
        for (int it = 0; it < RPW / RPI; ++it) {
          const int row = wave * RPW + it * RPI + rsub, gi = i0 + half * 128 + row;
          if (gi >= p.M || gj >= p.N) continue;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(Ct + row * GB + ((c0 ^ (row & 7)) << 2));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(Ct + row * GB + (((c0 + 1) ^ (row & 7)) << 2));
          f32x2 v[4] = {f32x2{lo[0], lo[1]} + bv[0], f32x2{lo[2], lo[3]} + bv[1], f32x2{hi[0], hi[1]} + bv[2], f32x2{hi[2], hi[3]} + bv[3]};
          float* cf = Cout + (int64_t)gi * ldo + gj;
          if (rmw) {
            f32x2 o[4];
            load8f(cf, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + o[e];
          }
          store8f(cf, v);
          if (p.colsum) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] = csum[e] + v[e];
          }
        }
      }
      lds_barrier();
    }
    if (p.colsum) {
      // a lane owns 8 fixed columns; lanes that differ only above the LPR bits share them.  Reduce the
      // waves through LDS first: ONE float atomic per column per workgroup.
      float* red = reinterpret_cast<float*>(smem);    // [NW][256]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = csum[e >> 1][e & 1];
        t += __shfl_xor(t, 32, 64);
        if (lane_e < LPR) red[wave * GB + lane_e * 8 + e] = t;
      }
      lds_barrier();
      int tid_e = tid;
      asm volatile("" : "+v"(tid_e));             // (laundered: the colsum address was hoisted to the kernel entry and SPILLED --
                                                  //  one pending scratch store puts a vmcnt(0) into the main loop, see dma16_asm)
      if (tid_e < GB) {
        float t = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) t += red[w8 * GB + tid_e];
        if (j0 + tid_e < p.N) atomicAdd(p.colsum + j0 + tid_e, t);
      }
      lds_barrier();
    }
  }
  if constexpr (!RING5) base = (base + nkt) & (TNST - 1);
  }  // work items
}

// =========================================================================================
// 64 x 64 kernel for SMALL outputs (the IAOG decoder: 768 decoder tokens x 768 features x K = 768, ~160 launches per step; the
// pruned fusion layers of the FCMF step).  On the 128 x 128 kernel such a product is 36 workgroups on 256 CUs, and each of them has to
// pull 2 x 128 x K x 2 bytes through ONE CU's L2 -> LDS path (~50-70 GB/s): 393 KB = 6-8 us of DMA alone, 20-23 us measured.  Here
// the tile is 64 x 64 (4 waves, 2 x 2, 32 x 32 outputs each): 144 workgroups that pull 196 KB each, 64-deep k-tiles (whole 128-byte
// lines per DMA row, half the barriers), the 4-stage ring of the 128 x 128 kernel (three k-tiles in flight), two workgroups per CU.
// K-contiguous operands only (y = x W^T: every forward and dX GEMM of the bf16 step), K % 64 == 0, no split-K.
// =========================================================================================
constexpr int SMALL_T = 64, SMALL_KB = 64, SMALL_TILE_BYTES = SMALL_T * SMALL_KB * 2, SMALL_STAGE_BYTES = 2 * SMALL_TILE_BYTES;
template <typename TC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_small_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NSTAGE x (A tile + B tile) = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + SMALL_T - 1) / SMALL_T;
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {   // XCD-aware order (as in the 128 x 128 kernel): the blocks of one XCD walk consecutive tiles, which share the A row panel
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int i0 = (bid / tiles_n) * SMALL_T, j0 = (bid % tiles_n) * SMALL_T;
  const int nkt = p.K / SMALL_KB;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);
  // an operand tile = 8 pieces of 8 rows x 128 B (the 64-deep image of the big kernels: 16-B chunk c of row r at c ^ ((r >> 1) & 7));
  // wave w brings pieces 2w, 2w + 1 of both tiles
  unsigned va[2], vb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    va[j] = dma_voffset_t<false, 64>(wave * 2 + j, lane, p.lda, i0, p.M);
    vb[j] = dma_voffset_t<false, 64>(wave * 2 + j, lane, p.ldb, j0, p.N);
  }
  auto issue = [&](int t) {
    char* st = smem + (t & (NSTAGE - 1)) * SMALL_STAGE_BYTES + (wave * 2) * 1024;
    const unsigned kk = (unsigned)t * (SMALL_KB * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t)(st + j * 1024), 16, va[j], kk, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t)(st + SMALL_TILE_BYTES + j * 1024), 16, vb[j], kk, 0, 0);
    }
  };
  const int rowl = lane & 15, g4 = lane >> 4;
  const int row_base = rowl * 128 + ((g4 ^ swz_row64(rowl)) << 4);
  const int a_lane = row_base + wm * 32 * 128, b_lane = row_base + wn * 32 * 128;
  f32x4 acc[2][2];   // [j frag][i frag]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nkt) issue(t);
  for (int t = 0; t < nkt; ++t) {
    const int younger = nkt - 1 - t;     // tile t has landed once at most the DMAs of the (<= 2) younger tiles are outstanding
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < nkt) issue(t + NSTAGE - 1);
    const char* la = smem + (t & (NSTAGE - 1)) * SMALL_STAGE_BYTES;
    const char* lb = la + SMALL_TILE_BYTES;
    bf16x8 fa[2][2], fb[2][2];           // [k half][fragment]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        fa[h][f] = *reinterpret_cast<const bf16x8*>(la + (a_lane ^ (h << 6)) + f * 2048);
        fb[h][f] = *reinterpret_cast<const bf16x8*>(lb + (b_lane ^ (h << 6)) + f * 2048);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int fj = 0; fj < 2; ++fj)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
          acc[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[h][fj], fa[h][fi], acc[fj][fi], 0, 0, 0);
  }
  // ---- epilogue: fragment layout (a lane holds 4 consecutive columns of one row), as in the 128 x 128 kernel ----------------
  TC* C = reinterpret_cast<TC*>(p.C);
  TC* AUX = reinterpret_cast<TC*>(p.aux);
  float4 cs[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#pragma unroll
  for (int fi = 0; fi < 2; ++fi) {
    const int i = i0 + wm * 32 + fi * 16 + (lane & 15);
    if (i >= p.M) continue;
#pragma unroll
    for (int fj = 0; fj < 2; ++fj) {
      const int j = j0 + wn * 32 + fj * 16 + (lane >> 4) * 4;
      if (j >= p.N) continue;
      float4 v = make_float4(acc[fj][fi][0], acc[fj][fi][1], acc[fj][fi][2], acc[fj][fi][3]);
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + j);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      const int64_t off = (int64_t)i * p.ldc + j;
      if (p.epilogue != FCMF_EPI_NONE) {
        float4 a = make_float4(0, 0, 0, 0);
        if (p.epilogue == FCMF_EPI_GELU) { if (AUX) Vec4<TC>::store(AUX + off, v); }
        else if (p.epilogue != FCMF_EPI_TANH) a = Vec4<TC>::load(AUX + off);
        if constexpr (sizeof(TC) == 2) {
          v.x = apply_epilogue_fast(v.x, p.epilogue, a.x); v.y = apply_epilogue_fast(v.y, p.epilogue, a.y);
          v.z = apply_epilogue_fast(v.z, p.epilogue, a.z); v.w = apply_epilogue_fast(v.w, p.epilogue, a.w);
        } else {
          v.x = apply_epilogue(v.x, p.epilogue, a.x); v.y = apply_epilogue(v.y, p.epilogue, a.y);
          v.z = apply_epilogue(v.z, p.epilogue, a.z); v.w = apply_epilogue(v.w, p.epilogue, a.w);
        }
      }
      if constexpr (sizeof(TC) == 4) {
        float* cf = reinterpret_cast<float*>(C) + off;
        if (p.accumulate) {
          float4 o = *reinterpret_cast<float4*>(cf);
          o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
          *reinterpret_cast<float4*>(cf) = o;
        } else {
          *reinterpret_cast<float4*>(cf) = v;
        }
      } else {
        Vec4<TC>::store(C + off, v);
      }
      if (p.colsum) { cs[fj].x += v.x; cs[fj].y += v.y; cs[fj].z += v.z; cs[fj].w += v.w; }
    }
  }
  if (p.colsum) {
#pragma unroll
    for (int fj = 0; fj < 2; ++fj) {
      float4 t = cs[fj];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
      }
      const int j = j0 + wn * 32 + fj * 16 + (lane >> 4) * 4;
      if ((lane & 15) == 0 && j < p.N) {
        atomicAdd(p.colsum + j, t.x); atomicAdd(p.colsum + j + 1, t.y); atomicAdd(p.colsum + j + 2, t.z); atomicAdd(p.colsum + j + 3, t.w);
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
  float* colsum;
  int kchunk;   // > 0: blockIdx.z owns the contraction range [z kchunk, (z + 1) kchunk) and ADDS its part into the float32 C with
                // atomics (accumulating outputs of a few tiles: the classifier's 4 x 768 weight gradient walked K = 384 in 24
                // dependent steps on 12 lone workgroups, 85 us for 1.2 MFLOP)
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

  const int kbeg = p.kchunk > 0 ? blockIdx.z * p.kchunk : 0;
  const int kend = p.kchunk > 0 ? min(p.K, kbeg + p.kchunk) : p.K;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int e = tid + 256 * it;  // 0..1023
      int ia, ka, jb, kb;
      if (a_kfast) { ka = e & 15; ia = e >> 4; } else { ia = e & 63; ka = e >> 6; }
      if (b_kfast) { kb = e & 15; jb = e >> 4; } else { jb = e & 63; kb = e >> 6; }
      float va = 0.f, vb = 0.f;
      if (i0 + ia < p.M && k0 + ka < kend) va = to_f32<TI>(A[(int64_t)(i0 + ia) * p.a_si + (int64_t)(k0 + ka) * p.a_sk]);
      if (j0 + jb < p.N && k0 + kb < kend) vb = to_f32<TI>(B[(int64_t)(j0 + jb) * p.b_sj + (int64_t)(k0 + kb) * p.b_sk]);
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
        if constexpr (sizeof(TC) == 4) {
          if (p.kchunk > 0) { atomicAdd(reinterpret_cast<float*>(C) + off, v); continue; }      // (host: accumulate, no epilogue / bias / colsum)
        }
        if (p.accumulate) v += to_f32<TC>(C[off]);
        C[off] = from_f32<TC>(v);
        if (p.colsum) atomicAdd(p.colsum + j, v);
      }
}

// column sums ----------------------------------------------------------------------------
// block = 256 columns x rows_per_block rows: lane -> 4 adjacent columns (8/16-byte loads, a wave
// reads 256 contiguous columns), the 4 waves stride the rows; LDS tree over the waves, then one
// float atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ out, int M, int N,
                                                     int64_t ldx, int rows_per_block, int vec) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && col + 3 < N) {
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {        // four rows in flight per wave (one dependent load per iteration is latency-bound)
      const float4 v0 = Vec4<T>::load(X + (int64_t)r * ldx + col), v1 = Vec4<T>::load(X + (int64_t)(r + 4) * ldx + col);
      const float4 v2 = Vec4<T>::load(X + (int64_t)(r + 8) * ldx + col), v3 = Vec4<T>::load(X + (int64_t)(r + 12) * ldx + col);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < r1; r += 4) {
      const float4 v = Vec4<T>::load(X + (int64_t)r * ldx + col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (col < N) {
    for (int r = r0 + wave; r < r1; r += 4) {
      const T* p = X + (int64_t)r * ldx + col;
      s.x += to_f32<T>(p[0]);
      if (col + 1 < N) s.y += to_f32<T>(p[1]);
      if (col + 2 < N) s.z += to_f32<T>(p[2]);
      if (col + 3 < N) s.w += to_f32<T>(p[3]);
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) {
    const float4 a = red[0][lane], b = red[1][lane], c = red[2][lane], d = red[3][lane];
    atomicAdd(out + col, a.x + b.x + c.x + d.x);
    if (col + 1 < N) atomicAdd(out + col + 1, a.y + b.y + c.y + d.y);
    if (col + 2 < N) atomicAdd(out + col + 2, a.z + b.z + c.z + d.z);
    if (col + 3 < N) atomicAdd(out + col + 3, a.w + b.w + c.w + d.w);
  }
}

// =========================================================================================
// host dispatch
// =========================================================================================
template <bool A_TR, bool B_TR>
static int launch_bf16(const GemmParams& p, int out_dtype, dim3 grid, hipStream_t st) {
  size_t smem = NSTAGE * STAGE_BYTES;
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

// (thin __global__ wrapper: the body holds AMDGPU inline-asm constraints, which the host pass must never see
// inside a kernel template -- it would silently drop the host-side kernel handle)
template <bool A_TR, bool B_TR, typename TC, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile256_kernel(GemmParams p) {
  gemm_bf16_tile256_body<A_TR, B_TR, TC, EPI, 8, 32>(p);
}
// narrow outputs (see the body): N <= 128 -> 4 x 2 waves of 64 x 64; N <= 64 -> 8 x 1 waves of 32 x 64
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile256k64_n128_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, false, bf16_t, FCMF_EPI_NONE, 4, 64, false, false, 2>(p);
}
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile256k64_n64_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, false, bf16_t, FCMF_EPI_NONE, 2, 64, false, false, 1>(p);
}
// up to BATCH_MAX same-shape weight gradients dW_i (+)= dY_i^T X_i in ONE launch: the tiles of all matrices form one work list
// (a layer's 768 x 768 gradient alone is 9 tiles: it filled the chip only through a 28-way split of K, whose partial tiles then
// cost a reduce pass per matrix)
__global__ __launch_bounds__(512, 1) void gemm_bf16_dw_batched_kernel(GemmParams p, BatchPtrs bp) {
  gemm_bf16_tile256_body<true, true, float, FCMF_EPI_NONE, 8, 32, false, true>(p, &bp);
}
template <bool B_TR, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile192_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, B_TR, bf16_t, EPI, 6, 32>(p);
}
// 64-deep k-tiles: both operands K-contiguous (every forward and dX GEMM of the bf16 step), bf16 output, K % 64 == 0
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile256k64_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, false, bf16_t, EPI, 8, 64>(p);
}
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile192k64_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, false, bf16_t, EPI, 6, 64>(p);
}

// e4m3 operands (see the body): K % 128 == 0, bf16 output.  192 x 256 block tile only: with 256 rows the wave's 128 accumulator
// registers + 64 fragment registers (32 bytes per fragment) + DMA offsets exceed 256 VGPRs and the main loop reloads from
// scratch (VMEM operations in the counted-vmcnt DMA pipeline); the 192-row tile (96 accumulators) compiles to 226 VGPRs, no scratch.
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_fp8_tile192_kernel(GemmParams p) {
  gemm_bf16_tile256_body<false, false, bf16_t, EPI, 6, 64, true>(p);
}
template <int EPI>
static void launch_fp8_tile(const GemmParams& p, dim3 grid, hipStream_t st) {
  const size_t smem = (size_t)RING_BYTES + 8 * 4096;
  auto k = gemm_fp8_tile192_kernel<EPI>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
}

template <bool A_TR, bool B_TR, typename TC, int EPI>
static void launch_bf16_tile_typed(const GemmParams& p, dim3 grid, hipStream_t st, int tm, int kb) {
  const size_t smem = (size_t)RING_BYTES + 8 * 4096;   // ring + per-wave transposition slices = 160 KiB
  if constexpr (!A_TR && !B_TR && sizeof(TC) == 2) {
    if (kb == 64) {
      if (tm == 192) {
        auto k = gemm_bf16_tile192k64_kernel<EPI>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
      } else {
        auto k = gemm_bf16_tile256k64_kernel<EPI>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
      }
      return;
    }
  }
  if constexpr (!A_TR && sizeof(TC) == 2) {
    if (tm == 192) {
      auto k = gemm_bf16_tile192_kernel<B_TR, EPI>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
      return;
    }
  }
  auto k = gemm_bf16_tile256_kernel<A_TR, B_TR, TC, EPI>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
}

template <bool A_TR, bool B_TR>
static int launch_bf16_tile(const GemmParams& p, int out_dtype, dim3 grid, hipStream_t st, int tm, int kb) {
  if (out_dtype == FCMF_F32) launch_bf16_tile_typed<A_TR, B_TR, float, FCMF_EPI_NONE>(p, grid, st, 256, 32);
  else switch (p.epilogue) {
    case FCMF_EPI_NONE: launch_bf16_tile_typed<A_TR, B_TR, bf16_t, FCMF_EPI_NONE>(p, grid, st, tm, kb); break;
    case FCMF_EPI_GELU: launch_bf16_tile_typed<A_TR, B_TR, bf16_t, FCMF_EPI_GELU>(p, grid, st, tm, kb); break;
    case FCMF_EPI_DGELU: launch_bf16_tile_typed<A_TR, B_TR, bf16_t, FCMF_EPI_DGELU>(p, grid, st, tm, kb); break;
    default: launch_bf16_tile_typed<A_TR, B_TR, bf16_t, FCMF_EPI_ADD>(p, grid, st, tm, kb); break;
  }
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
// C (+)= sum_z ws[z]: the reduce pass of the workspace split-K (partials were just written: L2 / Infinity Cache hits)
// C (+)= sum_z ws[z] for partial tiles stored in the FRAGMENT layout (the f32 epilogue above): ws[z][tile][wave][fj][fi][lane][4].
// One thread sums the `ksplit` copies of one 16-byte piece (coalesced KiB reads per wave) and writes it to its (row, column).
// cblk > 0: COLUMN-BLOCKED output -- column j of row i lives at C + (j / cblk) * cblk_stride + i * ldc + j % cblk: the per-head weight
// gradients of the IAOG decoder's Attention, dW^T [E, heads * d] = x^T dY, land directly in the parameters' [heads, E, d] layout
// (cblk = d, cblk_stride = E * d, ldc = d) instead of a [heads * d, E] buffer that autograd has to permute-copy per parameter.
__global__ __launch_bounds__(256) void splitk_reduce_frag_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                                                 int64_t ldc, int ksplit, int tiles, int tiles_n, int accumulate,
                                                                 int cblk, int64_t cblk_stride) {
  const int64_t total = (int64_t)tiles * (GB * GB / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int tile = (int)(i >> 14), r = (int)(i & 16383);          // 16384 pieces of 4 floats per 256 x 256 tile
    const int lane = r & 63, frag = (r >> 6) & 31, wave = r >> 11;
    const int fj = frag >> 3, fi = frag & 7, wm = wave >> 2, wn = wave & 3;
    const int row = (tile / tiles_n) * GB + wm * 128 + fi * 16 + (lane & 15);
    const int col = (tile % tiles_n) * GB + wn * 64 + fj * 16 + (lane >> 4) * 4;
    if (row >= M || col >= N) continue;
    float* c = cblk ? C + (int64_t)(col / cblk) * cblk_stride + (int64_t)row * ldc + (col % cblk) : C + (int64_t)row * ldc + col;
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = 0; z < ksplit; ++z) s += *reinterpret_cast<const f32x4*>(ws + ((int64_t)z * tiles + tile) * (GB * GB) + (int64_t)r * 4);
    *reinterpret_cast<f32x4*>(c) = s;
  }
}

// the same over the tiles of a batched launch: tile t belongs to matrix t / tiles_per_mat
__global__ __launch_bounds__(256) void splitk_reduce_frag_batched_kernel(const float* __restrict__ ws, BatchPtrs bp, int tiles_per_mat, int M,
                                                                         int N, int64_t ldc, int ksplit, int tiles, int tiles_n,
                                                                         int accumulate) {
  const int64_t total = (int64_t)tiles * (GB * GB / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int tile = (int)(i >> 14), r = (int)(i & 16383);
    const int b = tile / tiles_per_mat, local = tile - b * tiles_per_mat;
    const int lane = r & 63, frag = (r >> 6) & 31, wave = r >> 11;
    const int fj = frag >> 3, fi = frag & 7, wm = wave >> 2, wn = wave & 3;
    const int row = (local / tiles_n) * GB + wm * 128 + fi * 16 + (lane & 15);
    const int col = (local % tiles_n) * GB + wn * 64 + fj * 16 + (lane >> 4) * 4;
    if (row >= M || col >= N) continue;
    float* c = reinterpret_cast<float*>(bp.C[b]) + (int64_t)row * ldc + col;
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = 0; z < ksplit; ++z) s += *reinterpret_cast<const f32x4*>(ws + ((int64_t)z * tiles + tile) * (GB * GB) + (int64_t)r * 4);
    *reinterpret_cast<f32x4*>(c) = s;
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                                            int64_t ldc, int ksplit, int accumulate) {
  const int64_t total4 = (int64_t)M * N / 4;           // N % 8 == 0 on this path
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4, row = e / N, col = e - row * N;
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(C + row * ldc + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = 0; z < ksplit; ++z) s += *reinterpret_cast<const f32x4*>(ws + (int64_t)z * M * N + e);
    *reinterpret_cast<f32x4*>(C + row * ldc + col) = s;
  }
}

// ---- explicit GEMM context ------------------------------------------------------------------------------------------
// Everything fcmf_gemm uses beyond its arguments lives in a caller-owned context (include/fcmf_hip.h): the split-K
// workspace, the tuning knobs of benchmarks / tests and the name of the kernel the last call dispatched.  The library
// itself holds no mutable process-global state; a NULL context means defaults (no workspace: float atomics for split-K).
struct fcmf_gemm_ctx {
  float* ws = nullptr;          // split-K partial tiles (device memory owned by the caller)
  int64_t ws_bytes = 0;
  int force_tile = 0;           // 0 = heuristic, 128 / 192 / 256 = forced kernel
  int kb64 = 1;                 // 64-deep k-tiles where they apply
  int num_cus = 256;            // workgroups of the persistent kernels (MI355X: 8 XCDs x 32 CUs, one 160-KiB-LDS workgroup per CU)
  int64_t nt_min_bytes = 0;     // bf16 outputs of at least this many bytes leave with nontemporal stores
  char last_kernel[96] = "";
};
static const fcmf_gemm_ctx g_default_ctx;    // (const: the defaults of a NULL context)

extern "C" int fcmf_gemm_ctx_create(fcmf_gemm_ctx** ctx) {
  if (!ctx) return FCMF_ERR_ARG;
  *ctx = new fcmf_gemm_ctx();
  return FCMF_OK;
}
extern "C" int fcmf_gemm_ctx_destroy(fcmf_gemm_ctx* ctx) {
  delete ctx;
  return FCMF_OK;
}
extern "C" int fcmf_gemm_ctx_set_workspace(fcmf_gemm_ctx* ctx, void* ptr, int64_t bytes) {
  if (!ctx || bytes < 0 || (!ptr && bytes > 0)) return FCMF_ERR_ARG;
  ctx->ws = reinterpret_cast<float*>(ptr);
  ctx->ws_bytes = ptr ? bytes : 0;
  return FCMF_OK;
}
extern "C" int fcmf_gemm_ctx_tune(fcmf_gemm_ctx* ctx, int force_tile, int kb, int num_cus, int64_t nt_min_bytes) {
  if (!ctx) return FCMF_ERR_ARG;
  if (force_tile >= 0) {
    if (force_tile != 0 && force_tile != 128 && force_tile != 192 && force_tile != 256) return FCMF_ERR_ARG;
    ctx->force_tile = force_tile;
  }
  if (kb >= 0) {
    if (kb != 32 && kb != 64) return FCMF_ERR_ARG;
    ctx->kb64 = kb == 64;
  }
  if (num_cus >= 0) {
    if (num_cus < 8 || num_cus > 256) return FCMF_ERR_ARG;
    ctx->num_cus = num_cus;
  }
  if (nt_min_bytes >= 0) ctx->nt_min_bytes = nt_min_bytes;
  return FCMF_OK;
}
extern "C" const char* fcmf_gemm_ctx_last_kernel(const fcmf_gemm_ctx* ctx) { return ctx ? ctx->last_kernel : ""; }

struct ConvGeom { int C, logC, Hp, Wp, Ho, Wo, kw, stride; int64_t in_bytes; int logP; };

// shapes of the narrow-output layouts of the persistent kernel (N <= 64: 8 x 1 waves of 32 rows; N <= 128: 4 x 2 waves of 64 rows)
static bool narrow_shape(int M, int N, int K) { return N >= 32 && N <= 128 && N % 8 == 0 && K % 64 == 0 && M >= 8192; }

// rows per block of the statistics fcmf_gemm_colstats / fcmf_conv_gemm_colstats emit for an [M, N] output contracted over K:
// 128 (256-column tiles), 256 (narrow layouts: one block per tile), 0 where no kernel emits them (the GEMM call then returns FCMF_ERR_UNSUPPORTED)
extern "C" int fcmf_gemm_colstats_block_rows(const fcmf_gemm_ctx* ctx, int M, int N, int K) {
  const fcmf_gemm_ctx& cfg = ctx ? *ctx : g_default_ctx;
  if (cfg.force_tile == 0 && cfg.kb64 && narrow_shape(M, N, K)) return 256;
  return (M >= 256 && N >= 256 && N % 8 == 0) ? 128 : 0;
}

static int gemm_impl(fcmf_gemm_ctx* ctx, const void* A, const void* B, void* C, const float* bias, void* aux, float* colsum,
                     int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b, int in_dtype,
                     int out_dtype, int epilogue, int accumulate, void* stream, const ConvGeom* cv, float* colstats = nullptr,
                     int cblk = 0, int64_t cblk_stride = 0) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return FCMF_ERR_ARG;
  const fcmf_gemm_ctx& cfg = ctx ? *ctx : g_default_ctx;
  char name_sink[96];
  char* const last_kernel = ctx ? ctx->last_kernel : name_sink;
  constexpr size_t NAME = sizeof(name_sink);
  if (M == 0 || N == 0) return FCMF_OK;
  if (accumulate && out_dtype != FCMF_F32) return FCMF_ERR_ARG;
  if ((epilogue == FCMF_EPI_DGELU || epilogue == FCMF_EPI_DTANH || epilogue == FCMF_EPI_ADD) && !aux) return FCMF_ERR_ARG;
  if (colsum && accumulate) return FCMF_ERR_ARG;   // column sums are those of the final C, not of split-K partials
  if (in_dtype != FCMF_F32 && in_dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  if (out_dtype != FCMF_F32 && out_dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // contiguous extents that the 16-byte loaders walk must be multiples of 8 elements
  const int a_contig = trans_a ? M : K, b_contig = trans_b ? N : K;
  const bool fast = in_dtype == FCMF_BF16 && al16(A) && al16(B) && al16(C) && (!aux || al16(aux)) &&
                    (!bias || al16(bias)) && (lda % 8 == 0) && (ldb % 8 == 0) && (a_contig % 8 == 0) &&
                    (b_contig % 8 == 0) && (N % 4 == 0) && (ldc % 4 == 0) && K > 0 &&
                    // K-contiguous operands cannot zero-fill a partial k-tile; 32-bit byte offsets
                    (trans_a || K % BK == 0) && (trans_b || K % BK == 0) &&
                    (cv || ((int64_t)(trans_a ? K : M) * lda) < (1ll << 30)) && (((int64_t)(trans_b ? K : N) * ldb) < (1ll << 30));   // (cv: A's extent is the activation's, checked by the caller)
  if (fast) {
    GemmParams p{A, B, C, bias, aux, M, N, K, lda, ldb, ldc, epilogue, accumulate, 1, 0, 0, 0, 0, colsum, 0, 0, nullptr, 0, nullptr, nullptr,
                 0, 0, 0, 0, 0, 0, 0, 0, 0, colstats, 0, 0};
    if (cv) { p.cv_C = cv->C; p.cv_logC = cv->logC; p.cv_Hp = cv->Hp; p.cv_Wp = cv->Wp; p.cv_Ho = cv->Ho; p.cv_Wo = cv->Wo; p.cv_kw = cv->kw; p.cv_inv_kw = (65536 + cv->kw - 1) / cv->kw; p.cv_stride = cv->stride; p.cv_logP = cv->logP; }
    // bytes addressable through each operand: (rows - 1) * ld + contiguous extent
    p.a_bytes = cv ? (unsigned)cv->in_bytes : (unsigned)((((int64_t)(trans_a ? K : M) - 1) * lda + (trans_a ? M : K)) * 2);
    p.b_bytes = (unsigned)((((int64_t)(trans_b ? K : N) - 1) * ldb + (trans_b ? N : K)) * 2);
    const int64_t c_extent = (((int64_t)M - 1) * ldc + N) * 2;
    p.c_bytes = (unsigned)(c_extent < (1ll << 31) ? c_extent : 0);
    const int nk = (K + BK - 1) / BK;
    // 256x256 persistent ping-pong kernel for the big problems; 128x128 for small / ragged / narrow outputs,
    // f32 outputs with an activation epilogue and tanh epilogues (poolers)
    const bool tile_ok = (N % 8 == 0) && (ldc % 8 == 0) && (out_dtype == FCMF_BF16 || epilogue == FCMF_EPI_NONE) &&
                         epilogue != FCMF_EPI_TANH && epilogue != FCMF_EPI_DTANH &&
                         (out_dtype == FCMF_F32 || c_extent < (1ll << 31));
    // (weight gradients: the 256x256 kernel's row-wise f32 epilogue / 256-byte atomics beat the 128x128 kernel's
    // fragment-layout atomics from K = 1024 up -- 25 vs 97 us at 768x768x2048)
    // f32 outputs WITHOUT accumulate (a fresh weight-gradient buffer: no zero fill needed) may still split K when the context
    // owns a workspace: the reduce pass then writes the sum instead of adding it
    const bool splittable = epilogue == FCMF_EPI_NONE && out_dtype == FCMF_F32 && !colsum &&
                            (accumulate || (cfg.ws != nullptr && !bias));
    bool large = tile_ok && M >= 256 && N >= 256 &&
                 ((int64_t)M * N >= (int64_t)256 * 256 * 64 || (splittable && K >= 512));
    if (cfg.force_tile == 128) large = false;
    if (cfg.force_tile == 256 || cfg.force_tile == 192) large = tile_ok;
    if (cblk) {      // column-blocked output: written by the fragment-layout reduce pass only (f32, split K through the workspace)
      if (!(tile_ok && splittable && out_dtype == FCMF_F32 && cfg.ws && M >= 256 && N >= 256 && cblk % 4 == 0 && N % cblk == 0)) return FCMF_ERR_UNSUPPORTED;
      large = true;
    }
    // narrow outputs with many rows (the trunk's 64- / 128-channel convolutions): the persistent kernel's 4 x 2 / 8 x 1 wave layouts
    const bool narrow = cfg.force_tile == 0 && cfg.kb64 && tile_ok && !colsum && !aux && !accumulate && !trans_a && !trans_b &&
                        out_dtype == FCMF_BF16 && epilogue == FCMF_EPI_NONE && narrow_shape(M, N, K);
    if (narrow) {
      p.ksplit = 1; p.ktiles_per_split = K / 64;
      p.tiles = (M + GB - 1) / GB; p.total_items = p.tiles;
      p.ws = nullptr;
      p.nt_out = (int64_t)M * N * 2 >= cfg.nt_min_bytes;
      const dim3 grid(p.total_items < cfg.num_cus ? p.total_items : cfg.num_cus);
      const size_t smem = (size_t)RING_BYTES + 8 * 4096;
      snprintf(last_kernel, NAME, "gemm_bf16_tile256k64_n%d_kernel", N <= 64 ? 64 : 128);
      if (N <= 64) {
        auto k = gemm_bf16_tile256k64_n64_kernel;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
      } else {
        auto k = gemm_bf16_tile256k64_n128_kernel;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(512), smem, st, p);
      }
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
    if (colstats) {       // block statistics exist in the 256-row bf16 kernels only: the caller falls back to a statistics pass
      // (they were built into the 128 x 128 kernel too -- the trunk's 64 / 128-channel layers -- and measured: its epilogue, a
      //  fragment-layout one, pays as much for the rounding, the 128 shuffles and two barriers as the separate pass costs)
      if (!(tile_ok && M >= 256 && N >= 256 && out_dtype == FCMF_BF16 && !trans_a && !trans_b && epilogue == FCMF_EPI_NONE &&
            !accumulate && !aux))
        return FCMF_ERR_UNSUPPORTED;
      large = true;
    }
    if (large) {
      const int slots = cfg.num_cus;
      // block tile rows: 256, or 192 where that removes a nearly empty last round (cost model: rounds x
      // (k-loop time scaled by the tile rows + a fixed per-tile cost of ~8 k-tiles))
      int tm = 256;
      if (!trans_a && out_dtype == FCMF_BF16 && !accumulate) {
        auto cost = [&](int rows) {
          const int64_t t = (int64_t)((M + rows - 1) / rows) * ((N + GB - 1) / GB);
          return (double)((t + slots - 1) / slots) * (nk * (rows / 256.0) + 8.0);
        };
        if (cost(192) < 0.97 * cost(256) && !colstats) tm = 192;
      }
      if (cfg.force_tile == 192 && !trans_a && out_dtype == FCMF_BF16 && !accumulate && !colstats) tm = 192;
      const int tiles_l = ((M + tm - 1) / tm) * ((N + GB - 1) / GB);
      // 64-deep k-tiles where both operands are K-contiguous (whole-line DMA), the output is bf16 and K allows it
      // (weight gradients -- token-major operands, 512-B DMA rows already -- measured 4-8 % SLOWER on 64-deep k-tiles)
      const int kb = (cfg.kb64 && !trans_a && !trans_b && out_dtype == FCMF_BF16 && !accumulate && K % 64 == 0) ? 64 : 32;
      const int nk = (K + kb - 1) / kb;      // (shadows the 32-deep count above: the kernel counts k-tiles of ITS depth)
      int ksplit = 1;
      if (splittable && tiles_l < slots) {
        ksplit = slots / tiles_l;
        const int min_kt = (nk * (kb / 32) >= 64 ? 8 : 6) / (kb / 32);   // >= 256 (192) k per work item: short contractions (K = 768 rows) split 4 ways
        if (ksplit > nk / min_kt) ksplit = nk / min_kt > 0 ? nk / min_kt : 1;
        if (ksplit > 64) ksplit = 64;
      }
      p.ktiles_per_split = (nk + ksplit - 1) / ksplit;
      p.ksplit = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
      p.tiles = tiles_l;
      p.total_items = tiles_l * p.ksplit;
      dim3 grid(p.total_items < slots ? p.total_items : slots);
      p.ws = nullptr;
      p.nt_out = out_dtype == FCMF_BF16 && (int64_t)M * N * 2 >= cfg.nt_min_bytes;
      // (partial tiles are whole 256 x 256 fragment-layout tiles when the GEMM has neither bias nor column sums: size by tiles)
      const bool frag_ws = !bias && !colsum;
      const int64_t ws_need = frag_ws ? (int64_t)p.ksplit * tiles_l * GB * GB * 4 : (int64_t)p.ksplit * M * N * 4;
      if (p.ksplit > 1 && out_dtype == FCMF_F32 && cfg.ws && cfg.ws_bytes >= ws_need) p.ws = cfg.ws;
      if (p.ksplit > 1 && !accumulate && !p.ws) {       // (no workspace of that size: float atomics would need a zeroed C)
        p.ksplit = 1; p.ktiles_per_split = nk; p.total_items = tiles_l;
      }
      if (cblk && !(p.ksplit > 1 && p.ws && frag_ws)) return FCMF_ERR_UNSUPPORTED;   // (only the reduce pass knows the blocked layout)
      {
        static const char* const epi_names[] = {"NONE", "GELU", "TANH", "DGELU", "DTANH", "ADD"};
        if (kb == 64) snprintf(last_kernel, NAME, "gemm_bf16_tile%dk64_kernel<%s>", tm, epi_names[epilogue]);
        else if (tm == 192) snprintf(last_kernel, NAME, "gemm_bf16_tile192_kernel<%d,%s>", trans_b, epi_names[epilogue]);
        else snprintf(last_kernel, NAME, "gemm_bf16_tile256_kernel<%d,%d,%s,%s>", trans_a, trans_b, out_dtype == FCMF_F32 ? "f32" : "bf16", epi_names[epilogue]);
      }
      int rc;
      if (!trans_a && !trans_b) rc = launch_bf16_tile<false, false>(p, out_dtype, grid, st, tm, kb);
      else if (!trans_a && trans_b) rc = launch_bf16_tile<false, true>(p, out_dtype, grid, st, tm, kb);
      else if (trans_a && !trans_b) rc = launch_bf16_tile<true, false>(p, out_dtype, grid, st, tm, kb);
      else rc = launch_bf16_tile<true, true>(p, out_dtype, grid, st, tm, kb);
      if (rc == FCMF_OK && p.ws) {
        if (frag_ws) {
          const int64_t total4 = (int64_t)tiles_l * (GB * GB / 4);
          const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
          hipLaunchKernelGGL(splitk_reduce_frag_kernel, dim3(blocks), dim3(256), 0, st, p.ws, reinterpret_cast<float*>(C), M, N, ldc,
                             p.ksplit, tiles_l, (N + GB - 1) / GB, accumulate, cblk, cblk_stride);
        } else {
          const int64_t total4 = (int64_t)M * N / 4;
          const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.ws, reinterpret_cast<float*>(C), M, N, ldc,
                             p.ksplit, accumulate);
        }
        FCMF_CHECK_LAUNCH();
      }
      return rc;
    }
    if (colstats) return FCMF_ERR_UNSUPPORTED;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // small outputs (128 x 128 tiles would leave more than half of the CUs without a workgroup): the 64 x 64 kernel
    if (cfg.force_tile == 0 && cfg.kb64 && !trans_a && !trans_b && !cv && K % 64 == 0 && tiles <= cfg.num_cus / 2 &&
        (int64_t)M * N >= 64 * 64 && !(accumulate && epilogue == FCMF_EPI_NONE && nk >= 64)) {      // (long-K accumulations: split-K below)
      p.ksplit = 1; p.ktiles_per_split = K / 64;
      const dim3 grid(((M + SMALL_T - 1) / SMALL_T) * ((N + SMALL_T - 1) / SMALL_T));
      const size_t smem = (size_t)NSTAGE * SMALL_STAGE_BYTES;
      snprintf(last_kernel, NAME, "gemm_bf16_small_kernel<%s>", out_dtype == FCMF_F32 ? "f32" : "bf16");
      if (out_dtype == FCMF_F32) {
        auto k = gemm_bf16_small_kernel<float>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
      } else {
        auto k = gemm_bf16_small_kernel<bf16_t>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
      }
      FCMF_CHECK_LAUNCH();
      return FCMF_OK;
    }
    int ksplit = 1;
    // split K only where the output grid cannot fill the chip and C is an f32 accumulator
    // (weight gradients: K = number of tokens).
    // (unsplit, 36 lone workgroups of 24 k-tiles were measured SLOWER than 6-way split + float atomics on the IAOG
    // decoder's 768 x 768 x 768-row weight gradients: a single workgroup per CU has nothing to overlap its latencies with)
    if (accumulate && epilogue == FCMF_EPI_NONE && tiles < 512) {
      ksplit = 512 / tiles;   // one round of <= 512 resident blocks (256 CUs x 2)
      if (ksplit > nk / 4) ksplit = nk / 4 > 0 ? nk / 4 : 1;
      if (ksplit > 32) ksplit = 32;
    }
    p.ksplit = ksplit;
    p.ktiles_per_split = (nk + ksplit - 1) / ksplit;
    p.ksplit = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    dim3 grid(tiles, 1, p.ksplit);
    snprintf(last_kernel, NAME, "gemm_bf16_kernel<%d,%d,%s>", trans_a, trans_b, out_dtype == FCMF_F32 ? "f32" : "bf16");
    if (!trans_a && !trans_b) return launch_bf16<false, false>(p, out_dtype, grid, st);
    if (!trans_a && trans_b) return launch_bf16<false, true>(p, out_dtype, grid, st);
    if (trans_a && !trans_b) return launch_bf16<true, false>(p, out_dtype, grid, st);
    return launch_bf16<true, true>(p, out_dtype, grid, st);
  }
  if (cv || colstats) return FCMF_ERR_UNSUPPORTED;      // (the any-stride kernel has no implicit-convolution addressing, no block statistics)
  GenericParams g{A, B, C, bias, aux, M, N, K,
                  trans_a ? 1 : lda, trans_a ? lda : 1, trans_b ? 1 : ldb, trans_b ? ldb : 1, ldc,
                  epilogue, accumulate, colsum, 0};
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  if (accumulate && out_dtype == FCMF_F32 && epilogue == FCMF_EPI_NONE && !bias && !colsum && grid.x * grid.y < 128 && K >= 128) {
    int ks = 256 / (int)(grid.x * grid.y);               // about one workgroup per CU
    if (ks > K / 32) ks = K / 32;                        // >= 32 contraction steps each
    if (ks > 1) {
      g.kchunk = ((K + ks - 1) / ks + 15) / 16 * 16;     // multiple of the kernel's 16-deep k step
      grid.z = (K + g.kchunk - 1) / g.kchunk;
    }
  }
  snprintf(last_kernel, NAME, "gemm_generic_kernel");
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

// ---- e4m3 GEMM ---------------------------------------------------------------------------------------------------------
extern "C" int fcmf_gemm_fp8(fcmf_gemm_ctx* ctx, const void* A, const float* sa, const void* B, const float* sb, void* C,
                             const float* bias, void* aux, float* colsum, int M, int N, int K, int64_t lda, int64_t ldb,
                             int64_t ldc, int epilogue, void* stream) {
  if (!A || !B || !C || !sa || !sb || M < 0 || N < 0 || K <= 0) return FCMF_ERR_ARG;
  if (M == 0 || N == 0) return FCMF_OK;
  if ((epilogue == FCMF_EPI_DGELU || epilogue == FCMF_EPI_ADD) && !aux) return FCMF_ERR_ARG;
  if (epilogue == FCMF_EPI_TANH || epilogue == FCMF_EPI_DTANH) return FCMF_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int64_t c_extent = (((int64_t)M - 1) * ldc + N) * 2;
  // 128-byte k-tiles of whole cache lines; 16-byte DMA chunks; the persistent kernels' 8-column row pieces
  if (K % 128 || lda % 16 || ldb % 16 || N % 8 || ldc % 8 || !al16(A) || !al16(B) || !al16(C) || (aux && !al16(aux)) ||
      (bias && !al16(bias)) || !al16(sb) || c_extent >= (1ll << 31) || (int64_t)M * lda >= (1ll << 31) ||
      (int64_t)N * ldb >= (1ll << 31) || M < 256 || N < 256)
    return FCMF_ERR_UNSUPPORTED;
  const fcmf_gemm_ctx& cfg = ctx ? *ctx : g_default_ctx;
  GemmParams p{A, B, C, bias, aux, M, N, K, lda, ldb, ldc, epilogue, 0, 1, 0, 0, 0, 0, colsum, 0, 0, nullptr, 0, sa, sb, 0, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, 0};
  p.a_bytes = (unsigned)(((int64_t)M - 1) * lda + K);
  p.b_bytes = (unsigned)(((int64_t)N - 1) * ldb + K);
  p.c_bytes = (unsigned)c_extent;
  const int slots = cfg.num_cus, nk = K / 128;
  constexpr int tm = 192;
  p.tiles = ((M + tm - 1) / tm) * ((N + GB - 1) / GB);
  p.ktiles_per_split = nk;
  p.ksplit = 1;
  p.total_items = p.tiles;
  p.nt_out = (int64_t)M * N * 2 >= cfg.nt_min_bytes;
  dim3 grid(p.total_items < slots ? p.total_items : slots);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ctx) {
    static const char* const epi_names[] = {"NONE", "GELU", "TANH", "DGELU", "DTANH", "ADD"};
    snprintf(ctx->last_kernel, sizeof ctx->last_kernel, "gemm_fp8_tile%d_kernel<%s>", tm, epi_names[epilogue]);
  }
  switch (epilogue) {
    case FCMF_EPI_NONE: launch_fp8_tile<FCMF_EPI_NONE>(p, grid, st); break;
    case FCMF_EPI_GELU: launch_fp8_tile<FCMF_EPI_GELU>(p, grid, st); break;
    case FCMF_EPI_DGELU: launch_fp8_tile<FCMF_EPI_DGELU>(p, grid, st); break;
    default: launch_fp8_tile<FCMF_EPI_ADD>(p, grid, st); break;
  }
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

// x [rows, K] (bf16 / f32, row stride ldx) -> q [rows, K] e4m3 (row stride ldq bytes) with ONE scale per row:
// scale[r] = amax_r / 448 (1 for an all-zero row), q = round_to_nearest_even(x / scale[r]).  One wave per row, 8 elements per
// lane and trip; the row is read twice (the second pass hits L2).
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const T* __restrict__ x, int64_t ldx, unsigned char* __restrict__ q,
                                                             int64_t ldq, float* __restrict__ scale, int rows, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ldx;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const float4 a = Vec4<T>::load(xr + k), b = Vec4<T>::load(xr + k + 4);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  unsigned char* qr = q + (int64_t)row * ldq;
  for (int k = lane * 8; k < K; k += 512) {
    const float4 a = Vec4<T>::load(xr + k), b = Vec4<T>::load(xr + k + 4);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.x * inv, a.y * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.z * inv, a.w * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.x * inv, b.y * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.z * inv, b.w * inv, hi, true);
    *reinterpret_cast<int2*>(qr + k) = make_int2(lo, hi);
  }
}

// bf16 rows of up to 512 * NV elements held in registers: the row is read ONCE (16-byte loads, all in flight), then amax -> scale
// -> convert -> 8-byte stores.  (The two-pass kernel above re-reads the row; at [24576, 4096] that second pass misses L2.)
template <int NV>
__global__ __launch_bounds__(256) void quant_fp8_rows_reg_kernel(const bf16_t* __restrict__ x, int64_t ldx, unsigned char* __restrict__ q,
                                                                 int64_t ldq, float* __restrict__ scale, int rows, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  bf16x8 v[NV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int k = (lane + 64 * i) * 8;
    if (k < K) v[i] = *reinterpret_cast<const bf16x8*>(xr + k);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if ((lane + 64 * i) * 8 < K) {
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[i][e]));
    }
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  unsigned char* qr = q + (int64_t)row * ldq;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int k = (lane + 64 * i) * 8;
    if (k < K) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][0] * inv, (float)v[i][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][2] * inv, (float)v[i][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][4] * inv, (float)v[i][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][6] * inv, (float)v[i][7] * inv, hi, true);
      *reinterpret_cast<int2*>(qr + k) = make_int2(lo, hi);
    }
  }
}

extern "C" int fcmf_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, int dtype,
                                   void* stream) {
  if (!x || !q || !scale || rows < 0 || K <= 0) return FCMF_ERR_ARG;
  if (K % 8 || ldx % 8 || ldq % 8 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(q) & 7)) return FCMF_ERR_UNSUPPORTED;
  if (rows == 0) return FCMF_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((rows + 3) / 4);
#define FCMF_QREG(NV) hipLaunchKernelGGL((quant_fp8_rows_reg_kernel<NV>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (unsigned char*)q, ldq, scale, rows, K)
  if (dtype == FCMF_BF16 && K <= 4096) {
    if (K <= 512) FCMF_QREG(1); else if (K <= 1024) FCMF_QREG(2); else if (K <= 2048) FCMF_QREG(4); else FCMF_QREG(8);
  } else if (dtype == FCMF_BF16) hipLaunchKernelGGL((quant_fp8_rows_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (unsigned char*)q, ldq, scale, rows, K);
#undef FCMF_QREG
  else if (dtype == FCMF_F32) hipLaunchKernelGGL((quant_fp8_rows_kernel<float>), grid, dim3(256), 0, st, (const float*)x, ldx, (unsigned char*)q, ldq, scale, rows, K);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}

extern "C" int fcmf_gemm(fcmf_gemm_ctx* ctx, const void* A, const void* B, void* C, const float* bias, void* aux, float* colsum,
                         int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b, int in_dtype,
                         int out_dtype, int epilogue, int accumulate, void* stream) {
  return gemm_impl(ctx, A, B, C, bias, aux, colsum, M, N, K, lda, ldb, ldc, trans_a, trans_b, in_dtype, out_dtype, epilogue,
                   accumulate, stream, nullptr);
}

extern "C" int fcmf_gemm_colblocks(fcmf_gemm_ctx* ctx, const void* A, const void* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                                   int64_t ldc, int trans_a, int trans_b, int col_block, int64_t col_block_stride, int accumulate,
                                   void* stream) {
  if (col_block <= 0 || col_block_stride <= 0 || ldc < col_block) return FCMF_ERR_ARG;
  return gemm_impl(ctx, A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, trans_a, trans_b, FCMF_BF16, FCMF_F32, FCMF_EPI_NONE,
                   accumulate, stream, nullptr, nullptr, col_block, col_block_stride);
}

// ---- batched weight gradients -----------------------------------------------------------------------------------------
// C_i [M, N] float32 (+)= A_i^T B_i for `count` same-shape problems (A_i = dY_i [K, M], B_i = X_i [K, N], bf16, K = tokens): the
// weight gradients of the layers of an encoder, queued during the backward pass and multiplied together.  Chunks of up to
// BATCH_MAX matrices form one work list of (matrix, tile, k-split) items for the persistent 256 x 256 kernel; the split factor
// is chosen for whole rounds of the chip (a 768 x 768 gradient alone is 9 tiles: alone it needed a 28-way split and a 64 MB
// partial-tile round trip).  Shapes the persistent kernel does not take, and count == 1, run as `count` fcmf_gemm calls.
extern "C" int fcmf_gemm_dw_batched(fcmf_gemm_ctx* ctx, int count, const void* const* A, const void* const* B, void* const* C, int M, int N,
                                    int K, int64_t lda, int64_t ldb, int64_t ldc, int accumulate, void* stream) {
  if (count < 0 || !A || !B || !C || M < 0 || N < 0 || K < 0) return FCMF_ERR_ARG;
  for (int i = 0; i < count; ++i)
    if (!A[i] || !B[i] || !C[i]) return FCMF_ERR_ARG;
  if (count == 0 || M == 0 || N == 0) return FCMF_OK;
  const fcmf_gemm_ctx& cfg = ctx ? *ctx : g_default_ctx;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = count > 1 && cfg.force_tile != 128 && K > 0 && M >= 256 && N >= 256 && lda % 8 == 0 && ldb % 8 == 0 && M % 8 == 0 && N % 8 == 0 &&
            ldc % 8 == 0 && (int64_t)K * lda < (1ll << 30) && (int64_t)K * ldb < (1ll << 30);
  for (int i = 0; ok && i < count; ++i) ok = al16(A[i]) && al16(B[i]) && al16(C[i]);
  if (!ok) {
    for (int i = 0; i < count; ++i) {
      const int rc = gemm_impl(ctx, A[i], B[i], C[i], nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, 1, 1, FCMF_BF16, FCMF_F32, FCMF_EPI_NONE,
                               accumulate, stream, nullptr);
      if (rc != FCMF_OK) return rc;
    }
    return FCMF_OK;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slots = cfg.num_cus, tpm = ((M + GB - 1) / GB) * ((N + GB - 1) / GB), nk = (K + 31) / 32;
  const int chunks = (count + BATCH_MAX - 1) / BATCH_MAX;
  const size_t smem = (size_t)RING_BYTES + 8 * 4096;
  auto kern = gemm_bf16_dw_batched_kernel;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  for (int c = 0, first = 0; c < chunks; ++c) {
    const int n = (count - first + (chunks - c) - 1) / (chunks - c);          // near-equal chunks
    BatchPtrs bp{};
    for (int i = 0; i < n; ++i) { bp.A[i] = A[first + i]; bp.B[i] = B[first + i]; bp.C[i] = C[first + i]; }
    const int tiles = n * tpm;
    // k-split for whole rounds: the smallest split whose last round is at least 92 % full (else the fullest), each split at least
    // 8 k-tiles deep, partial tiles within the context's workspace
    int best = 1;
    double best_eff = 0.0;
    const int64_t ws_tiles = cfg.ws ? cfg.ws_bytes / ((int64_t)GB * GB * 4) : 0;
    for (int sp = 1; sp <= 16; ++sp) {
      if (sp > 1 && ((int64_t)sp * tiles > ws_tiles || nk / sp < 8)) break;
      const int64_t items = (int64_t)tiles * sp, rounds = (items + slots - 1) / slots;
      const double eff = (double)items / (double)(rounds * slots);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
      if (eff >= 0.92) { best = sp; break; }
    }
    GemmParams p{};
    p.A = bp.A[0]; p.B = bp.B[0]; p.C = bp.C[0];
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.epilogue = FCMF_EPI_NONE; p.accumulate = accumulate;
    p.a_bytes = (unsigned)((((int64_t)K - 1) * lda + M) * 2);
    p.b_bytes = (unsigned)((((int64_t)K - 1) * ldb + N) * 2);
    p.ktiles_per_split = (nk + best - 1) / best;
    p.ksplit = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    p.tiles = tiles; p.tiles_per_mat = tpm;
    p.total_items = tiles * p.ksplit;
    p.ws = p.ksplit > 1 ? cfg.ws : nullptr;
    const dim3 grid(p.total_items < slots ? p.total_items : slots);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, p, bp);
    if (p.ws) {
      const int64_t total4 = (int64_t)tiles * (GB * GB / 4);
      const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
      hipLaunchKernelGGL(splitk_reduce_frag_batched_kernel, dim3(blocks), dim3(256), 0, st, p.ws, bp, tpm, M, N, ldc, p.ksplit, tiles,
                         (N + GB - 1) / GB, accumulate);
    }
    FCMF_CHECK_LAUNCH();
    first += n;
  }
  if (ctx) snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "gemm_bf16_dw_batched_kernel");
  return FCMF_OK;
}

// Implicit-GEMM convolution on the MFMA GEMM kernels: y[(n, oy, ox), co] = sum_{ky, kx, c} x[n, oy s + ky, ox s + kx, c] w[co, (ky, kx, c)].
// x is the NHWC input INCLUDING its zero border ([n, Hp, Wp, C] with Hp = H + 2 pad: the producing kernel writes the interior,
// fcmf_bn_apply with `pad`); no patch matrix exists: the LDS-DMA of A's k-tile t reads the tap (ky, kx) of every row's
// receptive field straight from the activation (per-lane offset = the field's origin, fixed per work item; the tap is a
// wave-uniform scalar offset).
static int conv_gemm_impl(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, float* colstats, int n, int Hp, int Wp, int C, int Ho,
                          int Wo, int kh, int kw, int stride, int Cout, void* stream) {
  if (!x || !w || !y || n <= 0 || Hp <= 0 || Wp <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || Cout <= 0)
    return FCMF_ERR_ARG;
  if ((C & (C - 1)) || C < 64 || kh * kw > 64) return FCMF_ERR_UNSUPPORTED;             // k-tiles must not straddle taps
  if ((Ho - 1) * stride + kh > Hp || (Wo - 1) * stride + kw > Wp) return FCMF_ERR_ARG;   // every tap inside the (padded) input
  const int64_t in_bytes = (int64_t)n * Hp * Wp * C * 2, M = (int64_t)n * Ho * Wo;
  if (in_bytes >= (1ll << 31) || M >= (1ll << 31)) return FCMF_ERR_UNSUPPORTED;
  int logC = 0;
  while ((1 << logC) < C) ++logC;
  const ConvGeom cv{C, logC, Hp, Wp, Ho, Wo, kw, stride, in_bytes, logC};
  const int K = kh * kw * C;
  return gemm_impl(ctx, x, w, y, nullptr, nullptr, nullptr, (int)M, Cout, K, K, K, Cout, 0, 0, FCMF_BF16, FCMF_BF16, FCMF_EPI_NONE, 0,
                   stream, &cv, colstats);
}
extern "C" int fcmf_conv_gemm(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, int n, int Hp, int Wp, int C, int Ho,
                              int Wo, int kh, int kw, int stride, int Cout, void* stream) {
  return conv_gemm_impl(ctx, x, w, y, nullptr, n, Hp, Wp, C, Ho, Wo, kh, kw, stride, Cout, stream);
}

// The same two products with the BatchNorm statistics of the output as a by-product (ResNet trunks: every convolution feeds a
// train-mode BatchNorm, whose statistics pass re-read the whole activation): stats [ceil(M / 128)][N][2] float32 <- (sum, sum of
// squares) of each block of 128 output rows, per output channel, of the values AS STORED (bf16-rounded) -- plain stores from
// the epilogue's registers, deterministic, no atomics.  fcmf_bn_stats_blocks turns them into the per-group totals.  bf16,
// row-major A [M, K] and W [N, K], no epilogue; FCMF_ERR_UNSUPPORTED where the shape does not run on the 256-row persistent
// kernel (M or N < 256, unaligned): the caller then runs fcmf_bn_stats over the output as before.
extern "C" int fcmf_gemm_colstats(fcmf_gemm_ctx* ctx, const void* A, const void* B, void* C, float* stats, int M, int N, int K,
                                  int64_t lda, int64_t ldb, int64_t ldc, void* stream) {
  if (!stats) return FCMF_ERR_ARG;
  return gemm_impl(ctx, A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 0, FCMF_BF16, FCMF_BF16, FCMF_EPI_NONE, 0, stream,
                   nullptr, stats);
}
extern "C" int fcmf_conv_gemm_colstats(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, float* stats, int n, int Hp, int Wp,
                                       int C, int Ho, int Wo, int kh, int kw, int stride, int Cout, void* stream) {
  if (!stats) return FCMF_ERR_ARG;
  return conv_gemm_impl(ctx, x, w, y, stats, n, Hp, Wp, C, Ho, Wo, kh, kw, stride, Cout, stream);
}

// Implicit-GEMM convolution for inputs with FEW channels (the trunk's 7x7 / stride-2 stem on RGB crops, whose patch matrix was
// 1.8 GB per 448 crops): x is NHWC with `pix` (a power of two, 4 for RGB0) elements per pixel, zero border included, and the
// contraction walks, for every kernel row ky, ONE contiguous run of `run` elements (a power of two >= 32: kw x pix rounded up,
// e.g. 8 pixels x 4 = 32 for kw = 7) that starts at pixel (oy * stride + ky, ox * stride): K = kh * run, w [Cout, kh * run] with
// zeros where the run exceeds the kernel (kx >= kw, or the padding channel).  Same kernels as fcmf_conv_gemm: the tap offset is
// ky rows of the input, the per-lane offset the run's first pixel.
extern "C" int fcmf_conv_gemm_runs(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, float* stats, int n, int Hp, int Wp, int pix,
                                   int run, int Ho, int Wo, int kh, int stride, int Cout, void* stream) {
  if (!x || !w || !y || n <= 0 || Hp <= 0 || Wp <= 0 || pix <= 0 || run <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || stride <= 0 || Cout <= 0)
    return FCMF_ERR_ARG;
  if ((pix & (pix - 1)) || (run & (run - 1)) || run < 32 || run % pix || kh > 64 || (stride * pix) % 8) return FCMF_ERR_UNSUPPORTED;
  if ((Ho - 1) * stride + kh > Hp) return FCMF_ERR_ARG;
  const int64_t in_elems = (int64_t)n * Hp * Wp * pix, M = (int64_t)n * Ho * Wo;
  // the last run of the last row must end inside the buffer (runs may run past the end of THEIR row into the next one: those
  // elements meet zero weights)
  if ((((int64_t)(n - 1) * Hp + (Ho - 1) * stride + kh - 1) * Wp + (int64_t)(Wo - 1) * stride) * pix + run > in_elems) return FCMF_ERR_ARG;
  if (in_elems * 2 >= (1ll << 31) || M >= (1ll << 31)) return FCMF_ERR_UNSUPPORTED;
  int logC = 0, logP = 0;
  while ((1 << logC) < run) ++logC;
  while ((1 << logP) < pix) ++logP;
  const ConvGeom cv{run, logC, Hp, Wp, Ho, Wo, 1, stride, in_elems * 2, logP};
  const int K = kh * run;
  return gemm_impl(ctx, x, w, y, nullptr, nullptr, nullptr, (int)M, Cout, K, K, K, Cout, 0, 0, FCMF_BF16, FCMF_BF16, FCMF_EPI_NONE, 0,
                   stream, &cv, stats);
}

extern "C" int fcmf_colsum(const void* X, float* out, int M, int N, int64_t ldx, int dtype, int accumulate,
                           void* stream) {
  if (!X || !out || M < 0 || N <= 0) return FCMF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!accumulate) { if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) return FCMF_ERR_LAUNCH; }
  if (M == 0) return FCMF_OK;
  const int vec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  // rows per block: 256, or ~1024 blocks of >= 16 rows where 256-row blocks would not fill the chip (a 384-row matrix on
  // 18 blocks took 27 us: every wave walks its rows one dependent load at a time)
  const int bx = (N + 255) / 256;
  int rpb = 256;
  if ((int64_t)((M + 255) / 256) * bx < 256) {       // fewer blocks than CUs: shorter row runs
    rpb = (int)(((int64_t)M * bx / 1024 + 3) & ~3ll);
    rpb = rpb < 16 ? 16 : (rpb > 256 ? 256 : rpb);
  }
  dim3 grid(bx, (M + rpb - 1) / rpb);
  if (dtype == FCMF_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, st, (const float*)X, out, M, N, ldx, rpb, vec);
  else if (dtype == FCMF_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)X, out, M, N, ldx, rpb, vec);
  else return FCMF_ERR_UNSUPPORTED;
  FCMF_CHECK_LAUNCH();
  return FCMF_OK;
}
