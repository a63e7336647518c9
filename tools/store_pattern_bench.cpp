// Store-pattern micro-benchmark: how fast does the chip take the OUTPUT of a GEMM epilogue, by the shape of one wave's store
// instruction?  (diagnostic for the GELU / gelu' epilogue GEMMs: their second 302 MB output pass costs 91 us = 3.3 TB/s.)
//   hipcc -O2 --offload-arch=gfx950 tools/store_pattern_bench.cpp -o tools/bin/store_pattern_bench
// A [M, N] bf16 matrix (M = 49152, N = 3072 or 768) is written once by 256 persistent workgroups of 8 waves that walk 256 x 256
// tiles in the GEMM's order (XCD-aware, consecutive items share the row panel).  Patterns (per wave store instruction, 16 B/lane):
//   0  "wave-slice": the wave owns a 128-row x 64-column sub-tile (the GEMM's 2 x 4 wave grid) and stores 8 rows x 128 B
//   1  "row-512":    the wave owns 32 full tile rows (256 columns = 512 B) and stores 2 rows x 512 B
//   2  "row-1k":     tiles are 128 rows x 512 columns, the wave stores 1 row x 1 KiB
//   3  "linear":     no tiling: each wave streams contiguous KiBs (upper bound)
//   4  "frag-64":    the wave's 128 x 64 sub-tile, 16 rows x 64 B per instruction (lane = (row & 15, 16-B piece of a 64-B half line):
//                    what a store straight from permuted accumulator fragments would issue -- no LDS transposition)
// each with plain and with nt (aux = 2) stores; optional `gap_us` of ALU spin between tiles to mimic the main loop (burstiness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(512, 1) void store_kernel(void* out, int M, int N, int gap_ticks) {
  extern __shared__ char lds[];      // hold the whole CU's LDS like the GEMM does: one workgroup per CU
  lds[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
  const u32x4 v{(unsigned)lane, (unsigned)wave, blockIdx.x, 0x3f803f80u};
  const int nblk = gridDim.x;
  int slot = blockIdx.x;
  {
    int q = nblk >> 3, rr = nblk & 7, xcd = slot & 7, local = slot >> 3;
    slot = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  }
  const unsigned ld = (unsigned)N * 2u;
  auto st = [&](unsigned off) {
    if (NT) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
  };
  if (PAT == 3) {
    const long long total = (long long)M * N * 2 / 1024;       // KiB pieces
    for (long long p = (long long)slot * 8 + wave; p < total; p += (long long)nblk * 8) st((unsigned)(p * 1024 + lane * 16));
    return;
  }
  const int TMr = PAT == 2 ? 128 : 256, TNc = PAT == 2 ? 512 : 256;
  const int tiles_n = N / TNc, tiles = (M / TMr) * tiles_n;
  for (int item = slot; item < tiles; item += nblk) {
    const int i0 = (item / tiles_n) * TMr, j0 = (item % tiles_n) * TNc;
    if (gap_ticks > 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)gap_ticks) __builtin_amdgcn_s_sleep(8);
    }
    if (PAT == 0) {
      const int wm = wave >> 2, wn = wave & 3;
      const unsigned base = (unsigned)(i0 + wm * 128 + (lane >> 3)) * ld + (unsigned)(j0 + wn * 64 + (lane & 7) * 8) * 2u;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) st(base + (unsigned)(it * 8) * ld);
    } else if (PAT == 4) {
      const int wm = wave >> 2, wn = wave & 3;
      const unsigned base = (unsigned)(i0 + wm * 128 + (lane & 15)) * ld + (unsigned)(j0 + wn * 64 + (lane >> 4) * 8) * 2u;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) st(base + (unsigned)((it >> 1) * 16) * ld + (unsigned)(it & 1) * 64u);
    } else if (PAT == 1) {
      const unsigned base = (unsigned)(i0 + wave * 32 + (lane >> 5)) * ld + (unsigned)(j0 + (lane & 31) * 8) * 2u;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) st(base + (unsigned)(it * 2) * ld);
    } else {
      const unsigned base = (unsigned)(i0 + wave * 16) * ld + (unsigned)(j0 + lane * 8) * 2u;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) st(base + (unsigned)it * ld);
    }
  }
}

template <int PAT, bool NT>
static void run(const char* name, void* out, int M, int N, int gap, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)store_kernel<PAT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3(256), dim3(512), 160 * 1024, 0, out, M, N, gap);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3(256), dim3(512), 160 * 1024, 0, out, M, N, gap);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)M * N * 2;
  const int tiles_per_cu = (M / 256) * (N / 256) / 256;
  const double gap_ms = PAT == 3 ? 0.0 : tiles_per_cu * gap * 1e-5;      // 100 MHz ticks
  printf("N=%5d %-11s %-5s gap %3d us/tile: %.3f ms  (%.2f TB/s; minus the gaps %.3f ms = %.2f TB/s)\n", N, name, NT ? "nt" : "plain", gap / 100,
         ms, bytes / ms * 1e-9, ms - gap_ms, bytes / (ms - gap_ms) * 1e-9);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int M = 49152;
  void* out = nullptr;
  hipMalloc(&out, (size_t)M * 3072 * 2);
  for (int N : {3072, 768}) {
    for (int gap : {0, 1000, 2500}) {
      run<0, false>("wave-slice", out, M, N, gap, reps);
      run<0, true>("wave-slice", out, M, N, gap, reps);
      run<4, false>("frag-64", out, M, N, gap, reps);
      run<4, true>("frag-64", out, M, N, gap, reps);
      run<1, false>("row-512", out, M, N, gap, reps);
      run<1, true>("row-512", out, M, N, gap, reps);
      if (N % 512 == 0) {
        run<2, false>("row-1k", out, M, N, gap, reps);
        run<2, true>("row-1k", out, M, N, gap, reps);
      }
    }
    run<3, false>("linear", out, M, N, 0, reps);
    run<3, true>("linear", out, M, N, 0, reps);
  }
  hipFree(out);
  return 0;
}
