// Standalone GEMM micro-benchmark over the FCMF step's shapes (links libfcmf_hip.so).
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_bench.cpp -Iinclude -L<pkg>/fcmf_framework -lfcmf_hip -o tools/bin/gemm_bench   (tools/bin/ travels to the GPU box, gpurun_out/ does not)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <dlfcn.h>
#include "fcmf_hip.h"

// a stand-in for a communication kernel: `blocks` workgroups that each hold a CU's LDS (so no persistent GEMM
// workgroup fits beside them) and spin for `ticks` of the 100 MHz counter
__global__ void hog_kernel(unsigned long long ticks) {
  extern __shared__ char hog_lds[];
  hog_lds[threadIdx.x] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

struct Shape { const char* name; int M, N, K, ta, tb, epi, acc, out_f32; };

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 20;
  int tile = argc > 2 ? atoi(argv[2]) : 0;
  fcmf_gemm_ctx* ctx = nullptr;
  fcmf_gemm_ctx_create(&ctx);
  fcmf_gemm_ctx_tune(ctx, tile, getenv("FCMF_GEMM_KB") && atoi(getenv("FCMF_GEMM_KB")) == 32 ? 32 : -1,
                     getenv("FCMF_GEMM_CUS") ? atoi(getenv("FCMF_GEMM_CUS")) : -1,
                     getenv("FCMF_GEMM_NT_MIN_MB") ? ((int64_t)atoi(getenv("FCMF_GEMM_NT_MIN_MB")) << 20) : -1);
  printf("---- forced tile: %d (0 = heuristic)\n", tile);
  const int T = 49152;
  std::vector<Shape> shapes = {
      {"fwd  qkv/out   NT 49152x768x768", T, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"fwd  ffn1+gelu NT 49152x3072x768", T, 3072, 768, 0, 0, FCMF_EPI_GELU, 0, 0},
      {"fwd  ffn1 noepi NT 49152x3072x768", T, 3072, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"fwd  ffn2      NT 49152x768x3072", T, 768, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
      // (the bf16 step multiplies dY by the TRANSPOSED bf16 shadow of W: dX GEMMs are NT like the forward ones)
      {"dX   out       NT 49152x768x768", T, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"dX   ffn2+dgelu NT 49152x3072x768", T, 3072, 768, 0, 0, FCMF_EPI_DGELU, 0, 0},
      {"dX   ffn1      NT 49152x768x3072", T, 768, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"dX   qkv       NT 49152x768x2304", T, 768, 2304, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"dX   qkv+add   NT 49152x768x2304", T, 768, 2304, 0, 0, FCMF_EPI_ADD, 0, 0},
      {"f32-mode dX ffn1 NN 49152x768x3072", T, 768, 3072, 0, 1, FCMF_EPI_NONE, 0, 0},
      {"dW   768x768   TN k=49152", 768, 768, T, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"dW   3072x768  TN k=49152", 3072, 768, T, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"dW   768x3072  TN k=49152", 768, 3072, T, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"fwd  vismap    NT 21952x768x2048", 21952, 768, 2048, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"sq   NT 4096^3", 4096, 4096, 4096, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"sq   NN 4096^3", 4096, 4096, 4096, 0, 1, FCMF_EPI_NONE, 0, 0},
      {"sq   TN 4096^3 bf16 out", 4096, 4096, 4096, 1, 1, FCMF_EPI_NONE, 0, 0},
      {"sq   TN 4096^3 f32 acc", 4096, 4096, 4096, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"sq   TT 4096^3 (A^T, B [N,K])", 4096, 4096, 4096, 1, 0, FCMF_EPI_NONE, 0, 0},
      {"sq   NT 8192^3", 8192, 8192, 8192, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"sq   TN 8192^3 bf16 out", 8192, 8192, 8192, 1, 1, FCMF_EPI_NONE, 0, 0},

      {"iaog NT 8192x768x768", 8192, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"iaog NT 8192x2304x768", 8192, 2304, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"iaog NT 8192x3072x768 gelu", 8192, 3072, 768, 0, 0, FCMF_EPI_GELU, 0, 0},
      {"iaog NT 8192x768x3072", 8192, 768, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"iaog NT 8192x768x2304 add", 8192, 768, 2304, 0, 0, FCMF_EPI_ADD, 0, 0},
      {"iaog NT 768x768x768", 768, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"iaog NT 768x1536x768", 768, 1536, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"small NT 2048x768x768", 2048, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"small NT 2048x3072x768", 2048, 3072, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"small NT 2048x768x3072", 2048, 768, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"small NN 2048x768x768", 2048, 768, 768, 0, 1, FCMF_EPI_NONE, 0, 0},
      {"small dW 768x768 TN k=2048", 768, 768, 2048, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"small dW 3072x768 TN k=2048", 3072, 768, 2048, 1, 1, FCMF_EPI_NONE, 1, 1},
      {"small NT 2688x768x768", 2688, 768, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x3072 K=256", T, 3072, 256, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x3072 K=512", T, 3072, 512, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x3072 K=1536", T, 3072, 1536, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x3072 K=3072", T, 3072, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x768 K=256", T, 768, 256, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 49152x768 K=1536", T, 768, 1536, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 5376x3072 K=768 (252 tiles)", 5376, 3072, 768, 0, 0, FCMF_EPI_NONE, 0, 0},
      {"scan NT 5376x3072 K=3072 (252 tiles)", 5376, 3072, 3072, 0, 0, FCMF_EPI_NONE, 0, 0},
  };
  size_t maxA = (size_t)T * 3072, maxB = (size_t)T * 3072, maxC = (size_t)T * 3072;   // >= 8192^2 too
  unsigned short *A, *B; void *C, *AUX; float* bias;
  hipMalloc(&A, maxA * 2); hipMalloc(&B, maxB * 2); hipMalloc(&C, maxC * 4); hipMalloc(&AUX, maxC * 2); hipMalloc(&bias, 4096 * 4);
  {
    std::vector<unsigned short> h(maxA);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = f2bf(((s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    hipMemcpy(A, h.data(), maxA * 2, hipMemcpyHostToDevice);
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = f2bf((((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f); }
    hipMemcpy(B, h.data(), maxB * 2, hipMemcpyHostToDevice);
    hipMemcpy(AUX, h.data(), maxA * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096 * 4);
    hipMemset(C, 0, maxC * 4);
  }
  if (!getenv("FCMF_BENCH_NO_WS")) {   // split-K workspace (plain stores + reduce pass instead of float atomics)
    void* ws; const size_t wsb = 96u << 20;
    hipMalloc(&ws, wsb);
    fcmf_gemm_ctx_set_workspace(ctx, ws, (int64_t)wsb);
  }
  const int hog = getenv("FCMF_BENCH_HOG") ? atoi(getenv("FCMF_BENCH_HOG")) : 0;   // CUs taken away by a co-running kernel
  hipStream_t hs; hipStreamCreate(&hs);
  if (hog) hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* only = argc > 3 ? argv[3] : nullptr;   // run only the shapes whose name starts with this
  for (auto& sh : shapes) {
    if (only && strncmp(sh.name, only, strlen(only)) != 0) continue;
    int64_t lda = sh.ta ? sh.M : sh.K, ldb = sh.tb ? sh.N : sh.K, ldc = sh.N;
    auto run = [&]() {
      return fcmf_gemm(ctx, A, B, C, (sh.acc || getenv("FCMF_BENCH_NO_BIAS")) ? nullptr : bias, (sh.epi == FCMF_EPI_GELU || sh.epi == FCMF_EPI_DGELU || sh.epi == FCMF_EPI_ADD) ? AUX : nullptr, nullptr,
                       sh.M, sh.N, sh.K, lda, ldb, ldc, sh.ta, sh.tb, FCMF_BF16, sh.out_f32 ? FCMF_F32 : FCMF_BF16, sh.epi, sh.acc, nullptr);
    };
    int rc = run(); rc |= run();
    hipDeviceSynchronize();
    if (hog) {   // the hog outlives the timed launches (it needs the CUs' LDS: a resident persistent GEMM workgroup excludes it and vice versa)
      hipLaunchKernelGGL(hog_kernel, dim3(hog), dim3(256), 64 * 1024, hs, (unsigned long long)(100000ull * 30));   // 30 ms
      hipStreamQuery(hs);
      for (volatile int spin = 0; spin < 2000000; ++spin) {}
    }
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    if (hog) hipStreamSynchronize(hs);
    double tf = 2.0 * sh.M * sh.N * sh.K / (ms * 1e-3) / 1e12;
    printf("%-40s rc=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 2500)\n", sh.name, rc, ms, tf, tf / 25.0);
    // diagnostic library (python tools/diag/make_gemm_diag.py; LD_LIBRARY_PATH=tools/bin/diag): phase stamps of workgroup 8, waves 0 / 4
    typedef int (*rd_fn)(void*, int);
    static rd_fn rd = (rd_fn)dlsym(RTLD_DEFAULT, "fcmf_gemm_diag_read");
    if (rd) {
      static unsigned long long h[2 * 16 * 16];
      rd(h, 1);                      // clear what the timed launches left
      run();
      hipDeviceSynchronize();
      rd(h, 0);
      printf("    wave item:   loop  (waits: sum  t0  t1  t2)   prefetch  epilogue | item total   [us at the measured shader clock]\n");
      for (int g = 0; g < 2; ++g)
        for (int it = 0; it + 1 < 16 && h[(g * 16 + it + 1) * 16]; ++it) {
          const unsigned long long* r = &h[(g * 16 + it) * 16];
          const unsigned long long* n = &h[(g * 16 + it + 1) * 16];
          const double mhz = (double)(n[0] - r[0]) / ((double)(n[8] - r[8]) * 0.01);      // cycles per us
          auto us = [&](unsigned long long c) { return (double)c / mhz; };
          printf("    %4d %4d: %6.2f (%5.2f %5.2f %5.2f %5.2f)  %5.2f  %6.2f | %6.2f   clock %.0f MHz\n", g * 4, it, us(r[1] - r[0]), us(r[4]), us(r[5]), us(r[6]),
                 us(r[7]), us(r[2] - r[1]), us(r[3] - r[2]), us(n[0] - r[0]), mhz);
        }
    }
  }
  return 0;
}
