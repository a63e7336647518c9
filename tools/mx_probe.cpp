// Known-answer probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, E8M0 block scales) on gfx950: pins the operand layout
// the fp8 GEMM kernels of csrc/gemm_fp8.hip rely on, with exact arithmetic (small e4m3 values, power-of-two scales):
//   A operand: lane l holds A[row = l & 15][k = 32 (l >> 4) + j], j = 0..31, byte j of its 8 VGPRs (little endian);
//   B operand: lane l holds B[k = 32 (l >> 4) + j][col = l & 15];
//   scale operands: E8M0 127 (= 2^0) for every block -- the product kernels apply their per-row scales in the epilogue;
//   C/D: col = l & 15, row = 4 (l >> 4) + reg (as every 16x16 MFMA).
//   hipcc -O2 --offload-arch=gfx950 tools/mx_probe.cpp -o tools/bin/mx_probe && tools/bin/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPSEL>
__global__ void probe(const unsigned char* A /*[16][128]*/, const unsigned char* Bt /*[16][128]: B[k][col] at [col][k]*/,
                      const unsigned char* sA /*[16][4]*/, const unsigned char* sB /*[16][4]*/, float* D /*[16][16]*/) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  memcpy(&a, A + r * 128 + 32 * g, 32);
  memcpy(&b, Bt + r * 128 + 32 * g, 32);
  const int sa = (int)sA[r * 4 + g] << (8 * OPSEL), sb = (int)sB[r * 4 + g] << (8 * OPSEL);
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPSEL, sa, OPSEL, sb);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

static float e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int main() {
  srand(7);
  std::vector<unsigned char> A(16 * 128), Bt(16 * 128), sA(64), sB(64);
  int bad_total = 0;
  for (int trial = 0; trial < 8; ++trial) {
    for (auto& v : A) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v ^= 1; if (((v >> 3) & 15) > 9) v &= 0xC7 | (8 << 3) | 0x87; }
    for (auto& v : Bt) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v ^= 1; if (((v >> 3) & 15) > 9) v &= 0xC7 | (8 << 3) | 0x87; }
    // unit block scales (E8M0 127 = 2^0) in every byte: what the product kernels pass; per-row scales live in their epilogue
    for (auto& v : sA) v = 127;
    for (auto& v : sB) v = 127;
    unsigned char *dA, *dB, *dsA, *dsB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsA, sA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsB, sB.data(), 64, hipMemcpyHostToDevice);
    for (int opsel = 0; opsel < 4; opsel += 3) {
      if (opsel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, dD);
      else hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dA, dB, dsA, dsB, dD);
      float D[256];
      hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
      int bad = 0; double worst = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0;
          for (int g = 0; g < 4; ++g) {
            double blk = 0;
            for (int k = 0; k < 32; ++k) blk += (double)e4m3(A[i * 128 + 32 * g + k]) * (double)e4m3(Bt[j * 128 + 32 * g + k]);
            ref += blk * std::ldexp(1.0, sA[i * 4 + g] - 127) * std::ldexp(1.0, sB[j * 4 + g] - 127);
          }
          double mag = 0; for (int k = 0; k < 128; ++k) mag += std::fabs((double)e4m3(A[i * 128 + k]) * (double)e4m3(Bt[j * 128 + k]));
          const double err = std::fabs(D[i * 16 + j] - ref) / (mag + 1e-6);   // relative to the sum of |products|: a layout error is O(1)
          if (err > worst) worst = err;
          if (err > 1e-3) ++bad;
        }
      printf("trial %d opsel %d: %d / 256 mismatches, worst rel err %.2e\n", trial, opsel, bad, worst);
      bad_total += bad;
    }
    hipFree(dA); hipFree(dB); hipFree(dsA); hipFree(dsB); hipFree(dD);
  }
  printf(bad_total ? "MX PROBE FAILED\n" : "MX PROBE OK: layout as documented in tools/mx_probe.cpp\n");
  return bad_total ? 1 : 0;
}
