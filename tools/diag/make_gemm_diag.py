#!/usr/bin/env python3
"""Diagnostic build of the GEMM library with in-kernel time stamps (NOT the product: the product source carries no diagnostic code).
Copies csrc/gemm.hip to a scratch file, inserts s_memtime stamps at the phase boundaries of the persistent bf16-output kernel by exact
string replacement (asserted: a changed kernel breaks the build of this tool, never the product), builds tools/bin/diag/libfcmf_hip.so.

  python tools/diag/make_gemm_diag.py && LD_LIBRARY_PATH=tools/bin/diag tools/bin/gemm_bench 5 0 "fwd  ffn1"

Per work item of workgroup 8, waves 0 (group A) and 4 (group B), lane 0 -- shader-clock cycles:
  [0] item start   [1] main loop done   [2] next item's tiles in flight (prefetch issued)   [3] epilogue done (last store issued)
  [4] sum over k-tiles of (wait for the tile's DMA + barrier)   [5..7] that wait for k-tiles 0, 1, 2   [8] s_memrealtime at item start
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "multimodal-aspect-category-sentiment-analysis_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "bin", "diag")
src = open(os.path.join(CSRC, "gemm.hip")).read()


def sub(old, new, count=1):
    global src
    assert src.count(old) >= count, ("anchor not found", old[:70])
    src = src.replace(old, new, count)


sub('struct GemmParams {', '''__device__ unsigned long long fcmf_diag_buf[2 * 16 * 16];
#define DIAG_ON (blockIdx.x == 8 && (wave == 0 || wave == 4) && lane == 0)
#define DIAG_ROW (&fcmf_diag_buf[((wave >> 2) * 16 + (d_item < 16 ? d_item : 15)) * 16])
#define DIAG_STAMP(i) do { if (DIAG_ON) DIAG_ROW[i] = __builtin_amdgcn_s_memtime(); } while (0)
struct GemmParams {''')
sub('  for (int item = slot; item < p.total_items; item += nblk) {', '''  int d_item = -1;
  for (int item = slot; item < p.total_items; item += nblk) {
  ++d_item;
  unsigned long long d_wsum = 0, d_w0 = 0;''')
sub('  const Item w = decode(item);', '''  const Item w = decode(item);
  DIAG_STAMP(0);
  if (DIAG_ON) DIAG_ROW[8] = __builtin_amdgcn_s_memrealtime();''')
# group A loop
sub('''      wait_landed(t, IntTag<4>{});
      __builtin_amdgcn_s_barrier();            // tile t visible; the stage of tile t-1 is no longer read
''', '''      d_w0 = __builtin_amdgcn_s_memtime();
      wait_landed(t, IntTag<4>{});
      __builtin_amdgcn_s_barrier();            // tile t visible; the stage of tile t-1 is no longer read
      { const unsigned long long d = __builtin_amdgcn_s_memtime() - d_w0; d_wsum += d; if (t < 3 && DIAG_ON) DIAG_ROW[5 + t] = d; }
''')
sub('''      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads of tile t-1 have left LDS
      __builtin_amdgcn_s_barrier();
''', '''      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads of tile t-1 have left LDS
      __builtin_amdgcn_s_barrier();
      { const unsigned long long d = __builtin_amdgcn_s_memtime() - d_w0; d_wsum += d; if (t < 3 && DIAG_ON) DIAG_ROW[5 + t] = d; }
''')
sub('''      wait_landed(t, IntTag<(FULL_A ? 4 : 3)>{});
''', '''      d_w0 = __builtin_amdgcn_s_memtime();
      wait_landed(t, IntTag<(FULL_A ? 4 : 3)>{});
''')
sub('''  TC* C = reinterpret_cast<TC*>(BATCH ? bp->C[w.batch] : p.C);
''', '''  TC* C = reinterpret_cast<TC*>(BATCH ? bp->C[w.batch] : p.C);
  DIAG_STAMP(1);
  if (DIAG_ON) DIAG_ROW[4] = d_wsum;
''')
sub('''    if constexpr (FP8) {
      // acc[i][j] of quantised operands''', '''    DIAG_STAMP(2);
    if constexpr (FP8) {
      // acc[i][j] of quantised operands''')
sub('''    if (p.colsum) {
      // a lane owns 8 fixed columns for the rows it visited; lanes that share (lane & 7) share the columns''', '''    DIAG_STAMP(3);
    if (p.colsum) {
      // a lane owns 8 fixed columns for the rows it visited; lanes that share (lane & 7) share the columns''')
src += '''
extern "C" int fcmf_gemm_diag_read(void* dst, int clear) {
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(fcmf_diag_buf), sizeof(unsigned long long) * 2 * 16 * 16) != hipSuccess) return -1;
  if (clear) { static unsigned long long z[2 * 16 * 16]; (void)hipMemcpyToSymbol(HIP_SYMBOL(fcmf_diag_buf), z, sizeof(z)); }
  return 0;
}
'''
# ablations of the epilogue (sys.argv[1]): which resource is it made of?
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "base"
if VARIANT == "nostore":        # no global stores (LDS traffic and vector math stay)
    sub("        if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD,", "        if (p.M < 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD,")
    sub("        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD, sbase + (unsigned)(rnd * 32 + it * 8) * ldc2, 0, 0);",
        "        else if (p.M < -1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[it]), rD, sbase + (unsigned)(rnd * 32 + it * 8) * ldc2, 0, 0);\n        else asm volatile(\"\" :: \"v\"(x[it]));")
elif VARIANT == "nopoly":       # no polynomial (GELU / gelu' become plain products)
    sub("            v = v * phi_poly4(v);", "            v = v * v;")
    sub("            else v = v * dgelu_poly4(a);", "            else v = v * a;")
elif VARIANT != "base":
    raise SystemExit("variants: base, nostore, nopoly")
OUT = os.path.join(OUT, VARIANT) if VARIANT != "base" else OUT
os.makedirs(OUT, exist_ok=True)
scratch = os.path.join(OUT, "gemm_diag.hip")
open(scratch, "w").write(src)
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", scratch, "-o", os.path.join(OUT, "gemm_diag.o")])
objs = [os.path.join(CSRC, f) for f in ("attn_small.o", "attn_mfma.o", "norm.o", "box.o", "misc.o", "conv.o", "comm.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libfcmf_hip.so"),
                       os.path.join(OUT, "gemm_diag.o"), *objs, "-ldl"])
os.remove(os.path.join(OUT, "gemm_diag.o"))
print("built", os.path.join(OUT, "libfcmf_hip.so"))
