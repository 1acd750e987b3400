// probe: do FP64 MFMA and FP64 VALU share execution resources on gfx950?
//   A: every wave issues only v_mfma_f64_16x16x4_f64      B: every wave issues only v_fma_f64
//   C: 8 waves per CU; waves 0-3 MFMA, waves 4-7 VALU (waves w and w+4 of a 512-thread workgroup share a SIMD)
// If C's time ~= max(A/2, B/2)-ish scaled, the pipes are separate; if ~= sum, they are shared.
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int ITERS = 4096;
__global__ __launch_bounds__(512) void k(double* out, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || ((mode == 2 || mode == 4) && wave < 4);
  const bool do_valu = mode == 1 || ((mode == 2 || mode == 3) && wave >= 4);
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  f64x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
  if (do_mfma) {
    for (int i = 0; i < ITERS; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int i = 0; i < ITERS * 16; i++) {  // 16 independent-ish FMAs per MFMA pair slot: 8 chains x 2
      v0 = __builtin_fma(v0, b, a); v1 = __builtin_fma(v1, b, a); v2 = __builtin_fma(v2, b, a); v3 = __builtin_fma(v3, b, a);
      v4 = __builtin_fma(v4, b, a); v5 = __builtin_fma(v5, b, a); v6 = __builtin_fma(v6, b, a); v7 = __builtin_fma(v7, b, a);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
  double* out; hipMalloc(&out, 256 * 512 * 8);
  const char* names[5] = {"A: 8 waves/CU all MFMA f64   ", "B: 8 waves/CU all FMA f64    ", "C: 4 waves MFMA + 4 waves FMA", "D: 4 waves FMA only (C minus MFMA)", "E: 4 waves MFMA only (C minus FMA)"};
  for (int mode = 0; mode < 5; mode++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(out, mode);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<<<256, 512>>>(out, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double mf = (mode == 1 || mode == 3) ? 0 : (mode == 0 ? 8 : 4) * 256.0 * ITERS * 2 * 2048;   // MFMA flops
    double vf = (mode == 0 || mode == 4) ? 0 : (mode == 1 ? 8 : 4) * 256.0 * ITERS * 16 * 8 * 64 * 2;  // VALU flops
    printf("%s  %.3f ms   MFMA %.1f TF/s  VALU %.1f TF/s\n", names[mode], ms, mf / ms / 1e9, vf / ms / 1e9);
  }
  return 0;
}
