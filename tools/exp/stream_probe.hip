// probe: streaming-write ceilings for the emit kernel's working sets (plain vs nontemporal stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d2 = HIP_vector_type<double, 2>;
template <bool NT>
__global__ __launch_bounds__(256) void wr(double* dst, long long n2, double v) {
  d2 x; x.x = v; x.y = v;
  d2* d = reinterpret_cast<d2*>(dst);
  const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) { long long i = base + k * 256; if (i < n2) { if (NT) { __builtin_nontemporal_store(x.x, &d[i].x); __builtin_nontemporal_store(x.y, &d[i].y);} else d[i] = x; } }
}
int main() {
  for (double mb : {215.04, 430.0, 2150.4}) {
    long long n2 = (long long)(mb * 1e6 / 16);
    double* buf; hipMalloc(&buf, n2 * 16);
    for (int nt = 0; nt < 2; nt++) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      unsigned grid = (unsigned)((n2 + 1023) / 1024);
      for (int w = 0; w < 3; w++) { if (nt) wr<true><<<grid, 256>>>(buf, n2, 1.0); else wr<false><<<grid, 256>>>(buf, n2, 1.0); }
      hipEventRecord(a);
      const int reps = 50;
      for (int r = 0; r < reps; r++) { if (nt) wr<true><<<grid, 256>>>(buf, n2, 1.0); else wr<false><<<grid, 256>>>(buf, n2, 1.0); }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%8.1f MB  %s  %.1f us/launch  %.2f TB/s\n", mb, nt ? "nontemporal" : "plain      ", ms / reps * 1e3, n2 * 16.0 / (ms / reps * 1e-3) / 1e12);
    }
    hipFree(buf);
  }
  return 0;
}
