// probe: how fast can 215 MB (the emit kernel's output at 10 k images) be written?  Sweeps the store pattern:
//   per-workgroup contiguous run (bytes), workgroup size, one-shot vs persistent grid-stride, and the
//   blockIdx -> chunk mapping (linear, or XCD-contiguous: workgroups of one XCD write one eighth of the buffer).
// build: hipcc --offload-arch=gfx950 -O3 -o store_sweep store_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using d2 = HIP_vector_type<double, 2>;

template <int THREADS, int PER_LANE, bool XCD>
__global__ __launch_bounds__(THREADS) void wr(double* dst, long long n2, long long n_chunks, double v) {
  d2 x; x.x = v; x.y = v;
  d2* d = reinterpret_cast<d2*>(dst);
  for (long long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    long long chunk = c;
    if (XCD) { const long long per = (n_chunks + 7) / 8; chunk = (c & 7) * per + (c >> 3); if (chunk >= n_chunks) continue; }
    const long long base = chunk * (THREADS * PER_LANE) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < PER_LANE; k++) { const long long i = base + (long long)k * THREADS; if (i < n2) d[i] = x; }
  }
}

template <int THREADS, int PER_LANE, bool XCD>
void run(double* buf, long long n2, int grid_mult, const char* name) {
  const long long n_chunks = (n2 + THREADS * PER_LANE - 1) / (THREADS * PER_LANE);
  const unsigned grid = grid_mult ? 256u * grid_mult : (unsigned)n_chunks;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; w++) wr<THREADS, PER_LANE, XCD><<<grid, THREADS>>>(buf, n2, n_chunks, 1.0);
  hipEventRecord(a);
  const int reps = 50;
  for (int r = 0; r < reps; r++) wr<THREADS, PER_LANE, XCD><<<grid, THREADS>>>(buf, n2, n_chunks, 1.0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s grid %7u  %.1f us  %.2f TB/s\n", name, grid, ms / reps * 1e3, n2 * 16.0 / (ms / reps * 1e-3) / 1e12);
}

int main() {
  for (double mb : {215.04, 2150.4}) {
    const long long n2 = (long long)(mb * 1e6 / 16);
    double* buf; hipMalloc(&buf, n2 * 16);
    printf("--- %.1f MB\n", mb);
    run<256, 4, false>(buf, n2, 0, "256thr 16KiB/wg one-shot linear");
    run<256, 4, true>(buf, n2, 0, "256thr 16KiB/wg one-shot xcd-contiguous");
    run<256, 16, false>(buf, n2, 0, "256thr 64KiB/wg one-shot linear");
    run<256, 16, true>(buf, n2, 0, "256thr 64KiB/wg one-shot xcd-contiguous");
    run<64, 4, false>(buf, n2, 0, "64thr 4KiB/wg one-shot linear");
    run<1024, 4, false>(buf, n2, 0, "1024thr 64KiB/wg one-shot linear");
    run<256, 4, false>(buf, n2, 4, "256thr 16KiB persistent x4 linear");
    run<256, 4, true>(buf, n2, 4, "256thr 16KiB persistent x4 xcd-contiguous");
    run<256, 4, false>(buf, n2, 8, "256thr 16KiB persistent x8 linear");
    run<256, 16, false>(buf, n2, 8, "256thr 64KiB persistent x8 linear");
    run<1024, 4, false>(buf, n2, 2, "1024thr 64KiB persistent x2 linear");
    hipFree(buf);
  }
  return 0;
}
