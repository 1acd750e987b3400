// probe: the emit kernel's memory pattern without its arithmetic.  Per 256-observation tile: read 4 KiB of
// observations (optional), write 4 KiB of residuals + 24 KiB + 24 KiB of Jacobian rows into three separate arrays.
// Answers: is the gap between the emit kernel and a single streaming write (6.6 TB/s at 2 GB) the three streams,
// the interleaved read, or the kernel itself?
#include <hip/hip_runtime.h>
#include <cstdio>
using d2 = HIP_vector_type<double, 2>;

__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned n) {
  const unsigned x = b % 8, j = b / 8, q = n / 8, r = n % 8;
  return x * q + (x < r ? x : r) + j;
}

template <bool READ, bool XCD, int STREAMS>
__global__ __launch_bounds__(256) void pat(const d2* __restrict__ obs, d2* __restrict__ res, d2* __restrict__ ji,
                                           d2* __restrict__ jm, unsigned n_tiles, double v) {
  const unsigned t = XCD ? xcd_block(blockIdx.x, n_tiles) : blockIdx.x;
  const size_t o = (size_t)t * 256 + threadIdx.x;
  d2 x; x.x = v; x.y = v;
  if (READ) { const d2 ob = obs[o]; x.x += ob.x; x.y += ob.y; }
  res[o] = x;
  if (STREAMS == 3) {
    // each wave owns 64 observations = 6 KiB of each Jacobian array, written as six 1 KiB runs
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    d2* a = ji + ((size_t)t * 4 + wave) * 384;
    d2* b = jm + ((size_t)t * 4 + wave) * 384;
#pragma unroll
    for (int k = 0; k < 6; k++) a[k * 64 + lane] = x;
#pragma unroll
    for (int k = 0; k < 6; k++) b[k * 64 + lane] = x;
  } else {
    // same bytes into ONE array: 52 KiB per tile
    d2* a = ji + (size_t)t * 3072 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 12; k++) a[k * 256] = x;
  }
}

template <bool READ, bool XCD, int STREAMS>
void run(const d2* obs, d2* res, d2* ji, d2* jm, unsigned n_tiles, const char* name) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 3; w++) pat<READ, XCD, STREAMS><<<n_tiles, 256>>>(obs, res, ji, jm, n_tiles, 1.0);
  (void)hipEventRecord(a);
  const int reps = 30;
  for (int r = 0; r < reps; r++) pat<READ, XCD, STREAMS><<<n_tiles, 256>>>(obs, res, ji, jm, n_tiles, 1.0);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)n_tiles * 256 * (16 + 192 + (READ ? 16 : 0));
  printf("%-52s %.1f us  %.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
  for (unsigned images : {10000u, 100000u}) {
    const unsigned n_tiles = images * 96 / 256;
    const size_t no = (size_t)n_tiles * 256;
    d2 *obs, *res, *ji, *jm;
    (void)hipMalloc(&obs, no * 16); (void)hipMalloc(&res, no * 16); (void)hipMalloc(&ji, no * 16 * 12); (void)hipMalloc(&jm, no * 16 * 6);
    (void)hipMemset(obs, 0, no * 16);
    printf("--- %u images\n", images);
    run<false, false, 1>(obs, res, ji, jm, n_tiles, "write res + ONE 52 KiB/tile array, linear");
    run<false, true, 1>(obs, res, ji, jm, n_tiles, "write res + ONE 52 KiB/tile array, xcd");
    run<false, false, 3>(obs, res, ji, jm, n_tiles, "write res + two 24 KiB/tile arrays, linear");
    run<false, true, 3>(obs, res, ji, jm, n_tiles, "write res + two 24 KiB/tile arrays, xcd");
    run<true, false, 3>(obs, res, ji, jm, n_tiles, "read obs + write three arrays, linear");
    run<true, true, 3>(obs, res, ji, jm, n_tiles, "read obs + write three arrays, xcd");
    (void)hipFree(obs); (void)hipFree(res); (void)hipFree(ji); (void)hipFree(jm);
  }
  return 0;
}
