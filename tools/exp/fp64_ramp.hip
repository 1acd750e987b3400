// Does a launch that fills the chip with FP64 work run at full speed from its first microsecond?  Every wave executes the same
// chain of dependent FMAs (no memory, nothing to fetch but a 200-byte loop) and stamps the 100 MHz wall clock every `block` FMAs;
// a train of launches with a few microseconds between them is what an LM loop looks like.  Output: time per block of FMAs as a
// function of the time since the launch started, median over all waves.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/fp64_ramp.hip -o /tmp/fp64_ramp && /tmp/fp64_ramp [waves_per_simd] [stamps] [fmas_per_stamp]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

template <int ILP>
__global__ __launch_bounds__(256) void ramp(unsigned long long *out, int n_stamps, int fmas, double seed)
{
    extern __shared__ double pad[];   // dynamic LDS only to pin the occupancy: 160 KB / CU / wps workgroups
    double x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + i * 1e-3 + threadIdx.x * 1e-9;
    if (seed == -1.) pad[threadIdx.x] = seed;
    const double a = 1.0000001, b = 1e-9;
    unsigned long long *o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * n_stamps;
    for (int s = 0; s < n_stamps; s++) {
        if ((threadIdx.x & 63) == 0) o[s] = wall();
#pragma unroll 8
        for (int k = 0; k < fmas / ILP; k++)
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] = __builtin_fma(x[i], a, b);
    }
    double t = 0;
    for (int i = 0; i < ILP; i++) t += x[i];
    if (t == 12345.678) out[0] = 0;   // keep the chain alive
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 2, n_stamps = argc > 2 ? atoi(argv[2]) : 40, fmas = argc > 3 ? atoi(argv[3]) : 512;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_wg = prop.multiProcessorCount * wps;   // 4 waves per workgroup = wps waves per SIMD
    const size_t n_waves = (size_t)n_wg * 4;
    const size_t lds = (size_t)(150 * 1024 / wps) & ~(size_t)1023;   // exactly wps workgroups (4 waves each) fit a CU: wps waves per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void *>(ramp<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * n_waves * n_stamps);
    std::vector<unsigned long long> h(n_waves * n_stamps);
    for (int mode = 0; mode < 2; mode++) {   // 0: a train of launches back to back; 1: every launch behind an idle GPU
        for (int rep = 0; rep < 10; rep++) {
            ramp<4><<<n_wg, 256, lds>>>(d, n_stamps, fmas, 1.0 + rep);
            if (mode == 1) hipDeviceSynchronize();
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < n_waves; w++) t0 = std::min(t0, h[w * n_stamps]);
        printf("%s: %zu waves (%d per SIMD), %d FMAs (ILP 4) between stamps; full rate = %.3f us per stamp per wave x waves per SIMD\n",
               mode ? "idle GPU in front of the launch" : "train of launches", n_waves, wps, fmas, fmas * 4.0 / 2400.0);
        for (int s = 0; s + 1 < n_stamps; s++) {
            std::vector<double> dt(n_waves), at(n_waves);
            for (size_t w = 0; w < n_waves; w++) {
                dt[w] = (double)(h[w * n_stamps + s + 1] - h[w * n_stamps + s]) / 100.;
                at[w] = (double)(h[w * n_stamps + s] - t0) / 100.;
            }
            std::sort(dt.begin(), dt.end());
            std::sort(at.begin(), at.end());
            printf("  stamp %2d  starts at %6.2f us (median)  block takes p10 %.3f p50 %.3f p90 %.3f us\n", s, at[n_waves / 2], dt[n_waves / 10], dt[n_waves / 2],
                   dt[n_waves * 9 / 10]);
        }
    }
    return 0;
}
