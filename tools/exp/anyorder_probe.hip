// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  A = one wave spinning ~20 us, B = a streaming
// write of 64 MB (~15 us).  hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(double *x, int iters) { double v = x[threadIdx.x]; for (int i = 0; i < iters; i++) v = v * 1.0000001 + 1e-9; x[threadIdx.x] = v; }
__global__ void fill(double *x, long long n) { long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double *a, *b; const long long n = 8 << 20;
    hipMalloc(&a, 4096); hipMemset(a, 0, 4096); hipMalloc(&b, n * 8);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 4; mode++) {
        double t0 = 0; const int reps = 300;
        for (int r = -30; r < reps; r++) {
            if (r == 0) { hipStreamSynchronize(st); t0 = now(); }
            if (mode != 2) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, a, 6000);
            if (mode == 0 || mode == 2) hipLaunchKernelGGL(fill, dim3((unsigned)(n / 256)), dim3(256), 0, st, b, n);
            if (mode == 1) hipExtLaunchKernelGGL(fill, dim3((unsigned)(n / 256)), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b, n);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, a, 10);   // an ordered kernel behind both
        }
        hipStreamSynchronize(st);
        const char *names[] = {"A, B ordered, C", "A, B any-order, C", "B, C (no A)", "A, C (no B)"};
        printf("%-22s %.2f us per round\n", names[mode], (now() - t0) / reps * 1e6);
    }
    return 0;
}
