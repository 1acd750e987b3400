// Host round trip after a short kernel: hipStreamSynchronize / hipEventSynchronize vs spinning on a word the kernel (or a
// one-thread kernel behind it) writes into pinned host memory.   hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(double *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0000001 + 1e-9; }
__global__ void flag(volatile unsigned long long *f, unsigned long long v) { *f = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double *x; hipMalloc(&x, 8 << 20); hipMemset(x, 0, 8 << 20);
    unsigned long long *f; hipHostMalloc(&f, 64, hipHostMallocDefault); *f = 0;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int n = 1 << 20, reps = 2000;
    for (int mode = 0; mode < 4; mode++) {
        double t0 = 0;
        for (int r = -100; r < reps; r++) {
            if (r == 0) t0 = now();
            hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, x, n);
            if (mode == 0) hipStreamSynchronize(st);
            else if (mode == 1) { hipEventRecord(ev, st); hipEventSynchronize(ev); }
            else if (mode == 2) { hipLaunchKernelGGL(flag, dim3(1), dim3(1), 0, st, f, (unsigned long long)(r + 1000)); while (*(volatile unsigned long long *)f != (unsigned long long)(r + 1000)) {} }
            else { hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, x, n); hipStreamSynchronize(st); }
        }
        const char *names[] = {"kernel + hipStreamSynchronize", "kernel + event record + hipEventSynchronize", "kernel + flag kernel + host spin on pinned word", "two kernels + hipStreamSynchronize"};
        printf("%-50s %.2f us per round trip\n", names[mode], (now() - t0) / reps * 1e6);
    }
    return 0;
}
