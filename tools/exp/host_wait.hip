// host_wait.hip -- what one host round trip costs on MI355X / ROCm 7.2: a kernel of a few workgroups ends, the host learns of
// it, launches the next one.  The host-driven LM loop of wide problems (the rig, G = 45) makes two such trips per iteration
// (rocprofv3 trace, tools/exp/trace_gaps.py: 26 + 28 us between the kernels either side of them).
//   mode 0  hipStreamSynchronize
//   mode 1  hipEventRecord + hipEventSynchronize
//   mode 2  the kernel's LAST workgroup (device counter) stores a sequence word in pinned memory, system-scope release; host spins
//   mode 3  a one-thread kernel behind it stores the word; host spins
// Prints microseconds per trip with a ~1 us kernel and with a ~20 us kernel (the trip's overhead = the difference to the kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/host_wait.hip -o tools/exp/host_wait.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            std::exit(1);                                           \
        }                                                           \
    } while (0)

__global__ __launch_bounds__(256) void work_kernel(double *out, int spin, unsigned int *counter, unsigned long long *host_seq, unsigned long long seq,
                                                   double *host_payload)
{
    double v = (double)threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0000001 + 1e-9;
    out[blockIdx.x * 256 + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x < 64 && host_payload) host_payload[threadIdx.x] = v + (double)seq;   // what the host reads after the wait
    if (host_seq) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void flag_kernel(unsigned long long *host_seq, unsigned long long seq)
{
    __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    double *out, *payload;
    unsigned int *counter;
    unsigned long long *seq_w;
    CK(hipMalloc(&out, sizeof(double) * 256 * 64));
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(counter, 0, 4));
    CK(hipHostMalloc(reinterpret_cast<void **>(&seq_w), 64, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&payload), 8 * 64, hipHostMallocDefault));
    volatile unsigned long long *seq_v = seq_w;
    *seq_v = 0;
    hipStream_t created;
    CK(hipStreamCreateWithFlags(&created, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int trips = 3000;
    for (int which = 0; which < 2; which++) {
        hipStream_t st = which ? created : nullptr;
        for (int spin : {0, 6000}) {
            for (int grid : {1, 9}) {
                for (int mode = 0; mode < 4; mode++) {
                    unsigned long long seq = *seq_v;
                    double best = 1e30, sink = 0.;
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipStreamSynchronize(st));
                        const double t0 = now_us();
                        for (int i = 0; i < trips; i++) {
                            seq++;
                            hipLaunchKernelGGL(work_kernel, dim3(grid), dim3(256), 0, st, out, spin, counter, mode == 2 ? seq_w : nullptr, seq, payload);
                            if (mode == 0) CK(hipStreamSynchronize(st));
                            else if (mode == 1) {
                                CK(hipEventRecord(ev, st));
                                CK(hipEventSynchronize(ev));
                            } else {
                                if (mode == 3) hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, seq_w, seq);
                                while (*seq_v != seq) {
                                }
                                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                            }
                            sink += payload[i & 63];
                        }
                        const double t = (now_us() - t0) / trips;
                        best = t < best ? t : best;
                    }
                    CK(hipStreamSynchronize(st));
                    *seq_v = seq;
                    std::printf("%s stream  kernel spin %5d grid %d  mode %d: %7.2f us per trip  (sink %.3g)\n", which ? "created" : "null   ", spin, grid, mode, best,
                                sink);
                }
            }
        }
    }
    return 0;
}
