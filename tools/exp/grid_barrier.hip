// grid_barrier.hip -- what a device-wide hand-over costs INSIDE one launch on MI355X, against the ~4.5-5 us a dependent tiny
// launch costs (a cooperative one-launch LM loop for small problems needs ~5 grid barriers per iteration).
//
// All workgroups are co-resident (hipLaunchCooperativeKernel).  Each phase: every workgroup writes `payload` doubles, the
// grid meets at a barrier, every workgroup reads the payload of ANOTHER workgroup (another XCD: blockIdx + 1) and checks it.
//   mode 0  release / acquire at agent scope: the barrier's atomic add is a RELEASE (L2 write-back), the spin an ACQUIRE
//           (invalidate); payload through plain stores and loads.
//   mode 1  no cache maintenance at all: payload through agent-scope RELAXED atomic stores / loads (per-access coherence, the
//           sc1 forms), the barrier through relaxed atomics behind s_waitcnt vmcnt(0) + workgroup barrier.
//   mode 2  as mode 1, the spin sleeping between polls.
// Prints microseconds per phase and the number of stale reads (must be 0).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/grid_barrier.hip -o tools/exp/grid_barrier.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned int *counter, unsigned int target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            }
        } else {
            __builtin_amdgcn_s_waitcnt(0);  // every store of this lane has been acknowledged (the other lanes: the barrier above + their own waits)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (MODE == 2) __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void phases_kernel(double *buf, unsigned int *counter, int payload, int phases, unsigned int *stale)
{
    const unsigned int nwg = gridDim.x;
    double *mine = buf + (size_t)blockIdx.x * payload;
    const double *other = buf + (size_t)((blockIdx.x + 1) % nwg) * payload;
    unsigned int bad = 0;
    for (int ph = 1; ph <= phases; ph++) {
        for (int i = threadIdx.x; i < payload; i += 256) {
            const double v = (double)ph * 1000. + (double)blockIdx.x + (double)i * 1e-6;
            if (MODE == 0) mine[i] = v;
            else __hip_atomic_store(reinterpret_cast<unsigned long long *>(mine) + i, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE != 0) __builtin_amdgcn_s_waitcnt(0);  // this lane's stores are out before the workgroup barrier inside grid_barrier
        grid_barrier<MODE>(counter, (unsigned int)(2 * ph - 1) * nwg);
        for (int i = threadIdx.x; i < payload; i += 256) {
            double got;
            if (MODE == 0) got = other[i];
            else got = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(other) + i, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT));
            const double want = (double)ph * 1000. + (double)((blockIdx.x + 1) % nwg) + (double)i * 1e-6;
            bad += got != want;
        }
        grid_barrier<MODE>(counter, (unsigned int)(2 * ph) * nwg);  // nobody overwrites before everybody has read
    }
    if (bad) atomicAdd(stale, bad);
}

template <int MODE>
static void run(int nwg, int payload, int phases)
{
    double *buf;
    unsigned int *counter, *stale;
    CK(hipMalloc(&buf, sizeof(double) * (size_t)nwg * payload));
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&stale, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned int h_stale = 0;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(counter, 0, 4));
        CK(hipMemset(stale, 0, 4));
        void *args[] = {&buf, &counter, &payload, &phases, &stale};
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(phases_kernel<MODE>), dim3(nwg), dim3(256), args, 0, nullptr));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned int s;
        CK(hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost));
        h_stale += s;
    }
    std::printf("mode %d  %4d workgroups  payload %6d doubles  %d phases: %8.3f us per phase (2 barriers + write + read), stale reads %u\n", MODE, nwg,
                payload, phases, best * 1e3 / phases, h_stale);
    CK(hipFree(buf));
    CK(hipFree(counter));
    CK(hipFree(stale));
}

__global__ void tiny_kernel(double *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.; }

int main()
{
    // reference: dependent tiny launches back to back
    double *p;
    CK(hipMalloc(&p, 8));
    CK(hipMemset(p, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int grid : {1, 128, 256}) {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, 0, p);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, 0, p);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("dependent tiny launches, grid %3d: %.3f us each\n", grid, ms);
    }
    for (int nwg : {128, 256, 512})
        for (int payload : {64, 4096, 32768}) {
            run<0>(nwg, payload, 200);
            run<1>(nwg, payload, 200);
            run<2>(nwg, payload, 200);
        }
    return 0;
}
