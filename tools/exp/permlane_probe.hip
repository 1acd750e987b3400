// What v_permlane16_swap_b32 / masked DPP moves do on gfx950, lane by lane (semantics check for vg_gram_valu.hpp).
// hipcc --offload-arch=gfx950 -O2 tools/exp/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int l = threadIdx.x;
    const int A = l, B = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(A, B, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
    int a4 = __builtin_amdgcn_update_dpp(A, B, 0x114, 0xf, 0xA, false);  // row_shr:4, banks 1,3: old = A, src = B
    int b4 = __builtin_amdgcn_update_dpp(B, A, 0x104, 0xf, 0x5, false);  // row_shl:4, banks 0,2: old = B, src = A
    out[128 + l] = a4;
    out[192 + l] = b4;
    int a8 = __builtin_amdgcn_update_dpp(A, B, 0x128, 0xf, 0xC, false);  // row_ror:8, banks 2,3
    int b8 = __builtin_amdgcn_update_dpp(B, A, 0x128, 0xf, 0x3, false);
    out[256 + l] = a8;
    out[320 + l] = b8;
}
int main()
{
    int *d, h[384];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"swap16 vdst(A)", "swap16 src (B)", "shr4 A<-B b1,3", "shl4 B<-A b0,2", "ror8 A<-B b2,3", "ror8 B<-A b0,1"};
    for (int q = 0; q < 6; q++) {
        printf("%-16s", names[q]);
        for (int l = 0; l < 64; l++) printf(" %d", h[64 * q + l]);
        printf("\n");
    }
    return 0;
}
