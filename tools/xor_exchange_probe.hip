// Developer probe: lane exchanges without ds_bpermute on gfx950 (v_permlane32_swap / v_permlane16_swap / DPP) checked
// against __shfl_xor for every butterfly mask.  Prints "ok" per mask.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned xor_dw(unsigned v, int mask)
{
    switch (mask) {
    case 1: return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    case 2: return __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    case 4: {
        unsigned r = __builtin_amdgcn_update_dpp(0u, v, 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0, 2 (lane i gets i+4)
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);          // row_shr:4 -> banks 1, 3 (lane i gets i-4)
    }
    case 8: return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false);     // row_ror:8
    case 16: {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    }
    default: {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
    }
}

__global__ void k(int *bad)
{
    const unsigned v = 1000u + threadIdx.x * 7u;
    const int masks[6] = {1, 2, 4, 8, 16, 32};
    for (int m = 0; m < 6; m++) {
        const unsigned a = xor_dw(v, masks[m]);
        const unsigned b = (unsigned)__shfl_xor((int)v, masks[m], 64);
        if (a != b) atomicAdd(&bad[m], 1);
    }
    // the two-register form used by the butterfly: (X, Y) -> X' + Y' with X' = {X.lo, Y.lo}, Y' = {X.hi, Y.hi}
    const unsigned X = 10u + threadIdx.x, Y = 5000u + threadIdx.x * 3u;
    {
        auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
        const unsigned want = (threadIdx.x < 32) ? X + (unsigned)__shfl_xor((int)X, 32, 64) : Y + (unsigned)__shfl_xor((int)Y, 32, 64);
        if (r[0] + r[1] != want) atomicAdd(&bad[6], 1);
    }
    {
        auto r = __builtin_amdgcn_permlane16_swap(X, Y, false, false);
        const unsigned want = ((threadIdx.x & 16) == 0) ? X + (unsigned)__shfl_xor((int)X, 16, 64) : Y + (unsigned)__shfl_xor((int)Y, 16, 64);
        if (r[0] + r[1] != want) atomicAdd(&bad[7], 1);
    }
}

int main()
{
    int *d = nullptr, h[8] = {0};
    (void)hipMalloc(&d, sizeof(h));
    (void)hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"xor1", "xor2", "xor4", "xor8", "xor16", "xor32", "swap32-sum", "swap16-sum"};
    for (int i = 0; i < 8; i++) printf("%s %s (%d lanes differ)\n", names[i], h[i] == 0 ? "ok" : "WRONG", h[i]);
    return 0;
}
