// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // hwreg(HW_REG_HW_ID, 0, 32)
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 8 * 4);
    k<<<4096, 512>>>(d);
    static unsigned h[4096 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pair[4] = {0, 0, 0, 0};
    for (int b = 0; b < 4096; ++b) {
        if (b < 6 || b > 4090) { printf("wg %4d simd:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b*8] >> 8) & 15); }
        int same4 = 1, same1 = 1;
        for (int w = 0; w < 4; ++w) same4 &= (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
        for (int w = 0; w < 8; w += 2) same1 &= (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 1] >> 4) & 3));
        pair[0] += same4; pair[1] += same1;
    }
    printf("workgroups with (w, w+4) on one SIMD: %d / 4096; with (2i, 2i+1) on one SIMD: %d\n", pair[0], pair[1]);
}
