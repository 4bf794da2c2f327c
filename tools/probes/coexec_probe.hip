// Do an MFMA-only wave and a VALU-only wave that share a SIMD run at the same time?  8-wave workgroups (waves w and
// w+4 share a SIMD): role A = 32 dependent-chain MFMAs per step, role B = 64 exp + 64 add + 32 cvt per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float att_add(float x, float y) { float r = x + y; asm("" : "+v"(r)); return r; }

// MODE: 0 all waves MFMA role; 1 all waves VALU role; 2 waves 0-3 MFMA, 4-7 VALU; 3 waves 0-3 MFMA, 4-7 idle (exit);
//       4 waves 0-3 idle, 4-7 VALU; 5 even waves MFMA / odd waves VALU (partners on different SIMDs)
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ src, float* out, int steps) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool mf, va;
    if (MODE == 0) { mf = true; va = false; }
    else if (MODE == 1) { mf = false; va = true; }
    else if (MODE == 2) { mf = wave < 4; va = !mf; }
    else if (MODE == 3) { mf = wave < 4; va = false; }
    else if (MODE == 4) { mf = false; va = wave >= 4; }
    else { mf = (wave & 1) == 0; va = !mf; }
    const float base = src[lane];
    float r = 0;
    if (mf) {
        bf16x8 a[8], b[4];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(base * (0.5f + 0.01f * (i * 8 + e)));
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(base * (0.3f - 0.02f * (i * 8 + e)));
        f32x16 c0, c1, c2, c3;
        for (int v = 0; v < 16; ++v) { c0[v] = c1[v] = c2[v] = c3[v] = base; }
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c0 = MF(a[k], b[k], c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) c1 = MF(a[4 + k], b[k], c1);
#pragma unroll
            for (int k = 0; k < 4; ++k) { c2 = MF(a[k], b[k], c2); c3 = MF(a[4 + k], b[k], c3); }
#pragma unroll
            for (int k = 0; k < 4; ++k) c0 = MF(a[k], b[k], c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) c1 = MF(a[4 + k], b[k], c1);
#pragma unroll
            for (int k = 0; k < 4; ++k) { c2 = MF(a[k], b[k], c2); c3 = MF(a[4 + k], b[k], c3); }
        }
        for (int v = 0; v < 16; ++v) r += c0[v] + c1[v] + c2[v] + c3[v];
    } else if (va) {
        float s[32];
        for (int v = 0; v < 32; ++v) s[v] = base * 0.001f * v;
        float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
        bf16x8 pf[4];
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int v = 0; v < 32; ++v) s[v] = __builtin_amdgcn_exp2f(s[v] * 0.5f - 1.0f) ;
#pragma unroll
                for (int v = 0; v < 32; v += 4) { l0 = att_add(l0, s[v]); l1 = att_add(l1, s[v + 1]); l2 = att_add(l2, s[v + 2]); l3 = att_add(l3, s[v + 3]); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[i][e] = (__bf16)s[8 * i + e];
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(pf[i]));
            }
        }
        r = l0 + l1 + l2 + l3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, const float* src, float* out) {
    const int wgs = 256, steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<wgs, 512>>>(src, out, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<MODE><<<wgs, 512>>>(src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-60s %9.1f us\n", name, ms * 1e3);
}
int main() {
    float *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, (size_t)256 * 512 * sizeof(float));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("all 8 waves MFMA (2 per SIMD)", src, out);
        run<1>("all 8 waves VALU (2 per SIMD)", src, out);
        run<3>("waves 0-3 MFMA, 4-7 exit (1 MFMA wave per SIMD)", src, out);
        run<4>("waves 4-7 VALU, 0-3 exit (1 VALU wave per SIMD)", src, out);
        run<2>("waves 0-3 MFMA + waves 4-7 VALU (one of each per SIMD)", src, out);
        run<5>("even waves MFMA, odd waves VALU (same role per SIMD pair)", src, out);
    }
}
