// Probe: does a deeper software pipeline (every MFMA gap carrying ~5 softmax VALU of ANOTHER half-tile) beat the
// phase-separated step?  Register operands only.  hipcc --offload-arch=gfx950 -O3 -o attn_pipe_probe attn_pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float att_add(float x, float y) { float r = x + y; asm("" : "+v"(r)); return r; }

#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int W, int MODE>
__global__ __launch_bounds__(64 * W, 2) void probe(const float* __restrict__ src, float* out, int steps) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[4], pf[4], pn[2];
    const float base = src[lane];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(base * (0.5f + 0.01f * (i * 8 + e)));
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { b[i][e] = (__bf16)(base * (0.3f - 0.02f * (i * 8 + e))); pf[i][e] = (__bf16)(base * 0.1f * (e + 1 + i)); }
    for (int i = 0; i < 2; ++i) pn[i] = pf[i];
    f32x16 negm, o0, o1, s1p;
    for (int v = 0; v < 16; ++v) { negm[v] = src[lane & 1023]; o0[v] = 0; o1[v] = 0; s1p[v] = negm[v]; }
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    for (int s = 0; s < steps; ++s) {
        if (MODE <= 2) {
            f32x16 s0 = negm, s1 = negm;
#pragma unroll
            for (int k = 0; k < 4; ++k) s0 = MF(a[k], b[k], s0);
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 = MF(a[4 + k], b[k], s1);
#pragma unroll
            for (int k = 0; k < 4; ++k) { o0 = MF(a[k], pf[k], o0); o1 = MF(a[4 + k], pf[k], o1); }
            if (MODE >= 1) {
#pragma unroll
                for (int v = 0; v < 16; ++v) { s0[v] = __builtin_amdgcn_exp2f(s0[v]); s1[v] = __builtin_amdgcn_exp2f(s1[v]); }
#pragma unroll
                for (int v = 0; v < 16; v += 4) {
                    l0 = att_add(l0, s0[v]); l1 = att_add(l1, s0[v + 1]); l2 = att_add(l2, s0[v + 2]); l3 = att_add(l3, s0[v + 3]);
                    l0 = att_add(l0, s1[v]); l1 = att_add(l1, s1[v + 1]); l2 = att_add(l2, s1[v + 2]); l3 = att_add(l3, s1[v + 3]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pf[i][e] = (__bf16)s0[8 * i + e]; pf[2 + i][e] = (__bf16)s1[8 * i + e]; }
            } else {
                l0 += s0[0] + s1[0];
            }
            if (MODE == 2) __syncthreads();
        } else {
            // deeper pipeline: this step's 16 MFMAs carry the softmax of s1(prev) [under the score MFMAs] and of
            // s0(this) [under the PV MFMAs]
            f32x16 s0 = negm, s1 = negm;
            // -- scores (8 MFMAs)
#pragma unroll
            for (int k = 0; k < 4; ++k) s0 = MF(a[k], b[k], s0);
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 = MF(a[4 + k], b[k], s1);
            // -- softmax of s1p -> pf[2], pf[3]
#pragma unroll
            for (int v = 0; v < 16; ++v) s1p[v] = __builtin_amdgcn_exp2f(s1p[v]);
#pragma unroll
            for (int v = 0; v < 16; v += 4) { l0 = att_add(l0, s1p[v]); l1 = att_add(l1, s1p[v + 1]); l2 = att_add(l2, s1p[v + 2]); l3 = att_add(l3, s1p[v + 3]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[2 + i][e] = (__bf16)s1p[8 * i + e];
            pf[0] = pn[0]; pf[1] = pn[1];
            // -- PV of the previous tile (8 MFMAs)
#pragma unroll
            for (int k = 0; k < 4; ++k) { o0 = MF(a[k], pf[k], o0); o1 = MF(a[4 + k], pf[k], o1); }
            // -- softmax of s0 -> pn[0], pn[1]
#pragma unroll
            for (int v = 0; v < 16; ++v) s0[v] = __builtin_amdgcn_exp2f(s0[v]);
#pragma unroll
            for (int v = 0; v < 16; v += 4) { l0 = att_add(l0, s0[v]); l1 = att_add(l1, s0[v + 1]); l2 = att_add(l2, s0[v + 2]); l3 = att_add(l3, s0[v + 3]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) pn[i][e] = (__bf16)s0[8 * i + e];
            s1p = s1;
            if (MODE >= 4) {
                // 16 x [1 MFMA, 2 TRANS, 3 VALU]
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            if (MODE == 5) __syncthreads();
        }
    }
    float r = l0 + l1 + l2 + l3;
    for (int v = 0; v < 16; ++v) r += o0[v] + o1[v] + s1p[v];
    out[blockIdx.x * 64 * W + threadIdx.x] = r;
}

template <int W, int MODE>
void run(const char* name, const float* src, float* out, int wgs, int steps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<W, MODE><<<wgs, 64 * W>>>(src, out, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<W, MODE><<<wgs, 64 * W>>>(src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)wgs * W * steps * 16 * 32768.0;
    printf("%-44s waves/WG %d: %9.1f us  %7.1f TF/s\n", name, W, ms * 1e3, flops / ms / 1e9);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, (size_t)256 * 512 * sizeof(float));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 0>("mfma only", src, out, 256, 4000);
        run<8, 1>("phase-separated softmax work", src, out, 256, 4000);
        run<8, 2>("phase-separated + barrier", src, out, 256, 4000);
        run<8, 3>("deep pipeline, compiler order", src, out, 256, 4000);
        run<8, 4>("deep pipeline, sched_group_barrier", src, out, 256, 4000);
        run<8, 5>("deep pipeline, sgb + barrier", src, out, 256, 4000);
        run<4, 1>("phase-separated softmax work", src, out, 256, 4000);
        run<4, 4>("deep pipeline, sched_group_barrier", src, out, 256, 4000);
    }
    return 0;
}
