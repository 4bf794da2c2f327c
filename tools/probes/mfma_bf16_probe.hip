// Where does a wave's time go in the attention step?  Bare v_mfma_f32_32x32x16_bf16 loops in the dependency shape of
// k_attention_plain's fast step (two score chains of 4 from a constant block, two output chains of 4 each), 8 waves per
// workgroup (2 per SIMD) or 4 (1 per SIMD), with optional vector work per step.  Random operands in registers.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probes/mfma_bf16_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int W, int MODE>
__global__ __launch_bounds__(64 * W) void probe(const float* __restrict__ src, float* out, int steps) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[4], pf[4];
    const float base = src[lane];  // one load; every operand element is a cheap function of it
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(base * (0.5f + 0.01f * (i * 8 + e)));
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { b[i][e] = (__bf16)(base * (0.3f - 0.02f * (i * 8 + e))); pf[i][e] = (__bf16)(base * 0.1f * (e + 1 + i)); }
    f32x16 negm, o0, o1;
    for (int v = 0; v < 16; ++v) { negm[v] = src[lane & 1023]; o0[v] = 0; o1[v] = 0; }
    float l = 0;
    for (int s = 0; s < steps; ++s) {
        f32x16 s0 = negm, s1 = negm;
#pragma unroll
        for (int k = 0; k < 4; ++k) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], s0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4 + k], b[k], s1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], pf[k], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4 + k], pf[k], o1, 0, 0, 0);
        }
        if (MODE >= 1) {  // the softmax's vector work: 32 exp, 32 adds, 16 conversions feeding the next step's operand
#pragma unroll
            for (int v = 0; v < 16; ++v) { s0[v] = __builtin_amdgcn_exp2f(s0[v]); s1[v] = __builtin_amdgcn_exp2f(s1[v]); }
            float c = 0;
#pragma unroll
            for (int v = 0; v < 16; ++v) c += s0[v] + s1[v];
            l += c;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { pf[i][e] = (__bf16)s0[8 * i + e]; pf[2 + i][e] = (__bf16)s1[8 * i + e]; }
        } else {
            l += s0[0] + s1[0];  // keep the chains alive
        }
        if (MODE >= 2) __syncthreads();
    }
    float r = l;
    for (int v = 0; v < 16; ++v) r += o0[v] + o1[v];
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
    printf("%-34s waves/WG %d  WGs %5d steps %4d: %9.1f us  %7.1f TF/s\n", name, W, wgs, steps, ms * 1e3, flops / ms / 1e9);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, (size_t)256 * 42 * 512 * sizeof(float));  // the largest launch below: 256*42 workgroups of 512 threads
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<8, 0>("mfma only, 2 waves/SIMD", src, out, 256, 4000);
    run<4, 0>("mfma only, 1 wave/SIMD", src, out, 256, 4000);
    run<8, 1>("mfma + softmax work, 2 waves/SIMD", src, out, 256, 4000);
    run<4, 1>("mfma + softmax work, 1 wave/SIMD", src, out, 256, 4000);
    run<8, 2>("... + barrier per step, 2 w/SIMD", src, out, 256, 4000);
    run<8, 1>("42 launches-worth of 25-step WGs", src, out, 256 * 42, 25);
    return 0;
}
