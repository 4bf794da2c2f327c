// How far does v_mfma_f32_32x32x16_bf16's fp32 accumulation (4 instructions = 64 products per output) lie from the exact
// dot product of the same bf16 operands?  Reported relative to sum |a_k b_k| (the bound the matching's filter uses:
// csrc/tome_match_filter.h).  Operands: normal, wide exponent spread (2^-12 .. 2^12), and cancelling pairs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t* A, const uint16_t* B, float* out) {  // A, B: [32 rows][64 k] bf16 row-major
    const int lane = threadIdx.x, row = lane & 31, hf = lane >> 5;
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a, b;
        __builtin_memcpy(&a, A + row * 64 + 16 * ks + 8 * hf, 16);
        __builtin_memcpy(&b, B + row * 64 + 16 * ks + 8 * hf, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);  // rows of B x rows of A
    }
    for (int v = 0; v < 16; ++v) {
        const int j = (v & 3) + 8 * (v >> 2) + 4 * hf, i = row;
        out[i * 32 + j] = acc[v];
    }
}
static uint16_t to_bf16(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
int main() {
    uint16_t *dA, *dB; float* dO;
    hipMalloc(&dA, 32 * 64 * 2); hipMalloc(&dB, 32 * 64 * 2); hipMalloc(&dO, 32 * 32 * 4);
    srand(1);
    double worst[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode)
        for (int trial = 0; trial < 400; ++trial) {
            std::vector<uint16_t> A(2048), B(2048);
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 64; ++k) {
                    auto rnd = [] { return (rand() / (double)RAND_MAX) * 2 - 1; };
                    double a = rnd(), b = rnd();
                    if (mode == 1) { a *= std::ldexp(1.0, rand() % 25 - 12); b *= std::ldexp(1.0, rand() % 25 - 12); }
                    if (mode == 2 && (k & 1)) { a = -from_bf16(A[r * 64 + k - 1]) * (1 + 1e-2 * rnd()); b = from_bf16(B[r * 64 + k - 1]); }
                    A[r * 64 + k] = to_bf16((float)a); B[r * 64 + k] = to_bf16((float)b);
                }
            hipMemcpy(dA, A.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 4096, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dO);
            float O[1024]; hipMemcpy(O, dO, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ex = 0, ab = 0;
                    for (int kk = 0; kk < 64; ++kk) { double p = (double)from_bf16(A[i * 64 + kk]) * from_bf16(B[j * 64 + kk]); ex += p; ab += std::fabs(p); }
                    double e = std::fabs(O[i * 32 + j] - ex) / ab;
                    if (e > worst[mode]) worst[mode] = e;
                }
        }
    printf("max |mfma - exact| / sum|a_k b_k| over 400 x 1024 dot products of 64 bf16 terms:\n");
    printf("  uniform(-1,1)            %.3e  (= %.2f x 2^-24)\n", worst[0], worst[0] * 16777216.0);
    printf("  exponents 2^-12..2^12    %.3e  (= %.2f x 2^-24)\n", worst[1], worst[1] * 16777216.0);
    printf("  cancelling pairs         %.3e  (= %.2f x 2^-24)\n", worst[2], worst[2] * 16777216.0);
}
