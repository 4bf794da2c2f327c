// Issue cost of the vector instructions the LayerNorm tails are made of, one wave per SIMD and eight (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
// Every variant runs ITER iterations of 16 independent instructions of one kind; cycles per instruction per wave =
// s_memtime delta / (ITER * 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000

#define RUN16(ASM)                                                                                             \
    for (int i = 0; i < ITER; ++i) {                                                                           \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(8) ASM(9) ASM(10) ASM(11) ASM(12) \
                         ASM(13) ASM(14) ASM(15)                                                               \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                     : "v"(b), "v"(c));                                                                        \
    }

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define ADD(n) "v_add_f32 %" #n ", %" #n ", %16\n"
#define SHL(n) "v_lshlrev_b32 %" #n ", 16, %" #n "\n"
#define CND(n) "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
#define DOT(n) "v_dot2c_f32_bf16 %" #n ", %16, %17\n"
#define CVT(n) "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %16\n"
#define CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %16, s[10:11]\n"
#define AND(n) "v_and_b32 %" #n ", 0xffff0000, %" #n "\n"
#define MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %16\n"
#define CMP(n) "v_cmp_eq_u32 vcc, %" #n ", %16\n"
#define MAX3(n) "v_max3_f32 %" #n ", %" #n ", %16, %17\n"
#define CMPCND(n) "v_cmp_lt_f32 vcc, %" #n ", %16\nv_cndmask_b32 %" #n ", %" #n ", %17, vcc\n"
#define CMPCND64(n) "v_cmp_lt_f32_e64 s[10:11], %" #n ", %16\nv_cndmask_b32_e64 %" #n ", %" #n ", %17, s[10:11]\n"
#define DPP(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"

template <int KIND> __global__ void probe(float *out, unsigned long long *cyc, float b, float c) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) { RUN16(FMA) }
    if (KIND == 1) { RUN16(ADD) }
    if (KIND == 2) { RUN16(SHL) }
    if (KIND == 3) { RUN16(CND) }
    if (KIND == 4) { RUN16(DOT) }
    if (KIND == 5) { RUN16(CVT) }
    if (KIND == 6) { RUN16(DPP) }
    if (KIND == 7) { asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11"); RUN16(CND64) }
    if (KIND == 8) { RUN16(AND) }
    if (KIND == 9) { RUN16(MULLO) }
    if (KIND == 10) { RUN16(CMP) }
    if (KIND == 11) { RUN16(MAX3) }
    if (KIND == 13) { RUN16(CMPCND) }
    if (KIND == 14) { asm volatile("" ::: "s10", "s11"); RUN16(CMPCND64) }
    if (KIND == 12) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); RUN16(CND) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> __global__ void probe_pk(float *out, unsigned long long *cyc, float b, float c) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{threadIdx.x * 0.001f + i, 1.0f * i};
    const f32x2 bb = {b, b}, cc = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(bb), "v"(cc));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(bb));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(bb));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K> static void run(const char *name, K kern, int waves_per_block) {
    float *out;
    unsigned long long *cyc;
    const int blocks = 256 * 4;  // one block per SIMD-slot: 4 blocks per CU
    hipMalloc(&out, sizeof(float) * blocks * 64 * waves_per_block);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 64 * waves_per_block>>>(out, cyc, 1.0001f, 0.5f);
    hipEventRecord(e0);
    kern<<<blocks, 64 * waves_per_block>>>(out, cyc, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // wall: instructions issued per SIMD = waves on that SIMD * ITER * 16
    const double instr_per_simd = (double)waves_per_block * ITER * 16.0;  // 4 blocks per CU, waves_per_block/4... see note
    printf("%-36s waves/block %d: %8.1f us   memtime ticks per instr per wave %.3f   (wall ns per instr-slot %.3f)\n", name,
           waves_per_block, ms * 1e3, avg / (ITER * 16.0), ms * 1e6 / instr_per_simd);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {1, 4}) {
        run("v_fma_f32", probe<0>, w);
        run("v_add_f32", probe<1>, w);
        run("v_lshlrev_b32", probe<2>, w);
        run("v_cndmask_b32", probe<3>, w);
        run("v_dot2c_f32_bf16", probe<4>, w);
        run("v_cvt_pk_bf16_f32", probe<5>, w);
        run("v_add_f32_dpp", probe<6>, w);
        run("v_cndmask_b32_e64 sgpr", probe<7>, w);
        run("v_and_b32", probe<8>, w);
        run("v_mul_lo_u32", probe<9>, w);
        run("v_cmp_eq_u32", probe<10>, w);
        run("v_max3_f32", probe<11>, w);
        run("v_cndmask vcc preset", probe<12>, w);
        run("v_cmp vcc + v_cndmask vcc (pair)", probe<13>, w);
        run("v_cmp sgpr + v_cndmask sgpr (pair)", probe<14>, w);
        run("v_pk_fma_f32", probe_pk<0>, w);
        run("v_pk_add_f32", probe_pk<1>, w);
        run("v_pk_mul_f32", probe_pk<2>, w);
    }
    return 0;
}
