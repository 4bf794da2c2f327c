// tome_common.h -- part of the single translation unit csrc/tome_kernels.hip (element types, 16-byte packs, output-row layout).
#pragma once
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// element types
// ------------------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x.v; }

template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
    // round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32 on gfx950)
    __bf16 b = (__bf16)f;
    bf16_t r;
    __builtin_memcpy(&r.v, &b, 2);
    return r;
}
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) {
    f16_t r;
    r.v = (_Float16)f;
    return r;
}

// A lane's slice of a row: VEC consecutive elements moved with one 16-byte (or narrower) access.
template <typename T, int VEC> struct Pack { T e[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ void load_pack(const T *p, float (&out)[VEC]) {
    typedef Pack<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) P;
    P v = *reinterpret_cast<const P *>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = to_f32(v.e[i]);
}

template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T *p, const float (&in)[VEC]) {
    typedef Pack<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) P;
    P v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v.e[i] = from_f32<T>(in[i]);
    *reinterpret_cast<P *>(p) = v;
}

// Output layout of a merged sequence (merge.py:82-85): row of the k-th unmerged A token / of B token j.
__device__ __forceinline__ int out_row_unm(int k, int distill) { return (distill && k >= 1) ? k + 1 : k; }
__device__ __forceinline__ int out_row_dst(int j, int U, int distill) {
    if (!distill) return U + j;
    return j == 0 ? 1 : U + j;
}

