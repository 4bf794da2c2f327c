// tome_embed.h -- part of the single translation unit csrc/tome_kernels.hip.
//
// k_tubelet_rows: the regrouping in front of the models' patch embedding.  Every model of the reference embeds its clip
// with a convolution whose stride equals its kernel (VideoMAE `PatchEmbed.proj` Conv3d 2x16x16,
// slowfast/models/videomae_video_model_builder.py:137-166; TimeSformer's per-frame Conv2d 16x16; Motionformer
// `PatchEmbed3D`; ViViT's tubelet Conv3d), i.e. `rows[B*N, C*kt*kh*kw] @ weight^T` with
//     rows[b, (t', h', w'), (c, dt, dh, dw)] = x[b, c, t'*kt + dt, h'*kh + dh, w'*kw + dw].
// The hosts make `rows` and hand it to the library GEMM; as a framework permute-copy that is 1.78 ms for 384 VideoMAE
// clips (1.0 TB/s: the copy kernel walks the output element by element).  Here: a pure 16-byte move.  One lane owns one
// 16-byte chunk of the inner (c, dt, dh, dw) order for a whole strip of w' -- its sources lie kw elements apart in ONE
// input row, its destinations one token row apart -- so a wave's stores are whole contiguous token rows and every
// input row is consumed completely by the lanes of one workgroup (the 128-byte lines it shares between neighbouring
// tokens come from L2 / L1, HBM sees them once).  x may be any view with unit stride along W ({b, c, t, h} element strides
// are arguments): TimeSformer's frames (kt = 1) and ViViT's [B, T, C, H, W] clips are read where they lie.
#pragma once

#define TUBE_UNROLL 7

struct TubeArgs {
    int64_t sb, sc, st, sh;  // element strides of x along b, c, t, h
    int nt, nh, nw;          // tokens along t, h, w
    int kt, kh;              // tubelet extent along t, h
    int cpr;                 // 16-byte chunks per run of kw elements
    int chunks;              // chunks per token = C * kt * kh * cpr
    int64_t items;           // strips (b, t', h') x chunks
};

template <int ES>  // element size in bytes
__global__ __launch_bounds__(256) void k_tubelet_rows(const uint8_t *__restrict__ x, uint8_t *__restrict__ rows, TubeArgs a) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.items) return;
    const int64_t strip = g / a.chunks;
    const int e = (int)(g - strip * a.chunks);
    const int run = e / a.cpr, part = e - run * a.cpr;
    const int dh = run % a.kh, dt = (run / a.kh) % a.kt, c = run / (a.kh * a.kt);
    const int hp = (int)(strip % a.nh);
    const int64_t bt = strip / a.nh;
    const int tp = (int)(bt % a.nt);
    const int64_t b = bt / a.nt;
    const uint8_t *src = x + (b * a.sb + c * a.sc + (int64_t)(tp * a.kt + dt) * a.st + (int64_t)(hp * a.kh + dh) * a.sh) * ES
                         + part * 16;
    const int64_t token_bytes = (int64_t)a.chunks * 16;
    const int run_bytes = a.cpr * 16;
    uint8_t *dst = rows + strip * a.nw * token_bytes + (int64_t)e * 16;
    for (int j0 = 0; j0 < a.nw; j0 += TUBE_UNROLL) {
        uint4 v[TUBE_UNROLL];
#pragma unroll
        for (int u = 0; u < TUBE_UNROLL; ++u)
            if (j0 + u < a.nw) v[u] = *reinterpret_cast<const uint4 *>(src + (int64_t)(j0 + u) * run_bytes);
#pragma unroll
        for (int u = 0; u < TUBE_UNROLL; ++u)
            if (j0 + u < a.nw) st16(dst + (int64_t)(j0 + u) * token_bytes, v[u]);
    }
}
