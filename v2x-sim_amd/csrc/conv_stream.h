// Declarations of the streamed-weights kernel families of conv_stream.hip (4-wave, 8-wave ping-pong, wide forms): kernel arguments, LDS / LDS-DMA
// helpers, patch geometry.
#pragma once
#include "common.h"

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef const __attribute__((address_space(3))) float lds_cf_t;     // LDS-resident float (explicit address space: a generic pointer would be a FLAT load)
typedef const __attribute__((address_space(3))) f32x4_t lds_cf4_t;   // (the builtin vector type: HIP's float4 class cannot be read through an address-space pointer)
__device__ __forceinline__ float4 lds_ld4(lds_cf_t *p) {
    const f32x4_t v = *reinterpret_cast<lds_cf4_t *>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void glds16s(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
// the same with a cache-policy field (aux: 2 = nt, 1 = sc0, 16 = sc1 on gfx950): round 6's experiment on the ConvGRU's two streams -- a non-temporal PATCH stream so that
// the 5 MB of patches an XCD pulls per round stop evicting its 3.5 MB of weights from the 4-MiB L2 (the weights are re-fetched once per round: profiles/r06_fetch_calibration.txt).
// Compile-time (-DV2X_STREAM8G_PATCH_AUX=n / -DV2X_STREAM8G_WEIGHT_AUX=n, tools/ab_build.sh); 0 = the default policy = glds16s.
#ifndef V2X_STREAM8G_PATCH_AUX
#define V2X_STREAM8G_PATCH_AUX 0
#endif
#ifndef V2X_STREAM8G_WEIGHT_AUX
#define V2X_STREAM8G_WEIGHT_AUX 0
#endif
template <int AUX>
__device__ __forceinline__ void glds16s_aux(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, AUX);
}

struct StreamArgs {
    const uint16_t *in0, *in1;  // in0: first C0 channels (half resolution when up0), in1: next C1 channels
    int C0, C1, up0;
    int N, H, W;
    const uint16_t *w;  // [n_co_tiles][n_chunks][9][4][BCO][8] bf16, followed by 64 B of zeros (the zero page)
    const float *scale, *shift;
    int relu;
    void *out;
    int out_cstride, out_coff, Cout;
    int tiles_x, tiles_y, n_px_tiles, n_co_tiles;
    // chained 1x1 (SEPI_CHAIN, BCO == Cout == 64): hidden = relu(acc*scale+shift) never leaves the registers
    const uint16_t *w2;   // bf16 [64][64] row-major, K in the chain (kappa) order of conv_halo.hip
    const float *scale2, *shift2;
    int relu2;
    // split-K (small batches, conv3x3_stream_kernel<..., SPLITK = true>): blockIdx.y walks `ksplit` contiguous ranges of the 32-channel
    // chunks and stores its raw fp32 sums to ws[split][pixel][w_rows]; splitk_reduce_kernel adds them in split order and applies the epilogue
    int ksplit;
    float *ws;
    int w_rows;
    int x4;         // plain epilogue: 16-byte output stores through v_permlane16_swap_b32 (conv_stream.hip: stream_epilogue X4); set by the dispatch when the output rows allow
    int xcd_walk;   // stream8g, 8 channel tiles on 256 workgroups: 1 = an XCD walks 8 pixel tiles x 4 channel tiles per round instead of 4 x 8 (tuning switch GRU_XCD_WALK; the ConvGRU)
};

constexpr int PATCH_PIECES = 24;             // wave instructions (1 KiB each) per patch buffer
constexpr int PATCH_BYTES = PATCH_PIECES * 1024;
constexpr int RING = 4;                      // weight slices in flight + 1 being read

enum { SEPI_BF16 = 0, SEPI_CHAIN = 1, SEPI_GRU = 2 };

constexpr int PATCH8_PIECES = 40;            // 18 x 34 pixels x 64 B = 38.25 KiB; piece 39 is padding / dummy target
constexpr int PATCH8_BYTES = PATCH8_PIECES * 1024;

int v2x_num_cus();
