// Weight gradient of the 3x3 stride-1 pad-1 convolutions (the first hand-written backward
// kernel -- until now the whole backward pass was MIOpen's through torch.autograd).
//
//     dW[co][ky][kx][ci] = sum over (n, y, x) of  dY[n][y][x][co] * X[n][y + ky - 1][x + kx - 1][ci]        (zero padding)
//
// As a GEMM this contracts over PIXELS: M = Cout, N = 9*Cin, K = N*H*W, while both operands live in HBM channel-contiguous
// (NHWC).  The MFMA wants 8 consecutive k per lane, so both operands are TRANSPOSED on their way into LDS (16-B global
// loads, eight ds_write_b16 each): dY tile as [co][pixel], the input patch as [ci][patch row][column] in THREE copies
// shifted by the tap column kx, so that every fragment read is an aligned ds_read_b128 of 8 consecutive pixels.
//
// Work split: block = (64 output channels) x (32 input channels) x (a share of the 8x32-pixel tiles).  Wave w owns 16
// output channels x 9 taps x 2 input-channel sub-tiles = 18 accumulator tiles (72 VGPRs) that stay in registers over all
// the block's pixel tiles.  Per patch row pr the wave reads 6 B fragments (3 tap columns x 2 ci sub-tiles) and one new A
// fragment and issues the MFMAs of the (output row r, tap row ky) pairs with r + ky = pr: 7 fragment reads per 18 MFMAs.
// The partial sums of the blocks that share (co tile, ci tile) go to a workspace [split][Cout][9][Cin] fp32 and are added
// up by the caller in a fixed order -- deterministic, unlike an atomicAdd reduction (and unlike MIOpen's wgrad).
//
// 32-row form (Cout % 64 == 32: the full-resolution 32-channel layers conv_pre_2 / conv8_1 / conv8_2 and the head hiddens): the block
// covers 32 output channels; waves 0-1 take the tile's output rows 0-3, waves 2-3 rows 4-7 and write to two DIFFERENT workspace
// slots (2*split, 2*split + 1), so the halves meet in the caller's fixed-order sum like any other pair of partials.
#include "common.h"

constexpr int WG_TH = 8, WG_TW = 32;                 // pixel tile
constexpr int WG_CO = 64, WG_CI = 32;                // channel block (WG_CO: the 64-row form; LDS is sized for it)
constexpr int WG_PX = WG_TH * WG_TW;                 // 256 pixels = k extent of one tile
constexpr int WG_DY_ROW = WG_PX * 2 + 16;            // bytes per output channel of the transposed dY tile (+16: bank spread)
constexpr int WG_X_ROW = (WG_TH + 2) * WG_TW * 2 + 16;  // bytes per input channel of one shifted patch copy
constexpr int WG_SMEM = WG_CO * WG_DY_ROW + 3 * WG_CI * WG_X_ROW;

struct WgradArgs {
    const uint16_t *x;    // [N][H][W][Cin] bf16
    const uint16_t *dy;   // [N][H][W][Cout] bf16
    float *ws;            // [n_split][Cout][9][Cin] fp32 partial sums (every element written)
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, n_tiles, n_split;   // n_split = number of blocks that share one (co tile, ci tile) pair
};

template <int CO>   // 64: wave w owns co 16w..16w+15, all 8 tile rows;  32: wave w owns co 16(w&1).., rows 4(w>>1)..4(w>>1)+3
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradArgs a) {
    constexpr int RT = CO == 64 ? WG_TH : WG_TH / 2;   // output rows of the tile one wave contracts over
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_dy = smem;                                 // [64 co][256 px]
    char *s_x = smem + WG_CO * WG_DY_ROW;              // [3 kx][32 ci][10 rows][32 cols]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    const int wco = CO == 64 ? wave : (wave & 1);      // the wave's 16-channel sub-tile
    const int row0 = CO == 64 ? 0 : (wave >> 1) * RT;  // its first output row
    const int n_ci_t = a.Cin / WG_CI;
    const int blk = blockIdx.x;
    const int split = blk % a.n_split;
    const int cc = blk / a.n_split;
    const int ci_t = cc % n_ci_t, co_t = cc / n_ci_t;
    const int txy = a.tiles_x * a.tiles_y;

    f32x4_t acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int tile = split; tile < a.n_tiles; tile += a.n_split) {
        const int n = tile / txy;
        const int r0 = tile - n * txy;
        const int ty = r0 / a.tiles_x;
        const int y0 = ty * WG_TH, x0 = (r0 - ty * a.tiles_x) * WG_TW;
        __syncthreads();   // everyone is done reading the previous tile
        // ---- dY tile, transposed: thread = pixel, loop over the 8 groups of 8 output channels
        {
            const int pr = tid >> 5, pc = tid & 31;
            const uint16_t *src = a.dy + ((size_t)(n * a.H + y0 + pr) * a.W + x0 + pc) * a.Cout + co_t * CO;
#pragma unroll
            for (int cg = 0; cg < CO / 8; ++cg) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + cg * 8);
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<uint16_t *>(s_dy + (cg * 8 + e) * WG_DY_ROW + tid * 2) = (uint16_t)(wds[e >> 1] >> ((e & 1) * 16));
            }
        }
        // ---- input patch (10 x 34 pixels x 32 channels), transposed into the three kx-shifted copies
        for (int p = tid; p < (WG_TH + 2) * (WG_TW + 2); p += 256) {
            const int pr = p / (WG_TW + 2), pc = p - pr * (WG_TW + 2);
            const int y = y0 - 1 + pr, x = x0 - 1 + pc;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const uint16_t *src = a.x + ((size_t)(n * a.H + (ok ? y : 0)) * a.W + (ok ? x : 0)) * a.Cin + ci_t * WG_CI;
#pragma unroll
            for (int cg = 0; cg < WG_CI / 8; ++cg) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ok) v = *reinterpret_cast<const uint4 *>(src + cg * 8);
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int c = pc - kx;   // copy kx holds patch column c + kx at column c
                    if (c < 0 || c >= WG_TW) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        *reinterpret_cast<uint16_t *>(s_x + (kx * WG_CI + cg * 8 + e) * WG_X_ROW + (pr * WG_TW + c) * 2) =
                            (uint16_t)(wds[e >> 1] >> ((e & 1) * 16));
                }
            }
        }
        __syncthreads();
        // ---- MFMAs: patch row row0 + pr serves (r, ky) with r + ky = pr, r = output row relative to row0
        bf16x8_t A[3];   // A[pr % 3] = dY^T fragment of output row row0 + pr (16 co x 32 px)
#pragma unroll
        for (int pr = 0; pr < RT + 2; ++pr) {
            if (pr < RT)
                A[pr % 3] = *reinterpret_cast<const bf16x8_t *>(s_dy + (wco * 16 + fj) * WG_DY_ROW + ((row0 + pr) * WG_TW + fq * 8) * 2);
            bf16x8_t B[3][2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    B[kx][j] = *reinterpret_cast<const bf16x8_t *>(s_x + (kx * WG_CI + j * 16 + fj) * WG_X_ROW + ((row0 + pr) * WG_TW + fq * 8) * 2);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = pr - ky;
                if (r < 0 || r >= RT) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ky * 3 + kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r % 3], B[kx][j], acc[ky * 3 + kx][j], 0, 0, 0);
            }
        }
    }
    // ---- partial result of this block: ws[split][co][tap][ci]; lane (fj, fq) holds rows co = 4*fq + e, column ci = fj
    const int slot = CO == 64 ? split : 2 * split + (wave >> 1);
    float *dst = a.ws + (size_t)slot * a.Cout * 9 * a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co_t * CO + wco * 16 + fq * 4 + e;
                const int ci = ci_t * WG_CI + j * 16 + fj;
                dst[((size_t)co * 9 + t) * a.Cin + ci] = acc[t][j][e];
            }
}

// ---- second form (round 3): natural-layout tiles by LDS-DMA, fragments by the gfx950 TRANSPOSE READ -------------------------------------
// The kernel above spends as long transposing its operands into LDS (eight ds_write_b16 per 16 loaded bytes, three tap-shifted copies of
// the input patch: ~190 LDS stores per thread and tile, synchronous global loads in front of them) as on its MFMAs, with 94 KiB of LDS = one
// workgroup per CU and nothing to overlap with.  Here the dY tile [256 px][CO] and the input patch [10 x 34 px][32 ci] land in LDS in their
// NATURAL channel-contiguous layout by LDS-DMA (no VALU, no shifted copies: a tap shift is a pixel offset), and every MFMA operand -- 8
// consecutive PIXELS of one channel per lane -- is two ds_read_b64_tr_b16: within a 16-lane group lane i points at the 4 channels
// 4 (i & 3) .. of pixel p0 + (i >> 2) and receives channel i of the pixels p0 .. p0 + 3 (tools/tr_read_probe.hip).  53 KiB of LDS: THREE
// workgroups per CU cover each other's DMA waits.  Bank spread (a 32-lane half of the read = pixels p0 .. p0 + 3 and p0 + 8 .. p0 + 11 must
// touch 64 distinct banks): the 32-byte channel groups of a pixel are XOR-ed with pixel-column bits on the DMA's SOURCE side -- dY: group ^
// (bit 1 | bit 3 << 1 of the column), patch: half ^ bit 3 of the patch column.  Same sums in the same order as the first form: bit-identical.
typedef __attribute__((ext_vector_type(4))) short wg_s16x4_t;
typedef const __attribute__((address_space(1))) void *wg_gptr_t;
typedef __attribute__((address_space(3))) void *wg_lptr_t;
__device__ __attribute__((aligned(64))) const uint32_t g_wgrad_zero_page[16] = {0};
constexpr int WT_X_PIX = (WG_TH + 2) * (WG_TW + 2);                 // 340 patch pixels
constexpr int WT_X_BYTES = ((WT_X_PIX * 64 + 1023) / 1024) * 1024;   // 22 KiB (whole 1-KiB DMA pieces)

__device__ __forceinline__ bf16x8_t wg_tr_frag(const char *lds, int off_lo, int off_hi) {
    const wg_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4_t *)(lds + off_lo));
    const wg_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4_t *)(lds + off_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int CO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void conv3x3_wgrad_tr_kernel(const WgradArgs a) {
    constexpr int RT = CO == 64 ? WG_TH : WG_TH / 2;
    constexpr int DY_BYTES = WG_PX * CO * 2;             // 32 KiB (CO = 64) / 16 KiB
    constexpr int DY_PIECES = DY_BYTES / 1024;           // LDS-DMA wave instructions for the dY tile
    constexpr int X_PIECES = WT_X_BYTES / 1024;          // 22
    constexpr int CPP = CO / 8;                          // 16-byte chunks per dY pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_dy = smem;
    char *s_x = smem + DY_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    const int wco = CO == 64 ? wave : (wave & 1);
    const int row0 = CO == 64 ? 0 : (wave >> 1) * RT;
    const int n_ci_t = a.Cin / WG_CI;
    const int blk = blockIdx.x;
    const int split = blk % a.n_split;
    const int cc = blk / a.n_split;
    const int ci_t = cc % n_ci_t, co_t = cc / n_ci_t;
    const int txy = a.tiles_x * a.tiles_y;

    // per-lane read offsets (tile-invariant).  Lane (i = fj, k-slot fq), half h: pixel column 8 fq + 4 h + (i >> 2), channels 4 (i & 3) ..
    int dy_off[2], x_off[3][2];          // dY: [h];  patch: [kx][h], for ci sub-tile 0 (sub-tile 1 = the other 32-byte half: ^ 32)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = 8 * fq + 4 * h + (fj >> 2);
        // 128-byte pixels (CO = 64): rows p and p + 2 share their banks -> 2 swizzle bits (column bits 1 and 3); 64-byte pixels: bit 3 only
        const int sw = CO == 64 ? (((col >> 1) & 1) | (((col >> 3) & 1) << 1)) : ((col >> 3) & 1);
        dy_off[h] = col * (CO * 2) + ((wco ^ sw) * 32) + (fj & 3) * 8;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pc = col + kx;
            x_off[kx][h] = pc * 64 + (((pc >> 3) & 1) * 32) + (fj & 3) * 8;
        }
    }

    f32x4_t acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int tile = split; tile < a.n_tiles; tile += a.n_split) {
        const int n = tile / txy;
        const int r0 = tile - n * txy;
        const int ty = r0 / a.tiles_x;
        const int y0 = ty * WG_TH, x0 = (r0 - ty * a.tiles_x) * WG_TW;
        __syncthreads();   // everyone is done reading the previous tile
        // ---- dY tile: LDS slot L (16 B) = pixel L / CPP, physical chunk L % CPP; logical chunk = physical ^ (swizzle << 1)
#pragma unroll
        for (int t = 0; t < DY_PIECES / 4; ++t) {
            const int piece = wave + 4 * t;
            const int L = piece * 64 + lane;
            const int px = L / CPP, ch = L - px * CPP;
            const int col = px & 31;
            const int sw = CO == 64 ? (((col >> 1) & 1) | (((col >> 3) & 1) << 1)) : ((col >> 3) & 1);
            const int lch = ch ^ (sw << 1);
            const uint16_t *src = a.dy + ((size_t)(n * a.H + y0 + (px >> 5)) * a.W + x0 + col) * a.Cout + co_t * CO + lch * 8;
            __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(s_dy + piece * 1024), 16, 0, 0);
        }
        // ---- input patch: slot L = patch pixel L / 4, physical chunk L % 4; logical chunk = physical ^ (bit 3 of the patch column << 1)
#pragma unroll
        for (int t = 0; t < (X_PIECES + 3) / 4; ++t) {
            const int piece = wave + 4 * t;
            if (piece >= X_PIECES) break;     // wave-uniform
            const int L = piece * 64 + lane;
            const int p = L >> 2, ch = L & 3;
            const int pr = p / (WG_TW + 2), pc = p - pr * (WG_TW + 2);
            const int y = y0 - 1 + pr, x = x0 - 1 + pc;
            const bool ok = p < WT_X_PIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const int lch = ch ^ (((pc >> 3) & 1) << 1);
            const void *src = ok ? (const void *)(a.x + ((size_t)(n * a.H + y) * a.W + x) * a.Cin + ci_t * WG_CI + lch * 8) : (const void *)g_wgrad_zero_page;
            __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(s_x + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- MFMAs: patch row row0 + pr serves (r, ky) with r + ky = pr, r = output row relative to row0
        bf16x8_t A[3];
#pragma unroll
        for (int pr = 0; pr < RT + 2; ++pr) {
            if (pr < RT) A[pr % 3] = wg_tr_frag(s_dy + (row0 + pr) * (WG_TW * CO * 2), dy_off[0], dy_off[1]);
            bf16x8_t B[3][2];
            const char *xrow = s_x + (row0 + pr) * ((WG_TW + 2) * 64);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 2; ++j) B[kx][j] = wg_tr_frag(xrow, x_off[kx][0] ^ (j * 32), x_off[kx][1] ^ (j * 32));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = pr - ky;
                if (r < 0 || r >= RT) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ky * 3 + kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r % 3], B[kx][j], acc[ky * 3 + kx][j], 0, 0, 0);
            }
        }
    }
    const int slot = CO == 64 ? split : 2 * split + (wave >> 1);
    float *dst = a.ws + (size_t)slot * a.Cout * 9 * a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co_t * CO + wco * 16 + fq * 4 + e;
                const int ci = ci_t * WG_CI + j * 16 + fj;
                dst[((size_t)co * 9 + t) * a.Cin + ci] = acc[t][j][e];
            }
}

// dW[co][ci][ky][kx] (the parameter's own OIHW layout, ci < cin_out) = sum over the n_split partials ws[s][co][tap][ci], in slot order
// (deterministic).  Replaces the caller's `ws.sum(0).permute(0, 3, 1, 2).contiguous()[:, :cin]` (a reduction and a permuting copy per layer).
// A block = 32 output elements (ci fastest: coalesced reads of the partials) x 8 slices of the split range: a thread adds its slice's
// partials in slot order, the 8 slice sums of an element are added in slice order -- a fixed tree, so the result does not depend on the
// launch geometry (the small layers have up to 512 partials per element: one thread per element walked them as 512 dependent loads).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ ws, int n_split, int Cout, int Cin, int cin_out,
                                                           float *__restrict__ dw) {
    __shared__ float part[8][32];
    const long long total = (long long)Cout * 9 * cin_out;
    const size_t slab = (size_t)Cout * 9 * Cin;
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int per = (n_split + 7) / 8;
    for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {
        const long long t = base + e;
        float s = 0.0f;
        long long ct = 0;
        int ci = 0;
        if (t < total) {
            ci = (int)(t % cin_out);
            ct = t / cin_out;                       // co * 9 + tap
            const int k1 = (q + 1) * per < n_split ? (q + 1) * per : n_split;
            for (int k = q * per; k < k1; ++k) s += ws[(size_t)k * slab + (size_t)ct * Cin + ci];
        }
        part[q][e] = s;
        __syncthreads();
        if (q == 0 && t < total) {
            float v = part[0][e];
#pragma unroll
            for (int j = 1; j < 8; ++j) v += part[j][e];
            const int tap = (int)(ct % 9);
            const int co = (int)(ct / 9);
            dw[((size_t)co * cin_out + ci) * 9 + tap] = v;
        }
        __syncthreads();
    }
}

// The same sums for FOUR consecutive ci per thread (cin_out % 4 == 0: every layer but the 13-channel first one), 16-byte loads, eight partials in flight per thread
// (round 6).  The scalar kernel above walks a slice's partials as one 4-byte load per add -- 48-64 dependent load-add pairs per thread: 15 us per launch for 19-28 MB of
// partials (1.8 TB/s; 24 launches = 7 % of a 10-map FaFNet step, profiles/r06_train_step_profile.txt).  Same association as the scalar kernel -- a slice's partials in slot
// order, the 8 slice sums in slice order -- so the two forms give the same bits (tests/test_gpu_train_kernels.py).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float *__restrict__ ws, int n_split, int Cout, int Cin, int cin_out, float *__restrict__ dw) {
    __shared__ float4 part[8][32];
    const int c4 = cin_out >> 2;
    const long long total = (long long)Cout * 9 * c4;          // float4 elements
    const size_t slab = (size_t)Cout * 9 * Cin;
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int per = (n_split + 7) / 8;
    for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {
        const long long t = base + e;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        long long ct = 0;
        int ci = 0;
        if (t < total) {
            ci = (int)(t % c4) * 4;
            ct = t / c4;                            // co * 9 + tap
            const int k0 = q * per;
            const int k1 = (q + 1) * per < n_split ? (q + 1) * per : n_split;
            const float *p = ws + (size_t)ct * Cin + ci;
            int k = k0;
            for (; k + 8 <= k1; k += 8) {           // eight loads in flight, added in slot order
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(p + (size_t)(k + u) * slab);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s.x += v[u].x;
                    s.y += v[u].y;
                    s.z += v[u].z;
                    s.w += v[u].w;
                }
            }
            for (; k < k1; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)k * slab);
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
        }
        part[q][e] = s;
        __syncthreads();
        if (q == 0 && t < total) {
            float4 v = part[0][e];
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                v.x += part[j][e].x;
                v.y += part[j][e].y;
                v.z += part[j][e].z;
                v.w += part[j][e].w;
            }
            const int tap = (int)(ct % 9);
            const int co = (int)(ct / 9);
            float *d = dw + ((size_t)co * cin_out + ci) * 9 + tap;      // OIHW: consecutive ci are 9 floats apart
            d[0] = v.x;
            d[9] = v.y;
            d[18] = v.z;
            d[27] = v.w;
        }
        __syncthreads();
    }
}

extern "C" int v2x_conv3x3_wgrad_reduce(const float *workspace, int n_split, int Cout, int Cin, int cin_out, float *dw_oihw, v2x_stream_t stream) {
    V2X_REQUIRE(workspace && dw_oihw && n_split >= 1 && Cout > 0 && Cin > 0 && cin_out > 0 && cin_out <= Cin, "v2x_conv3x3_wgrad_reduce: bad arguments");
    if (cin_out % 4 == 0 && Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && v2x_tune(V2X_TUNE_WGRAD_REDUCE4) != 0) {
        const long long total4 = (long long)Cout * 9 * (cin_out / 4);
        const long long blocks4 = (total4 + 31) / 32;
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)(blocks4 < 8192 ? blocks4 : 8192)), dim3(256), 0, (hipStream_t)stream, workspace, n_split, Cout, Cin,
                           cin_out, dw_oihw);
        V2X_CHECK_LAUNCH("wgrad_reduce4_kernel");
        return V2X_OK;
    }
    const long long total = (long long)Cout * 9 * cin_out;
    const long long blocks = (total + 31) / 32;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, workspace, n_split, Cout, Cin,
                       cin_out, dw_oihw);
    V2X_CHECK_LAUNCH("wgrad_reduce_kernel");
    return V2X_OK;
}

extern "C" int v2x_conv3x3_wgrad_splits(int N, int H, int W, int Cin, int Cout) {
    // number of workspace slots the library would use: enough blocks for one and a half rounds of the 256 CUs, at most one block per pixel tile
    if (N <= 0 || H <= 0 || W <= 0 || H % WG_TH || W % WG_TW || Cin <= 0 || Cin % WG_CI || Cout <= 0 || Cout % 32) return 0;
    const bool rows32 = Cout % WG_CO != 0;
    const long long tiles = (long long)N * (H / WG_TH) * (W / WG_TW);
    const long long pairs = (long long)(Cout / (rows32 ? 32 : WG_CO)) * (Cin / WG_CI);
    // Round 6, measured and rejected (commit 943fb9d holds the code; profiles/r06_wgrad_tr2_rejected.txt): a third form of the kernel -- two tile buffers (the
    // DMAs of tile i + 1 under the MFMAs of tile i, counted vmcnt) and a register-pipelined row loop (counted lgkmcnt(14): a row's transpose reads under the
    // previous row's MFMAs), grid = the residency (1 workgroup per CU at 108 KiB, 2 at 76 KiB) -- bit-identical, and NO faster: FaFNet step 11.45 against 11.39 ms
    // at 40 maps, 4.616 against 4.612 at 10.  The kernel is not waiting where that form helps: a 40-map step's weight gradients must read >= 4.5 GB of
    // activations and activation gradients once (>= 0.9 ms at 5 TB/s) and read 7.1 GB because a dY tile is fetched once per 32-channel input tile of its layer
    // (conv7_1: 6 x, conv8_1: 3 x) -- 2.3 ms is 1.6 x that stream, not 4 x its MFMAs.
    // (384 workgroups = one and a half rounds of the CUs: every workgroup writes a 74-KiB partial that the reduce reads back -- 0.9 GB each way per
    // 10-map FaFNet step at 512 -- so fewer, longer workgroups win until the chip under-fills: captured step 5.19 / 5.13 / 5.17 / 5.20 ms at
    // 512 / 384 / 320 / 256, V2VNet 6.34 / 6.26 / 6.28 / 6.33)
    long long n = (384 + pairs - 1) / pairs;
    if (n > 256) n = 256;
    if (n > tiles) n = tiles;
    if (n < 1) n = 1;
    return (int)(rows32 ? 2 * n : n);
}

extern "C" int v2x_conv3x3_wgrad(const uint16_t *x, const uint16_t *dy, int N, int H, int W, int Cin, int Cout, float *workspace,
                                 int n_split, v2x_stream_t stream) {
    V2X_REQUIRE(x && dy && workspace, "v2x_conv3x3_wgrad: null pointer");
    V2X_REQUIRE(N > 0 && H > 0 && W > 0 && H % WG_TH == 0 && W % WG_TW == 0 && Cin > 0 && Cin % WG_CI == 0 && Cout > 0 && Cout % 32 == 0,
                "v2x_conv3x3_wgrad: needs H %% 8 == 0, W %% 32 == 0, Cin %% 32 == 0, Cout %% 32 == 0");
    const bool rows32 = Cout % WG_CO != 0;
    WgradArgs a;
    a.x = x;
    a.dy = dy;
    a.ws = workspace;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = Cin;
    a.Cout = Cout;
    a.tiles_x = W / WG_TW;
    a.tiles_y = H / WG_TH;
    a.n_tiles = N * a.tiles_x * a.tiles_y;
    if (rows32) {
        V2X_REQUIRE(n_split >= 2 && n_split % 2 == 0 && n_split / 2 <= a.n_tiles,
                    "v2x_conv3x3_wgrad: Cout %% 64 != 0 takes an even n_split in [2, %d] (two workspace slots per block), got %d", 2 * a.n_tiles, n_split);
        a.n_split = n_split / 2;
    } else {
        V2X_REQUIRE(n_split >= 1 && n_split <= a.n_tiles, "v2x_conv3x3_wgrad: n_split=%d outside [1, %d]", n_split, a.n_tiles);
        a.n_split = n_split;
    }
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wgrad_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wgrad_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
    }
    const int grid = (Cout / (rows32 ? 32 : WG_CO)) * (Cin / WG_CI) * a.n_split;
    if (v2x_tune(V2X_TUNE_WGRAD_TR) != 0) {   // the transpose-read form (default); 0: the first form (A/B, bitwise-equality test)
        static v2x_once_per_device tr_once;
        if (v2x_first_use_on_device(tr_once)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wgrad_tr_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_PX * 64 * 2 + WT_X_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wgrad_tr_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_PX * 32 * 2 + WT_X_BYTES);
        }
        if (rows32)
            hipLaunchKernelGGL(conv3x3_wgrad_tr_kernel<32>, dim3(grid), dim3(256), WG_PX * 32 * 2 + WT_X_BYTES, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(conv3x3_wgrad_tr_kernel<64>, dim3(grid), dim3(256), WG_PX * 64 * 2 + WT_X_BYTES, (hipStream_t)stream, a);
        V2X_CHECK_LAUNCH("conv3x3_wgrad_tr_kernel");
        return V2X_OK;
    }
    if (rows32)
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<32>, dim3(grid), dim3(256), WG_SMEM, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<64>, dim3(grid), dim3(256), WG_SMEM, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("conv3x3_wgrad_kernel");
    return V2X_OK;
}
