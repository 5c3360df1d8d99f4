// Two back-to-back HBM-bound 3x3 layers as ONE kernel: conv_pre_1 (13 -> 32, input = the voxelizer's bit grid) followed by
// conv_pre_2 (32 -> 32) of upstream coperception/models/det/backbone/Backbone.py::LidarEncoder (code absent from
// /root/reference, see include/v2x_amd.h).
//
// Why: both layers run at the HBM ceiling in conv_halo.hip (0.51 + 0.55 ms per 320 maps: the first WRITES 1.34 GB, the
// second reads it back and writes another 1.34 GB).  Here the intermediate 32-channel map never leaves the CU:
//   * per 8x32 output tile the workgroup expands the (8+4)x(32+4) window of occupancy words into LDS (16 channels as two
//     8-channel planes; channels 13..15 are zero, the padded layer's channels 16..31 never enter an MFMA: layer A packs
//     TWO taps x 16 channels into each K = 32 step, 5 steps instead of 9);
//   * layer A is evaluated on the (8+2)x(32+2) halo region the second layer needs: its 340 pixels are walked as 22
//     linear 16-pixel MFMA fragments (a fragment may wrap to the next region row: every lane addresses its own pixel),
//     scale/shift/ReLU applied, rounded to bf16 exactly as the stand-alone layer stores it, ZEROED where the pixel lies
//     outside the image (= the second layer's zero padding), and written into LDS in the pixel-major swizzled patch
//     layout of conv_halo.hip;
//   * layer B then runs on that patch like the stand-alone kernel (group walk over kx, B fragments shared by the three
//     ky taps) and stores bf16 NHWC.
// Both weight tensors stay resident in LDS (2 x 18 KiB).  76.4 KiB of LDS per workgroup -> 2 workgroups per CU; the next
// tile's occupancy words are prefetched into registers under the current tile's MFMAs, barriers are raw s_barrier with
// lgkmcnt waits only, so the output stores of a tile drain under the next tile.
// Traffic per tile: 1.7 KB read + 16 KB written (the stand-alone pair: 1.7 + 16 | 22 + 16).  The recompute of layer A
// on the halo ring costs 22/16 of its MFMAs.  Results are bit-identical to running the two layers separately
// (tests/test_gpu_bits_input.py::test_conv_pair_equals_two_layers_bitwise).
// Measured and dropped (same-box A/B, 0.69 ms for this form): scale/shift read from LDS instead of held in registers,
// operand double-buffering in registers, an interior-tile path without the image-border predicate, 8 waves per workgroup
// -- together 0.79 ms.  Switching phases off one at a time showed the phases of a tile adding up almost linearly (no
// stores -0.09, no layer-A MFMAs -0.07, no layer-B MFMAs -0.13, no barriers -0.08, no window fetch/expansion -0.06, no
// layer-A epilogue -0.15 ms): the two waves a SIMD holds overlap little; the MFMA work alone would be 0.31 ms.
#include "common.h"
#include <cstdlib>

typedef const __attribute__((address_space(1))) void *gptr_p_t;
typedef __attribute__((address_space(3))) void *lptr_p_t;

struct PairArgs {
    const uint32_t *bits;   // [N][H][W] occupancy words
    int zbits;
    int N, H, W;
    const uint16_t *wA, *wB;            // k-slot-major [36][32][8] bf16 each (conv_halo.hip layout, C = 32 -> 32)
    const float *scA, *shA, *scB, *shB; // [32]
    int reluA, reluB;
    uint16_t *out;                      // NHWC [N][H][W][out_cstride] (+out_coff)
    int out_cstride, out_coff;
    int tiles_x, tiles_y, n_tiles;
    int xcd_walk;                       // 1: XCD-contiguous tile walk (common.h; tuning switch HALO_XCD)
    int x4;                             // 16-byte output stores (common.h: v2x_store_pair_x4)
};

namespace pair {
constexpr int TH = 8, TW = 32;          // output tile
constexpr int MH = TH + 2, MW = TW + 2; // layer-A region (= layer B's input patch)
constexpr int IH = TH + 4, IW = TW + 4; // input window
constexpr int NMID = MH * MW;           // 340 pixels
constexpr int NFRAG = (NMID + 15) / 16; // 22 linear fragments
constexpr int FPW = (NFRAG + 3) / 4;    // fragments per wave (6; waves 2,3 use 5)
constexpr int W_BYTES = 36 * 32 * 16;   // 18 432
constexpr int IN_PLANE = IH * IW * 16;  // one 8-channel plane of the input window: [pixel][8] bf16
constexpr int IN_BYTES = 2 * IN_PLANE; // 13 824: channels 0..7 | 8..15, planar so that 16 consecutive pixels are 256 contiguous bytes
constexpr int MID_BYTES = NMID * 64;    // 21 760
constexpr int LUT_BYTES = 256 * 16;     // byte of occupancy bits -> 8 bf16 {0,1}: the expansion is one ds_read_b128 instead of ~60 VALU ops
#ifndef V2X_PAIR_LDS_PAD_BUILD
#define V2X_PAIR_LDS_PAD_BUILD 0   // occupancy experiment (round 6, profiles/r06_pair_occupancy.txt): extra dynamic LDS per workgroup; 10240 -> 86 KiB = ONE workgroup per CU instead of two
#endif
constexpr int SMEM = 2 * W_BYTES + IN_BYTES + MID_BYTES + LUT_BYTES + V2X_PAIR_LDS_PAD_BUILD;
#ifndef V2X_HALO_PSWZ_BUILD
#define V2X_HALO_PSWZ_BUILD 1
#endif
// 1: (x >> 1) & 3 (round 1);  2: (x >> 2) & 3, whose fragment reads time 30 % faster in isolation and change nothing here (conv_halo.hip)
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> V2X_HALO_PSWZ_BUILD) & 3); }
// ReLU on two packed bf16: as 16-bit integers a negative bf16 (sign bit) is a negative short, so max(x, 0) per half is
// ONE v_pk_max_i16 for two values (fmaxf on the fp32 values costs two instructions EACH: it canonicalises first).
// relu(round(v)) == round(relu(v)): rounding is monotone and -0 maps to +0 either way.
typedef short v2x_s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t x) {
    const v2x_s16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(v2x_s16x2_t, x), (v2x_s16x2_t){0, 0});
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
}  // namespace pair

// X4: 16-byte output stores (common.h: v2x_store_pair_x4).  A TEMPLATE parameter, not a run-time flag: the compiler counts this kernel's vector-memory
// operations itself (the next tile's occupancy words are requested before this tile's stores: vmcnt(4 / 5) here, vmcnt(8 / 9) with 8-byte stores), and
// with both store paths behind a run-time branch it can no longer count -- it drained the stores in every tile (tests/test_build_invariants_cpu.py).
template <bool X4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_pair_bits_kernel(const PairArgs a) {
    using namespace pair;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_wA = smem, *s_wB = smem + W_BYTES, *s_in = smem + 2 * W_BYTES, *s_mid = smem + 2 * W_BYTES + IN_BYTES;
    char *s_lut = s_mid + MID_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;

    // weights: linear LDS-DMA copies, resident for the whole kernel
    for (int off = wave * 1024; off < W_BYTES; off += 4096) {
        __builtin_amdgcn_global_load_lds((gptr_p_t)(reinterpret_cast<const char *>(a.wA) + off + lane * 16), (lptr_p_t)(s_wA + off), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_p_t)(reinterpret_cast<const char *>(a.wB) + off + lane * 16), (lptr_p_t)(s_wB + off), 16, 0, 0);
    }

    // tile-invariant per-lane geometry of the layer-A fragments of this wave
    int in_off[FPW];      // byte offset of the lane's pixel (tap 0,0) in its channel plane of s_in
    int mid_off[FPW][2];  // byte offset of the lane's 8-B result piece in s_mid, per channel tile
    int rc[FPW];          // region row | column << 8 | valid << 16
#pragma unroll
    for (int t = 0; t < FPW; ++t) {
        const int p = (wave + 4 * t) * 16 + fj;
        const int pc = p < NMID ? p : NMID - 1;
        const int r = pc / MW, c = pc - r * MW;
        in_off[t] = (fq & 1) * IN_PLANE + (r * IW + c) * 16;
        rc[t] = r | (c << 8) | ((p < NMID ? 1 : 0) << 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) mid_off[t][i] = ((r * MW + c) * 4 + swz4(i * 2 + (fq >> 1), c)) * 16 + (fq & 1) * 8;
    }
    // layer A walks K in 5 steps of TWO taps: k-groups 0,1 = the 16 channels of tap 2s, k-groups 2,3 = those of tap 2s+1
    // (channels 16..31 of the padded layer never enter an MFMA).  Per lane: window shift and weight slot of its tap.
    int tap_off[5], wA_off[5][2];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = 2 * st + (fq >> 1);
        const int tc = tap < 9 ? tap : 8;
        tap_off[st] = ((tc / 3) * IW + (tc % 3)) * 16;
        // the 10th tap does not exist: its lanes read entry 0 of the expansion table (16 zero bytes) as their weights
#pragma unroll
        for (int i = 0; i < 2; ++i)
            wA_off[st][i] = tap < 9 ? ((tap * 4 + (fq & 1)) * 32 + i * 16 + fj) * 16 : 2 * W_BYTES + IN_BYTES + MID_BYTES - 0;
    }
    const uint32_t zmask = (a.zbits >= 16) ? 0xffffu : ((1u << a.zbits) - 1u);   // 16 channels exist in LDS (zbits <= 16)
    const int txy = a.tiles_x * a.tiles_y;

    // The loads are UNCONDITIONAL (clamped address) and the in-image predicate is applied when the words are expanded:
    // `w = ok ? load : 0` makes the compiler zero the register first, and a VALU write to a register with a possibly
    // pending load costs an s_waitcnt vmcnt(0) at the top of every tile -- which, vmcnt being in-order, waits for the
    // previous tile's output stores.
    auto fetch_words = [&](int tile, uint32_t (&w)[2], uint32_t &okmask) {
        const int n = tile / txy;
        const int r = tile - n * txy;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        uint32_t m = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int p = min(tid + 256 * s, IH * IW - 1);
            const int pr = p / IW, pcx = p - pr * IW;
            const int y = ty * TH - 2 + pr, x = tx * TW - 2 + pcx;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
            w[s] = a.bits[(size_t)(n * a.H + yc) * a.W + xc];
            m |= (ok ? 1u : 0u) << s;
        }
        okmask = m;
    };

    float4 scA[2], shA[2], scB[2], shB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        scA[i] = *reinterpret_cast<const float4 *>(a.scA + i * 16 + fq * 4);
        shA[i] = *reinterpret_cast<const float4 *>(a.shA + i * 16 + fq * 4);
        scB[i] = *reinterpret_cast<const float4 *>(a.scB + i * 16 + fq * 4);
        shB[i] = *reinterpret_cast<const float4 *>(a.shB + i * 16 + fq * 4);
    }

    {   // expansion table: entry b = the 8 channels of occupancy byte b as bf16 {0, 1}
        const uint32_t b = tid, one = 0x3f80u;
        uint4 v;
        v.x = ((b & 1u) ? one : 0u) | ((b & 2u) ? (one << 16) : 0u);
        v.y = ((b & 4u) ? one : 0u) | ((b & 8u) ? (one << 16) : 0u);
        v.z = ((b & 16u) ? one : 0u) | ((b & 32u) ? (one << 16) : 0u);
        v.w = ((b & 64u) ? one : 0u) | ((b & 128u) ? (one << 16) : 0u);
        *reinterpret_cast<uint4 *>(s_lut + tid * 16) = v;
    }
    __syncthreads();
    auto expand_words = [&](const uint32_t (&w)[2], uint32_t okmask) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int p = tid + 256 * s;
            if (p < IH * IW) {
                const uint32_t word = ((okmask >> s) & 1u) ? (w[s] & zmask) : 0u;
#pragma unroll
                for (int slot = 0; slot < 2; ++slot)
                    *reinterpret_cast<uint4 *>(s_in + slot * IN_PLANE + p * 16) =
                        *reinterpret_cast<const uint4 *>(s_lut + ((word >> (8 * slot)) & 0xffu) * 16);
            }
        }
    };

    // Vector-memory waits are in-order (vmcnt): a wait for ANY load inside the tile loop also waits for the previous
    // tile's output stores (~1-2 us each).  So (1) everything loaded once is consumed here, before the loop -- otherwise
    // the compiler re-waits for it with vmcnt(0) in every iteration -- and (2) the occupancy words of the next tile are
    // requested BEFORE this tile's stores and expanded AFTER them, which the compiler resolves to vmcnt(8), not 0.
    const v2x_tile_walk walk = v2x_xcd_tile_walk(a.n_tiles, a.xcd_walk);
    int tile = walk.first;
    uint32_t words[2] = {0u, 0u}, okmask = 0;
    if (tile < walk.end) {
        fetch_words(tile, words, okmask);
        expand_words(words, okmask);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        asm volatile("" ::"v"(scA[i].x), "v"(scA[i].w), "v"(shA[i].x), "v"(shA[i].w), "v"(scB[i].x), "v"(scB[i].w), "v"(shB[i].x), "v"(shB[i].w));
    __syncthreads();   // input window of the first tile written; the weight DMA has landed

    for (; tile < walk.end; tile += walk.step) {
        const int next = tile + walk.step;
        if (next < walk.end) fetch_words(next, words, okmask);   // lands under this tile's MFMAs

        const int n = tile / txy;
        const int rr_ = tile - n * txy;
        const int ty = rr_ / a.tiles_x, tx = rr_ - ty * a.tiles_x;

        // ---- (c) layer A on the 10 x 34 region, 22 linear fragments ---------------------------------------------
        {
            f32x4_t acc[FPW][2];
#pragma unroll
            for (int t = 0; t < FPW; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // waves 2 and 3 own 5 fragments: their 6th (k = 22, 23) lies behind the region, every lane is clamped to the last
            // pixel and its result dropped (rc valid bit) -- cheaper than a conditionally defined operand array
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                bf16x8_t A[2], B[FPW];
#pragma unroll
                for (int i = 0; i < 2; ++i) A[i] = *reinterpret_cast<const bf16x8_t *>(smem + wA_off[st][i]);
#pragma unroll
                for (int t = 0; t < FPW; ++t) {
                    B[t] = *reinterpret_cast<const bf16x8_t *>(s_in + in_off[t] + tap_off[st]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FPW; ++t) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[t], acc[t][i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue A: bf16 as the stand-alone layer stores it; zero outside the image (layer B's padding)
#pragma unroll
            for (int t = 0; t < FPW; ++t) {
                const int r = rc[t] & 0xff, c = (rc[t] >> 8) & 0xff;
                const int y = ty * TH - 1 + r, x = tx * TW - 1 + c;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                if (!(rc[t] >> 16)) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v0 = acc[t][i][0] * scA[i].x + shA[i].x, v1 = acc[t][i][1] * scA[i].y + shA[i].y;
                    float v2 = acc[t][i][2] * scA[i].z + shA[i].z, v3 = acc[t][i][3] * scA[i].w + shA[i].w;
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    if (a.reluA) {
                        o.x = relu_bf16x2(o.x);
                        o.y = relu_bf16x2(o.y);
                    }
                    o.x = inside ? o.x : 0u;
                    o.y = inside ? o.y : 0u;
                    *reinterpret_cast<uint2 *>(s_mid + mid_off[t][i]) = o;
                }
            }
        }
        lds_barrier();

        // ---- (e) layer B on the region, as conv_halo.hip's 32 -> 32 form ----------------------------------------
        {
            f32x4_t acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8_t A[3][2], B[8];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_wB + (((ky * 3 + kx) * 4 + fq) * 32 + i * 16 + fj) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int pr = 2 * wave + q, pc = ch * 16 + fj + kx;
                        B[q * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(s_mid + ((pr * MW + pc) * 4 + swz4(fq, pc)) * 16);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], B[((f >> 1) + ky) * 2 + (f & 1)], acc[i][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (X4) {   // 16-byte stores: the two channel tiles exchanged between the k-slot quarters (common.h: v2x_store_pair_x4)
                const uint32_t floorB = a.reluB ? 0u : 0x80008000u;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                    uint32_t ox[2], oy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[h][f][0] * scB[h].x + shB[h].x, acc[h][f][1] * scB[h].y + shB[h].y), floorB);
                        oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[h][f][2] * scB[h].z + shB[h].z, acc[h][f][3] * scB[h].w + shB[h].w), floorB);
                    }
                    v2x_store_pair_x4(a.out + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + fq * 4, fq, ox[0], oy[0], ox[1], oy[1]);
                }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = i * 16 + fq * 4;
                const float4 sc = scB[i], sf = shB[i];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                    float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                    float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    if (a.reluB) {
                        o.x = relu_bf16x2(o.x);
                        o.y = relu_bf16x2(o.y);
                    }
                    *reinterpret_cast<uint2 *>(a.out + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + co) = o;
                }
            }
            }
        }
        // the next tile's input window: every wave is past the barrier above, i.e. done reading s_in for this tile
        if (next < a.n_tiles) expand_words(words, okmask);
        lds_barrier();   // window visible; and every wave is done reading s_mid before the next tile's layer A rewrites it
    }
}

int v2x_num_cus();   // conv_stream.hip

// first: 3x3 s1 p1, bit-grid input (in_format 1), 32 (padded) -> 32, w_layout 1, bf16 epilogue; second: 32 -> 32 likewise.
// The intermediate map is never stored: first->out is ignored.
int v2x_conv_tail_dispatch(const v2x_conv_desc *first, const v2x_conv_desc *second, hipStream_t stream);   // conv_tail.hip

extern "C" int v2x_conv2d_pair(const v2x_conv_desc *first, const v2x_conv_desc *second, v2x_stream_t stream) {
    V2X_REQUIRE(first && second, "v2x_conv2d_pair: null descriptor");
    if (second->Cout2 > 0 && first->in_format == 0) return v2x_conv_tail_dispatch(first, second, (hipStream_t)stream);   // conv8_2 o detection heads
    const v2x_conv_desc *ds[2] = {first, second};
    for (int k = 0; k < 2; ++k) {
        const v2x_conv_desc *d = ds[k];
        V2X_REQUIRE(d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->w_layout == 1 && d->C0 == 32 && d->C1 == 0 &&
                        d->Cout == 32 && d->Cout2 == 0 && d->epilogue == V2X_EPI_BF16 && d->up0 == 0 && d->split == 0,
                    "v2x_conv2d_pair: layer %d must be a halo-packed 3x3 stride-1 32 -> 32 bf16 layer", k);
        V2X_REQUIRE(d->weight && d->scale && d->shift, "v2x_conv2d_pair: layer %d has null parameters", k);
    }
    V2X_REQUIRE(first->in_format == 1 && first->in0 && first->in_zbits >= 1 && first->in_zbits <= 16,
                "v2x_conv2d_pair: the first layer reads the voxelizer's bit grid (1 <= zbits <= 16)");
    V2X_REQUIRE(second->out && second->out_cstride >= 32 + second->out_coff && second->out_cstride % 4 == 0 && second->out_coff % 4 == 0,
                "v2x_conv2d_pair: bad output view");
    V2X_REQUIRE(first->N == second->N && first->H == second->H && first->W == second->W, "v2x_conv2d_pair: extents differ");
    V2X_REQUIRE(first->N >= 0 && first->H > 0 && first->W > 0 && first->H % pair::TH == 0 && first->W % pair::TW == 0,
                "v2x_conv2d_pair: H %% 8 == 0 and W %% 32 == 0 required (H=%d W=%d)", first->H, first->W);
    V2X_REQUIRE((long long)first->N * first->H * first->W < (1ll << 31), "v2x_conv2d_pair: N*H*W must fit 31 bits");
    if (first->N == 0) return V2X_OK;
    PairArgs a;
    a.xcd_walk = v2x_tune(V2X_TUNE_HALO_XCD);
    a.x4 = v2x_x4_ok(second->out, second->out_cstride, second->out_coff, 32);
    a.bits = reinterpret_cast<const uint32_t *>(first->in0);
    a.zbits = first->in_zbits;
    a.N = first->N;
    a.H = first->H;
    a.W = first->W;
    a.wA = first->weight;
    a.wB = second->weight;
    a.scA = first->scale;
    a.shA = first->shift;
    a.scB = second->scale;
    a.shB = second->shift;
    a.reluA = first->relu;
    a.reluB = second->relu;
    a.out = reinterpret_cast<uint16_t *>(second->out);
    a.out_cstride = second->out_cstride;
    a.out_coff = second->out_coff;
    a.tiles_x = a.W / pair::TW;
    a.tiles_y = a.H / pair::TH;
    a.n_tiles = a.N * a.tiles_x * a.tiles_y;
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_pair_bits_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, pair::SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_pair_bits_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, pair::SMEM);
    }
    int grid = v2x_num_cus() * 2;
    if (grid > a.n_tiles) grid = a.n_tiles;
    if (a.x4) hipLaunchKernelGGL(conv3x3_pair_bits_kernel<true>, dim3(grid), dim3(256), pair::SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv3x3_pair_bits_kernel<false>, dim3(grid), dim3(256), pair::SMEM, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("conv3x3_pair_bits_kernel");
    return V2X_OK;
}
