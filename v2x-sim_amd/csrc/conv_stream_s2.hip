// Streamed-weights kernel for the STRIDE-2 3x3 convolutions of the encoder (conv1_1, conv2_1, conv3_1 of upstream
// Backbone.py::LidarEncoder; code absent from /root/reference, see include/v2x_amd.h).
//
// The gather kernel (conv_igemm.hip) re-fetches every input pixel of a stride-2 layer 2.25 times through L2 -> LDS, one
// im2col slot at a time, and tops out at 290-590 TFLOP/s on these layers (conv1_1 additionally at 2.9 TB/s of a 5+ TB/s
// HBM-bound layer).  Here, as in conv_stream.hip, a workgroup owns an output tile (4 rows x 32 columns) x BCO output
// channels and walks K as (32-channel chunk) x (9 taps):
//   * the (2*4+1) x (2*32+1) input patch of a chunk is brought in ONCE by LDS-DMA and serves all 9 taps;
//   * the patch columns are DE-INTERLEAVED BY PARITY while they are written (the DMA's per-lane source address is free):
//     a patch row is [33 even columns][32 odd columns], so tap kx reads 16 CONSECUTIVE entries of one half (kx = 0: even
//     half, entry c; kx = 1: odd half, entry c; kx = 2: even half, entry c + 1) and the stride-2 fragment reads are as
//     conflict-free as the stride-1 ones (same 16-B slot XOR ((entry >> 1) & 3));
//   * weights stream through the same 4-slot ring with counted s_waitcnt vmcnt(N) and raw s_barrier; the weight layout is
//     the one of conv_stream.hip (packing.pack_conv_stream), so a layer is packed the same way for either stride.
// One patch buffer (37 KiB) + ring (16 / 32 KiB) = 53 / 69 KiB -> three / two workgroups per CU; the patch refill at a
// chunk boundary (everything drains there) is what the other workgroups cover.
#include "common.h"
#ifndef V2X_S2_PSWZ_BUILD
#define V2X_S2_PSWZ_BUILD 1   // patch swizzle (entry >> 1) & 3; 2 = (entry >> 2) & 3 (conv_stream.hip PSWZ): measured no different here (297 vs 299 us)
#endif
#include <cstdlib>

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16q(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

struct S2Args {
    const uint16_t *in;   // [N][H][W][C] bf16
    int C, N, H, W;       // input extent (H, W even)
    const uint16_t *w;    // [n_co_tiles][n_chunks][9][4][BCO][8] bf16 + 64 B zero page
    const float *scale, *shift;
    int relu;
    uint16_t *out;        // [N][H/2][W/2][out_cstride]
    int out_cstride, out_coff, Cout;
    int tiles_x, tiles_y, n_px_tiles, n_co_tiles;
};

constexpr int S2_TH = 4, S2_TW = 32;
constexpr int S2_PH = 2 * S2_TH + 1;             // 9 patch rows
constexpr int S2_NE = S2_TW + 1, S2_NO = S2_TW;  // 33 even + 32 odd patch columns
constexpr int S2_ROW_SLOTS = (S2_NE + S2_NO) * 4;            // 16-B slots per patch row (260)
constexpr int S2_SLOTS = S2_PH * S2_ROW_SLOTS;               // 2340
constexpr int S2_PIECES = (S2_SLOTS + 63) / 64;              // 37 wave instructions of 1 KiB
constexpr int S2_PPW = (S2_PIECES + 3) / 4;                  // pieces per wave (10)
constexpr int S2_PATCH_BYTES = S2_PIECES * 1024;             // 37 KiB (pieces 37..39 of the per-wave loop are never issued)
constexpr int S2_RING = 4;

// Output-tile geometry of the streamed kernel.  4 x 32 (the 128^2 / 64^2 / 32^2 outputs of conv1_1 .. conv3_1) or 8 x 16
// (conv4_1: 16 x 16 outputs -- before, that layer fell back to the gather kernel, 31 % of the MFMA peak).  Both are 128
// output pixels = two 16-pixel fragments per wave, and both patches fit the same 37-KiB buffer (9 x 65 vs 17 x 33 pixels).
template <int TH, int TW>
struct S2Geom {
    static constexpr int PH = 2 * TH + 1;
    static constexpr int NE = TW + 1, NO = TW;
    static constexpr int ROW_SLOTS = (NE + NO) * 4;
    static constexpr int SLOTS = PH * ROW_SLOTS;
    static constexpr int PIECES = (SLOTS + 63) / 64;
    static constexpr int PPW = (PIECES + 3) / 4;
    static_assert(TH * TW == 128 && (TW == 32 || TW == 16), "128 output pixels: 4 x 32 or 8 x 16");
    static_assert(PIECES * 1024 <= S2_PATCH_BYTES, "patch must fit the common buffer");
};

template <int N>
__device__ __forceinline__ void s2_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BCO, int TH = S2_TH, int TW = S2_TW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void conv3x3_s2_stream_kernel(const S2Args a) {
    using G = S2Geom<TH, TW>;
    constexpr int TCO = BCO / 16;
    constexpr int W_PIECES = BCO / 16;           // 1 KiB pieces per weight slice
    constexpr int NW = W_PIECES / 4;             // weight DMAs per wave and step (1 or 2)
    constexpr int SLICE_BYTES = BCO * 64;
    static_assert(BCO == 64 || BCO == 128, "channel tiles of 64 or 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;   // v2x_relu_bf16x2_floor: identity when the layer has no ReLU
    char *s_ring = smem;
    char *s_patch = smem + S2_RING * SLICE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int co_tile = bid % a.n_co_tiles;
    const int px_tile = bid / a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int n = px_tile / txy;
    const int trem = px_tile - n * txy;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;       // output coordinates

    const int nchunks = a.C >> 5;
    const int S = nchunks * 9;
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);
    // scale/shift rows of this channel tile in LDS (read in the epilogue, many barriers later): a global load between the
    // stores of two channel tiles can only be waited for together with those stores (in-order vmcnt)
    float *s_ss = reinterpret_cast<float *>(smem + S2_RING * SLICE_BYTES + S2_PATCH_BYTES);   // [BCO scale | BCO shift]
    for (int i = tid; i < BCO; i += 256) {
        s_ss[i] = a.scale[co_tile * BCO + i];       // both arrays hold n_co_tiles * BCO entries
        s_ss[BCO + i] = a.shift[co_tile * BCO + i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // before the counted-DMA regime starts

    // per-lane DMA descriptors of this wave's patch pieces (piece = wave + 4t): (input pixel index << 5) | (logical 16-B
    // slot * 8 elements), -1 = zero page (outside the image, or the padding slots behind the patch)
    int pd[G::PPW];
#pragma unroll
    for (int t = 0; t < G::PPW; ++t) {
        const int L = (wave + 4 * t) * 64 + lane;
        const int r = L / G::ROW_SLOTS, q = L - r * G::ROW_SLOTS;
        const int ent = q >> 2, phys = q & 3;
        const bool odd = ent >= G::NE;
        const int idx = odd ? ent - G::NE : ent;
        const int pc = 2 * idx + (odd ? 1 : 0);               // patch column
        const int y = 2 * y0 - 1 + r, x = 2 * x0 - 1 + pc;   // input pixel
        const bool ok = L < G::SLOTS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        pd[t] = ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ ((idx >> V2X_S2_PSWZ_BUILD) & 3)) << 3)) : -1;
    }
    // fragment f of a wave: 4 x 32 tile -> output row `wave`, columns f*16 + fj;  8 x 16 tile -> output row 2*wave + f,
    // columns fj.  Column offsets: tap kx reads entry (col + (kx == 2)) of the even (kx = 0, 2) or odd (kx = 1) half
    int ct[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int idx = (TW == 32 ? f * 16 : 0) + fj + (kx == 2 ? 1 : 0);
            ct[f][kx] = (((kx == 1 ? G::NE : 0) + idx) * 4 + (fq ^ ((idx >> V2X_S2_PSWZ_BUILD) & 3))) * 16;
        }
    const int frow0 = (TW == 32) ? wave : 2 * wave, frow1 = (TW == 32) ? wave : 2 * wave + 1;   // output row of fragment 0 / 1

    const uint16_t *wsrc = wbase + lane * 8 + wave * 512;
    auto issue_weights = [&](int s) {   // NW DMAs
        char *dst = s_ring + (s & (S2_RING - 1)) * SLICE_BYTES;
        const uint16_t *src = wsrc + (size_t)s * (BCO * 32);
        glds16q(src, dst + wave * 1024);
        if (NW == 2) glds16q(src + 4 * 512, dst + (wave + 4) * 1024);
    };
    auto issue_patch = [&](int kc) {    // this wave's pieces of chunk kc (pieces >= S2_PIECES do not exist)
#pragma unroll
        for (int t = 0; t < G::PPW; ++t) {
            if (wave + 4 * t >= G::PIECES) break;            // wave-uniform
            const int d = pd[t];
            const unsigned off = (unsigned)(d >> 5) * (unsigned)a.C + (unsigned)(kc * 32 + (d & 31));
            glds16q(d >= 0 ? (const void *)(a.in + off) : zero_page, s_patch + (wave + 4 * t) * 1024);
        }
    };

    f32x4_t acc[TCO][2];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // prologue: weight slices of steps 0..2 (S >= 9), then the patch of chunk 0
    issue_weights(0);
    issue_weights(1);
    issue_weights(2);

    int s = 0;
    for (int kc = 0; kc < nchunks; ++kc) {
        // ---- chunk boundary: everyone has left the previous chunk's patch (and ring slot s-1) -> refill, drain, meet
        if (kc > 0) __builtin_amdgcn_s_barrier();
        issue_patch(kc);
        s2_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const int rowoff0 = (2 * frow0 + ky) * (G::ROW_SLOTS * 16), rowoff1 = (2 * frow1 + ky) * (G::ROW_SLOTS * 16);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx, ++s) {
                const int tap = ky * 3 + kx;
                if (tap > 0) {
                    // this step's slice has landed (groups s+1, s+2 may be in flight); everyone is done with slot s-1
                    if (s + 2 >= S) s2_wait_vmcnt<0>();
                    else s2_wait_vmcnt<2 * NW>();
                    __builtin_amdgcn_s_barrier();
                }
                if (s + 3 < S) issue_weights(s + 3);
                const char *ws = s_ring + (s & (S2_RING - 1)) * SLICE_BYTES;
                bf16x8_t fa[TCO], fb[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) fb[f] = *reinterpret_cast<const bf16x8_t *>(s_patch + (f ? rowoff1 : rowoff0) + ct[f][kx]);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8_t *>(ws + (fq * BCO + i * 16 + fj) * 16);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[f], acc[i][f], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TCO + 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TCO * 2, 0);
            }
        }
    }

    // ---- epilogue: BN / ReLU, bf16, 8-byte NHWC stores
    const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
        const int co = co_tile * BCO + i * 16 + fq * 4;
        if (co >= a.Cout) continue;
        const float4 sc = *reinterpret_cast<const float4 *>(s_ss + i * 16 + fq * 4);
        const float4 sf = *reinterpret_cast<const float4 *>(s_ss + BCO + i * 16 + fq * 4);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
            float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            o.x = v2x_relu_bf16x2_floor(o.x, relu_floor);   // ReLU on the packed bf16 pairs; the floor is the identity when the layer has none
            o.y = v2x_relu_bf16x2_floor(o.y, relu_floor);
            const size_t pix = (size_t)(n * Ho + y0 + (f ? frow1 : frow0)) * Wo + x0 + (TW == 32 ? f * 16 : 0) + fj;
            *reinterpret_cast<uint2 *>(a.out + pix * a.out_cstride + a.out_coff + co) = o;
        }
    }
}

// ---- resident-weights form for ONE-chunk layers (conv1_1: 32 -> 64) ------------------------------------
// With 32 input channels the whole K walk is 9 steps of 8 MFMAs: in the streamed form above every one of them sits behind
// its own barrier + counted wait, and the three weight slices of the prologue are re-fetched for every 128-pixel tile.
// Here the 9 slices (36 KiB) are loaded ONCE per workgroup and stay in LDS; the workgroup is persistent over tiles and a
// tile is: refill the patch (37 KiB, single buffer), one wait + barrier, 72 MFMAs without any synchronisation, 8 stores.
// 73 KiB -> two workgroups per CU cover each other's refill; the layer is HBM-bound (6.3 MB per map).
template <int BCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_s2_resident_kernel(const S2Args a) {
    constexpr int TCO = BCO / 16;
    constexpr int SLICE_BYTES = BCO * 64;
    constexpr int W_BYTES = 9 * SLICE_BYTES;
    static_assert(BCO == 64, "one 64-row channel tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;   // v2x_relu_bf16x2_floor: identity when the layer has no ReLU
    char *s_w = smem;
    char *s_patch = smem + W_BYTES;
    // scale/shift in LDS: a global load between the stores of two channel tiles can only be waited for (in-order vmcnt)
    // together with the stores before it -- four store drains per tile
    float *s_ss = reinterpret_cast<float *>(smem + W_BYTES + S2_PATCH_BYTES);   // [64 scale | 64 shift]
    if (threadIdx.x < 64) {
        s_ss[threadIdx.x] = a.scale[threadIdx.x];
        s_ss[64 + threadIdx.x] = a.shift[threadIdx.x];
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    const void *zero_page = a.w + (size_t)a.n_co_tiles * 9 * (BCO * 32);

    // weights of the (single) channel tile: one linear LDS-DMA copy
    for (int off = wave * 1024; off < W_BYTES; off += 4096)
        glds16q(reinterpret_cast<const char *>(a.w) + off + lane * 16, s_w + off);

    int ct[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int idx = f * 16 + fj + (kx == 2 ? 1 : 0);
            ct[f][kx] = (((kx == 1 ? S2_NE : 0) + idx) * 4 + (fq ^ ((idx >> V2X_S2_PSWZ_BUILD) & 3))) * 16;
        }
    // tile-independent part of the patch descriptors: (patch row << 20) | (patch column << 8) | (swizzled slot << 3), -1 = padding
    int pdt[S2_PPW];
#pragma unroll
    for (int t = 0; t < S2_PPW; ++t) {
        const int L = (wave + 4 * t) * 64 + lane;
        const int r = L / S2_ROW_SLOTS, q = L - r * S2_ROW_SLOTS;
        const int ent = q >> 2, phys = q & 3;
        const bool odd = ent >= S2_NE;
        const int idx = odd ? ent - S2_NE : ent;
        const int pc = 2 * idx + (odd ? 1 : 0);
        pdt[t] = L < S2_SLOTS ? ((r << 20) | (pc << 8) | ((phys ^ ((idx >> V2X_S2_PSWZ_BUILD) & 3)) << 3)) : -1;
    }
    const int txy = a.tiles_x * a.tiles_y;
    const int Ho = a.H >> 1, Wo = a.W >> 1;

    bool first = true;
    for (int tile = blockIdx.x; tile < a.n_px_tiles; tile += gridDim.x) {
        const int n = tile / txy;
        const int trem = tile - n * txy;
        const int ty = trem / a.tiles_x;
        const int y0 = ty * S2_TH, x0 = (trem - ty * a.tiles_x) * S2_TW;
        // everyone has left the patch of the previous tile (its MFMAs consumed their fragments)
        if (!first) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < S2_PPW; ++t) {
            if (wave + 4 * t >= S2_PIECES) break;            // wave-uniform
            const int d = pdt[t];
            const int y = 2 * y0 - 1 + (d >> 20), x = 2 * x0 - 1 + ((d >> 8) & 0xfff);
            const bool ok = d >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ((unsigned)(n * a.H + y) * (unsigned)a.W + (unsigned)x) * (unsigned)a.C + (unsigned)(d & 0xff);
            glds16q(ok ? (const void *)(a.in + off) : zero_page, s_patch + (wave + 4 * t) * 1024);
        }
        s2_wait_vmcnt<0>();     // the patch (and, the first time, the weights; later also the previous tile's stores)
        __builtin_amdgcn_s_barrier();
        first = false;

        f32x4_t acc[TCO][2];
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int rowoff = (2 * wave + ky) * (S2_ROW_SLOTS * 16);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const char *ws = s_w + (ky * 3 + kx) * SLICE_BYTES;
                bf16x8_t fa[TCO], fb[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) fb[f] = *reinterpret_cast<const bf16x8_t *>(s_patch + rowoff + ct[f][kx]);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8_t *>(ws + (fq * BCO + i * 16 + fj) * 16);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[f], acc[i][f], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = i * 16 + fq * 4;
            const float4 sc = *reinterpret_cast<const float4 *>(s_ss + co), sf = *reinterpret_cast<const float4 *>(s_ss + 64 + co);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                o.x = v2x_relu_bf16x2_floor(o.x, relu_floor);   // ReLU on the packed bf16 pairs; the floor is the identity when the layer has none
                o.y = v2x_relu_bf16x2_floor(o.y, relu_floor);
                const size_t pix = (size_t)(n * Ho + y0 + wave) * Wo + x0 + f * 16 + fj;
                *reinterpret_cast<uint2 *>(a.out + pix * a.out_cstride + a.out_coff + co) = o;
            }
        }
    }
}

static int launch_s2_resident(const S2Args &a, hipStream_t s) {
    constexpr int smem = 9 * 64 * 64 + S2_PATCH_BYTES + 512;   // 36 + 37 KiB + scale/shift
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_s2_resident_kernel<64>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    int grid = 512;                                       // two workgroups per CU
    if (grid > a.n_px_tiles) grid = a.n_px_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_s2_resident_kernel");
    return V2X_OK;
}

template <int BCO, int TH = S2_TH, int TW = S2_TW>
static int launch_s2(const S2Args &a, hipStream_t s) {
    constexpr int smem = S2_RING * BCO * 64 + S2_PATCH_BYTES + 2 * BCO * 4;
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_s2_stream_kernel<BCO, TH, TW>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_s2_stream_kernel");
    return V2X_OK;
}

// Returns V2X_OK if handled, 1 if the shape is not covered (caller reports).
int v2x_conv_stream_s2_dispatch(const v2x_conv_desc *d, hipStream_t s) {
    if (d->C1 != 0 || d->up0 != 0 || d->Cout2 != 0 || d->epilogue != V2X_EPI_BF16) return 1;
    const bool t32 = d->H % (2 * S2_TH) == 0 && d->W % (2 * S2_TW) == 0;     // 4 x 32 output tiles
    const bool t16 = !t32 && d->H % 16 == 0 && d->W % 32 == 0;                // 8 x 16 output tiles (16 x 16 outputs: conv4_1)
    if (!t32 && !t16) return 1;
    const int rows = (d->Cout % 128 == 0) ? 128 : ((d->Cout % 64 == 0) ? 64 : 0);
    if (rows == 0 || d->w_rows != d->Cout) return 1;
    S2Args a;
    a.in = d->in0;
    a.C = d->C0;
    a.N = d->N;
    a.H = d->H;
    a.W = d->W;
    a.w = d->weight;
    a.scale = d->scale;
    a.shift = d->shift;
    a.relu = d->relu;
    a.out = reinterpret_cast<uint16_t *>(d->out);
    a.out_cstride = d->out_cstride;
    a.out_coff = d->out_coff;
    a.Cout = d->Cout;
    a.tiles_x = (d->W / 2) / (t32 ? S2_TW : 16);
    a.tiles_y = (d->H / 2) / (t32 ? S2_TH : 8);
    a.n_px_tiles = d->N * a.tiles_x * a.tiles_y;
    a.n_co_tiles = d->Cout / rows;
    if (t16) return rows == 128 ? launch_s2<128, 8, 16>(a, s) : launch_s2<64, 8, 16>(a, s);
    if (rows == 64 && a.n_co_tiles == 1 && d->C0 == 32) {   // one chunk, one channel tile: resident weights (conv1_1)
        if (v2x_tune(V2X_TUNE_S2_RESIDENT) != 0) return launch_s2_resident(a, s);
    }
    return rows == 128 ? launch_s2<128>(a, s) : launch_s2<64>(a, s);
}
