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
#ifndef V2X_S2G_DBG_BUILD
#define V2X_S2G_DBG_BUILD 0   // phase probe of conv3x3_s2g_kernel (tools/s2g_phase_probe.sh; results are garbage): 1 no weight DMAs, 2 no patch DMAs, 4 no pixel-fragment reads, 8 no MFMAs, 16 no weight-fragment reads, 32 no counted waits
#endif
#include <cstdlib>

int v2x_num_cus();   // conv_stream.hip
#if (V2X_S2G_DBG_BUILD & 64)
// timestamp build (tools/s2g_timeline.sh): lane 0 of wave 0 of each group of workgroup 0 stamps the shader clock at 6 points of each of its
// first S2G_T_STEPS steps: 0 top of the load phase, 1 DMAs issued, 2 fragments read + waits done, 3 first barrier passed, 4 MFMAs done, 5 end-of-phase wait done
constexpr int S2G_T_STEPS = 96;
__device__ unsigned v2x_s2g_timeline[2 * S2G_T_STEPS * 8];
extern "C" int v2x_debug_s2g_timeline(unsigned *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(v2x_s2g_timeline), sizeof(unsigned) * 2 * S2G_T_STEPS * 8);
}
#endif

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
    int xcd_walk;         // resident form: 1 = XCD-contiguous tile walk (tuning switch HALO_XCD)
    int x4;               // 16-byte output stores (common.h: v2x_store_pair_x4); set by the dispatch when the output view allows
    // split-K (latency mode, conv3x3_s2_stream_kernel<..., SPLITK = true>): blockIdx.y walks `ksplit` contiguous ranges of the 32-channel chunks and
    // stores its raw fp32 sums to ws[split][pixel][w_rows]; splitk_reduce_kernel (conv_stream.hip) adds them in split order and applies the epilogue
    int ksplit;
    float *ws;
    int w_rows;
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

template <int BCO, int TH = S2_TH, int TW = S2_TW, bool SPLITK = false>
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
    // SPLITK: this workgroup walks the chunks [kc_lo, kc_hi) of the layer's K (blockIdx.y = its range)
    const int kper = SPLITK ? (nchunks + a.ksplit - 1) / a.ksplit : nchunks;
    const int kc_lo = SPLITK ? (int)blockIdx.y * kper : 0;
    const int kc_hi = SPLITK ? (kc_lo + kper < nchunks ? kc_lo + kper : nchunks) : nchunks;
    const int S = (kc_hi - kc_lo) * 9;
    const uint16_t *wbase = a.w + ((size_t)co_tile * nchunks + kc_lo) * 9 * (BCO * 32);
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
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        // ---- chunk boundary: everyone has left the previous chunk's patch (and ring slot s-1) -> refill, drain, meet
        if (kc > kc_lo) __builtin_amdgcn_s_barrier();
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
    if constexpr (SPLITK) {
        // raw fp32 sums of this chunk range: ws[split][pixel][w_rows], 16 bytes per lane (4 channels of its pixel)
        const size_t npix = (size_t)a.N * Ho * Wo;
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = co_tile * BCO + i * 16 + fq * 4;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const size_t pix = (size_t)(n * Ho + y0 + (f ? frow1 : frow0)) * Wo + x0 + (TW == 32 ? f * 16 : 0) + fj;
                *reinterpret_cast<f32x4_t *>(a.ws + ((size_t)blockIdx.y * npix + pix) * a.w_rows + co) = acc[i][f];
            }
        }
        return;
    }
    if (a.x4) {   // 16-byte stores: channel tiles i, i + 1 exchanged between the k-slot quarters (common.h: v2x_store_pair_x4)
#pragma unroll
        for (int i = 0; i < TCO; i += 2) {
            float4 sc[2], sf[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sc[h] = *reinterpret_cast<const float4 *>(s_ss + (i + h) * 16 + fq * 4);
                sf[h] = *reinterpret_cast<const float4 *>(s_ss + BCO + (i + h) * 16 + fq * 4);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                uint32_t ox[2], oy[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), relu_floor);
                    oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), relu_floor);
                }
                const size_t pix = (size_t)(n * Ho + y0 + (f ? frow1 : frow0)) * Wo + x0 + (TW == 32 ? f * 16 : 0) + fj;
                v2x_store_pair_x4(a.out + pix * a.out_cstride + a.out_coff + co_tile * BCO + i * 16 + fq * 4, fq, ox[0], oy[0], ox[1], oy[1]);
            }
        }
        return;
    }
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
    const v2x_tile_walk walk = v2x_xcd_tile_walk(a.n_px_tiles, a.xcd_walk);     // neighbouring tiles (shared halo rows / columns) on one XCD: common.h
    for (int tile = walk.first; tile < walk.end; tile += walk.step) {
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
        if (a.x4) {   // 16-byte stores (common.h: v2x_store_pair_x4)
#pragma unroll
            for (int i = 0; i < TCO; i += 2) {
                float4 sc[2], sf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    sc[h] = *reinterpret_cast<const float4 *>(s_ss + (i + h) * 16 + fq * 4);
                    sf[h] = *reinterpret_cast<const float4 *>(s_ss + 64 + (i + h) * 16 + fq * 4);
                }
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    uint32_t ox[2], oy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), relu_floor);
                        oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), relu_floor);
                    }
                    const size_t pix = (size_t)(n * Ho + y0 + wave) * Wo + x0 + f * 16 + fj;
                    v2x_store_pair_x4(a.out + pix * a.out_cstride + a.out_coff + i * 16 + fq * 4, fq, ox[0], oy[0], ox[1], oy[1]);
                }
            }
        } else
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

// ---- 8-wave ping-pong form with THREE taps per synchronisation ("s2g") ---------------------------------------------------------------
// The streamed kernel above synchronises once per tap (16 MFMAs per wave between barriers) and DRAINS at every chunk boundary: its
// single patch buffer is refilled with nothing but the other workgroup of the CU to cover the HBM latency (wave-wait 46 % of the
// cycles, 0.27-0.36 of the MFMA peak at 320 maps, profiles/r02_pmc_sq_h.csv).  Here, as in conv3x3_stream8g_kernel (conv_stream.hip):
//   * 8 waves in two groups half a step apart -- a group's load phase (pixel fragment reads) runs under the other group's MFMA phase;
//     the workgroup owns 128 channels x 256 output pixels (8 x 32 or 16 x 16), a wave 64 channels x 64 pixels;
//   * a step is a tap COLUMN of a 32-channel chunk (ky = 0..2 of one kx): 48 MFMAs per wave per synchronisation, the wave's pixel
//     fragments shared between the tap rows (output row r, tap row ky -> patch row 2r + ky: 5 / 9 distinct rows per wave);
//   * the patch is split by column PARITY into two LDS regions that are refilled separately, and a chunk walks kx = 0, 2, 1: the
//     EVEN region serves kx = 0 (entry c) and kx = 2 (entry c + 1) and BOTH sets of fragments are read in the first load phase of
//     the chunk (the wave has 64 accumulator registers, so 20 fragments fit), after which the region is free: the next chunk's even
//     columns are DMA'd during step 1 and have until step 0 of the next chunk to land; the ODD region serves kx = 1 (step 2) and is
//     refilled during step 0 of its own chunk.  One patch buffer (70 KiB) therefore behaves like a double buffer and nothing drains
//     at a chunk boundary.
//   * a stride-2 tile moves FOUR times the bytes of a stride-1 tile per MFMA (47 KiB of LDS-DMA per step and workgroup: 750 cycles of the
//     CU's 64 B/clk vector-memory path against 1 536 cycles of MFMAs).  Issued in the load phases (the first forms of this kernel: 9
//     DMAs ~ 1 000 cycles, tools/s2g_timeline.py) that path is the critical one; here every DMA is issued INSIDE an MFMA phase, one
//     after each block of four MFMAs, so the path is busy evenly and a load phase is fragment reads only.
//   * division of labour: GROUP 0 fills the patch (odd region of chunk kc in M0(s0), even region of chunk kc+1 in M0(s1): 8-9 pieces per
//     wave), GROUP 1 streams the weights (the 24 pieces of step g+2 in M1(g), 6 per wave, 3-step ring).  Waits are counted and sit at the
//     END of the MFMA phases: group 0 waits for the odd region at the end of M0(s1) and for the even region at the end of M0(s2); group
//     1 waits at the end of M1(g) for what it issued in M1(g-1).  Only tap row 0 of a step is read in a LOAD phase (the MFMA phase starts
//     without an exposed LDS latency): its pieces are the first two of a wave's six and are waited for at the end of L1(g) (vmcnt(4)).
// Hazards (interval 2s = L0(s) | M1(s-1), 2s+1 = M0(s) | L1(s); steps s0, s1, s2 of chunk kc are global steps 3kc..3kc+2):
//   even region: last read L1(s0) @6kc+1; rewritten in M0(s1) @6kc+3, landed by the end of M0(s2) @6kc+5, read L0(s0') @6kc+6.
//   odd region: last read L1(s2) @6kc-1; rewritten in M0(s0) @6kc+1, landed by the end of M0(s1) @6kc+3, read L0(s2) @6kc+4.
//   ring slot t%3: last read M1(t-3) @2t-4; rewritten in M1(t-2) @2t-2; tap row 0 landed by the end of L1(t-1) @2t-1, read L0(t) @2t; the
//     rest landed by the end of M1(t-1) @2t, read from M0(t) @2t+1.
// K order is (chunk, kx in {0, 2, 1}, ky): the fp32 sums differ from the 1-tap kernel in their last bits (one bf16 rounding of the output).
template <int TH, int TW>
struct S2GGeom {
    static constexpr int PR = 2 * TH + 1;                        // patch rows
    static constexpr int NE = TW + 1, NO = TW;                   // even / odd patch columns
    static constexpr int E_SLOTS = PR * NE * 4, O_SLOTS = PR * NO * 4;
    static constexpr int E_PIECES = (E_SLOTS + 63) / 64, O_PIECES = (O_SLOTS + 63) / 64;
    static constexpr int RW = TH / 4;                            // output rows of a wave
    static constexpr int NCF = TW / 16;                          // 16-pixel column fragments of a wave
    static constexpr int NBR = 2 * RW + 1;                       // distinct patch rows of a wave
    static constexpr int NB = NBR * NCF;                         // pixel fragments per tap column
    static constexpr int NPW = 9;                                // patch pieces per wave of group 0 and region, at most = DMA slots of an MFMA phase
    static_assert(TH * TW == 256 && RW * NCF == 4, "256 output pixels, 4 fragments per wave");
    static_assert((E_PIECES + 3) / 4 <= NPW && (O_PIECES + 3) / 4 <= NPW, "pieces per wave");
};

template <int TH, int TW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_s2g_kernel(const S2Args a) {
    using G = S2GGeom<TH, TW>;
    constexpr int BCO = 128;
    constexpr int SLICE_BYTES = BCO * 64, STEP_BYTES = 3 * SLICE_BYTES;
    constexpr int N_ST = 8;                                // output stores per wave and tile in the 16-byte form (a.x4); with 8-byte stores (16) the relaxed waits are merely stricter
    constexpr int NWD = 6;                                 // weight DMAs per wave of group 1 and step
    constexpr int DBG = V2X_S2G_DBG_BUILD;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;                                   // 3 x STEP_BYTES
    char *s_E = smem + 3 * STEP_BYTES;
    char *s_O = s_E + G::E_PIECES * 1024;
    float *s_ss = reinterpret_cast<float *>(s_O + G::O_PIECES * 1024);   // [BCO scale | BCO shift]
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wv = wave & 3;
    const int coh = wv & 1, ph = wv >> 1;
    const int R0 = grp * (TH / 2) + ph * G::RW;            // the wave's first output row of the tile

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    const int co_tile = bid % a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int nchunks = a.C >> 5;
    const int S3 = nchunks * 3;
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);

    auto tile_coords = [&](int t, int &n, int &y0, int &x0) {
        const int px_tile = t / a.n_co_tiles;
        n = px_tile / txy;
        const int trem = px_tile - n * txy;
        const int ty = trem / a.tiles_x;
        y0 = ty * TH;
        x0 = (trem - ty * a.tiles_x) * TW;
    };
    // lane id from volatile asm: what is derived from it is not hoisted out of the loops and kept in registers across the MFMA phases
    auto fresh_lane = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // tile-independent part of the DMA source descriptor of piece p of a patch region: (patch row << 16) | (patch column << 4) | logical
    // 16-B slot, -1 behind the region
    auto desc_const = [&](bool odd, int p, int lane) -> int {
        const int L = p * 64 + lane;
        const int RS = (odd ? G::NO : G::NE) * 4;
        const int r = L / RS, q = L - r * RS;
        const int ent = q >> 2, phys = q & 3;
        return r < G::PR ? ((r << 16) | ((2 * ent + (odd ? 1 : 0)) << 4) | (phys ^ ((ent >> 1) & 3))) : -1;
    };
    // ... completed for a tile: (input pixel index << 5) | (logical slot * 8 elements), -1 = zero page
    auto desc_tile = [&](int pk, int n, int y0, int x0) -> int {
        const int y = 2 * y0 - 1 + (pk >> 16), x = 2 * x0 - 1 + ((pk >> 4) & 0xfff);
        const bool ok = pk >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        return ok ? ((((n * a.H + y) * a.W + x) << 5) | ((pk & 15) << 3)) : -1;
    };
    auto issue_piece = [&](int d, int kc, char *dst) {
        if constexpr ((DBG & 2) != 0) return;
        const unsigned off = (unsigned)(d >> 5) * (unsigned)a.C + (unsigned)(kc * 32 + (d & 31));
        glds16q(d >= 0 ? (const void *)(a.in + off) : zero_page, dst);
    };
    // piece p = wv + 4u of step t (chunk t / 3, tap column j = t % 3 -> kx = 0, 2, 1); piece p = (tap row p / 8, piece p % 8 of its slice):
    // a wave's first two pieces (u = 0, 1) are tap row 0
    auto issue_weight = [&](int t, int slot, int u, int lw) {
        if constexpr ((DBG & 1) != 0) return;
        const int kc = t / 3, j = t - kc * 3;
        const int kx = j == 0 ? 0 : (j == 1 ? 2 : 1);
        const int p = wv + 4 * u;
        const int ky = p >> 3, pis = p & 7;
        glds16q(wbase + (size_t)(kc * 9 + ky * 3 + kx) * (BCO * 32) + pis * 512 + lw * 8, s_ring + slot * STEP_BYTES + ky * SLICE_BYTES + pis * 1024);
    };
    // s_waitcnt vmcnt(K) for the wave-uniform run-time K in {0, 4, 6, 8, 9} (+ N_ST)
    auto wait_keep = [&](int k, bool plus_stores) {
        if constexpr ((DBG & 32) != 0) return;
#define S2G_WAIT_CASE(K) case K: if (plus_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K + N_ST) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
        switch (k) {
            S2G_WAIT_CASE(4) S2G_WAIT_CASE(6) S2G_WAIT_CASE(8) S2G_WAIT_CASE(9)
            default: S2G_WAIT_CASE(0)
        }
#undef S2G_WAIT_CASE
    };

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    // group 0: this wave's pieces of the two regions, piece = wv + 4t
    // (the odd region's rows are 128 / 64 slots: its constant part is shifts and masks, recomputed where needed; the even region's rows of
    // 132 / 68 slots need a division, kept in registers)
    int pkE[G::NPW], pdE[G::NPW], pdO[G::NPW];
    {
        const int l0 = fresh_lane();
#pragma unroll
        for (int t = 0; t < G::NPW; ++t) {
            pkE[t] = desc_const(false, wv + 4 * t, l0);
            pdE[t] = desc_tile(pkE[t], n, y0, x0);
            pdO[t] = desc_tile(desc_const(true, wv + 4 * t, l0), n, y0, x0);
        }
    }
    if (grp == 0) {
        // prologue: even region of chunk 0
#pragma unroll
        for (int t = 0; t < G::NPW; ++t)
            if (wv + 4 * t < G::E_PIECES) issue_piece(pdE[t], 0, s_E + (wv + 4 * t) * 1024);
    } else {
        // prologue: weights of steps 0 and 1
        const int lw = fresh_lane();
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
            issue_weight(0, 0, u, lw);
            issue_weight(1, 1, u, lw);
        }
    }
    for (int i = tid; i < BCO; i += 512) {
        s_ss[i] = a.scale[co_tile * BCO + i];
        s_ss[BCO + i] = a.shift[co_tile * BCO + i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // half-step offset

#if (V2X_S2G_DBG_BUILD & 64)
    unsigned *s_tl = reinterpret_cast<unsigned *>(s_ss + 2 * BCO);
    int tl_n = 0;
#define S2G_STAMP(pt) do { if (blockIdx.x == 0 && wv == 0 && tl_n < S2G_T_STEPS) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); if (fresh_lane() == 0) s_tl[(grp * S2G_T_STEPS + tl_n) * 8 + (pt)] = t_; } } while (0)
#else
#define S2G_STAMP(pt) do { } while (0)
#endif
    bool relaxed = false;   // group 1, first step after an epilogue: what its two waits are for is OLDER than the tile's output stores, which may stay in flight
    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < n_tiles;
        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) tile_coords(next, nn, ny0, nx0);

        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int kc = 0; kc < nchunks; ++kc) {
            const bool last_chunk = kc + 1 == nchunks;
            const bool fill_e = !last_chunk || has_next;
            const int kcn = last_chunk ? 0 : kc + 1;
            bf16x8_t Bq[2][G::NB];                         // [0]: kx = 0 (step 0), then kx = 1 (step 2); [1]: kx = 2 (step 1)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                S2G_STAMP(0);
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                // DMAs of this phase.  Group 1: the weights of step g+2; group 0: the odd region of this chunk (step 0), the even region of the next (step 1)
                int wt = kc * 3 + j + 2;
                bool w_ok = true;
                if (wt >= S3) {
                    wt -= S3;
                    w_ok = has_next;
                }
                const int nd = grp ? (w_ok ? NWD : 0)
                                   : (j == 0 ? (G::O_PIECES - wv + 3) / 4 : ((j == 1 && fill_e) ? (G::E_PIECES - wv + 3) / 4 : 0));   // DMAs of this wave in this phase
                __builtin_amdgcn_sched_barrier(0);
                S2G_STAMP(1);
                // ---- L: pixel fragments.  Fragment (patch row q of the wave, column fragment ch) at Bq[.][q * NCF + ch]
                if constexpr ((DBG & 4) != 0) {
                    if (j == 0) {
#pragma unroll
                        for (int q = 0; q < G::NB; ++q) Bq[0][q] = Bq[1][q] = __builtin_bit_cast(bf16x8_t, make_uint4(ln, q, kc, j));
                    }
                } else if (j == 0) {
#pragma unroll
                    for (int ch = 0; ch < G::NCF; ++ch) {
                        const int i0 = ch * 16 + fjl, i2 = i0 + 1;
                        const int c0 = (i0 * 4 + (fql ^ ((i0 >> 1) & 3))) * 16, c2 = (i2 * 4 + (fql ^ ((i2 >> 1) & 3))) * 16;
#pragma unroll
                        for (int q = 0; q < G::NBR; ++q) {
                            const char *prow = s_E + (2 * R0 + q) * (G::NE * 64);
                            Bq[0][q * G::NCF + ch] = *reinterpret_cast<const bf16x8_t *>(prow + c0);
                            Bq[1][q * G::NCF + ch] = *reinterpret_cast<const bf16x8_t *>(prow + c2);
                        }
                    }
                } else if (j == 2) {
#pragma unroll
                    for (int ch = 0; ch < G::NCF; ++ch) {
                        const int i1 = ch * 16 + fjl;
                        const int c1 = (i1 * 4 + (fql ^ ((i1 >> 1) & 3))) * 16;
#pragma unroll
                        for (int q = 0; q < G::NBR; ++q)
                            Bq[0][q * G::NCF + ch] = *reinterpret_cast<const bf16x8_t *>(s_O + (2 * R0 + q) * (G::NO * 64) + c1);
                    }
                }
                const char *ws = s_ring + j * STEP_BYTES + (fql * BCO + fjl) * 16 + coh * 1024;
                bf16x8_t A[2][4];
                // tap row 0 of this step: complete and visible from interval 2g+1 on (group 1 waited for it at the end of M1(g-1)) -- group 1 reads
                // it here, in its load phase L1(g) @2g+1; group 0 (L0(g) @2g) at the top of its MFMA phase
                auto read_a0 = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr ((DBG & 16) == 0) A[0][i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
                        else A[0][i] = A[1][i] = __builtin_bit_cast(bf16x8_t, make_uint4(ln, i, kc, j));
                    }
                };
                if (grp == 1) read_a0();
                __builtin_amdgcn_sched_barrier(0);
                // the DMAs after the fragment reads (nothing issued in a phase is written where that phase reads): the reads land under the DMA issue
                if (grp == 1) {
                    if (w_ok) {
#pragma unroll
                        for (int u = 0; u < NWD; ++u) issue_weight(wt, (j + 2) % 3, u, ln);
                    }
                } else if (j == 0) {
#pragma unroll
                    for (int t = 0; t < G::NPW; ++t)
                        if (wv + 4 * t < G::O_PIECES) issue_piece(pdO[t], kc, s_O + (wv + 4 * t) * 1024);
                } else if (j == 1 && fill_e) {
#pragma unroll
                    for (int t = 0; t < G::NPW; ++t)
                        if (wv + 4 * t < G::E_PIECES) issue_piece(pdE[t], kcn, s_E + (wv + 4 * t) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                // group 0, while the fragments are in flight: complete the patch descriptors that change with the tile -- the odd region's in the
                // last load phase of the previous tile (or the first of the kernel), the even region's (which the last chunk fills for the NEXT tile)
                // in the first load phase of the last chunk
                if (grp == 0) {
                    if (j == 2 && last_chunk && has_next) {
#pragma unroll
                        for (int t = 0; t < G::NPW; ++t) pdO[t] = desc_tile(desc_const(true, wv + 4 * t, ln), nn, ny0, nx0);
                    }
                    if (j == 0 && last_chunk && has_next) {
#pragma unroll
                        for (int t = 0; t < G::NPW; ++t) pdE[t] = desc_tile(pkE[t], nn, ny0, nx0);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): fragments in registers before the regions may be overwritten
                S2G_STAMP(2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                S2G_STAMP(3);
                // ---- M: three taps of 16 MFMAs; the next tap's weight fragments are read after the first four MFMAs of a tap
                if (grp == 0) read_a0();
                auto &B = Bq[j == 1 ? 1 : 0];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    auto mma = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = i0; i < i1; ++i)
#pragma unroll
                            for (int f = 0; f < 4; ++f) {
                                if constexpr ((DBG & 8) == 0)
                                    acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky & 1][i], B[(2 * (f / G::NCF) + ky) * G::NCF + (f % G::NCF)], acc[i][f], 0, 0, 0);
                                else if (f == 0)
                                    acc[i][0] += __builtin_bit_cast(f32x4_t, A[ky & 1][i]) + __builtin_bit_cast(f32x4_t, B[(2 * (f / G::NCF) + ky) * G::NCF]);
                            }
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    mma(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ky < 2 && (DBG & 16) == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) A[(ky + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(ws + (ky + 1) * SLICE_BYTES + i * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma(1, 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                S2G_STAMP(4);
                // end of the MFMA phase: group 1 -- everything but this phase's own DMAs (the weights of the next step are complete); group 0 --
                // the odd region (issued in M(s0)) after step 1, the next even region (issued in M(s1)) after step 2
                if (grp == 1) {
                    wait_keep(nd, relaxed);
                    relaxed = false;
                } else if (j == 1) {
                    wait_keep(nd, false);
                } else if (j == 2 && (DBG & 32) == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                S2G_STAMP(5);
#if (V2X_S2G_DBG_BUILD & 64)
                ++tl_n;
#endif
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: BN / ReLU, bf16, 8-byte NHWC stores (16 per wave)
        {
            const int le = fresh_lane();
            const int fje = le & 15, fqe = le >> 4;
            const int Ho = a.H >> 1, Wo = a.W >> 1;
            if (a.x4) {   // 16-byte stores (common.h: v2x_store_pair_x4)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int cl = coh * 64 + i * 16 + fqe * 4;
                    float4 sc[2], sf[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sc[h] = *reinterpret_cast<const float4 *>(s_ss + cl + h * 16);
                        sf[h] = *reinterpret_cast<const float4 *>(s_ss + BCO + cl + h * 16);
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), relu_floor);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), relu_floor);
                        }
                        const size_t pix = (size_t)(n * Ho + y0 + R0 + f / G::NCF) * Wo + x0 + (f % G::NCF) * 16 + fje;
                        v2x_store_pair_x4(a.out + pix * a.out_cstride + a.out_coff + co_tile * BCO + cl, fqe, ox[0], oy[0], ox[1], oy[1]);
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = coh * 64 + i * 16 + fqe * 4;
                const int co = co_tile * BCO + cl;
                const float4 sc = *reinterpret_cast<const float4 *>(s_ss + cl);
                const float4 sf = *reinterpret_cast<const float4 *>(s_ss + BCO + cl);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                    float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                    uint2 o;
                    o.x = v2x_relu_bf16x2_floor(pack_bf16x2(v0, v1), relu_floor);
                    o.y = v2x_relu_bf16x2_floor(pack_bf16x2(v2, v3), relu_floor);
                    const size_t pix = (size_t)(n * Ho + y0 + R0 + f / G::NCF) * Wo + x0 + (f % G::NCF) * 16 + fje;
                    *reinterpret_cast<uint2 *>(a.out + pix * a.out_cstride + a.out_coff + co) = o;
                }
            }
        }
        if (!has_next) break;
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
        relaxed = true;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // balance the offset barrier of group 1
#if (V2X_S2G_DBG_BUILD & 64)
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * S2G_T_STEPS * 8; i += 512) v2x_s2g_timeline[i] = s_tl[i];
#endif
#undef S2G_STAMP
}

template <int TH, int TW>
static int launch_s2g(const S2Args &a, hipStream_t s) {
    using G = S2GGeom<TH, TW>;
    constexpr int smem = 3 * 3 * 128 * 64 + (G::E_PIECES + G::O_PIECES) * 1024 + 2 * 128 * 4 + ((V2X_S2G_DBG_BUILD & 64) ? 2 * 96 * 8 * 4 : 0);   // 72 + 70 + 1 KiB
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_s2g_kernel<TH, TW>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    int grid = n_tiles;
    const int g = v2x_num_cus() / a.n_co_tiles * a.n_co_tiles;   // persistent: a workgroup's tiles share one channel tile
    if (g > 0 && g < n_tiles) grid = g;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_s2g_kernel");
    return V2X_OK;
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

int v2x_launch_splitk_reduce(const float *ws, int ksplit, long long npix, int w_rows, int Cout, const float *scale, const float *shift, int relu,
                             uint16_t *out, int out_cstride, int out_coff, hipStream_t s);   // conv_stream.hip

template <int BCO, int TH, int TW>
static int launch_s2_splitk(const S2Args &a, hipStream_t s) {
    constexpr int smem = S2_RING * BCO * 64 + S2_PATCH_BYTES + 2 * BCO * 4;
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_s2_stream_kernel<BCO, TH, TW, true>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles, a.ksplit), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_s2_stream_kernel (split-K)");
    return v2x_launch_splitk_reduce(a.ws, a.ksplit, (long long)a.N * (a.H / 2) * (a.W / 2), a.w_rows, a.Cout, a.scale, a.shift, a.relu, a.out,
                                    a.out_cstride, a.out_coff, s);
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
    a.xcd_walk = v2x_tune(V2X_TUNE_HALO_XCD);
    a.x4 = v2x_x4_ok(d->out, d->out_cstride, d->out_coff, d->Cout);
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
    a.ksplit = d->splitk;
    a.ws = d->splitk_ws;
    a.w_rows = d->w_rows;
    if (d->splitk > 1) {
        // latency mode: the 1-tap kernel with the chunk range divided over blockIdx.y + the reduce kernel (sums added in split order: results differ
        // from the unsplit kernels in fp32 summation order).  Every split at least one chunk.
        const int nchunks = d->C0 >> 5, per = (nchunks + d->splitk - 1) / d->splitk;
        if (!d->splitk_ws || d->splitk > nchunks || per * (d->splitk - 1) >= nchunks) return 1;
        if (t16) return rows == 128 ? launch_s2_splitk<128, 8, 16>(a, s) : launch_s2_splitk<64, 8, 16>(a, s);
        return rows == 128 ? launch_s2_splitk<128, S2_TH, S2_TW>(a, s) : launch_s2_splitk<64, S2_TH, S2_TW>(a, s);
    }
    // 8-wave three-tap form: 128-row tiles, >= 2 chunks, 256-pixel output tiles (8 x 32, or 16 x 16 for the 16 x 16 maps).  The choice follows
    // the layer SHAPE only: the two forms differ in fp32 summation order, and a choice that followed N (the items a rank owns) would break the
    // R-rank == 1-rank bitwise equality and give one frame different bits in a small batch than in a large one.  Only a caller that DECLARES a
    // latency launch (desc->small_batch, the host's SMALL_BATCH mode) gets the batch-dependent rule: below four rounds of the persistent grid
    // (one workgroup per CU: at 2.5 tiles per workgroup -- conv2_1 at 8 frames) the 128-pixel kernel with twice the workgroups is 20 % faster.
    // (The N * H * W bound is the kernel's 32-bit pixel arithmetic, 2^26 input pixels = 1 024 maps of 256 x 256: maps, not items per rank, in
    // practice -- the host never batches that many, and a slice of such a batch falls under it on every rank alike.)
    if (rows == 128 && d->C0 >= 64 && v2x_tune(V2X_TUNE_S2_G) != 0 && (long long)d->N * d->H * d->W < (1ll << 26)) {
        const int Ho = d->H / 2, Wo = d->W / 2;
        const bool g32 = Ho % 8 == 0 && Wo % 32 == 0, g16 = !g32 && Ho % 16 == 0 && Wo % 16 == 0;
        const long long tiles = (long long)d->N * (Ho * Wo / 256) * (d->Cout / 128);
        if ((g32 || g16) && !(d->small_batch && tiles < 4 * v2x_num_cus())) {
            S2Args b = a;
            b.tiles_x = Wo / (g32 ? 32 : 16);
            b.tiles_y = Ho / (g32 ? 8 : 16);
            b.n_px_tiles = d->N * b.tiles_x * b.tiles_y;
            b.n_co_tiles = d->Cout / 128;
            return g32 ? launch_s2g<8, 32>(b, s) : launch_s2g<16, 16>(b, s);
        }
    }
    if (t16) return rows == 128 ? launch_s2<128, 8, 16>(a, s) : launch_s2<64, 8, 16>(a, s);
    if (rows == 64 && a.n_co_tiles == 1 && d->C0 == 32) {   // one chunk, one channel tile: resident weights (conv1_1)
        return launch_s2_resident(a, s);   // (the switch S2_RESIDENT = 0 -> streamed form was retired in round 6: no test or tool exercised it since round 3)
    }
    return rows == 128 ? launch_s2<128>(a, s) : launch_s2<64>(a, s);
}
