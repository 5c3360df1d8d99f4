// The detection tail as ONE kernel: conv8_2 (32 -> 32, the decoder's last layer) followed by the fused detection heads (3x3 32 -> 64 hidden,
// block-diagonal 1x1 64 -> 12 + 36, fp32 logits) of upstream coperception/models/det/backbone/Backbone.py::LidarDecoder and
// models/det/base/DetModelBase.py::ClassificationHead / SingleRegressionHead (code absent from /root/reference, see include/v2x_amd.h).
//
// Why: both launches are HBM-bound (conv8_2 reads 1.5 GB and writes 1.34 GB per 320 maps, the heads read those 1.34 GB back 0.5 ms later and
// write 4.03 GB of logits: 530 + 1 100 us).  Here conv8_2's output never leaves the CU: per 8 x 32 output tile
//   * the 12 x 36-pixel window of conv8_1's output is moved into LDS by LDS-DMA (pixel-major, swizzled: conv_halo.hip's patch layout);
//   * STAGE A evaluates conv8_2 on the 10 x 34 region the heads need -- 340 pixels walked as 22 linear 16-pixel fragments (conv_halo_pair.hip's
//     scheme), scale / shift / ReLU, rounded to bf16 exactly as the stand-alone layer stores it, ZERO where the pixel lies outside the image
//     (= the heads' zero padding) -- into a second LDS patch;
//   * STAGE B is conv_halo.hip's heads form on that patch: tap-column groups, the hidden rows in kappa order so that a lane's accumulators are
//     its B fragment of the 1x1, whose weights live in registers; fp32 split stores (cls | loc).
// 8 waves = two 4-wave groups, each with its OWN tile, window and patch, on ONE resident copy of the three weight sets (54 KiB); the groups run
// one stage apart (raw s_barrier, group 1 starts one barrier late): while one group's waves multiply stage A of their next tile, the other's
// are in stage B and its 12 logit stores per wave -- one workgroup per CU, 152 KiB of LDS.  The next tile's window is requested right after the
// barrier that ends stage A and waited for with a COUNTED vmcnt that leaves the tile's own stores in flight.
// K order and epilogue arithmetic are those of the stand-alone kernels: the logits are bit-identical to v2x_conv2d(conv8_2) followed by
// v2x_conv2d(heads) (tests/test_gpu_tail.py).  The recompute of conv8_2 on the halo ring costs 22/16 of its MFMAs.
#include "conv_stream.h"   // lds_ld4: LDS table reads through an explicit address-space pointer
#include <cstdlib>

typedef const __attribute__((address_space(1))) void *gptr_tl_t;
typedef __attribute__((address_space(3))) void *lptr_tl_t;

typedef unsigned int u32x2_tl_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2_tl_t lds_u2_t;   // stage A's results go to LDS through an EXPLICIT address-space pointer: behind a generic
                                                                  // one the compiler orders the write after every LDS-DMA in flight (s_waitcnt vmcnt(0))

static __device__ __attribute__((aligned(64))) unsigned int g_zero_page_tail[16];

struct TailArgs {
    const uint16_t *in;                 // conv8_1's output [N][H][W][in_cstride] bf16 (32 channels at in_coff)
    int in_cstride, in_coff;
    int N, H, W;
    const uint16_t *wA;                 // conv8_2: k-slot-major [36][32][8]
    const float *scA, *shA;
    int reluA;
    const uint16_t *wB;                 // heads hidden: k-slot-major [36][64][8], rows in kappa order
    const float *scB, *shB;             // [64] in kappa order
    int reluB;
    const uint16_t *w2;                 // chained 1x1: row-major [48][64] bf16, K in kappa order
    const float *sc2, *sh2;             // [48]
    int relu2, split;                   // rows < split -> out, the others -> out2
    float *out, *out2;
    int out_cstride, out_coff, out2_cstride;
    int tiles_x, tiles_y, n_tiles;
    int xcd_walk;
};

#ifndef V2X_TAIL_FENCE_BUILD
#define V2X_TAIL_FENCE_BUILD 1
#endif
#if V2X_TAIL_FENCE_BUILD
#define TAIL_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define TAIL_FENCE
#endif
namespace tail {
constexpr int TH = 8, TW = 32;
constexpr int MH = TH + 2, MW = TW + 2;     // stage-A region = stage B's patch
constexpr int IH = TH + 4, IW = TW + 4;     // input window
constexpr int NMID = MH * MW;               // 340
constexpr int NFRAG = (NMID + 15) / 16;     // 22
constexpr int FPW = (NFRAG + 3) / 4;        // 6 (waves 2, 3 of a group own 5)
constexpr int WA_BYTES = 36 * 32 * 16;      // 18 432
constexpr int WB_BYTES = 36 * 64 * 16;      // 36 864
constexpr int IN_BYTES = IH * IW * 64;      // 27 648 = 27 one-KiB pieces
constexpr int IN_PIECES = IN_BYTES / 1024;
constexpr int PPW = (IN_PIECES + 3) / 4;    // pieces per wave (7; the last wave's 7th is a dummy)
constexpr int MID_BYTES = NMID * 64;        // 21 760
constexpr int GRP_BYTES = IN_BYTES + MID_BYTES;
constexpr int OFF_GRP = WA_BYTES + WB_BYTES;
constexpr int OFF_DUMMY = OFF_GRP + 2 * GRP_BYTES;
constexpr int OFF_TAB = OFF_DUMMY + 1024;
constexpr int SMEM = OFF_TAB + 288 * 4;     // (12 fp32 dwordx4 stores per wave and tile: 3 channel tiles x 4 fragments -- the counted wait below)
static_assert(IN_BYTES % 1024 == 0 && SMEM <= 160 * 1024, "LDS map");
#ifndef V2X_TAIL_PSWZ_BUILD
#define V2X_TAIL_PSWZ_BUILD 1
#endif
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> V2X_TAIL_PSWZ_BUILD) & 3); }
}  // namespace tail

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_tail_kernel(const TailArgs a) {
    using namespace tail;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int fj = lane & 15, fq = lane >> 4;
    char *s_wA = smem, *s_wB = smem + WA_BYTES;
    char *s_in = smem + OFF_GRP + grp * GRP_BYTES, *s_mid = s_in + IN_BYTES;

    // the three weight sets: wA and wB by LDS-DMA (resident), the 1x1's fragments in registers
    for (int off = wave8 * 1024; off < WA_BYTES + WB_BYTES; off += 8192) {
        const char *src = off < WA_BYTES ? reinterpret_cast<const char *>(a.wA) + off : reinterpret_cast<const char *>(a.wB) + (off - WA_BYTES);
        __builtin_amdgcn_global_load_lds((gptr_tl_t)(src + lane * 16), (lptr_tl_t)(smem + off), 16, 0, 0);
    }
    bf16x8_t w2f[3][2];
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
        for (int s = 0; s < 2; ++s) w2f[i2][s] = *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(i2 * 16 + fj) * 64 + s * 32 + fq * 8);
    // scale / shift vectors of the three layers: a 1.1-KiB LDS table [scA 32 | shA 32 | scB 64 | shB 64 | sc2 48 | sh2 48] (in registers they are 72 VGPRs
    // the two stages cannot spare; read from global memory per tile they would be waited for together with the previous tile's stores)
    float *s_tab = reinterpret_cast<float *>(smem + OFF_TAB);
    for (int i = tid; i < 288; i += 512) {
        float v;
        if (i < 32) v = a.scA[i];
        else if (i < 64) v = a.shA[i - 32];
        else if (i < 128) v = a.scB[i - 64];
        else if (i < 192) v = a.shB[i - 128];
        else if (i < 240) v = a.sc2[i - 192];
        else v = a.sh2[i - 240];
        s_tab[i] = v;
    }
    const uint32_t floorA = a.reluA ? 0u : 0x80008000u, floorB = a.reluB ? 0u : 0x80008000u;
    // everything loaded once is CONSUMED before the loop: vmcnt is in-order, and a load the compiler still has to wait for inside the loop would
    // be waited for together with the previous tile's stores (conv_halo_pair.hip)
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(w2f[i][0]), "v"(w2f[i][1]));

    // tile-invariant lane geometry.  Stage A: the lane's pixel of each of its fragments, its window offsets per tap column (the swizzle follows
    // the column).
    int in_off[FPW][3], rc[FPW];
#pragma unroll
    for (int t = 0; t < FPW; ++t) {
        const int p = (wave + 4 * t) * 16 + fj;
        const int pc = p < NMID ? p : NMID - 1;
        const int r = pc / MW, c = pc - r * MW;
        rc[t] = r | (c << 8) | ((p < NMID ? 1 : 0) << 16);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) in_off[t][kx] = ((r * IW + c + kx) * 4 + swz4(fq, c + kx)) * 16;
    }
    const int txy = a.tiles_x * a.tiles_y;
    auto tile_coords = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / txy;
        const int r = tile - n * txy;
        ty = r / a.tiles_x;
        tx = r - ty * a.tiles_x;
    };
    auto fresh_lane = [&]() -> int {   // recomputed where it is used: lane-derived DMA addresses hoisted out of the tile loop cost ~30 registers (spills)
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto load_window = [&](int tile) {   // exactly PPW DMAs per wave (the fourth wave's 7th goes to a dummy page)
        int n, ty, tx;
        tile_coords(tile, n, ty, tx);
        const int ln = fresh_lane();
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int piece = wave + 4 * u;
            const int sidx = piece * 64 + ln;
            const int pix = sidx >> 2, phys = sidx & 3;
            const int pr = pix / IW, pcx = pix - pr * IW;
            const int y = ty * TH - 2 + pr, x = tx * TW - 2 + pcx;
            const bool ok = piece < IN_PIECES && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = (unsigned)((n * a.H + y) * a.W + x) * (unsigned)a.in_cstride + (unsigned)(a.in_coff + swz4(phys, pcx) * 8);
            char *dst = piece < IN_PIECES ? s_in + piece * 1024 : smem + OFF_DUMMY;   // wave-uniform
            __builtin_amdgcn_global_load_lds((gptr_tl_t)(ok ? (const void *)(a.in + off) : (const void *)g_zero_page_tail), (lptr_tl_t)dst, 16, 0, 0);
        }
    };

    // tiles are walked in PAIRS (group g owns tile 2 p + g: x-neighbours, their windows overlap in L2); every group runs the same number of
    // iterations -- one without a tile (odd tile count) still meets the barriers
    const int n_pairs = (a.n_tiles + 1) >> 1;
    const v2x_tile_walk walk = v2x_xcd_tile_walk(n_pairs, a.xcd_walk);
    int pair_i = walk.first;
    {
        const int tile = 2 * pair_i + grp;
        if (pair_i < walk.end && tile < a.n_tiles) load_window(tile);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // weights and both first windows have landed
#ifndef V2X_TAIL_OFFSET_BUILD
#define V2X_TAIL_OFFSET_BUILD 1
#endif
    if (V2X_TAIL_OFFSET_BUILD && grp == 1) __builtin_amdgcn_s_barrier();   // one stage behind group 0

    for (; pair_i < walk.end; pair_i += walk.step) {
        const int tile = 2 * pair_i + grp;
        const bool has = tile < a.n_tiles;
        const int next_pair = pair_i + walk.step;
        const int next = 2 * next_pair + grp;
        const bool has_next = next_pair < walk.end && next < a.n_tiles;
        int n = 0, ty = 0, tx = 0;
        tile_coords(has ? tile : 0, n, ty, tx);

        // ================= STAGE A: conv8_2 on the 10 x 34 region =================
        if (has) {
            f32x4_t acc[FPW][2];
#pragma unroll
            for (int t = 0; t < FPW; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // per tap: 2 weight + 6 pixel fragments in front of 12 MFMAs; the next tap's fragments are read under them (two alternating sets)
                bf16x8_t A[2][2], B[2][FPW];
#pragma unroll
                for (int i = 0; i < 2; ++i) A[0][i] = *reinterpret_cast<const bf16x8_t *>(s_wA + ((kx * 4 + fq) * 32 + i * 16 + fj) * 16);
#pragma unroll
                for (int t = 0; t < FPW; ++t) B[0][t] = *reinterpret_cast<const bf16x8_t *>(s_in + in_off[t][kx]);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    TAIL_FENCE;
                    if (ky < 2) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            A[(ky + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(s_wA + ((((ky + 1) * 3 + kx) * 4 + fq) * 32 + i * 16 + fj) * 16);
#pragma unroll
                        for (int t = 0; t < FPW; ++t) B[(ky + 1) & 1][t] = *reinterpret_cast<const bf16x8_t *>(s_in + in_off[t][kx] + (ky + 1) * (IW * 64));
                    }
                    TAIL_FENCE;
#pragma unroll
                    for (int t = 0; t < FPW; ++t)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky & 1][i], B[ky & 1][t], acc[t][i], 0, 0, 0);
                }
                TAIL_FENCE;
            }
            float4 scA[2], shA[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                scA[i] = lds_ld4((lds_cf_t *)s_tab + i * 16 + fq * 4);
                shA[i] = lds_ld4((lds_cf_t *)s_tab + 32 + i * 16 + fq * 4);
            }
#pragma unroll
            for (int t = 0; t < FPW; ++t) {
                const int r = rc[t] & 0xff, c = (rc[t] >> 8) & 0xff;
                const int y = ty * TH - 1 + r, x = tx * TW - 1 + c;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                if (!(rc[t] >> 16)) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint2 o;
                    o.x = v2x_relu_bf16x2_floor(pack_bf16x2(acc[t][i][0] * scA[i].x + shA[i].x, acc[t][i][1] * scA[i].y + shA[i].y), floorA);
                    o.y = v2x_relu_bf16x2_floor(pack_bf16x2(acc[t][i][2] * scA[i].z + shA[i].z, acc[t][i][3] * scA[i].w + shA[i].w), floorA);
                    o.x = inside ? o.x : 0u;
                    o.y = inside ? o.y : 0u;
                    *(lds_u2_t *)(s_mid + ((r * MW + c) * 4 + swz4(i * 2 + (fq >> 1), c)) * 16 + (fq & 1) * 8) = (u32x2_tl_t){o.x, o.y};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // (X) this group's patch is complete; every wave of the group is done reading its window
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) load_window(next);   // lands under stage B

        // ================= STAGE B: heads hidden 3x3 32 -> 64, chained 1x1, fp32 logits =================
        if (has) {
            f32x4_t acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // the tap column's 8 pixel fragments (4 patch rows x 2 column halves) serve its three taps; the weight fragments of tap ky + 1 are read
                // under the 16 MFMAs of tap ky (two alternating sets: 32 registers instead of 48)
                bf16x8_t A[2][4], B[8];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int pr = 2 * wave + rr, pc = ch * 16 + fj + kx;
                        B[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(s_mid + ((pr * MW + pc) * 4 + swz4(fq, pc)) * 16);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) A[0][i] = *reinterpret_cast<const bf16x8_t *>(s_wB + ((kx * 4 + fq) * 64 + i * 16 + fj) * 16);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    TAIL_FENCE;
                    if (ky < 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            A[(ky + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(s_wB + ((((ky + 1) * 3 + kx) * 4 + fq) * 64 + i * 16 + fj) * 16);
                    }
                    TAIL_FENCE;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky & 1][i], B[((f >> 1) + ky) * 2 + (f & 1)], acc[i][f], 0, 0, 0);
                }
                TAIL_FENCE;
            }
            float4 scB[4], shB[4], s2v[3], t2v[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // packed row 16 i + 4 q + r computes hidden channel kappa = 32 (i >> 1) + 8 q + 4 (i & 1) + r
                const int kappa = 32 * (i >> 1) + 8 * fq + 4 * (i & 1);
                scB[i] = lds_ld4((lds_cf_t *)s_tab + 64 + kappa);
                shB[i] = lds_ld4((lds_cf_t *)s_tab + 128 + kappa);
            }
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) {
                s2v[i2] = lds_ld4((lds_cf_t *)s_tab + 192 + i2 * 16 + fq * 4);
                t2v[i2] = lds_ld4((lds_cf_t *)s_tab + 240 + i2 * 16 + fq * 4);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                bf16x8_t hb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float h[8];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int i = 2 * s + half;
                        h[half * 4 + 0] = acc[i][f][0] * scB[i].x + shB[i].x;
                        h[half * 4 + 1] = acc[i][f][1] * scB[i].y + shB[i].y;
                        h[half * 4 + 2] = acc[i][f][2] * scB[i].z + shB[i].z;
                        h[half * 4 + 3] = acc[i][f][3] * scB[i].w + shB[i].w;
                    }
                    uint4 p;
                    p.x = v2x_relu_bf16x2_floor(pack_bf16x2(h[0], h[1]), floorB);
                    p.y = v2x_relu_bf16x2_floor(pack_bf16x2(h[2], h[3]), floorB);
                    p.z = v2x_relu_bf16x2_floor(pack_bf16x2(h[4], h[5]), floorB);
                    p.w = v2x_relu_bf16x2_floor(pack_bf16x2(h[6], h[7]), floorB);
                    hb[s] = __builtin_bit_cast(bf16x8_t, p);
                }
                const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                const size_t pix = (size_t)(n * a.H + y) * a.W + x;
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) {
                    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2][s], hb[s], d, 0, 0, 0);
                    const int co = i2 * 16 + fq * 4;
                    const float4 s2 = s2v[i2], t2 = t2v[i2];
                    float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                    if (a.relu2) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                        v2 = fmaxf(v2, 0.f);
                        v3 = fmaxf(v3, 0.f);
                    }
                    const bool second = co >= a.split;
                    float *dst = second ? a.out2 + pix * a.out2_cstride + (co - a.split) : a.out + pix * a.out_cstride + a.out_coff + co;
                    *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                }
            }
        }
        // the next window's pieces are OLDER than this tile's stores (in-order vmcnt): they have landed, the stores stay in flight
        if (has) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // (Y) the patch may be rewritten; the next window is visible to the whole group
        __builtin_amdgcn_sched_barrier(0);
    }
    if (V2X_TAIL_OFFSET_BUILD && grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's offset barrier
}

// ---- the WAVE-PRIVATE form: every wave owns its own 8 x 16 output tile, window and patch; all three weight sets live in its registers -----------------------
// What bounds the two-group form above (profiles/r05_tail_phase_probe.txt): not HBM (dropping the stores: -1 %) and not the MFMAs alone (dropping them all:
// -27 %) but the LDS: a wave re-reads all 54 KiB of weights for every tile (54 of its 156 fragment reads), 1.2 MiB of LDS reads per pair of tiles at the
// ~85 B/clk the CU delivers.  Here a wave holds conv8_2's 18 and the 1x1's 6 weight fragments in registers (one wave per SIMD: 512 registers), works on twice
// the pixels per hidden-layer weight fragment (8 fragments instead of 4), and the linear walk wastes less: 108 + 30 + 36 reads per 128 pixels instead of 312.  Nothing is shared between waves -- no
// workgroup barrier anywhere in the loop; the four waves of a CU drift apart and their LDS, MFMA, VALU and store phases overlap by themselves.
// Stage A: the 10 x 18 region as 12 linear 16-pixel fragments (the last has 4 valid lanes) x 2 channel tiles, one tap = 12 reads in front of 24 MFMAs, the
// next tap's reads under them.  Stage B: 8 rows x 16 pixels x 4 channel tiles; a tap column's 10 patch rows serve its three taps (96 MFMAs).
// Same K order and epilogue arithmetic as the stand-alone kernels: bit-identical logits.
namespace tailw {
constexpr int TH = 8, TW = 16;
constexpr int MH = TH + 2, MW = TW + 2;     // 10 x 18 region
constexpr int IH = TH + 4, IW = TW + 4;     // 12 x 20 window
constexpr int NMID = MH * MW;               // 180
constexpr int NFRAG = (NMID + 15) / 16;     // 12
constexpr int IN_BYTES = IH * IW * 64;      // 15 360 = 15 one-KiB pieces
constexpr int IN_PIECES = IN_BYTES / 1024;
constexpr int MID_BYTES = NMID * 64;        // 11 520
constexpr int WAVE_BYTES = IN_BYTES + MID_BYTES;
constexpr int OFF_TAB = 4 * WAVE_BYTES;     // 107 520
constexpr int OFF_WB = OFF_TAB + 288 * 4;   // the hidden layer's weights stay in LDS (36 KiB, shared by the four waves): 144 registers more do not fit
constexpr int SMEM = OFF_WB + 36 * 64 * 16; // (24 fp32 dwordx4 stores per tile: 8 fragments x 3 channel tiles -- the counted wait below)
static_assert(IN_BYTES % 1024 == 0 && SMEM <= 160 * 1024, "LDS map");
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> 1) & 3); }
typedef const __attribute__((address_space(3))) bf16x8_t lds_cfrag_t;   // fragment reads through an explicit address-space pointer (see lds_u2_t above)
}  // namespace tailw

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_tail_wave_kernel(const TailArgs a) {
    using namespace tailw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    char *s_in = smem + wave * WAVE_BYTES, *s_mid = s_in + IN_BYTES;
    float *s_tab = reinterpret_cast<float *>(smem + OFF_TAB);

    // the three weight sets: this lane's fragments, resident in registers for the whole kernel
    bf16x8_t WA[9][2], W2[3][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < 2; ++i) WA[tap][i] = *reinterpret_cast<const bf16x8_t *>(a.wA + ((tap * 4 + fq) * 32 + i * 16 + fj) * 8);
    for (int off = wave * 1024; off < 36 * 64 * 16; off += 4096)
        __builtin_amdgcn_global_load_lds((gptr_tl_t)(reinterpret_cast<const char *>(a.wB) + off + lane * 16), (lptr_tl_t)(smem + OFF_WB + off), 16, 0, 0);
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
        for (int s = 0; s < 2; ++s) W2[i2][s] = *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(i2 * 16 + fj) * 64 + s * 32 + fq * 8);
    for (int i = tid; i < 288; i += 256) {
        float v;
        if (i < 32) v = a.scA[i];
        else if (i < 64) v = a.shA[i - 32];
        else if (i < 128) v = a.scB[i - 64];
        else if (i < 192) v = a.shB[i - 128];
        else if (i < 240) v = a.sc2[i - 192];
        else v = a.sh2[i - 240];
        s_tab[i] = v;
    }
    const uint32_t floorA = a.reluA ? 0u : 0x80008000u, floorB = a.reluB ? 0u : 0x80008000u;
    // everything loaded once is consumed before the loop (in-order vmcnt: a load still pending inside the loop would be waited for with the stores)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) asm volatile("" ::"v"(WA[tap][0]), "v"(WA[tap][1]));
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) asm volatile("" ::"v"(W2[i2][0]), "v"(W2[i2][1]));

    // tile-invariant lane geometry of stage A: region row | column << 8 | valid << 16 | window pixel index << 20 of the lane's pixel per fragment
    int rc[NFRAG];
#pragma unroll
    for (int t = 0; t < NFRAG; ++t) {
        const int p = t * 16 + fj;
        const int pc = p < NMID ? p : NMID - 1;
        const int r = pc / MW, c = pc - r * MW;
        rc[t] = r | (c << 8) | ((p < NMID ? 1 : 0) << 16) | ((r * IW + c) << 20);
    }
    const int txy = a.tiles_x * a.tiles_y;   // here: tiles of 8 x 16
    auto tile_coords = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / txy;
        const int r = tile - n * txy;
        ty = r / a.tiles_x;
        tx = r - ty * a.tiles_x;
    };
    auto load_window = [&](int tile) {   // 15 DMAs (the lane's pixel per piece is recomputed here: 15 hoisted registers are better spent on fragments)
        int n, ty, tx;
        tile_coords(tile, n, ty, tx);
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int u = 0; u < IN_PIECES; ++u) {
            const int sidx = u * 64 + ln;
            const int pix = sidx >> 2, phys = sidx & 3;
            const int pr = pix / IW, pcx = pix - pr * IW, slot = swz4(phys, pcx);
            const int y = ty * TH - 2 + pr, x = tx * TW - 2 + pcx;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = (unsigned)((n * a.H + y) * a.W + x) * (unsigned)a.in_cstride + (unsigned)(a.in_coff + slot * 8);
            __builtin_amdgcn_global_load_lds((gptr_tl_t)(ok ? (const void *)(a.in + off) : (const void *)g_zero_page_tail), (lptr_tl_t)(s_in + u * 1024), 16, 0, 0);
        }
    };

    // a workgroup owns a contiguous run of tiles (XCD-contiguous: v2x_xcd_tile_walk over groups of four), its waves take neighbouring tiles
    const int n_quads = (a.n_tiles + 3) >> 2;
    const v2x_tile_walk walk = v2x_xcd_tile_walk(n_quads, a.xcd_walk);
    int quad = walk.first;
    if (quad < walk.end && 4 * quad + wave < a.n_tiles) load_window(4 * quad + wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the table and the hidden layer's weights (the only data shared between the waves)

    for (; quad < walk.end; quad += walk.step) {
        const int tile = 4 * quad + wave;
        if (tile >= a.n_tiles) break;   // (only in the last quad; this wave had no window in flight)
        const int next = tile + 4 * walk.step;
        const bool has_next = quad + walk.step < walk.end && next < a.n_tiles;
        int n, ty, tx;
        tile_coords(tile, n, ty, tx);

        // ================= STAGE A =================
        {
            f32x4_t acc[NFRAG][2];
#pragma unroll
            for (int t = 0; t < NFRAG; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // K order of the stand-alone layer: tap column outer, tap row inner.  A tap is walked as two halves of 6 fragments; a ring of three half-tap
            // fragment sets keeps the reads two halves (24 MFMAs) ahead of their use
            constexpr int HF = NFRAG / 2;
            bf16x8_t B[3][HF];
            auto read_half = [&](int h, bf16x8_t (&dst)[HF]) __attribute__((always_inline)) {   // h = 2 (3 kx + ky) + half
                const int g = h >> 1, kx = g / 3, ky = g - 3 * kx, t0 = (h & 1) * HF;
#pragma unroll
                for (int t = 0; t < HF; ++t) {
                    const int col = ((rc[t0 + t] >> 8) & 0xff) + kx;
                    dst[t] = *(lds_cfrag_t *)(s_in + (((rc[t0 + t] >> 20) + kx) * 4 + swz4(fq, col)) * 16 + ky * (IW * 64));
                }
            };
            read_half(0, B[0]);
            read_half(1, B[1]);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                const int g = h >> 1, kx = g / 3, ky = g - 3 * kx, t0 = (h & 1) * HF;
                __builtin_amdgcn_sched_barrier(0);
                if (h + 2 < 18) read_half(h + 2, B[(h + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < HF; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[t0 + t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WA[ky * 3 + kx][i], B[h % 3][t], acc[t0 + t][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 scA[2], shA[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                scA[i] = lds_ld4((lds_cf_t *)s_tab + i * 16 + fq * 4);
                shA[i] = lds_ld4((lds_cf_t *)s_tab + 32 + i * 16 + fq * 4);
            }
#pragma unroll
            for (int t = 0; t < NFRAG; ++t) {
                const int r = rc[t] & 0xff, c = (rc[t] >> 8) & 0xff;
                const int y = ty * TH - 1 + r, x = tx * TW - 1 + c;
                const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && ((rc[t] >> 16) & 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint2 o;
                    o.x = v2x_relu_bf16x2_floor(pack_bf16x2(acc[t][i][0] * scA[i].x + shA[i].x, acc[t][i][1] * scA[i].y + shA[i].y), floorA);
                    o.y = v2x_relu_bf16x2_floor(pack_bf16x2(acc[t][i][2] * scA[i].z + shA[i].z, acc[t][i][3] * scA[i].w + shA[i].w), floorA);
                    o.x = inside ? o.x : 0u;
                    o.y = inside ? o.y : 0u;
                    // (lanes behind the region's last pixel rewrite pixel 179 with zeros... they must not: they are masked off)
                    if ((rc[t] >> 16) & 1) *(lds_u2_t *)(s_mid + ((r * MW + c) * 4 + swz4(i * 2 + (fq >> 1), c)) * 16 + (fq & 1) * 8) = (u32x2_tl_t){o.x, o.y};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is written, the window read for the last time
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) load_window(next);   // lands under stage B

        // ================= STAGE B =================
        {
            f32x4_t acc[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 8; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // a tap column's 10 patch rows serve its three taps; the next tap's 4 weight fragments and (during the column's last tap) the next column's
            // rows are read under the 32 MFMAs of a tap
            bf16x8_t B[2][MH], A[2][4];
            auto read_col = [&](int kx, bf16x8_t (&dst)[MH]) __attribute__((always_inline)) {
                const int pc = fj + kx;
#pragma unroll
                for (int pr = 0; pr < MH; ++pr) dst[pr] = *(lds_cfrag_t *)(s_mid + ((pr * MW + pc) * 4 + swz4(fq, pc)) * 16);
            };
            auto read_w = [&](int g, bf16x8_t (&dst)[4]) __attribute__((always_inline)) {   // g = 3 kx + ky
                const int kx = g / 3, ky = g - 3 * kx;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[i] = *(lds_cfrag_t *)(smem + OFF_WB + (((ky * 3 + kx) * 4 + fq) * 64 + i * 16 + fj) * 16);
            };
            read_col(0, B[0]);
            read_w(0, A[0]);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                const int kx = g / 3, ky = g - 3 * kx;
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < 9) read_w(g + 1, A[(g + 1) & 1]);
                if (ky == 2 && kx + 1 < 3) read_col(kx + 1, B[(kx + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int f = 0; f < 8; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g & 1][i], B[kx & 1][f + ky], acc[i][f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 scB[4], shB[4], s2v[3], t2v[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // packed row 16 i + 4 q + r computes hidden channel kappa = 32 (i >> 1) + 8 q + 4 (i & 1) + r
                const int kappa = 32 * (i >> 1) + 8 * fq + 4 * (i & 1);
                scB[i] = lds_ld4((lds_cf_t *)s_tab + 64 + kappa);
                shB[i] = lds_ld4((lds_cf_t *)s_tab + 128 + kappa);
            }
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) {
                s2v[i2] = lds_ld4((lds_cf_t *)s_tab + 192 + i2 * 16 + fq * 4);
                t2v[i2] = lds_ld4((lds_cf_t *)s_tab + 240 + i2 * 16 + fq * 4);
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                bf16x8_t hb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float h[8];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int i = 2 * s + half;
                        h[half * 4 + 0] = acc[i][f][0] * scB[i].x + shB[i].x;
                        h[half * 4 + 1] = acc[i][f][1] * scB[i].y + shB[i].y;
                        h[half * 4 + 2] = acc[i][f][2] * scB[i].z + shB[i].z;
                        h[half * 4 + 3] = acc[i][f][3] * scB[i].w + shB[i].w;
                    }
                    uint4 p;
                    p.x = v2x_relu_bf16x2_floor(pack_bf16x2(h[0], h[1]), floorB);
                    p.y = v2x_relu_bf16x2_floor(pack_bf16x2(h[2], h[3]), floorB);
                    p.z = v2x_relu_bf16x2_floor(pack_bf16x2(h[4], h[5]), floorB);
                    p.w = v2x_relu_bf16x2_floor(pack_bf16x2(h[6], h[7]), floorB);
                    hb[s] = __builtin_bit_cast(bf16x8_t, p);
                }
                const int y = ty * TH + f, x = tx * TW + fj;
                const size_t pix = (size_t)(n * a.H + y) * a.W + x;
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) {
                    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2[i2][s], hb[s], d, 0, 0, 0);
                    const int co = i2 * 16 + fq * 4;
                    const float4 s2 = s2v[i2], t2 = t2v[i2];
                    float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                    if (a.relu2) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                        v2 = fmaxf(v2, 0.f);
                        v3 = fmaxf(v3, 0.f);
                    }
                    const bool second = co >= a.split;
                    float *dst = second ? a.out2 + pix * a.out2_cstride + (co - a.split) : a.out + pix * a.out_cstride + a.out_coff + co;
                    *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                }
            }
        }
        // the next window's pieces are OLDER than this tile's 24 stores (in-order vmcnt): they have landed, the stores stay in flight
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

int v2x_num_cus();   // conv_stream.hip

static int tail_launch(const TailArgs &a, hipStream_t s) {
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tail::SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_tail_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tailw::SMEM);
    }
    if (v2x_tune(V2X_TUNE_TAIL_WAVE) != 0 && a.W % tailw::TW == 0) {   // the wave-private form (default): tiles of 8 x 16, four per workgroup and step
        TailArgs w = a;
        w.tiles_x = a.W / tailw::TW;
        w.tiles_y = a.H / tailw::TH;
        w.n_tiles = a.N * w.tiles_x * w.tiles_y;
        const int n_quads = (w.n_tiles + 3) / 4;
        int grid = v2x_num_cus();
        if (grid > n_quads) grid = n_quads;
        hipLaunchKernelGGL(conv3x3_tail_wave_kernel, dim3(grid), dim3(256), tailw::SMEM, s, w);
        V2X_CHECK_LAUNCH("conv3x3_tail_wave_kernel");
        return V2X_OK;
    }
    const int n_pairs = (a.n_tiles + 1) / 2;
    int grid = v2x_num_cus();
    if (grid > n_pairs) grid = n_pairs;
    hipLaunchKernelGGL(conv3x3_tail_kernel, dim3(grid), dim3(512), tail::SMEM, s, a);
    V2X_CHECK_LAUNCH("conv3x3_tail_kernel");
    return V2X_OK;
}

// v2x_conv2d_pair's second form (include/v2x_amd.h): first = conv8_2 (3x3 s1 p1, 32 -> 32, w_layout 1, bf16 NHWC input), second = the fused detection heads
// (w_layout 1, 32 -> 64 hidden in kappa order, chained 1x1 -> 48 = split + 36 fp32 channels).  first->out is ignored.
int v2x_conv_tail_dispatch(const v2x_conv_desc *first, const v2x_conv_desc *second, hipStream_t stream) {
    V2X_REQUIRE(first->ksize == 3 && first->stride == 1 && first->pad == 1 && first->w_layout == 1 && first->C0 == 32 && first->C1 == 0 && first->Cout == 32 &&
                    first->Cout2 == 0 && first->epilogue == V2X_EPI_BF16 && first->up0 == 0 && first->in_format == 0,
                "v2x_conv2d_pair (tail form): the first layer must be a halo-packed 3x3 stride-1 32 -> 32 bf16 layer on a bf16 NHWC input");
    V2X_REQUIRE(second->ksize == 3 && second->stride == 1 && second->pad == 1 && second->w_layout == 1 && second->C0 == 32 && second->C1 == 0 &&
                    second->Cout == 64 && second->Cout2 == 48 && second->epilogue == V2X_EPI_F32 && second->up0 == 0 && second->split > 0 &&
                    second->split % 4 == 0 && second->split < 48,
                "v2x_conv2d_pair (tail form): the second layer must be the fused heads (3x3 32 -> 64 chained with a 1x1 -> 48, fp32 split outputs)");
    V2X_REQUIRE(first->in0 && first->weight && first->scale && first->shift && second->weight && second->scale && second->shift && second->weight2 &&
                    second->scale2 && second->shift2 && second->out && second->out2,
                "v2x_conv2d_pair (tail form): null pointer");
    V2X_REQUIRE(first->N == second->N && first->H == second->H && first->W == second->W, "v2x_conv2d_pair (tail form): extents differ");
    V2X_REQUIRE(first->N >= 0 && first->H > 0 && first->W > 0 && first->H % tail::TH == 0 && first->W % tail::TW == 0,
                "v2x_conv2d_pair (tail form): H %% 8 == 0 and W %% 32 == 0 required (H=%d W=%d)", first->H, first->W);
    V2X_REQUIRE((long long)first->N * first->H * first->W < (1ll << 27), "v2x_conv2d_pair (tail form): N*H*W must stay below 2^27 (32-bit element offsets)");
    V2X_REQUIRE(second->out_cstride >= second->split + second->out_coff && second->out_cstride % 4 == 0 && second->out_coff % 4 == 0 &&
                    second->out2_cstride >= 48 - second->split && second->out2_cstride % 4 == 0 && (reinterpret_cast<uintptr_t>(second->out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(second->out2) & 15) == 0,
                "v2x_conv2d_pair (tail form): bad output views (16-byte stores)");
    if (first->N == 0) return V2X_OK;
    TailArgs a;
    a.in = first->in0;
    a.in_cstride = 32;
    a.in_coff = 0;
    a.N = first->N;
    a.H = first->H;
    a.W = first->W;
    a.wA = first->weight;
    a.scA = first->scale;
    a.shA = first->shift;
    a.reluA = first->relu;
    a.wB = second->weight;
    a.scB = second->scale;
    a.shB = second->shift;
    a.reluB = second->relu;
    a.w2 = second->weight2;
    a.sc2 = second->scale2;
    a.sh2 = second->shift2;
    a.relu2 = second->relu2;
    a.split = second->split;
    a.out = reinterpret_cast<float *>(second->out);
    a.out2 = reinterpret_cast<float *>(second->out2);
    a.out_cstride = second->out_cstride;
    a.out_coff = second->out_coff;
    a.out2_cstride = second->out2_cstride;
    a.tiles_x = a.W / tail::TW;
    a.tiles_y = a.H / tail::TH;
    a.n_tiles = a.N * a.tiles_x * a.tiles_y;
    a.xcd_walk = v2x_tune(V2X_TUNE_HALO_XCD);
    return tail_launch(a, stream);
}
