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

// a phase ends: the fragments it read are in registers (the windows / patches may be rewritten behind the barrier), nothing moves across the barrier
#define TAIL_PHASE_END()                                   \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
#ifndef V2X_TAIL_PHASE_OFFSET_BUILD
#define V2X_TAIL_PHASE_OFFSET_BUILD 7
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
constexpr int OFF_W2 = OFF_TAB + 288 * 4;   // the 1x1's weights [48][64] bf16, row-major as the caller holds them (6 KiB)
constexpr int SMEM = OFF_W2 + 48 * 64 * 2;     // (12 fp32 dwordx4 stores per wave and tile: 3 channel tiles x 4 fragments -- the counted wait below)
constexpr int PHASE_OFFSET = V2X_TAIL_PHASE_OFFSET_BUILD;   // group 1 runs this many phases behind group 0 (odd: L opposite M; 7 = half a tile: the groups' store bursts alternate)
static_assert(IN_BYTES % 1024 == 0 && SMEM <= 160 * 1024 && (PHASE_OFFSET & 1) == 1, "LDS map; odd phase offset");
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> 1) & 3); }
}  // namespace tail

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_tail_kernel(const TailArgs a) {
    using namespace tail;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int fj = lane & 15;
    char *s_wA = smem, *s_wB = smem + WA_BYTES;
    char *s_in = smem + OFF_GRP + grp * GRP_BYTES, *s_mid = s_in + IN_BYTES;

    // the three weight sets by LDS-DMA, resident for the whole kernel
    for (int off = wave8 * 1024; off < WA_BYTES + WB_BYTES; off += 8192) {
        const char *src = off < WA_BYTES ? reinterpret_cast<const char *>(a.wA) + off : reinterpret_cast<const char *>(a.wB) + (off - WA_BYTES);
        __builtin_amdgcn_global_load_lds((gptr_tl_t)(src + lane * 16), (lptr_tl_t)(smem + off), 16, 0, 0);
    }
    for (int off = wave8 * 1024; off < 48 * 64 * 2; off += 8192)
        __builtin_amdgcn_global_load_lds((gptr_tl_t)(reinterpret_cast<const char *>(a.w2) + off + lane * 16), (lptr_tl_t)(smem + OFF_W2 + off), 16, 0, 0);
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
    // tile-invariant lane geometry of stage A: the lane's pixel of each of its fragments
    int rc[FPW];   // region row | column << 8 | valid << 16 | window pixel index (r IW + c) << 20
#pragma unroll
    for (int t = 0; t < FPW; ++t) {
        const int p = (wave + 4 * t) * 16 + fj;
        const int pc = p < NMID ? p : NMID - 1;
        const int r = pc / MW, c = pc - r * MW;
        rc[t] = r | (c << 8) | ((p < NMID ? 1 : 0) << 16) | ((r * IW + c) << 20);
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
    if (grp == 1)
        for (int i = 0; i < PHASE_OFFSET; ++i) __builtin_amdgcn_s_barrier();   // an odd number of phases behind group 0: its load phases meet the other's MFMA phases

    // ---- the phases of a tile (every one ends with a workgroup barrier; L = LDS / VALU work, M = MFMAs from registers) ----------------------------------
    //   0 L  the PREVIOUS tile's stage-B epilogue (hidden -> 1x1 -> 12 logit stores), then the fragments of stage-A tap column 0
    //   1 M  its 36 MFMAs          2 L / 3 M  tap column 1          4 L / 5 M  tap column 2
    //   6 L  the next window's DMAs (every wave of the group is past its last window read), stage A's epilogue -> patch
    //   7 M  (nothing)             8 L / 9 M, 10 L / 11 M, 12 L / 13 M  stage B's three tap columns, 48 MFMAs each; phase 13 ends with the wait for the window
    // One extra pass of phase 0 after the last tile writes its logits.
    f32x4_t accB[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) accB[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bool have_prev = false;
    int pn = 0, pty = 0, ptx = 0;
    for (;; pair_i += walk.step) {
        const bool live = pair_i < walk.end;
        const int tile = 2 * pair_i + grp;
        const bool has = live && tile < a.n_tiles;
        const int next_pair = pair_i + walk.step;
        const int next = 2 * next_pair + grp;
        const bool has_next = live && next_pair < walk.end && next < a.n_tiles;
        int n = 0, ty = 0, tx = 0;
        tile_coords(has ? tile : 0, n, ty, tx);
        // lane coordinates recomputed per tile (opaque to the compiler): with the 14 phases unrolled it otherwise hoists >100 registers of tile-invariant
        // addresses out of this loop and spills them -- and scratch reloads are vector-memory operations (in-order vmcnt behind the logit stores)
        const int ln_t = fresh_lane();
        const int fj = ln_t & 15, fq = ln_t >> 4;

        // ================= phase 0: stage B's epilogue of the previous tile =================
        {   // (straight-line even without a previous tile: only the stores are predicated -- conditional regions around the phases cost the allocator
            //  ~140 registers of copies, i.e. spills)
            float4 scB[4], shB[4], s2v[3], t2v[3];
            bf16x8_t w2f[3][2];
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
                for (int s = 0; s < 2; ++s) w2f[i2][s] = *reinterpret_cast<const bf16x8_t *>(smem + OFF_W2 + ((i2 * 16 + fj) * 64 + s * 32 + fq * 8) * 2);
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
                        h[half * 4 + 0] = accB[i][f][0] * scB[i].x + shB[i].x;
                        h[half * 4 + 1] = accB[i][f][1] * scB[i].y + shB[i].y;
                        h[half * 4 + 2] = accB[i][f][2] * scB[i].z + shB[i].z;
                        h[half * 4 + 3] = accB[i][f][3] * scB[i].w + shB[i].w;
                    }
                    uint4 p;
                    p.x = v2x_relu_bf16x2_floor(pack_bf16x2(h[0], h[1]), floorB);
                    p.y = v2x_relu_bf16x2_floor(pack_bf16x2(h[2], h[3]), floorB);
                    p.z = v2x_relu_bf16x2_floor(pack_bf16x2(h[4], h[5]), floorB);
                    p.w = v2x_relu_bf16x2_floor(pack_bf16x2(h[6], h[7]), floorB);
                    hb[s] = __builtin_bit_cast(bf16x8_t, p);
                }
                const int y = pty * TH + 2 * wave + (f >> 1), x = ptx * TW + (f & 1) * 16 + fj;
                const size_t pix = (size_t)(pn * a.H + y) * a.W + x;
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
                    if (have_prev) *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                }
            }
        }
        if (!live) break;
        have_prev = has;
        pn = n;
        pty = ty;
        ptx = tx;

        // ================= phases 0 .. 5: stage A, conv8_2 on the 10 x 34 region, one tap column per L / M pair =================
        f32x4_t accA[FPW][2];
#pragma unroll
        for (int t = 0; t < FPW; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) accA[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8_t A[3][2], B[3][FPW];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int i = 0; i < 2; ++i) A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_wA + (((ky * 3 + kx) * 4 + fq) * 32 + i * 16 + fj) * 16);
#pragma unroll
                for (int t = 0; t < FPW; ++t) {
                    const int col = ((rc[t] >> 8) & 0xff) + kx;
                    B[ky][t] = *reinterpret_cast<const bf16x8_t *>(s_in + (((rc[t] >> 20) + kx) * 4 + swz4(fq, col)) * 16 + ky * (IW * 64));
                }
            }
            TAIL_PHASE_END();   // L
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < FPW; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) accA[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], B[ky][t], accA[t][i], 0, 0, 0);
            TAIL_PHASE_END();   // M
        }

        // ================= phase 6: the next window's DMAs; stage A's epilogue -> the patch =================
        if (has_next) load_window(next);
        {
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
                    o.x = v2x_relu_bf16x2_floor(pack_bf16x2(accA[t][i][0] * scA[i].x + shA[i].x, accA[t][i][1] * scA[i].y + shA[i].y), floorA);
                    o.y = v2x_relu_bf16x2_floor(pack_bf16x2(accA[t][i][2] * scA[i].z + shA[i].z, accA[t][i][3] * scA[i].w + shA[i].w), floorA);
                    o.x = inside ? o.x : 0u;
                    o.y = inside ? o.y : 0u;
                    *(lds_u2_t *)(s_mid + ((r * MW + c) * 4 + swz4(i * 2 + (fq >> 1), c)) * 16 + (fq & 1) * 8) = (u32x2_tl_t){o.x, o.y};
                }
            }
        }
        TAIL_PHASE_END();   // L: the group's patch is complete
        TAIL_PHASE_END();   // M: (nothing -- keeps this group's load phases opposite the other group's MFMA phases)

        // ================= phases 8 .. 13: stage B, the heads' hidden 3x3 32 -> 64 =================
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) accB[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8_t A[3][4], B[8];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int pr = 2 * wave + rr, pc = ch * 16 + fj + kx;
                    B[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(s_mid + ((pr * MW + pc) * 4 + swz4(fq, pc)) * 16);
                }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < 4; ++i) A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_wB + (((ky * 3 + kx) * 4 + fq) * 64 + i * 16 + fj) * 16);
            TAIL_PHASE_END();   // L
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        accB[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], B[((f >> 1) + ky) * 2 + (f & 1)], accB[i][f], 0, 0, 0);
            if (kx == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next window (requested 7 phases ago) and the previous tile's stores (13 phases ago)
            TAIL_PHASE_END();   // M
        }
    }
    if (grp == 0)
        for (int i = 0; i < PHASE_OFFSET; ++i) __builtin_amdgcn_s_barrier();   // balance group 1's offset
}

int v2x_num_cus();   // conv_stream.hip

static int tail_launch(const TailArgs &a, hipStream_t s) {
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tail::SMEM);
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
