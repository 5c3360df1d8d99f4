// The detection tail as ONE kernel: conv8_2 (32 -> 32, the decoder's last layer) followed by the fused detection heads (3x3 32 -> 64 hidden,
// block-diagonal 1x1 64 -> 12 + 36, fp32 logits) of upstream coperception/models/det/backbone/Backbone.py::LidarDecoder and
// models/det/base/DetModelBase.py::ClassificationHead / SingleRegressionHead (code absent from /root/reference, see include/v2x_amd.h).
//
// Why: both launches are HBM-bound (conv8_2 reads 1.5 GB and writes 1.34 GB per 320 maps, the heads read those 1.34 GB back 0.5 ms later and
// write 4.03 GB of logits: 530 + 1 100 us).  Here conv8_2's output never leaves the CU: per 8 x 32 output tile
//   * the 12 x 36-pixel window of conv8_1's output is moved into LDS by LDS-DMA (pixel-major, swizzled: conv_halo.hip's patch layout);
//   * STAGE A evaluates conv8_2 on the 10 x 34 region the heads need -- 20 row-segment fragments of the tile's own columns + 2 fragments for the two
//     halo columns (22 fragments for 340 pixels; row segments share their window reads across the tap rows), scale / shift / ReLU, rounded to bf16 exactly as the stand-alone layer stores it, ZERO where the pixel lies outside the image
//     (= the heads' zero padding) -- into a second LDS patch;
//   * STAGE B is conv_halo.hip's heads form on that patch: tap-column groups, the hidden rows in kappa order so that a lane's accumulators are
//     its B fragment of the 1x1, whose weights live in registers; fp32 split stores (cls | loc).
// 8 waves = two 4-wave groups, each with its OWN tile, window and patch, on ONE resident copy of the three weight sets (54 KiB); the groups run
// one stage apart (raw s_barrier, group 1 starts one barrier late): while one group's waves multiply stage A of their next tile, the other's
// are in stage B and its 12 logit stores per wave -- one workgroup per CU, 152 KiB of LDS.  The next tile's window is requested right after the
// barrier that ends stage A and waited for with a COUNTED vmcnt that leaves the tile's own stores in flight.
// What bounds it (profiles/r05_tail_probe.txt; phase-removal and time-stamp builds of tools/probes/conv_tail_probe.hip, tools/tail_timeline.sh): not HBM (no
// logit stores: -1 %) and not one pipe -- a SIMD's two waves (one per group) issue ~480 VALU, ~130 LDS and 276 MFMA instructions per tile each, and the
// stage times are close to the SUM of those three (stage A ~5 us for 0.8 us of MFMAs).  What helped: fewer instructions -- one output address per tensor and
// pixel with the channel tiles at immediate offsets instead of a `co >= split` select per store, no second ReLU, a window fill without bounds tests for tiles
// away from the image border (lane offsets precomputed), stage A on row segments (wave-uniform border tests, a third fewer pixel reads): 1 555 -> 1 422 us.
// Measured and dropped (same probe file; all bit-identical):
//   * the stages cut into 14 barrier-separated load / MFMA phases with the groups an odd number of phases apart (stream8g's scheme;
//     git show c68e3aa:tools/probes/conv_tail_phases_probe.hip): 1 841 us against 1 577 -- a phase is then one tap column, and 14 barriers per tile add their skew;
//   * a WAVE-PRIVATE form (git show c68e3aa:tools/probes/conv_tail_wave_probe.hip): one wave per SIMD with 512 registers, its own 8 x 16 tile, window and patch, conv8_2's and the
//     1x1's weights in registers, 8 fragments per hidden-layer weight fragment, no workgroup barrier in the loop -- 174 instead of 312 fragment reads per 128
//     pixels, and 1 735 us against 1 511: with one wave per SIMD its ~1 600 VALU instructions per tile (364 of them AGPR -> VGPR copies for the epilogues) run
//     in series with its 552 MFMAs;
//   * the other patch swizzle ((x >> 2) & 3, the faster one in isolation): +3 %;  no scheduling fences inside the stages: +-1 %;  no stage offset between the groups: +2.5 %.
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
    int split;                          // rows < split (4, 8 or 12) -> out, the others -> out2; no ReLU on the logits
    float *out, *out2;
    int out_cstride, out_coff, out2_cstride;
    int tiles_x, tiles_y, n_tiles;
    int xcd_walk;
};

// (the build-time experiment switches of this kernel -- scheduling fences, group offset, phase removal, time stamps -- live in tools/probes/conv_tail_probe.hip)
namespace tail {
constexpr int TH = 8, TW = 32;
constexpr int MH = TH + 2, MW = TW + 2;     // stage-A region = stage B's patch
constexpr int IH = TH + 4, IW = TW + 4;     // input window
constexpr int NMID = MH * MW;               // 340
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
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> 1) & 3); }
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

    // Stage A's fragments.  The 10 x 34 region = BODY (columns 1 .. 32 = the tile's own 32 columns: two 16-pixel segments per row, 20 fragments) +
    // the two halo COLUMNS 0 and 33 (20 pixels: two more fragments).  A body fragment of region row r and tap (ky, kx) reads window row r + ky at
    // columns seg + kx .. + 15 -- the same read for every (r, ky) with the same sum, so a wave that owns consecutive rows reads each window row once
    // per tap column (the linear walk of conv_halo_pair.hip read one fragment per MFMA pair: 54 instead of 30-33 pixel reads per wave and tile).
    // Waves 0, 1 own three body rows (6 fragments), waves 2, 3 two body rows and one halo-column fragment (5): rows 3w.. resp. 6 + 2 (w - 2)..
    const int r0 = wave < 2 ? 3 * wave : 6 + 2 * (wave - 2);                       // wave-uniform
    int body_off[2][3];                                                             // lane offset of (segment, kx) inside a window row, swizzled
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int col = 1 + 16 * sg + fj + kx;
            body_off[sg][kx] = (col * 4 + swz4(fq, col)) * 16;
        }
    // halo fragment: wave 2 -- lanes 0 .. 9 = column 0 rows 0 .. 9, lanes 10 .. 15 = column 33 rows 0 .. 5;  wave 3 -- lanes 0 .. 3 = column 33 rows 6 .. 9
    int hr, hc, hvalid;
    {
        const int q = (wave == 3 ? 16 : 0) + fj;                                    // 0 .. 19: column 0 rows 0..9, then column 33 rows 0..9
        const int qc = q < 20 ? q : 19;
        hc = qc < 10 ? 0 : MW - 1;
        hr = qc < 10 ? qc : qc - 10;
        hvalid = (wave >= 2 && q < 20) ? 1 : 0;
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
    // window fill: the lane's element offset from the window's first pixel per piece (tile-invariant: 7 registers); a tile away from the image border
    // (70 % of them at 256 x 256) needs no bounds test -- 3 VALU instructions per piece instead of ~15
    unsigned woff[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int sidx = (wave + 4 * u) * 64 + lane;
        const int pix = sidx >> 2, phys = sidx & 3;
        const int pr = pix / IW, pcx = pix - pr * IW;
        woff[u] = (unsigned)(pr * a.W + pcx) * (unsigned)a.in_cstride + (unsigned)(swz4(phys, pcx) * 8);
    }
    auto load_window = [&](int tile) {   // exactly PPW DMAs per wave (the fourth wave's 7th goes to a dummy page)
        int n, ty, tx;
        tile_coords(tile, n, ty, tx);
        const bool interior = ty > 0 && ty + 1 < a.tiles_y && tx > 0 && tx + 1 < a.tiles_x;   // wave-uniform
        if (interior) {
            const uint16_t *base = a.in + (size_t)((n * a.H + ty * TH - 2) * a.W + tx * TW - 2) * a.in_cstride + a.in_coff;
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int piece = wave + 4 * u;
                char *dst = piece < IN_PIECES ? s_in + piece * 1024 : smem + OFF_DUMMY;   // wave-uniform
                __builtin_amdgcn_global_load_lds((gptr_tl_t)(piece < IN_PIECES ? (const void *)(base + woff[u]) : (const void *)g_zero_page_tail), (lptr_tl_t)dst, 16, 0, 0);
            }
            return;
        }
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
    if (grp == 1) __builtin_amdgcn_s_barrier();   // one stage behind group 0

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
            f32x4_t acc[6][2];   // [2 j + segment] for the wave's body rows j = 0, 1 (and 2: waves 0, 1);  waves 2, 3: [4] = the halo-column fragment
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const char *wrow = s_in + r0 * (IW * 64);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {   // K order of the stand-alone layer: tap column outer, tap row inner
                // X: waves 0, 1 -- the fifth window row's two segments (their third body row's last tap row);  waves 2, 3 -- the halo fragment's three tap rows
                bf16x8_t A[3][2], Bb[4][2], X[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < 2; ++i) A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_wA + (((ky * 3 + kx) * 4 + fq) * 32 + i * 16 + fj) * 16);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) Bb[rr][sg] = *reinterpret_cast<const bf16x8_t *>(wrow + rr * (IW * 64) + body_off[sg][kx]);
                if (wave < 2) {
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) X[sg] = *reinterpret_cast<const bf16x8_t *>(wrow + 4 * (IW * 64) + body_off[sg][kx]);
                    X[2] = X[1];
                } else {
                    const int col = hc + kx;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) X[ky] = *reinterpret_cast<const bf16x8_t *>(s_in + (((hr + ky) * IW + col) * 4 + swz4(fq, col)) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                acc[2 * j + sg][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], Bb[j + ky][sg], acc[2 * j + sg][i], 0, 0, 0);
                    if (wave < 2) {
#pragma unroll
                        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                acc[4 + sg][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], ky == 2 ? X[sg] : Bb[2 + ky][sg], acc[4 + sg][i], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[4][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], X[ky], acc[4][i], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float4 scA[2], shA[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                scA[i] = lds_ld4((lds_cf_t *)s_tab + i * 16 + fq * 4);
                shA[i] = lds_ld4((lds_cf_t *)s_tab + 32 + i * 16 + fq * 4);
            }
            auto emit = [&](const f32x4_t (&v)[2], int r, int c, bool keep, bool inside) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint2 o;
                    o.x = v2x_relu_bf16x2_floor(pack_bf16x2(v[i][0] * scA[i].x + shA[i].x, v[i][1] * scA[i].y + shA[i].y), floorA);
                    o.y = v2x_relu_bf16x2_floor(pack_bf16x2(v[i][2] * scA[i].z + shA[i].z, v[i][3] * scA[i].w + shA[i].w), floorA);
                    o.x = inside ? o.x : 0u;
                    o.y = inside ? o.y : 0u;
                    if (keep) *(lds_u2_t *)(s_mid + ((r * MW + c) * 4 + swz4(i * 2 + (fq >> 1), c)) * 16 + (fq & 1) * 8) = (u32x2_tl_t){o.x, o.y};
                }
            };
            // body fragments: their columns are the tile's own (always inside the image); a row is outside only for the first / last tile row (wave-uniform)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j == 2 && wave >= 2) break;
                const int r = r0 + j;
                const int y = ty * TH - 1 + r;
                const bool inside = (unsigned)y < (unsigned)a.H;
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) emit(acc[2 * j + sg], r, 1 + 16 * sg + fj, true, inside);
            }
            if (wave >= 2) {
                const int y = ty * TH - 1 + hr, x = tx * TW - 1 + hc;
                emit(acc[4], hr, hc, hvalid != 0, (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W);
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
                    __builtin_amdgcn_sched_barrier(0);
                    if (ky < 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            A[(ky + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(s_wB + ((((ky + 1) * 3 + kx) * 4 + fq) * 64 + i * 16 + fj) * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky & 1][i], B[((f >> 1) + ky) * 2 + (f & 1)], acc[i][f], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
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
                const uint32_t pix = (uint32_t)((n * a.H + y) * a.W + x);
                // channel tile 0 straddles the split (its lanes with 4 fq >= split belong to out2), tiles 1 and 2 lie behind it: ONE address per output
                // tensor and pixel, the tiles at immediate offsets (the general `co >= split` select per store cost 8 VALU instructions each)
                float *p1 = a.out + (size_t)pix * a.out_cstride + a.out_coff + fq * 4;
                float *p2 = a.out2 + (size_t)pix * a.out2_cstride + fq * 4 - a.split;
                float *p0 = (fq * 4 >= a.split) ? p2 : p1;
#pragma unroll
                for (int i2 = 0; i2 < 3; ++i2) {
                    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2][s], hb[s], d, 0, 0, 0);
                    const float4 s2 = s2v[i2], t2 = t2v[i2];
                    const float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;   // (no second ReLU: the dispatch requires relu2 == 0)
                    *reinterpret_cast<float4 *>(i2 == 0 ? p0 : p2 + i2 * 16) = make_float4(v0, v1, v2, v3);
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
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's offset barrier
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
                    second->split % 4 == 0 && second->split <= 12 && second->relu2 == 0,
                "v2x_conv2d_pair (tail form): the second layer must be the fused heads (3x3 32 -> 64 chained with a 1x1 -> 48 without ReLU, fp32 outputs split at 4, 8 or 12 channels)");
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
