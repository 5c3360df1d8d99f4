// Parity-class form of the decoder's STREAMED upsample -> concat -> 3x3 layers: conv5_1 (512 half-res + 256 -> 256 @32x32) and conv6_1
// (256 half-res + 128 -> 128 @64x64) of upstream Backbone.py::LidarDecoder (code absent from /root/reference, see include/v2x_amd.h).
//
// conv_halo.hip (conv3x3_halo_ppc_kernel) explains the identity: on a x2 NEAREST-upsampled operand the 3x3 is, per output parity class (py, px),
// a 2x2-tap convolution on the half-resolution map with pre-summed weights -- 4 C0 + 9 C1 instead of 9 (C0 + C1) multiply-adds per output,
// -37 % for these two layers.  There the weights are resident; here they stream (5.4 / 1.4 MB per layer), and the class decomposition multiplies
// the weight bytes of the upsampled half by 16 / 9 while its MFMAs shrink by 4 / 9: the kernel is built around that ratio.
//
//   * 512 threads, one workgroup per CU, a 16 x 32-pixel tile x 128 output channels (conv3x3_stream8g_kernel's tile).  WAVE = (half g of the tile,
//     CLASS c): wave (g, c) owns the 4 x 16 pixels (8 g + 2 Y + py, 2 X + px) and ALL 128 channels -- 8 channel tiles x 4 pixel fragments = 128
//     accumulator registers, a weight fragment in front of 4 MFMAs, a pixel fragment in front of 8.
//   * UP PHASE (C0 / 32 chunks): a step is ONE class tap (a, b) for all four classes -- a 32-KiB slot [class][4 k-slots][128][8] of which a
//     wave reads only its class's quarter -- 32 MFMAs per wave from 8 + 4 fragment reads.  Three slots; the four waves of group 1 stream the slot
//     of step s + 2 (8 one-KiB LDS-DMA pieces each) in their load phase, group 0 the next chunk's 11.5-KiB half-resolution patch.
//   * SKIP PHASE (C1 / 32 chunks): conv3x3_stream8g_kernel's step -- a tap column kx, three taps, 96 MFMAs from 24 + 9 reads -- on a patch stored
//     as two COLUMN-PARITY planes (the DMA descriptors permute the pixels on their way in), so that the class's stride-2 pixels are again 16
//     consecutive entries of a plane row and the swizzle of the dense form applies.
//   * the two wave groups run half a step apart (one computes while the other loads: stream8g's ping-pong); the LDS holds either phase's
//     ring + patches in the same 151 KiB.  Each phase has its own prologue; the NEXT tile's up prologue is issued BEFORE this tile's output
//     stores (vmcnt is in order: the wait then covers the DMAs and leaves the stores in flight).
// What bounds it (profiles/r05_stream8p_probe.txt, phase-removal builds of an instrumented copy, retired in round 6: git show c68e3aa:tools/probes/conv_stream_pc_probe.hip): the up phase's MFMAs are hidden
// entirely (removing them: -1 %); its time is the 32 + 3 one-KiB LDS-DMA pieces per step -- 0.9 us per step = ~22 B/clk per CU, the LDS-DMA rate
// these kernels see everywhere -- so the class decomposition's 4x weight bytes per MFMA, not its MFMA count, sets the layer's time: -19 % against
// the 9-tap kernel where the MFMA count alone says -37 %.  Variants measured and dropped: four up slots (above), the weight pieces split over
// both groups (no change), a drain-free up -> skip transition with per-phase waits in group 0 (-8 % instead of -19 %: group 0 then stalls on its
// own patch piece every step), a LOCKSTEP up phase (one barrier per step, all eight waves issuing the DMAs of step s + 2, reading, multiplying and then
// waiting under their MFMAs: -13 % instead of -19 % -- the two groups' alternation is worth more than the second barrier costs).
// K order: (up chunk, class tap), then (skip chunk, kx, ky) -- results differ from the 9-tap kernels in the weights (pre-summed, rounded once)
// and in fp32 summation order; tests/test_gpu_parity_class.py holds this kernel to the unmodified fp32 9-tap oracle layer at the 9-tap
// kernel's tolerance, and to a torch evaluation of the same bf16 operands.
// Weight layout (w_layout 4, v2x_pack_conv / packing.pack_conv_stream_parity): per 128-row channel tile
//   [C0 / 32 chunks][4 class taps 2 a + b][4 classes 2 py + px][4 k-slots][128 rows][8]   then   [C1 / 32 chunks][kx][ky][4 k-slots][128 rows][8],
// followed (after the last tile) by 64 B of zeros, the zero page.
#include "conv_stream.h"

namespace pcs {
constexpr int TH = 16, TW = 32, BCO = 128, TCO = 8, HCO = 4;
constexpr int PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;      // half-resolution patch 10 x 18
constexpr int PHF = TH + 2, PWH = (TW + 2) / 2;        // full-resolution patch as two column-parity planes of 18 x 17
constexpr int SLICE = BCO * 64;                        // one tap of one 32-channel chunk: 8 KiB
constexpr int USTEP = 4 * SLICE, SSTEP = 3 * SLICE;    // bytes a step streams
constexpr int UP_PIECES = (PH0 * PW0 * 4 + 63) / 64;   // 12 one-KiB pieces
constexpr int SP_PIECES = (2 * PHF * PWH * 4 + 63) / 64;   // 39
#ifndef V2X_PCS_USLOTS_BUILD
#define V2X_PCS_USLOTS_BUILD 3
#endif
// up-phase ring: NU slots, the image of step s + NU - 1 streams in while step s computes.  Measured (profiles/r05_stream8p_probe.txt):
// four slots are no faster than three (2-4 % slower): the up phase is bound by the CU's LDS-DMA throughput (35 KiB per 32-MFMA step at ~22 B/clk),
// not by the time a piece has to land.
constexpr int NU = V2X_PCS_USLOTS_BUILD;
constexpr int OFF_P0 = 3 * SSTEP, OFF_P1 = OFF_P0 + SP_PIECES * 1024;
constexpr int OFF_UPA = NU * USTEP, OFF_UPB = OFF_UPA + UP_PIECES * 1024;
constexpr int OFF_SS = (OFF_UPB + UP_PIECES * 1024 > OFF_P1 + SP_PIECES * 1024) ? OFF_UPB + UP_PIECES * 1024 : OFF_P1 + SP_PIECES * 1024;
constexpr int SMEM = OFF_SS + 2 * BCO * 4;
static_assert(SMEM <= 160 * 1024 && (NU == 3 || NU == 4), "LDS map");
#define PCS_SWZ(pc) (((pc) >> 1) & 3)

}  // namespace pcs

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_stream8p_kernel(const StreamArgs a) {
    using namespace pcs;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, wv = wave & 3;
    const int py = wv >> 1, px = wv & 1;
    const int fj = lane & 15, fq = lane >> 4;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    const int co_tile = bid % a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int n_up = a.C0 >> 5, n_sk = a.C1 >> 5;
    const int S_up = 4 * n_up, S_sk = 3 * n_sk;
    const size_t tile_bytes = (size_t)S_up * USTEP + (size_t)S_sk * SSTEP;
    const char *w_up = reinterpret_cast<const char *>(a.w) + (size_t)co_tile * tile_bytes;
    const char *w_sk = w_up + (size_t)S_up * USTEP;
    const void *zero_page = reinterpret_cast<const char *>(a.w) + (size_t)a.n_co_tiles * tile_bytes;
    const int Hs = a.H >> 1, Ws = a.W >> 1;

    auto tile_coords = [&](int t, int &n, int &y0, int &x0) {
        const int px_tile = t / a.n_co_tiles;
        n = px_tile / txy;
        const int trem = px_tile - n * txy;
        const int ty = trem / a.tiles_x;
        y0 = ty * TH;
        x0 = (trem - ty * a.tiles_x) * TW;
    };
    // lane id from volatile asm: what is derived from it is recomputed where it is used instead of living in registers across the MFMA phases
    auto fresh_lane = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // ---- LDS-DMA issue helpers (one instruction = one 1-KiB piece) ------------------------------------------------------------------------
    auto issue_up_patch_piece = [&](int n, int y0, int x0, int kc, int piece, int buf_off) {   // half-resolution patch of up chunk kc
        const int L = piece * 64 + fresh_lane();
        const int pix = L >> 2, phys = L & 3;
        const int pr = pix / PW0, pc = pix - pr * PW0;
        const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
        const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const unsigned off = (unsigned)((n * Hs + y) * Ws + x) * (unsigned)a.C0 + (unsigned)(kc * 32 + ((phys ^ PCS_SWZ(pc)) << 3));
        glds16s(ok ? (const void *)(a.in0 + off) : zero_page, smem + buf_off + piece * 1024);
    };
    auto issue_sk_patch_piece = [&](int n, int y0, int x0, int kc, int piece, int buf_off) {   // full-resolution patch of skip chunk kc, two parity planes
        const int L = piece * 64 + fresh_lane();
        const int pix = L >> 2, phys = L & 3;
        const int q = pix / (PHF * PWH), rem = pix - q * (PHF * PWH);
        const int pr = rem / PWH, cc = rem - pr * PWH;
        const int y = y0 - 1 + pr, x = x0 - 1 + 2 * cc + q;
        const bool ok = pix < 2 * PHF * PWH && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const unsigned off = (unsigned)((n * a.H + y) * a.W + x) * (unsigned)a.C1 + (unsigned)(kc * 32 + ((phys ^ PCS_SWZ(cc)) << 3));
        glds16s(ok ? (const void *)(a.in1 + off) : zero_page, smem + buf_off + piece * 1024);
    };
    auto issue_up_weights = [&](int st, int first, int count) {   // pieces [first, first + count) of up step st's 32-KiB slot image
        const int lane_w = fresh_lane();
        const char *src = w_up + (size_t)st * USTEP + lane_w * 16;
        char *dst = smem + (st % NU) * USTEP;
#pragma unroll
        for (int u = 0; u < 12; ++u)
            if (u < count) glds16s(src + (first + u) * 1024, dst + (first + u) * 1024);
    };
    auto issue_sk_weights = [&](int st, int first, int count) {   // pieces of skip step st's 24-KiB slot image
        const int lane_w = fresh_lane();
        const char *src = w_sk + (size_t)st * SSTEP + lane_w * 16;
        char *dst = smem + (st % 3) * SSTEP;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (u < count) glds16s(src + (first + u) * 1024, dst + (first + u) * 1024);
    };
    auto up_buf = [&](int kc) { return ((n_up - 1 - kc) & 1) ? OFF_UPB : OFF_UPA; };   // the last up chunk sits in UPA
    auto sk_buf = [&](int kc) { return (kc & 1) ? OFF_P0 : OFF_P1; };                  // the first skip chunk in P1
    auto up_prologue = [&](int n, int y0, int x0) {   // patch of up chunk 0 (group 0), slot images of up steps 0 and 1 (all eight waves)
        if (grp == 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u) issue_up_patch_piece(n, y0, x0, 0, wv + 4 * u, up_buf(0));
        }
        if constexpr (NU == 3) {
            issue_up_weights(wave >> 2, (wave & 3) * 8, 8);   // 64 pieces: waves 0-3 step 0, waves 4-7 step 1
        } else {                                              // 96 pieces, 12 per wave: piece index wave * 12 + u -> step (index / 32)
            issue_up_weights((wave * 12) / 32, (wave * 12) % 32, ((wave * 12) % 32 + 12 <= 32) ? 12 : 32 - (wave * 12) % 32);
            if ((wave * 12) % 32 + 12 > 32) issue_up_weights((wave * 12) / 32 + 1, 0, (wave * 12) % 32 + 12 - 32);
        }
    };

    // epilogue parameters of this workgroup's channel tile
    float *s_ss = reinterpret_cast<float *>(smem + OFF_SS);
    for (int i = tid; i < BCO; i += 512) {
        const int co = co_tile * BCO + i;
        s_ss[i] = co < a.Cout ? a.scale[co] : 0.f;
        s_ss[BCO + i] = co < a.Cout ? a.shift[co] : 0.f;
    }
    const uint32_t floor_bits = a.relu ? 0u : 0x80008000u;

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    up_prologue(n, y0, x0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < n_tiles;
        f32x4_t acc[TCO][4];
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t)(0.f);

        // =============================================== UP PHASE ================================================
        __builtin_amdgcn_s_barrier();                     // the prologue's pieces of every wave have landed
        if (grp == 1) __builtin_amdgcn_s_barrier();       // half-step offset
        for (int kc = 0; kc < n_up; ++kc) {
            const char *pb = smem + up_buf(kc);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int st = kc * 4 + t;
                const int ta = t >> 1, tb = t & 1;
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                // ---- L: group 1 streams the slot of step st + 2, group 0 the next chunk's patch; the step's 12 fragments
                [[maybe_unused]] int nw = 0;
                if (grp == 1) {
                    if (st + NU - 1 < S_up) {
                        issue_up_weights(st + NU - 1, wv * 8, 8);
                        nw = 8;
                    }
                } else if (t < 3 && kc + 1 < n_up) {
                    issue_up_patch_piece(n, y0, x0, kc + 1, wv + 4 * t, up_buf(kc + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t B[4], A[TCO];
                {
                    const int pc = fjl + tb + px;
                    const char *p = pb + ((4 * grp + py + ta) * PW0 * 4 + pc * 4 + (fql ^ PCS_SWZ(pc))) * 16;
#pragma unroll
                    for (int y = 0; y < 4; ++y) B[y] = *reinterpret_cast<const bf16x8_t *>(p + y * (PW0 * 64));
                    const char *ws = smem + (st % NU) * USTEP + wv * SLICE + (fql * BCO + fjl) * 16;
#pragma unroll
                    for (int i = 0; i < TCO; ++i) A[i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
                if (grp == 1) {
                    int keep = 0;
                    for (int d = 2; d <= NU - 1; ++d) keep += (st + d < S_up) ? 8 : 0;
                    const int stores = (st < NU - 2) ? (a.x4 ? 16 : 32) : 0;
                    switch (keep + stores) {
                        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                        case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                        case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
                } else if (t == 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next chunk's patch
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[i][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[y], acc[i][y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();       // realign: group 1's last MFMA phase

        // =============================================== SKIP PHASE ==============================================
        // prologue: patch of skip chunk 0 (group 0), slot images of skip steps 0 and 1 (all eight waves: 48 pieces)
        if (grp == 0) {
#pragma unroll
            for (int u = 0; u < 10; ++u)
                if (wv + 4 * u < SP_PIECES) issue_sk_patch_piece(n, y0, x0, 0, wv + 4 * u, sk_buf(0));
        }
        issue_sk_weights(wave >> 2, (wave & 3) * 6, 6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int kc = 0; kc < n_sk; ++kc) {
            const char *pb = smem + sk_buf(kc);
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {
                const int st = kc * 3 + kx;
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                int nw = 0;
                if (grp == 1) {
                    if (st + 2 < S_sk) {
                        issue_sk_weights(st + 2, wv * 6, 6);
                        nw = 6;
                    }
                } else if (kx < 2 && kc + 1 < n_sk) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int piece = wv + 4 * (kx * 5 + u);
                        if (piece < SP_PIECES) issue_sk_patch_piece(n, y0, x0, kc + 1, piece, sk_buf(kc + 1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t B[9], A[2][HCO];
                const char *ws = smem + (st % 3) * SSTEP + (fql * BCO + fjl) * 16;
                {
                    const int q = (px + kx) & 1, cc = fjl + ((px + kx) >> 1);
                    const char *p = pb + (((q * PHF + 8 * grp + py) * PWH + cc) * 4 + (fql ^ PCS_SWZ(cc))) * 16;
#pragma unroll
                    for (int r = 0; r < 9; ++r) B[r] = *reinterpret_cast<const bf16x8_t *>(p + r * (PWH * 64));
#pragma unroll
                    for (int i = 0; i < HCO; ++i) A[0][i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (grp == 1) {
                    if (nw) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (kx == 2) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M: six half taps; the next half's weight fragments are read behind the first channel tile's MFMAs
#pragma unroll
                for (int h = 0; h < 6; ++h) {
                    const int ky = h >> 1, hh = h & 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[hh * HCO][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[h & 1][0], B[2 * y + ky], acc[hh * HCO][y], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (h + 1 < 6) {
                        const int ky1 = (h + 1) >> 1, hh1 = (h + 1) & 1;
#pragma unroll
                        for (int i = 0; i < HCO; ++i) A[(h + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(ws + ky1 * SLICE + (hh1 * HCO + i) * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 1; i < HCO; ++i)
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                            acc[hh * HCO + i][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[h & 1][i], B[2 * y + ky], acc[hh * HCO + i][y], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();       // realign

        // ---- the NEXT tile's up prologue first (every wave is past its last read of the ring and the patches), then this tile's stores
        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) {
            tile_coords(next, nn, ny0, nx0);
            up_prologue(nn, ny0, nx0);
        }
        {
            lds_cf_t *lss = (lds_cf_t *)s_ss;
#pragma unroll
            for (int i = 0; i < TCO; i += 2) {
                float4 sc[2], sf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    sc[h] = lds_ld4(lss + (i + h) * 16 + fq * 4);
                    sf[h] = lds_ld4(lss + BCO + (i + h) * 16 + fq * 4);
                }
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    uint32_t ox[2], oy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][y][0] * sc[h].x + sf[h].x, acc[i + h][y][1] * sc[h].y + sf[h].y), floor_bits);
                        oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][y][2] * sc[h].z + sf[h].z, acc[i + h][y][3] * sc[h].w + sf[h].w), floor_bits);
                    }
                    const size_t pix = (size_t)(n * a.H + y0 + 8 * grp + 2 * y + py) * a.W + x0 + 2 * fj + px;
                    uint16_t *p = reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + co_tile * BCO + i * 16 + fq * 4;
                    if (a.x4) {
                        v2x_store_pair_x4(p, fq, ox[0], oy[0], ox[1], oy[1]);
                    } else {
                        *reinterpret_cast<uint2 *>(p) = make_uint2(ox[0], oy[0]);
                        *reinterpret_cast<uint2 *>(p + 16) = make_uint2(ox[1], oy[1]);
                    }
                }
            }
        }
        if (!has_next) break;
        // the prologue's DMAs are OLDER than the stores: they have landed, the stores may stay in flight (16 dwordx4 or 32 dwordx2 per wave)
        if (a.x4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
    }
}

// ---- 64 output channels (conv7_1: 128 half-res + 64 -> 64 @128x128): TWO 16 x 32 tiles per workgroup on ONE weight stream ------------------------------
// With 64 rows a 16 x 32 tile fills only half of the accumulators and every weight byte would stand in front of half as many MFMAs -- and the parity-class
// up phase is bound by exactly those bytes (above).  Here a workgroup owns a 16 x 64 region: group g is the 16 x 32 tile of column half g, wave (g, c)
// class c of it -- 8 x 16 pixels = 8 fragments x 4 channel tiles, the same 128 accumulators and 32 MFMAs per class tap as the 128-row kernel, from a
// 16-KiB step image that now serves 1 024 pixels.  Each group fills its OWN patches (an up patch per group and chunk, double-buffered; a full-resolution
// parity-plane patch per group, single-buffered: four of them do not fit -- the refill is issued behind the barrier that ends the chunk's last MFMA
// phase and waited for on the spot, with one extra barrier per chunk boundary; the groups run half a step apart, so group 0's wait lies under group 1's
// MFMA phase and only group 1's is exposed).  Skip step: 96 MFMAs in two halves of the wave's rows (9 + 9 pixel fragments through one register set).
namespace pcq {
constexpr int BCO = 64, TCO = 4;
constexpr int SLICE = BCO * 64;                        // 4 KiB
constexpr int USTEP = 4 * SLICE, SSTEP = 3 * SLICE;    // 16 KiB, 12 KiB
constexpr int OFF_UP = 3 * USTEP;                      // up patches [group][chunk parity], 12 KiB each
constexpr int OFF_SP = 3 * SSTEP;                      // skip patches [group], 39 KiB each
constexpr int OFF_SS = OFF_SP + 2 * pcs::SP_PIECES * 1024, SMEM = OFF_SS + 2 * BCO * 4;
static_assert(OFF_UP + 4 * pcs::UP_PIECES * 1024 <= OFF_SS && SMEM <= 160 * 1024, "LDS map");
}  // namespace pcq

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_stream8q_kernel(const StreamArgs a) {
    using namespace pcq;
    using pcs::PW0; using pcs::PH0; using pcs::PHF; using pcs::PWH; using pcs::UP_PIECES; using pcs::SP_PIECES; using pcs::TH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wv = wave & 3;
    const int py = wv >> 1, px = wv & 1;
    const int fj = lane & 15, fq = lane >> 4;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int n_tiles = a.n_px_tiles;                 // 16 x 64 regions; one channel tile
    const int txy = a.tiles_x * a.tiles_y;
    const int n_up = a.C0 >> 5, n_sk = a.C1 >> 5;
    const int S_up = 4 * n_up, S_sk = 3 * n_sk;
    const char *w_up = reinterpret_cast<const char *>(a.w);
    const char *w_sk = w_up + (size_t)S_up * USTEP;
    const void *zero_page = w_sk + (size_t)S_sk * SSTEP;
    const int Hs = a.H >> 1, Ws = a.W >> 1;

    auto tile_coords = [&](int t, int &n, int &y0, int &x0) {   // x0 = this GROUP's tile
        n = t / txy;
        const int trem = t - n * txy;
        const int ty = trem / a.tiles_x;
        y0 = ty * TH;
        x0 = (trem - ty * a.tiles_x) * 64 + 32 * grp;
    };
    auto fresh_lane = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto issue_up_patch_piece = [&](int n, int y0, int x0, int kc, int piece, int buf_off) {
        const int L = piece * 64 + fresh_lane();
        const int pix = L >> 2, phys = L & 3;
        const int pr = pix / PW0, pc = pix - pr * PW0;
        const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
        const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const unsigned off = (unsigned)((n * Hs + y) * Ws + x) * (unsigned)a.C0 + (unsigned)(kc * 32 + ((phys ^ PCS_SWZ(pc)) << 3));
        glds16s(ok ? (const void *)(a.in0 + off) : zero_page, smem + buf_off + piece * 1024);
    };
    auto issue_sk_patch_piece = [&](int n, int y0, int x0, int kc, int piece, int buf_off) {
        const int L = piece * 64 + fresh_lane();
        const int pix = L >> 2, phys = L & 3;
        const int q = pix / (PHF * PWH), rem = pix - q * (PHF * PWH);
        const int pr = rem / PWH, cc = rem - pr * PWH;
        const int y = y0 - 1 + pr, x = x0 - 1 + 2 * cc + q;
        const bool ok = pix < 2 * PHF * PWH && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const unsigned off = (unsigned)((n * a.H + y) * a.W + x) * (unsigned)a.C1 + (unsigned)(kc * 32 + ((phys ^ PCS_SWZ(cc)) << 3));
        glds16s(ok ? (const void *)(a.in1 + off) : zero_page, smem + buf_off + piece * 1024);
    };
    auto issue_up_weights = [&](int st, int first, int count) {
        const int lane_w = fresh_lane();
        const char *src = w_up + (size_t)st * USTEP + lane_w * 16;
        char *dst = smem + (st % 3) * USTEP;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < count) glds16s(src + (first + u) * 1024, dst + (first + u) * 1024);
    };
    auto issue_sk_weights = [&](int st, int first, int count) {
        const int lane_w = fresh_lane();
        const char *src = w_sk + (size_t)st * SSTEP + lane_w * 16;
        char *dst = smem + (st % 3) * SSTEP;
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < count) glds16s(src + (first + u) * 1024, dst + (first + u) * 1024);
    };
    auto up_buf = [&](int kc) { return OFF_UP + (grp * 2 + (kc & 1)) * (UP_PIECES * 1024); };
    const int sk_buf = OFF_SP + grp * (SP_PIECES * 1024);
    auto up_prologue = [&](int n, int y0, int x0) {   // every group: its up patch of chunk 0; all eight waves: the images of up steps 0 and 1 (32 pieces)
#pragma unroll
        for (int u = 0; u < 3; ++u) issue_up_patch_piece(n, y0, x0, 0, wv + 4 * u, up_buf(0));
        issue_up_weights(wave >> 2, (wave & 3) * 4, 4);
    };

    float *s_ss = reinterpret_cast<float *>(smem + OFF_SS);
    for (int i = tid; i < BCO; i += 512) {
        s_ss[i] = i < a.Cout ? a.scale[i] : 0.f;
        s_ss[BCO + i] = i < a.Cout ? a.shift[i] : 0.f;
    }
    const uint32_t floor_bits = a.relu ? 0u : 0x80008000u;

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    up_prologue(n, y0, x0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < n_tiles;
        f32x4_t acc[TCO][8];
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[i][f] = (f32x4_t)(0.f);

        // =============================================== UP PHASE ================================================
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();       // half-step offset
        for (int kc = 0; kc < n_up; ++kc) {
            const char *pb = smem + up_buf(kc);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int st = kc * 4 + t;
                const int ta = t >> 1, tb = t & 1;
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                // ---- L: group 1 streams the image of step st + 2 (4 pieces per wave); every wave one piece of its group's next up patch
                int keep = 0;
                if (grp == 1 && st + 2 < S_up) {
                    issue_up_weights(st + 2, wv * 4, 4);
                    keep = 4;
                }
                if (t < 3 && kc + 1 < n_up) {
                    issue_up_patch_piece(n, y0, x0, kc + 1, wv + 4 * t, up_buf(kc + 1));
                    keep += 1;
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t B[8], A[TCO];
                {
                    const int pc = fjl + tb + px;
                    const char *p = pb + ((py + ta) * PW0 * 4 + pc * 4 + (fql ^ PCS_SWZ(pc))) * 16;
#pragma unroll
                    for (int y = 0; y < 8; ++y) B[y] = *reinterpret_cast<const bf16x8_t *>(p + y * (PW0 * 64));
                    const char *ws = smem + (st % 3) * USTEP + wv * SLICE + (fql * BCO + fjl) * 16;
#pragma unroll
                    for (int i = 0; i < TCO; ++i) A[i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
                // group 1: everything older than this load phase's own DMAs has landed (the image of step st + 1; at t == 3 the whole next patch); in the
                // tile's first step the previous tile's output stores may stay in flight.  Group 0 has only its patch pieces: it waits at t == 3.
                if (grp == 1) {
                    const int nst = st == 0 ? (a.x4 ? 16 : 32) : 0;
                    switch (keep + nst) {
                        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
                        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
                } else if (t == 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int y = 0; y < 8; ++y) acc[i][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[y], acc[i][y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();       // realign

        // =============================================== SKIP PHASE ==============================================
#pragma unroll
        for (int u = 0; u < 10; ++u)
            if (wv + 4 * u < SP_PIECES) issue_sk_patch_piece(n, y0, x0, 0, wv + 4 * u, sk_buf);
        issue_sk_weights(wave >> 2, (wave & 3) * 3, 3);   // the images of skip steps 0 and 1: 24 pieces
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int kc = 0; kc < n_sk; ++kc) {
            const char *pb = smem + sk_buf;
            if (kc > 0) {   // every wave of this group is past its last read of the patch (the barrier that ended the MFMA phase): refill it, wait, tell the group
#pragma unroll
                for (int u = 0; u < 10; ++u)
                    if (wv + 4 * u < SP_PIECES) issue_sk_patch_piece(n, y0, x0, kc, wv + 4 * u, sk_buf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {
                const int st = kc * 3 + kx;
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                int nw = 0;
                if (grp == 1 && st + 2 < S_sk) {
                    issue_sk_weights(st + 2, wv * 3, 3);
                    nw = 3;
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t B[9], A[3][TCO];
                const char *ws = smem + (st % 3) * SSTEP + (fql * BCO + fjl) * 16;
                const int q = (px + kx) & 1, cc = fjl + ((px + kx) >> 1);
                const char *p = pb + (((q * PHF + py) * PWH + cc) * 4 + (fql ^ PCS_SWZ(cc))) * 16;
#pragma unroll
                for (int r = 0; r < 9; ++r) B[r] = *reinterpret_cast<const bf16x8_t *>(p + r * (PWH * 64));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < TCO; ++i) A[ky][i] = *reinterpret_cast<const bf16x8_t *>(ws + ky * SLICE + i * 256);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (grp == 1) {
                    if (nw) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M: rows Y = 0..3 from B[0..8], then rows 4..7 from the plane rows 8..16 read into the same registers
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
#pragma unroll
                        for (int y = 0; y < 4; ++y) acc[i][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], B[2 * y + ky], acc[i][y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 9; ++r) B[r] = *reinterpret_cast<const bf16x8_t *>(p + (8 + r) * (PWH * 64));
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
#pragma unroll
                        for (int y = 0; y < 4; ++y) acc[i][4 + y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky][i], B[2 * y + ky], acc[i][4 + y], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();       // realign

        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) {
            tile_coords(next, nn, ny0, nx0);
            up_prologue(nn, ny0, nx0);
        }
        {
            lds_cf_t *lss = (lds_cf_t *)s_ss;
#pragma unroll
            for (int i = 0; i < TCO; i += 2) {
                float4 sc[2], sf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    sc[h] = lds_ld4(lss + (i + h) * 16 + fq * 4);
                    sf[h] = lds_ld4(lss + BCO + (i + h) * 16 + fq * 4);
                }
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    uint32_t ox[2], oy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][y][0] * sc[h].x + sf[h].x, acc[i + h][y][1] * sc[h].y + sf[h].y), floor_bits);
                        oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][y][2] * sc[h].z + sf[h].z, acc[i + h][y][3] * sc[h].w + sf[h].w), floor_bits);
                    }
                    const size_t pix = (size_t)(n * a.H + y0 + 2 * y + py) * a.W + x0 + 2 * fj + px;
                    uint16_t *p = reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + i * 16 + fq * 4;
                    if (a.x4) {
                        v2x_store_pair_x4(p, fq, ox[0], oy[0], ox[1], oy[1]);
                    } else {
                        *reinterpret_cast<uint2 *>(p) = make_uint2(ox[0], oy[0]);
                        *reinterpret_cast<uint2 *>(p + 16) = make_uint2(ox[1], oy[1]);
                    }
                }
            }
        }
        if (!has_next) break;
        if (a.x4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
    }
}

int v2x_conv_stream_pc_launch(const StreamArgs &a, hipStream_t s) {
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_stream8p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, pcs::SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_stream8q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, pcq::SMEM);
    }
    if (a.Cout == 64) {   // 16 x 64 regions (the dispatcher set tiles_x = W / 64), one channel tile
        int grid = a.n_px_tiles;
        const int g = v2x_num_cus();
        if (g > 0 && g < grid) grid = g;
        hipLaunchKernelGGL(conv3x3_stream8q_kernel, dim3(grid), dim3(512), pcq::SMEM, s, a);
        V2X_CHECK_LAUNCH("conv3x3_stream8q_kernel");
        return V2X_OK;
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    int grid = n_tiles;
    const int g = v2x_num_cus() / a.n_co_tiles * a.n_co_tiles;   // persistent: a workgroup's tiles share one channel tile
    if (g > 0 && g < n_tiles) grid = g;
    hipLaunchKernelGGL(conv3x3_stream8p_kernel, dim3(grid), dim3(512), pcs::SMEM, s, a);
    V2X_CHECK_LAUNCH("conv3x3_stream8p_kernel");
    return V2X_OK;
}
