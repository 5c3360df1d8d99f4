// One-wave-per-SIMD form of the streamed 3x3 kernels ("w1", round 4): conv3_2 / conv5_x / conv6_x and the ConvGRU of upstream Backbone.py /
// V2VNet.py (code absent from /root/reference, see include/v2x_amd.h).
//
// conv3x3_stream8g_kernel (conv_stream.hip) pairs two waves per SIMD: one in its MFMA phase, one in its load phase.  Its MFMA phases run at
// ~70 % of back-to-back issue (16-cycle 16x16x32 MFMAs leave four issue slots each for the weight-fragment reads), MfmaUtil 65 %.  Here a CU
// holds FOUR waves, one per SIMD, 512 registers each:
//   * wave tile = all BCO channels (128, GRU 96) x 128 pixels (4 rows of the 16 x 32 tile) on v_mfma_f32_32x32x16_bf16: BCO / 32 x 4
//     accumulators of 16 registers; a 32-cycle MFMA hides up to ~5 single-issue instructions, and a step (tap column kx of a 32-channel chunk:
//     3 taps x 2 K halves x BCO / 32 x 4 MFMAs = 96 at 128 rows) carries 24 weight + 12 pixel fragment reads and <= 11 LDS-DMA pieces, issued BY
//     THE SAME WAVE between its MFMAs -- there is no co-resident wave whose DMAs the 32x32 MFMAs would starve (what killed the 32x32 form of
//     stream8g in round 3, profiles/r03_m32_rejected.txt).  tools/w1_probe.hip measured this step skeleton at 0.89 of bare MFMA issue.
//   * same LDS image as stream8g: 3-slot weight ring of whole tap columns (72 KiB at 128 rows), two patch buffers of 40 KiB, K order
//     (chunk, kx, ky, K half) = stream8g's: the fp32 sums are the same (32-channel dot products as two 16-channel halves, as the round-3 32x32
//     form, which was bit-identical).  Patch swizzle (pc >> 2) & 3: conflict-free 32-pixel fragments at both resolutions.
//   * ONE s_barrier per step, two thirds in (after tap row 1): before it a wave waits (counted vmcnt) for the DMAs it issued one step ago --
//     the weights of step s+1, the next chunk's patch pieces -- and for its own LDS reads, which include ALL of this step's weight fragments
//     (tap row 2's are read two blocks ahead); after it the pixel fragments and the first weight fragments of step s+1 are read under the
//     MFMAs of tap row 2.  DMAs of step s (weights of step s+2 into the slot last read before
//     barrier s-1; patch pieces of the next chunk into the buffer last read before barrier 3c-2) are issued in the first third.
//   * persistent over tiles like stream8g (ring and patch fill wrap into the next tile); the epilogue is the only part nothing overlaps.
#include "conv_stream.h"
#include <type_traits>

// TIMING EXPERIMENTS ONLY (-DV2X_W1_DBG_BUILD=n through tools/ab_inproc.sh; results are garbage): which of the step's companion work costs the
// MFMA stream how much?  1 = no patch pieces (descriptor stages + DMA), 2 = no weight DMAs, 4 = no counted wait / barrier, 8 = no pixel-fragment reads.
#ifndef V2X_W1_DBG_BUILD
#define V2X_W1_DBG_BUILD 0
#endif
namespace {
constexpr int W1_DBG = V2X_W1_DBG_BUILD;
constexpr int W1_TH = 16, W1_TW = 32, W1_PW = W1_TW + 2, W1_PH = W1_TH + 2, W1_PW0 = W1_TW / 2 + 2, W1_PH0 = W1_TH / 2 + 2;
constexpr int W1_PSH = 2;     // patch swizzle shift

template <int N>
__device__ __forceinline__ void w1_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `keep` (0 .. 11) vector-memory operations [+ EXTRA] are outstanding: the immediate must be a constant, `keep` is wave-uniform
template <int EXTRA>
__device__ __forceinline__ void w1_wait_keep(int keep) {
    switch (keep) {
        case 0: w1_wait_vmcnt<EXTRA + 0>(); break;
        case 1: w1_wait_vmcnt<EXTRA + 1>(); break;
        case 2: w1_wait_vmcnt<EXTRA + 2>(); break;
        case 3: w1_wait_vmcnt<EXTRA + 3>(); break;
        case 4: w1_wait_vmcnt<EXTRA + 4>(); break;
        case 5: w1_wait_vmcnt<EXTRA + 5>(); break;
        case 6: w1_wait_vmcnt<EXTRA + 6>(); break;
        case 7: w1_wait_vmcnt<EXTRA + 7>(); break;
        case 8: w1_wait_vmcnt<EXTRA + 8>(); break;
        case 9: w1_wait_vmcnt<EXTRA + 9>(); break;
        case 10: w1_wait_vmcnt<EXTRA + 10>(); break;
        default: w1_wait_vmcnt<EXTRA + 11>(); break;
    }
}
}  // namespace

// Epilogue: acc[mt][r] = row tile mt (32 channels; GRU: gate mt of the workgroup's 32 hidden channels), pixel row r of the wave; lane (jl = pixel
// column, hl): register i = channel 16 hl + i of the tile (the kernel permutes the MFMA rows so) -> 16 consecutive channels = two 16-byte stores
// per (tile, row).  co0: first output channel of the tile's rows (GRU: first hidden channel), yr: image row of the wave's first pixel row.
template <int MT, int NR, int EPI>
__device__ __forceinline__ void w1_epilogue(const StreamArgs &a, f32x16_t (&acc)[MT][NR], int co0, int n, int yr, int x0, int jl, int hl, lds_cf_t *lss,
                                            int lss_stride) {
    uint16_t *outp = reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + yr) * a.W + x0 + jl) * a.out_cstride + a.out_coff;
    const size_t row_stride = (size_t)a.W * a.out_cstride;
    if constexpr (EPI == SEPI_GRU) {
        static_assert(EPI != SEPI_GRU || MT == 3, "gates r, z, n");
        const int hc = co0 + 16 * hl;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            uint32_t o[8];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                float h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i4 * 4 + k;
                    const float4 bias = reinterpret_cast<const float4 *>(a.scale)[hc + i];
                    const float rg = 1.0f / (1.0f + __expf(-(acc[0][r][i] + bias.x)));
                    const float zg = 1.0f / (1.0f + __expf(-(acc[1][r][i] + bias.y)));
                    const float ng = tanhf(acc[2][r][i] + bias.z + rg * bias.w);
                    h[k] = ng + zg * (0.0f - ng);
                }
                o[i4 * 2] = pack_bf16x2(h[0], h[1]);
                o[i4 * 2 + 1] = pack_bf16x2(h[2], h[3]);
            }
            uint16_t *q = outp + r * row_stride + hc;
            *reinterpret_cast<uint4 *>(q) = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4 *>(q + 8) = make_uint4(o[4], o[5], o[6], o[7]);
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c = mt * 32 + 16 * hl;
            float sc[16], sf[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 s4 = lds_ld4(lss + c + i4 * 4);
                const float4 t4 = lds_ld4(lss + lss_stride + c + i4 * 4);
                sc[i4 * 4] = s4.x, sc[i4 * 4 + 1] = s4.y, sc[i4 * 4 + 2] = s4.z, sc[i4 * 4 + 3] = s4.w;
                sf[i4 * 4] = t4.x, sf[i4 * 4 + 1] = t4.y, sf[i4 * 4 + 2] = t4.z, sf[i4 * 4 + 3] = t4.w;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                uint32_t o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    o[i] = pack_bf16x2(acc[mt][r][2 * i] * sc[2 * i] + sf[2 * i], acc[mt][r][2 * i + 1] * sc[2 * i + 1] + sf[2 * i + 1]);
                    if (a.relu) o[i] = v2x_relu_bf16x2(o[i]);
                }
                uint16_t *q = outp + r * row_stride + co0 + c;
                *reinterpret_cast<uint4 *>(q) = make_uint4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<uint4 *>(q + 8) = make_uint4(o[4], o[5], o[6], o[7]);
            }
        }
    }
}

template <int BCO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_w1_kernel(const StreamArgs a) {
    constexpr int PW = W1_PW, PH = W1_PH, PW0 = W1_PW0, PH0 = W1_PH0, PSH = W1_PSH, TH = W1_TH, TW = W1_TW;
    constexpr int MT = BCO / 32, NR = 4;                   // row tiles of 32 channels (GRU: the three gates), pixel rows of 32 pixels per wave
    constexpr int W_PIECES = BCO / 16;                     // 1-KiB pieces per tap slice
    constexpr int SLICE_BYTES = BCO * 64;
    constexpr int STEP_BYTES = 3 * SLICE_BYTES;            // one tap column of a chunk
    constexpr int NWD = (3 * W_PIECES + 3) / 4;            // weight DMAs per wave and step (6 at 128 rows, <= 5 at 96)
    constexpr int N_ST = (EPI == SEPI_GRU) ? NR * 2 : MT * NR * 2;   // output stores per wave and tile
    constexpr int NBLK = 6, BLK = MT * NR;                 // MFMA blocks per step: (tap row ky, K half kh); MFMAs per block
    static_assert(BCO % 32 == 0 && PH * PW * 4 <= (PATCH8_PIECES - 1) * 64, "row tiles of 32; patch fits its buffer");
    static_assert(N_ST + 11 <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;                                   // 3 x STEP_BYTES
    char *s_patch = smem + 3 * STEP_BYTES;                 // 2 x PATCH8_BYTES
    float *s_ss = reinterpret_cast<float *>(smem + 3 * STEP_BYTES + 2 * PATCH8_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3, one per SIMD
    const int jl = lane & 31, hl = lane >> 5;              // MFMA row / pixel of the lane, K-slot half

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        if (a.xcd_walk != 0 && a.n_co_tiles == 8 && r == 0 && q == 32) {   // the ConvGRU: 8 pixel tiles x 4 channel tiles per XCD and round (conv_stream.hip)
            const int x = bid >> 5, i2 = bid & 31;
            bid = (((x >> 1) * 8 + (i2 >> 2)) << 3) + (x & 1) * 4 + (i2 & 3);
        }
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    const int co_tile = bid % a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    const int S3 = nchunks * 3;                            // steps (tap columns) per tile
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
    // lane id from VOLATILE asm: values derived from it cannot be hoisted out of the step loop (as lane constants they would be -- the patch
    // descriptors of ten pieces kept, and spilled, beside 400 registers of accumulators and operands)
    auto fresh_lane = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // DMA source descriptor of patch piece (wave + 4t), computed where it is issued.  A piece is 16 consecutive pixels of the linear patch
    // (pixel = row * width + column, 4 lanes of 16 B each).  Its first pixel 64 t + 16 wave is wave-uniform and t is a compile-time constant at
    // every call (unrolled loops): row and column of that pixel = constants + a few scalar compares on 16 wave, no division; a lane only adds
    // its pixel offset (at most one row wrap: 16 < width) -- ~15 vector instructions per piece beside the MFMAs.
    auto desc = [&](int n, int y0, int x0, int t, bool hf) -> int {
        const int ln = fresh_lane();
        int pr0, pc0;
        if (!hf) {
            const int v = (64 * t) % PW + 16 * wave, f = (v >= PW ? 1 : 0) + (v >= 2 * PW ? 1 : 0);
            pr0 = (64 * t) / PW + f;
            pc0 = v - f * PW;
        } else {
            const int v = (64 * t) % PW0 + 16 * wave, f = (v >= PW0 ? 1 : 0) + (v >= 2 * PW0 ? 1 : 0) + (v >= 3 * PW0 ? 1 : 0);
            pr0 = (64 * t) / PW0 + f;
            pc0 = v - f * PW0;
        }
        const int pw = hf ? PW0 : PW, npx = hf ? PH0 * PW0 : PH * PW;
        const int Hs = hf ? a.H >> 1 : a.H, Ws = hf ? a.W >> 1 : a.W;
        const int yb = (hf ? y0 >> 1 : y0) - 1 + pr0, xb = (hf ? x0 >> 1 : x0) - 1;
        int pc = pc0 + (ln >> 2);
        const bool wrap = pc >= pw;
        pc = wrap ? pc - pw : pc;
        const int y = yb + (wrap ? 1 : 0), x = xb + pc;
        const bool ok = (64 * t + 16 * wave) + (ln >> 2) < npx && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const int v = (((n * Hs + y) * Ws + x) << 5) | (((ln & 3) ^ ((pc >> PSH) & 3)) << 3);
        return ok ? v : -1;
    };
    auto issue_piece = [&](int d, int kc, int t, int buf) {
        const bool first = kc < nc0;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
        glds16s(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + buf * PATCH8_BYTES + (wave + 4 * t) * 1024);
    };
    // weight piece u of this wave for step `st` (chunk st / 3, tap column st % 3): piece p = wave + 4 u < 3 * W_PIECES,
    // p = (tap row ky = p / W_PIECES, 1-KiB piece p % W_PIECES of that tap's slice)
    auto issue_weight = [&](int st, int slot, int u) {
        const int kc = st / 3, kx = st - kc * 3;
        const int p = wave + 4 * u;
        const int ky = p / W_PIECES, pis = p - ky * W_PIECES;
        glds16s(wbase + (size_t)(kc * 9 + ky * 3 + kx) * (BCO * 32) + pis * 512 + lane * 8, s_ring + slot * STEP_BYTES + ky * SLICE_BYTES + pis * 1024);
    };
    const int n_wd = (3 * W_PIECES - wave + 3) / 4;        // weight pieces of this wave per step (wave-uniform)

    const int R0 = 4 * wave;                               // the wave's first output row
    const bool chunk0_half = (nc0 > 0) && a.up0;

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
#pragma unroll
    for (int t = 0; t < 10; ++t) issue_piece(desc(n, y0, x0, t, chunk0_half), 0, t, 0);
    // prologue: weights of steps 0 and 1
    for (int stp = 0; stp < (S3 > 1 ? 2 : 1); ++stp)
#pragma unroll
        for (int u = 0; u < NWD; ++u)
            if (u < n_wd) issue_weight(stp, stp, u);
    if constexpr (EPI != SEPI_GRU) {
        for (int i = tid; i < BCO; i += 256) {
            const int co = co_tile * BCO + i;
            s_ss[i] = co < a.Cout ? a.scale[co] : 0.f;
            s_ss[BCO + i] = co < a.Cout ? a.shift[co] : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // channel row of MFMA row jl inside its 32-row tile: 16 * ((jl >> 2) & 1) + (jl & 3) + 4 * (jl >> 3), so that a lane's 16 accumulators are 16
    // consecutive channels; the GRU's 32 hidden channels are rows {0..15} and {48..63} of the gate's 16-row blocks (packing.pack_gru_stream)
    const int prm = ((jl >> 2) & 1) * (EPI == SEPI_GRU ? 48 : 16) + (jl & 3) + 4 * (jl >> 3);
    const int a_lane = (hl * BCO + prm) * 16;              // + slot * STEP_BYTES + a_off(ky, kh, mt)
    auto a_off = [](int ky, int kh, int mt) constexpr -> int { return ky * SLICE_BYTES + kh * (2 * BCO * 16) + mt * (EPI == SEPI_GRU ? 256 : 512); };

    bf16x8_t A[3][MT];      // weight fragments: block b computes from set b % 3 (three sets: block 5's are read two blocks ahead, before the barrier)
    bf16x8_t B[2][12];      // pixel fragments of the current / the next step: [patch row q = r + ky][K half]
    // pixel fragments of a step: patch buffer pb, source resolution sh (1 = half), tap column kx
    auto read_B = [&](bf16x8_t (&Bs)[12], const char *pb, int sh, int kx, int q0, int q1) __attribute__((always_inline)) {
        const int row_bytes = (sh ? PW0 : PW) * 64;
        const int col = jl + kx;
        const int pc = ((col - sh) >> sh) + sh;            // full: col;  half: ((col - 1) >> 1) + 1
        const int coff = ((pc << 2) + (hl ^ ((pc >> PSH) & 3))) * 16;   // K half 1: slot ^ 2 = byte offset ^ 32
        const char *prow0 = pb + (R0 >> sh) * row_bytes;
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const char *prow = prow0 + (((q - sh) >> sh) + sh) * row_bytes;
            Bs[q * 2] = *reinterpret_cast<const bf16x8_t *>(prow + coff);
            Bs[q * 2 + 1] = *reinterpret_cast<const bf16x8_t *>(prow + (coff ^ 32));
        }
    };
    // first step's operands
    read_B(B[0], s_patch, chunk0_half ? 1 : 0, 0, 0, 6);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) A[0][mt] = *reinterpret_cast<const bf16x8_t *>(s_ring + a_lane + a_off(0, 0, mt));

    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < n_tiles;
        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) tile_coords(next, nn, ny0, nx0);

        f32x16_t acc[MT][NR];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int f = 0; f < NR; ++f) acc[i][f] = (f32x16_t)(0.f);

        // Two chunks (six steps) per iteration of the chunk loop: with three steps per chunk the ring slot of a step is its tap column kx, the
        // patch buffer the chunk's parity CP and the pixel-fragment register set (CP + kx) & 1 -- ALL compile-time inside step<CP, KX>.  (The
        // host only sends layers with an even chunk count.)  Every per-step quantity is a by-value local: mutable state captured by reference
        // ended up in scratch memory, and scratch loads share vmcnt with the LDS-DMA pieces.
        const bool later_tile = tile != bid;                // the previous tile's output stores may still be in flight at the first wait
        auto step = [&](auto CPC, auto KXC, int kc) __attribute__((always_inline)) {
            constexpr int CP = decltype(CPC)::value, KX = decltype(KXC)::value;
            constexpr int PAR = (CP + KX) & 1;
            constexpr int SLOT = KX, WSLOT = (KX + 2) % 3, NSLOT = (KX + 1) % 3;
            const bool half = (kc < nc0) && a.up0;
            const bool last_chunk = kc + 1 == nchunks;
            const int kcn = last_chunk ? 0 : kc + 1;
            const bool fill = !last_chunk || has_next;
            const bool hfn = (kcn < nc0) && a.up0;
            const int dn = last_chunk ? nn : n, dy = last_chunk ? ny0 : y0, dx = last_chunk ? nx0 : x0;   // tile whose patch is filled during this chunk
            const int npieces = hfn ? (PH0 * PW0 * 4 + 63) / 64 : PATCH8_PIECES;
            const int st = kc * 3 + KX;
            // ---- what this step issues: weights of step st + 2 (wrapping into the next tile), patch pieces of the next chunk
            int wst = -1;                                   // step whose weights are streamed now
            {
                const int ahead = st + 2;
                if (ahead < S3) wst = ahead;
                else if (has_next && ahead - S3 < S3) wst = ahead - S3;
            }
            const int np = (fill && KX < 2) ? max(0, min(5, (npieces - wave - 20 * KX + 3) / 4)) : 0;   // patch pieces t = 5 KX .. 5 KX + np - 1
            const int nd = (wst >= 0 ? n_wd : 0) + np;     // DMAs of this wave in this step
            const bool relaxed = later_tile && st == 0;
            // ---- the step after this one (whose first operands are read under tap row 2)
            const bool pre = !(last_chunk && KX == 2) || has_next;
            constexpr int NKX = KX == 2 ? 0 : KX + 1;
            const int nsh = (KX == 2 ? hfn : half) ? 1 : 0;
            const char *npb = s_patch + (KX == 2 ? CP ^ 1 : CP) * PATCH8_BYTES;
            const char *ws = s_ring + SLOT * STEP_BYTES + a_lane;
            const char *nws = s_ring + NSLOT * STEP_BYTES + a_lane;
            // the patch piece whose source address is being formed (stages desc_pos -> desc_index -> desc_addr -> issue)
            int p_pc = 0, p_y = 0, p_x = 0, p_d = 0, p_l4 = 0;
            const void *p_src = nullptr;
            auto desc_pos = [&](int y0t, int x0t, int t, bool hf) __attribute__((always_inline)) {
                const int ln = fresh_lane();
                int pr0, pc0;
                if (!hf) {
                    const int v = (64 * t) % PW + 16 * wave, f = (v >= PW ? 1 : 0) + (v >= 2 * PW ? 1 : 0);
                    pr0 = (64 * t) / PW + f;
                    pc0 = v - f * PW;
                } else {
                    const int v = (64 * t) % PW0 + 16 * wave, f = (v >= PW0 ? 1 : 0) + (v >= 2 * PW0 ? 1 : 0) + (v >= 3 * PW0 ? 1 : 0);
                    pr0 = (64 * t) / PW0 + f;
                    pc0 = v - f * PW0;
                }
                const int pw = hf ? PW0 : PW;
                int pc = pc0 + (ln >> 2);
                const bool wrap = pc >= pw;
                pc = wrap ? pc - pw : pc;
                p_pc = pc;
                p_l4 = ln;
                p_y = (hf ? y0t >> 1 : y0t) - 1 + pr0 + (wrap ? 1 : 0);
                p_x = (hf ? x0t >> 1 : x0t) - 1 + pc;
            };
            auto desc_index = [&](int nt, int t, bool hf) __attribute__((always_inline)) {
                const int npx = hf ? PH0 * PW0 : PH * PW;
                const int Hs = hf ? a.H >> 1 : a.H, Ws = hf ? a.W >> 1 : a.W;
                const bool ok = (64 * t + 16 * wave) + (p_l4 >> 2) < npx && (unsigned)p_y < (unsigned)Hs && (unsigned)p_x < (unsigned)Ws;
                const int v = (((nt * Hs + p_y) * Ws + p_x) << 5) | (((p_l4 & 3) ^ ((p_pc >> PSH) & 3)) << 3);
                p_d = ok ? v : -1;
            };
            auto desc_addr = [&](int kcp) __attribute__((always_inline)) {
                const bool first = kcp < nc0;
                const uint16_t *src = first ? a.in0 : a.in1;
                const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
                const unsigned off = (unsigned)(p_d >> 5) * cs + (unsigned)((first ? kcp : kcp - nc0) * 32 + (p_d & 31));
                p_src = p_d >= 0 ? (const void *)(src + off) : zero_page;
            };
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int ky = b >> 1, kh = b & 1;
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const int mt = i / NR, r = i - mt * NR;
                    const int j = b * BLK + i;          // MFMA index in the step
                    __builtin_amdgcn_sched_barrier(0);
                    acc[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[b % 3][mt], B[PAR][(r + ky) * 2 + kh], acc[mt][r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // weight fragments, one read per MFMA: under block b those of block b + 1 -- except that block 5's are read under
                    // block 3 too, BEFORE the barrier (every read of this step's ring slot precedes it: the slot is overwritten by the
                    // DMAs of the next step), and block 0 of the next step under block 5
                    if (i >= 1 && i <= MT) {
                        const int m2 = i - 1;
                        if (b < 4) A[(b + 1) % 3][m2] = *reinterpret_cast<const bf16x8_t *>(ws + a_off((b + 1) >> 1, (b + 1) & 1, m2));
                        else if (b == 5 && pre) A[0][m2] = *reinterpret_cast<const bf16x8_t *>(nws + a_off(0, 0, m2));
                    }
                    if (b == 3 && i >= MT + 1 && i <= 2 * MT) A[2][i - MT - 1] = *reinterpret_cast<const bf16x8_t *>(ws + a_off(2, 1, i - MT - 1));
                    // pixel fragments of the next step, after the barrier: patch rows 0..2 under block 4, 3..5 under block 5
                    if (b >= 4 && i >= MT + 1 && i <= MT + 3) {
                        const int q = (b - 4) * 3 + (i - MT - 1);
                        if (pre && !(W1_DBG & 8)) read_B(B[PAR ^ 1], npb, nsh, NKX, q, q + 1);
                    }
                    // LDS-DMA pieces of this step: one per 5 (4 at 96 rows) MFMAs in the first two tap rows, weights and patch alternating.
                    // A patch piece's source address takes ~40 instructions: computed in three stages under the three MFMAs before its slot
                    // (a 32-cycle MFMA hides about five instructions; all of it in one gap stalled the pipe for ~170 cycles per piece).
                    constexpr int DSTRIDE = BLK >= 16 ? 5 : 4;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        const int jj = j + k;                // the slot this gap works for, k MFMAs ahead
                        if (jj < 2 || (jj - 2) % DSTRIDE != 0 || (jj - 2) / DSTRIDE >= 11) continue;
                        const int d = (jj - 2) / DSTRIDE;
                        if ((d & 1) == 0) {
                            const int u = d >> 1;           // weight piece u = 0..5
                            if (k == 0 && u < NWD && wst >= 0 && u < n_wd && !(W1_DBG & 2)) issue_weight(wst, WSLOT, u);
                        } else if (KX < 2 && !(W1_DBG & 1)) {
                            const int t = d >> 1;           // patch piece t = 0..4 of this tap column
                            if (t < np) {
                                if (k == 3) desc_pos(dy, dx, KX * 5 + t, hfn);
                                else if (k == 2) desc_index(dn, KX * 5 + t, hfn);
                                else if (k == 1) desc_addr(kcn);
                                else glds16s(p_src, s_patch + (CP ^ 1) * PATCH8_BYTES + (wave + 4 * (KX * 5 + t)) * 1024);
                            }
                        }
                    }
                    if (j == 4 * BLK - 1) {
                        // two thirds in: this wave's DMAs of the PREVIOUS step have landed (this step's nd -- and, right after an epilogue,
                        // the tile's output stores, which are younger -- may stay in flight); its LDS reads are done; then everybody's are
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!(W1_DBG & 4)) {
                            if (relaxed) w1_wait_keep<N_ST>(nd);
                            else w1_wait_keep<0>(nd);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
        for (int kc = 0; kc < nchunks; kc += 2) {
            step(I0{}, I0{}, kc);
            step(I0{}, I1{}, kc);
            step(I0{}, I2{}, kc);
            step(I1{}, I0{}, kc + 1);
            step(I1{}, I1{}, kc + 1);
            step(I1{}, I2{}, kc + 1);
        }
        w1_epilogue<MT, NR, EPI>(a, acc, EPI == SEPI_GRU ? co_tile * 32 : co_tile * BCO, n, y0 + R0, x0, jl, hl, (lds_cf_t *)s_ss, BCO);
        if (!has_next) break;
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
    }
}

template <int BCO, int EPI>
static int launch_w1(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = 3 * 3 * BCO * 64 + 2 * PATCH8_BYTES + 1024;   // 153 KiB at 128 rows, 135 KiB at 96 (+1 KiB: epilogue parameters)
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_w1_kernel<BCO, EPI>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    int grid = n_tiles;
    const int g = v2x_num_cus() / a.n_co_tiles * a.n_co_tiles;   // persistent: a workgroup's tiles share one channel tile
    if (g > 0 && g < n_tiles) grid = g;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_w1_kernel");
    return V2X_OK;
}

// rows: the layer's row tile (128 plain, 96 = the ConvGRU's 32 hidden channels x 3 gates); a.tiles_* / n_px_tiles describe 16 x 32 pixel tiles
int v2x_launch_stream_w1(const StreamArgs &a, int rows, int gru, hipStream_t s) {
    if (gru) return rows == 96 ? launch_w1<96, SEPI_GRU>(a, s) : 1;
    return rows == 128 ? launch_w1<128, SEPI_BF16>(a, s) : 1;
}
