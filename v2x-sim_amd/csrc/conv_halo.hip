// Halo-tile 3x3 stride-1 convolution for the full-resolution, small-channel layers
// (conv_pre_1/2, conv8_1/2, conv1_2, conv7_2 and the detection heads of upstream
// coperception/models/det/backbone/Backbone.py + DetModelBase.py; code absent from
// /root/reference, see include/v2x_amd.h).
//
// Why a second conv kernel: with Cin <= 96 the gather-style implicit GEMM (conv_igemm.hip) re-reads
// every input pixel 9x through L2 and spends its issue slots on im2col address math; these layers
// are HBM-bound (AI 100-350 FLOP/B), so the design goal is "each input byte crosses the memory
// system once":
//   * a persistent workgroup keeps the WHOLE weight tensor of the layer in LDS (18-55 KiB) and
//     walks 8x32-pixel output tiles;
//   * per tile the (8+2)x(32+2) input patch (and for conv8_1 the 6x18 patch of the half-resolution
//     source: nearest x2 upsample + concat are address math on the patch) is brought in by LDS-DMA
//     (global_load_lds_dwordx4), double-buffered so the next tile's patch streams in under the MFMAs;
//     out-of-image pixels read a zero page;
//   * all 9 taps x Cin/32 k-steps read their MFMA operands from that patch: 16 consecutive pixels
//     of one patch row are one ds_read_b128 fragment.  LDS layouts found by exhaustive search over
//     the ds_read_b128 lane groups (MI355X_MICROARCH.md): pixel-major patch with the 16-B slot XOR
//     ((x>>1)&3) for 4/12 slots per pixel and (x&7) for 8 -> conflict-free at every tap alignment;
//     weights k-slot-major [kslot][cout][8] -> conflict-free;
//   * optional chained 1x1 conv in the epilogue (det heads 32->64->48, conv1_2 -> conv3d_1): the rows
//     of the first GEMM are ordered so that each lane's accumulators ARE its B-operand fragment of
//     the second MFMA (hidden channel kappa = 32*(i>>1) + 8*q + 4*(i&1) + r for tile i, lane group q,
//     register r) -- no cross-lane traffic, the hidden map never exists in memory.
#include "common.h"
#include <cstdlib>

// 64 B of zeros for out-of-image patch pixels (own copy: no relocatable device code needed)
static __device__ __attribute__((aligned(64))) unsigned int g_zero_page_h[16];

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void glds16h(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

struct HaloArgs {
    const uint16_t *in0;  // [N][H/2][W/2][C0] (nearest-x2 upsampled on the fly) or nullptr
    const uint16_t *in1;  // [N][H][W][C1]
    const uint32_t *bits; // BITS form: in1 is replaced by the occupancy words [N][H][W] (bit z = channel z)
    int zbits;            // number of valid bits (BEV height bins)
    int N, H, W;
    const uint16_t *w;    // k-slot-major [9*(C0+C1)/8][COUT][8] bf16
    const float *scale, *shift;  // [COUT] (chain: in kappa order)
    int relu;
    void *out;            // NHWC [N][H][W][out_cstride] (+out_coff)
    int out_cstride, out_coff;
    // chained 1x1 (COUT2 > 0)
    const uint16_t *w2;   // row-major [COUT2][COUT] bf16, K in kappa order
    const float *scale2, *shift2;
    int relu2, cout2_real, split;
    void *out2;
    int out2_cstride;
    int tiles_x, tiles_y, n_tiles;
    // EPI2 == 3 (fused detection heads): out = candidate keys [N][det_cap] u64, out2 = candidate box codes [N][det_cap][6] f32
    int32_t *det_counts;  // [N], zeroed by the caller
    float det_thr, det_margin;   // score threshold; logit(det_thr) - 1e-3 = the conservative pre-test on c1 - c0
    int det_cap;
    int xcd_walk;         // 1: XCD-contiguous tile walk (common.h: v2x_xcd_tile_walk; tuning switch HALO_XCD)
    int x4;               // bf16 outputs: 16-byte stores (common.h: v2x_store_pair_x4); set by the dispatch when the output view allows
};

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int PH0 = TH / 2 + 2, PW0 = TW / 2 + 2;

// Patch swizzle (slot permutation per pixel x).  Chosen by TIMING the fragment reads (tools/lds_conflict_probe.hip: lane (fj, fq) reads
// 16 bytes of pixel x0 + fj -- or ((c + fj) >> 1) + 1 for a half-resolution source -- at slot 4*kc + fq; 8 reads in flight per wave):
//   4 slots per pixel:  (x >> 2) & 3             17.5 ns per read at every alignment of both forms;  the round-1 choice (x >> 1) & 3:
//                                                22.9 ns for EVERY full-resolution read, 22.6 for two of three half-resolution alignments
//   8 slots per pixel:  ((x >> 1) & 1) * 4 ^ ((x >> 2) & 3)   17.2-17.5 ns;  round 1's x & 7: 22.7 / 19.2 ns
// SQ_LDS_BANK_CONFLICT reads 0 for the slow full-resolution patterns: the counter does not see whatever pairing rule ds_read_b128
// applies, which is how round 1's search (driven by that counter) settled on them.  INSIDE the kernels the faster reads change nothing
// (same-box A/B, two runs each: heads 1 132-1 154 vs 1 133-1 139 us, conv8_1 967-974 vs 975-977, pair kernel 687-713 vs 723-727 (worse: its
// layer-A epilogue WRITES this layout), conv7_2 365-367 vs 364-369): the fragment reads are not what these kernels wait for.  Default stays
// the round-1 swizzle (V2X_HALO_PSWZ_BUILD=1); =2 builds the timing-derived one.
#ifndef V2X_HALO_PSWZ_BUILD
#define V2X_HALO_PSWZ_BUILD 1
#endif
template <int SPP>
__device__ __forceinline__ int swz(int slot, int x) {
    if constexpr (V2X_HALO_PSWZ_BUILD == 1) {
        if constexpr (SPP == 8) return slot ^ (x & 7);
        else return slot ^ ((x >> 1) & 3);
    } else {
        if constexpr (SPP == 8) return slot ^ ((((x >> 1) & 1) << 2) ^ ((x >> 2) & 3));
        else return slot ^ ((x >> 2) & 3);  // SPP 4 or 12: permute inside each aligned group of 4 slots
    }
}

constexpr int round64(int v) { return (v + 63) / 64 * 64; }

// EPI2: 0 = none (plain epilogue, bf16 out), 1 = chained 1x1 with bf16 out, 2 = chained 1x1 with fp32 split out,
//       3 = detection heads: the chained 1x1's rows are in "det order" (include/v2x_amd.h, V2X_EPI_DET) so that lane (fj, fq < 3) holds
//           the two class logits AND the six box codes of anchors 2 fq and 2 fq + 1 of its pixel; the foreground score is thresholded
//           here and only the candidates (key = ~score | anchor | slot, six codes) leave the chip -- the 4 GB of fp32 logits per 320 maps
//           that EPI2 = 2 writes and det_candidates_kernel reads back are never materialised
// DB:   true  = two patch buffers, the next tile's patch streams in under this tile's MFMAs (MFMA-heavy variants);
//       false = ONE patch buffer and a raw barrier after the output stores: the per-tile __syncthreads of the DB
//               form drains vmcnt(0), i.e. also waits for the tile's own output stores, which serialises ~2 us of
//               store latency into every tile of an HBM-bound layer.  The single-buffer form halves the LDS
//               footprint instead (40 KiB for 32 -> 32) so 3-4 workgroups per CU hide each other's load AND store
//               latency.
// BITS: the full-resolution source is the voxelizer's bit grid; the patch fill expands bit z -> bf16 {0,1} channel z
//       while writing LDS (ordinary loads + ds_write_b128), so the first layer reads 4 B instead of 64 B per pixel
//       and the separate expansion kernel disappears from the points -> logits path.
template <int C0, int C1, int COUT, int COUT2, int EPI2, bool DB, bool BITS = false>
__device__ __forceinline__ void conv3x3_halo_body(const HaloArgs &a) {
    constexpr int SPP0 = C0 / 8, SPP1 = C1 / 8;
    constexpr int NS1 = round64(PH * PW * SPP1);            // 16-B slots of the full-res patch (padded to whole waves)
    constexpr int NS0 = C0 ? round64(PH0 * PW0 * SPP0) : 0; // ... of the half-res patch
    constexpr int PATCH_BYTES = (NS0 + NS1) * 16;
    constexpr int KSLOTS = 9 * (SPP0 + SPP1);
    constexpr int W_BYTES = KSLOTS * COUT * 16;
    constexpr int TCO = COUT / 16;
    constexpr int TCO2 = COUT2 / 16;
    static_assert(C1 % 32 == 0 && C0 % 32 == 0, "channel groups of 32 (one MFMA k-step)");
    static_assert(COUT % 32 == 0, "output channel tiles come in pairs (chain layout)");
    static_assert(W_BYTES % 1024 == 0, "weights are moved 1 KiB per wave instruction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;   // v2x_relu_bf16x2_floor: identity when the layer has no ReLU
    char *s_w = smem;
    char *s_patch = smem + W_BYTES;  // DB: two buffers of PATCH_BYTES, else one
    // Plain epilogue: scale/shift live in LDS.  vmcnt is in-order, so a GLOBAL load in the epilogue can only be waited for
    // together with every store issued before it -- reading them per channel tile right before use drained the first
    // channel tile's output stores (~1-2 us of HBM write latency) before the second could be written (same-box A/B at 320
    // maps: conv8_1 1.33 -> 1.28 ms, conv8_2 0.54 -> 0.53).  The chained (heads) epilogue reads its vectors from global memory
    // ONCE per tile, before its first store (registers): 1.29 -> 1.09 ms same-box, whereas the LDS form measured 1.36.
    constexpr int PATCH_ALLOC = DB ? 2 * PATCH_BYTES : (BITS ? PH * PW * SPP1 * 16 : PATCH_BYTES);
    float *s_ss = reinterpret_cast<float *>(smem + W_BYTES + PATCH_ALLOC);   // [scale | shift]
    if constexpr (COUT2 == 0) {
        for (int i = threadIdx.x; i < COUT; i += 256) {
            s_ss[i] = a.scale[i];
            s_ss[COUT + i] = a.shift[i];
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15;  // pixel inside a 16-pixel fragment
    const int fq = lane >> 4;  // k-group (operands) / channel quad (accumulators)

    // ---- weights: one linear LDS-DMA copy, resident for the whole kernel --------------------------
    for (int off = wave * 1024; off < W_BYTES; off += 4096)
        glds16h(reinterpret_cast<const char *>(a.w) + off + lane * 16, s_w + off);

    // ---- per-lane DMA tables, computed once (the per-tile patch fill is then ~14 instructions per 1-KiB piece instead
    // of ~60: two integer divisions by constants, the swizzle and the 64-bit address used to be redone for every piece
    // of every tile -- ~560 of the ~1200 instructions a wave executed per conv8_1 tile).  Piece t of this wave covers
    // LDS slots [wave*64 + t*256, +64); one packed word per piece: bits 0-19 = element offset of the lane's 16 B relative
    // to the patch origin pixel (the dispatcher bounds W so that it fits), bits 20-23 = patch row, bits 24-29 = patch
    // column (for the image-bounds test), negative = padding slot behind the patch.
    constexpr bool TABLES = DB && !BITS;   // the single-buffer form lives on a 128-VGPR budget: it keeps the per-piece arithmetic
    constexpr int NP1 = TABLES ? (NS1 + 255) / 256 : 0, NP0 = TABLES ? (NS0 + 255) / 256 : 0;
    int tb1[NP1 ? NP1 : 1], tb0[NP0 ? NP0 : 1];
    if constexpr (TABLES) {
#pragma unroll
        for (int t = 0; t < NP1; ++t) {
            const int L = wave * 64 + t * 256 + lane;
            const int pix = L / SPP1, phys = L - pix * SPP1;
            const int pr = pix / PW, pc = pix - pr * PW;
            tb1[t] = pix < PH * PW ? (((pr * a.W + pc) * C1 + swz<SPP1>(phys, pc) * 8) | (pr << 20) | (pc << 24)) : -1;
        }
        if constexpr (C0 > 0) {
            constexpr int S0 = SPP0 ? SPP0 : 1;
#pragma unroll
            for (int t = 0; t < NP0; ++t) {
                const int L = wave * 64 + t * 256 + lane;
                const int pix = L / S0, phys = L - pix * S0;
                const int pr = pix / PW0, pc = pix - pr * PW0;
                tb0[t] = pix < PH0 * PW0 ? (((pr * (a.W >> 1) + pc) * C0 + swz<(SPP0 ? SPP0 : 4)>(phys, pc) * 8) | (pr << 20) | (pc << 24)) : -1;
            }
        }
    }

    auto load_patch = [&](int tile, int buf) {
        const int txy = a.tiles_x * a.tiles_y;
        const int n = tile / txy;
        const int r = tile - n * txy;
        const int ty = r / a.tiles_x;
        const int tx = r - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        char *pb = s_patch + buf * PATCH_BYTES;
        if constexpr (BITS) {
            static_assert(!BITS || (C0 == 0 && SPP1 == 4), "bit-grid input: single 32-channel source");
            const uint32_t zmask = (a.zbits >= 32) ? 0xffffffffu : ((1u << a.zbits) - 1u);
            for (int p = tid; p < PH * PW; p += 256) {
                const int pr = p / PW, pc = p - pr * PW;
                const int y = y0 - 1 + pr, x = x0 - 1 + pc;
                const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const uint32_t word = ok ? (a.bits[(size_t)(n * a.H + y) * a.W + x] & zmask) : 0u;
#pragma unroll
                for (int slot = 0; slot < 4; ++slot) {
                    const uint32_t b = (word >> (8 * slot)) & 0xffu;
                    const uint32_t one = 0x3f80u;  // bf16(1.0)
                    uint4 v;
                    v.x = ((b & 1u) ? one : 0u) | ((b & 2u) ? (one << 16) : 0u);
                    v.y = ((b & 4u) ? one : 0u) | ((b & 8u) ? (one << 16) : 0u);
                    v.z = ((b & 16u) ? one : 0u) | ((b & 32u) ? (one << 16) : 0u);
                    v.w = ((b & 64u) ? one : 0u) | ((b & 128u) ? (one << 16) : 0u);
                    *reinterpret_cast<uint4 *>(pb + NS0 * 16 + (p * 4 + swz<4>(slot, pc)) * 16) = v;
                }
            }
            return;
        }
        if constexpr (!TABLES) {
            // full-resolution source: patch pixel (pr, pc) <- image pixel (y0-1+pr, x0-1+pc)
            for (int base = wave * 64; base < NS1; base += 256) {
                const int L = base + lane;
                const int pix = L / SPP1;
                const int phys = L - pix * SPP1;
                const int pr = pix / PW;
                const int pc = pix - pr * PW;
                const int y = y0 - 1 + pr, x = x0 - 1 + pc;
                const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const unsigned off = ((unsigned)(n * a.H + y) * (unsigned)a.W + (unsigned)x) * (unsigned)C1 + swz<SPP1>(phys, pc) * 8;
                glds16h(ok ? (const void *)(a.in1 + off) : (const void *)g_zero_page_h, pb + NS0 * 16 + base * 16);
            }
            static_assert(TABLES || C0 == 0, "the single-buffer form has one source");
            return;
        }
        // full-resolution source: patch pixel (pr, pc) <- image pixel (y0-1+pr, x0-1+pc)
        {
            const unsigned base = ((unsigned)(n * a.H + y0 - 1) * (unsigned)a.W + (unsigned)(x0 - 1)) * (unsigned)C1;  // may wrap: only used in-bounds
#pragma unroll
            for (int t = 0; t < NP1; ++t) {
                if (wave * 64 + t * 256 >= NS1) break;   // wave-uniform
                const int y = y0 - 1 + ((tb1[t] >> 20) & 15), x = x0 - 1 + ((tb1[t] >> 24) & 63);
                const bool ok = tb1[t] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const unsigned off = base + (unsigned)(tb1[t] & 0xfffff);
                glds16h(ok ? (const void *)(a.in1 + off) : (const void *)g_zero_page_h, pb + NS0 * 16 + (wave * 64 + t * 256) * 16);
            }
        }
        if constexpr (C0 > 0) {
            // half-resolution source: patch pixel (pr, pc) <- source pixel (y0/2-1+pr, x0/2-1+pc)
            const int Hs = a.H >> 1, Ws = a.W >> 1;
            const unsigned base = ((unsigned)(n * Hs + (y0 >> 1) - 1) * (unsigned)Ws + (unsigned)((x0 >> 1) - 1)) * (unsigned)C0;
#pragma unroll
            for (int t = 0; t < NP0; ++t) {
                if (wave * 64 + t * 256 >= NS0) break;   // wave-uniform
                const int y = (y0 >> 1) - 1 + ((tb0[t] >> 20) & 15), x = (x0 >> 1) - 1 + ((tb0[t] >> 24) & 63);
                const bool ok = tb0[t] >= 0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
                const unsigned off = base + (unsigned)(tb0[t] & 0xfffff);
                glds16h(ok ? (const void *)(a.in0 + off) : (const void *)g_zero_page_h, pb + (wave * 64 + t * 256) * 16);
            }
        }
    };

    // chained 1x1: its weight fragments live in registers for the whole kernel
    bf16x8_t w2f[(TCO2 > 0 && EPI2 != 3) ? TCO2 : 1][COUT / 32];
    // detection heads: the block-diagonal 1x1 -- the class tile multiplies the cls hidden half (k-step 0) only, the three code tiles the
    // reg hidden half (k-step 1) only; the other halves of the rows are zeros and are skipped (exact: they contribute +0)
    bf16x8_t wdet[EPI2 == 3 ? 4 : 1];
    if constexpr (EPI2 == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            wdet[t] = *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(t * 16 + fj) * COUT + (t == 0 ? 0 : 32) + fq * 8);
    } else if constexpr (COUT2 > 0) {
#pragma unroll
        for (int i2 = 0; i2 < TCO2; ++i2)
#pragma unroll
            for (int s = 0; s < COUT / 32; ++s)
                w2f[i2][s] = *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(i2 * 16 + fj) * COUT + s * 32 + fq * 8);
    }

    const v2x_tile_walk walk = v2x_xcd_tile_walk(a.n_tiles, a.xcd_walk);
    int tile = walk.first;
    int cur = 0;
    bool first_tile = true;
    // output-store instructions a wave issues per tile (static: every channel tile is written; the det heads' 48 real
    // channels fill their 3 tiles exactly -- other padded widths fall back to the full drain)
    // (the detection heads store only their rare candidates: full drain, which then waits for nothing but the next patch)
    // (COUT2 == 16: the seg head, 8 real classes of one padded tile -- lanes of the upper two k-slot quarters are masked off, the store instruction is issued)
    constexpr int N_STORES = EPI2 == 3 ? 0 : (COUT2 == 0 ? TCO * 4 : ((COUT2 == 48 || COUT2 == 64 || COUT2 == 16) ? TCO2 * 4 : 0));
    if (DB && tile < walk.end) load_patch(tile, 0);

    for (; tile < walk.end; tile += walk.step) {
        if constexpr (DB) {
            // patch[cur] (and, first time, the weights) have landed; everyone left patch[cur^1].  After the first tile
            // this is a COUNTED wait + raw barrier: the only vector-memory operations younger than patch[cur]'s DMA are
            // the previous tile's output stores (the epilogue's scale/shift loads were consumed before the stores were
            // issued), and a __syncthreads (vmcnt(0)) would serialise their ~1-2 us HBM write latency into every tile.
            if (first_tile) {
                __syncthreads();
                first_tile = false;
            } else {
                // (a.x4: the 16-byte form issues HALF as many stores -- the count must follow, or the wait would let patch pieces stay in flight)
                if constexpr (N_STORES == 8) {
                    if (a.x4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else if constexpr (N_STORES == 16) {
                    if (a.x4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                } else if constexpr (N_STORES == 24) {
                    if (a.x4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                } else if constexpr (N_STORES == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // (fp32 heads: never x4)
                else if constexpr (N_STORES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (fp32 seg head)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const int next = tile + walk.step;
            if (next < walk.end) load_patch(next, cur ^ 1);
        } else {
            load_patch(tile, 0);
            __syncthreads();  // this tile's patch (and, first time, the weights) have landed
        }

        const char *pb = s_patch + cur * PATCH_BYTES;
        // detection heads (EPI2 == 3): the main loop computes only the CLASSIFICATION head's hidden rows (channel tiles 0, 1); the
        // regression head's (tiles 2, 3) are computed in the epilogue, per 16-pixel fragment, only where a candidate may exist
        constexpr int TCOM = (EPI2 == 3) ? 2 : TCO;
        f32x4_t acc[TCOM][4];
#pragma unroll
        for (int i = 0; i < TCOM; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // wave w owns tile rows 2w, 2w+1; fragment f = (row r = f>>1, column half ch = f&1).
        // K is walked in GROUPS = (tap column kx, 32-channel chunk); a group covers the three tap rows ky.  Its pixel
        // operands are read ONCE: the three ky shifts of the wave's two rows touch only 4 distinct patch rows (3 for the
        // half-resolution source, where rows 2w+r+ky-1 fold pairwise), so a group reads 8 (6) B fragments instead of 12,
        // plus its 3 x TCO weight fragments: 114 instead of 162 ds_read_b128 per tile for conv8_1, in batches of 12-20
        // reads per wait (the LDS reaches its rate only with >= 16 reads in flight per wave, MI355X_MICROARCH.md).
        // With TCO == 2 the fragments of group g+1 are read into a second register set before the 24 MFMAs of group g
        // are issued (hard scheduling fences keep that order), so the LDS latency runs under the matrix work even with
        // one wave per SIMD (conv8_1: 129 KiB of LDS -> 4 waves per CU).
        constexpr int KC0 = C0 / 32, KC1 = C1 / 32, KC = KC0 + KC1, NG = 3 * KC;
        constexpr bool PIPE = (TCOM == 2) && DB;   // the single-buffer form runs 4 workgroups per CU on a 128-VGPR budget
        struct Frags {
            bf16x8_t A[3][TCOM];
            bf16x8_t B[8];
        };
        Frags fr[PIPE ? 2 : 1];
        auto load_group = [&](int g, Frags &F) {
            const int kx = g / KC, kk = g - kx * KC;
            if (kk < KC0) {   // half-resolution (x2 upsampled) source, chunk kk
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int kslot = (ky * 3 + kx) * (SPP0 + SPP1) + kk * 4 + fq;
#pragma unroll
                    for (int i = 0; i < TCOM; ++i)
                        F.A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + i * 16 + fj) * 16);
                }
#pragma unroll
                for (int hr = 0; hr < 3; ++hr)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int pr = wave + hr;                           // = ((2w + r + ky - 1) >> 1) + 1
                        const int pc = ((ch * 16 + fj + kx - 1) >> 1) + 1;
                        F.B[hr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(
                            pb + ((pr * PW0 + pc) * (SPP0 ? SPP0 : 1) + swz<(SPP0 ? SPP0 : 4)>(kk * 4 + fq, pc)) * 16);
                    }
            } else {
                const int kc = kk - KC0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int kslot = (ky * 3 + kx) * (SPP0 + SPP1) + SPP0 + kc * 4 + fq;
#pragma unroll
                    for (int i = 0; i < TCOM; ++i)
                        F.A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + i * 16 + fj) * 16);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int pr = 2 * wave + rr;                       // = 2w + r + ky
                        const int pc = ch * 16 + fj + kx;
                        F.B[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(
                            pb + NS0 * 16 + ((pr * PW + pc) * SPP1 + swz<SPP1>(kc * 4 + fq, pc)) * 16);
                    }
            }
        };
        auto mma_group = [&](int g, const Frags &F) {
            const bool half = (g % KC) < KC0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < TCOM; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int r = f >> 1, ch = f & 1;
                        const int row = half ? (((r + ky - 1) >> 1) + 1) : (r + ky);
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.A[ky][i], F.B[row * 2 + ch], acc[i][f], 0, 0, 0);
                    }
        };
        if constexpr (PIPE) {
            load_group(0, fr[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, fr[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // reads of group g+1 stay above the MFMA block of group g
                mma_group(g, fr[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                load_group(g, fr[0]);
                __builtin_amdgcn_sched_barrier(0);   // all reads of the group in one batch, then its 12 x TCO MFMAs
                mma_group(g, fr[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue -----------------------------------------------------------------------------
        const int txy = a.tiles_x * a.tiles_y;
        const int n = tile / txy;
        const int rr = tile - n * txy;
        const int ty = rr / a.tiles_x;
        const int tx = rr - ty * a.tiles_x;
        if constexpr (EPI2 == 3) {
            static_assert(EPI2 != 3 || (C0 == 0 && C1 == 32 && COUT == 64 && COUT2 == 64), "detection heads: 32 -> (32 | 32) -> det order");
            auto hidden = [&](const f32x4_t &lo, const f32x4_t &hi, const float4 &s0, const float4 &t0, const float4 &s1, const float4 &t1) {
                uint4 p;   // the arithmetic of the logits path's hidden layer, value for value
                p.x = pack_bf16x2(lo[0] * s0.x + t0.x, lo[1] * s0.y + t0.y);
                p.y = pack_bf16x2(lo[2] * s0.z + t0.z, lo[3] * s0.w + t0.w);
                p.z = pack_bf16x2(hi[0] * s1.x + t1.x, hi[1] * s1.y + t1.y);
                p.w = pack_bf16x2(hi[2] * s1.z + t1.z, hi[3] * s1.w + t1.w);
                p.x = v2x_relu_bf16x2_floor(p.x, relu_floor);
                p.y = v2x_relu_bf16x2_floor(p.y, relu_floor);
                p.z = v2x_relu_bf16x2_floor(p.z, relu_floor);
                p.w = v2x_relu_bf16x2_floor(p.w, relu_floor);
                return __builtin_bit_cast(bf16x8_t, p);
            };
            // first-pass parameters: hidden scale / shift of tiles 0, 1 (channels 8 fq .. 8 fq + 7) and the class tile of the 1x1
            const float4 det_sc0 = *reinterpret_cast<const float4 *>(a.scale + 8 * fq), det_sf0 = *reinterpret_cast<const float4 *>(a.shift + 8 * fq);
            const float4 det_sc1 = *reinterpret_cast<const float4 *>(a.scale + 8 * fq + 4), det_sf1 = *reinterpret_cast<const float4 *>(a.shift + 8 * fq + 4);
            const float4 det_s2 = *reinterpret_cast<const float4 *>(a.scale2 + fq * 4), det_t2 = *reinterpret_cast<const float4 *>(a.shift2 + fq * 4);
            // ---- first pass, all four fragments at once (independent chains): class logits of the lane's two anchors and the conservative
            // pre-test on the logit margin (score >= thr <=> c1 - c0 >= logit(thr); det_margin = that minus 1e-3).  The exact softmax formula
            // below decides, but only fragments that can hold a candidate pay for it and for the regression half.
            float4 cl[4];
            unsigned long long any_f[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const bf16x8_t hb0 = hidden(acc[0][f], acc[1][f], det_sc0, det_sf0, det_sc1, det_sf1);
                const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdet[0], hb0, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cl[f] = make_float4(d[0] * det_s2.x + det_t2.x, d[1] * det_s2.y + det_t2.y, d[2] * det_s2.z + det_t2.z, d[3] * det_s2.w + det_t2.w);
                const bool maybe = fq < 3 && ((cl[f].y - cl[f].x >= a.det_margin) || (cl[f].w - cl[f].z >= a.det_margin));
                any_f[f] = __builtin_amdgcn_ballot_w64(maybe);
            }
            if ((any_f[0] | any_f[1] | any_f[2] | any_f[3]) != 0) {     // wave-uniform; false for almost every wave of a trained detector
                // ---- second pass, per fragment with a possible candidate: the regression head's hidden rows (tiles 2, 3) from the patch still
                // in LDS, in the main loop's K order (kx, then ky), then the six codes of each anchor.  Its parameters are loaded here
                // (rare path): hidden scale / shift of tile i -> channels kappa = 32 (i >> 1) + 8 fq + 4 (i & 1) (chain order).
                const float4 sc2 = *reinterpret_cast<const float4 *>(a.scale + 32 + 8 * fq), sf2 = *reinterpret_cast<const float4 *>(a.shift + 32 + 8 * fq);
                const float4 sc3 = *reinterpret_cast<const float4 *>(a.scale + 36 + 8 * fq), sf3 = *reinterpret_cast<const float4 *>(a.shift + 36 + 8 * fq);
                float4 s2v[3], t2v[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    s2v[t] = *reinterpret_cast<const float4 *>(a.scale2 + (t + 1) * 16 + fq * 4);
                    t2v[t] = *reinterpret_cast<const float4 *>(a.shift2 + (t + 1) * 16 + fq * 4);
                }
#pragma unroll   // (four copies of the rare path: a run-time fragment index would put cl / any_f into scratch)
                for (int f = 0; f < 4; ++f) {
                    const float4 c = cl[f];
                    if (any_f[f] == 0) continue;
                    const int r = f >> 1, ch = f & 1;
                    f32x4_t g2 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, g3 = g2;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int kslot = (ky * 3 + kx) * SPP1 + fq;
                            const int pr = 2 * wave + r + ky, pc = ch * 16 + fj + kx;
                            const bf16x8_t Bf = *reinterpret_cast<const bf16x8_t *>(pb + NS0 * 16 + ((pr * PW + pc) * SPP1 + swz<SPP1>(fq, pc)) * 16);
                            const bf16x8_t A2 = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + 32 + fj) * 16);
                            const bf16x8_t A3 = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + 48 + fj) * 16);
                            g2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, Bf, g2, 0, 0, 0);
                            g3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, Bf, g3, 0, 0, 0);
                        }
                    const bf16x8_t hb1 = hidden(g2, g3, sc2, sf2, sc3, sf3);
                    float4 cd[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const f32x4_t e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdet[t + 1], hb1, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        cd[t] = make_float4(e[0] * s2v[t].x + t2v[t].x, e[1] * s2v[t].y + t2v[t].y, e[2] * s2v[t].z + t2v[t].z, e[3] * s2v[t].w + t2v[t].w);
                    }
                    if (fq < 3) {
                        const int y = ty * TH + 2 * wave + r, x = tx * TW + ch * 16 + fj;
                        const float c0[2] = {c.x, c.z}, c1[2] = {c.y, c.w};
                        const float code[2][6] = {{cd[0].x, cd[0].y, cd[0].z, cd[0].w, cd[1].x, cd[1].y},
                                                  {cd[1].z, cd[1].w, cd[2].x, cd[2].y, cd[2].z, cd[2].w}};
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            // det_candidates_kernel's formula, operation for operation: identical candidates and scores
                            const float mx = fmaxf(c0[u], c1[u]);
                            const float e0 = expf(c0[u] - mx), e1 = expf(c1[u] - mx);
                            const float fg = e1 / (e0 + e1);
                            if (fg >= a.det_thr) {
                                const int pos = atomicAdd(&a.det_counts[n], 1);
                                if (pos < a.det_cap) {
                                    const unsigned m = (unsigned)((y * a.W + x) * 6 + 2 * fq + u);   // anchor index inside the map
                                    reinterpret_cast<unsigned long long *>(a.out)[(size_t)n * a.det_cap + pos] =
                                        ((unsigned long long)(~__float_as_uint(fg)) << 32) | (m << 12) | (unsigned)pos;
                                    float *co = reinterpret_cast<float *>(a.out2) + ((size_t)n * a.det_cap + pos) * 6;
                                    *reinterpret_cast<float2 *>(co) = make_float2(code[u][0], code[u][1]);
                                    *reinterpret_cast<float2 *>(co + 2) = make_float2(code[u][2], code[u][3]);
                                    *reinterpret_cast<float2 *>(co + 4) = make_float2(code[u][4], code[u][5]);
                                }
                            }
                        }
                    }
                }
            }
        } else if constexpr (COUT2 == 0) {
            if (a.x4) {   // 16-byte stores: channel tiles i, i + 1 exchanged between the k-slot quarters (common.h: v2x_store_pair_x4)
#pragma unroll
                for (int i = 0; i < TCO; i += 2) {
                    float4 sc[2], sf[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sc[h] = *reinterpret_cast<const float4 *>(s_ss + (i + h) * 16 + fq * 4);
                        sf[h] = *reinterpret_cast<const float4 *>(s_ss + COUT + (i + h) * 16 + fq * 4);
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), relu_floor);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), relu_floor);
                        }
                        v2x_store_pair_x4(reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + i * 16 + fq * 4, fq,
                                          ox[0], oy[0], ox[1], oy[1]);
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int co = i * 16 + fq * 4;
                const float4 sc = *reinterpret_cast<const float4 *>(s_ss + co);
                const float4 sf = *reinterpret_cast<const float4 *>(s_ss + COUT + co);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                    float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                    float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    o.x = v2x_relu_bf16x2_floor(o.x, relu_floor);   // ReLU on the packed bf16 pairs; the floor is the identity when the layer has none
                    o.y = v2x_relu_bf16x2_floor(o.y, relu_floor);
                    uint16_t *dst = reinterpret_cast<uint16_t *>(a.out) +
                                    ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + co;
                    *reinterpret_cast<uint2 *>(dst) = o;
                }
            }
        } else {
            // hidden = relu(acc*scale+shift) -> bf16 -> directly the B fragments of the 1x1 GEMM
            float4 sc[TCO], sf[TCO];
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                // packed row 16i + 4q + r computes hidden channel kappa = 32*(i>>1) + 8q + 4*(i&1) + r
                const int kappa = 32 * (i >> 1) + 8 * fq + 4 * (i & 1);
                sc[i] = *reinterpret_cast<const float4 *>(a.scale + kappa);
                sf[i] = *reinterpret_cast<const float4 *>(a.shift + kappa);
            }
            // the chained layer's scale/shift for all its channel tiles, loaded ONCE per tile before the first store (a load
            // issued between stores can only be waited for together with them: in-order vmcnt)
            float4 s2v[TCO2 > 0 ? TCO2 : 1], t2v[TCO2 > 0 ? TCO2 : 1];
#pragma unroll
            for (int i2 = 0; i2 < TCO2; ++i2) {
                s2v[i2] = *reinterpret_cast<const float4 *>(a.scale2 + i2 * 16 + fq * 4);
                t2v[i2] = *reinterpret_cast<const float4 *>(a.shift2 + i2 * 16 + fq * 4);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                bf16x8_t hb[COUT / 32];
#pragma unroll
                for (int s = 0; s < COUT / 32; ++s) {
                    float h[8];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int i = 2 * s + half;
                        h[half * 4 + 0] = acc[i][f][0] * sc[i].x + sf[i].x;
                        h[half * 4 + 1] = acc[i][f][1] * sc[i].y + sf[i].y;
                        h[half * 4 + 2] = acc[i][f][2] * sc[i].z + sf[i].z;
                        h[half * 4 + 3] = acc[i][f][3] * sc[i].w + sf[i].w;
                    }
                    uint4 p;
                    p.x = pack_bf16x2(h[0], h[1]);
                    p.y = pack_bf16x2(h[2], h[3]);
                    p.z = pack_bf16x2(h[4], h[5]);
                    p.w = pack_bf16x2(h[6], h[7]);
                    p.x = v2x_relu_bf16x2_floor(p.x, relu_floor);
                    p.y = v2x_relu_bf16x2_floor(p.y, relu_floor);
                    p.z = v2x_relu_bf16x2_floor(p.z, relu_floor);
                    p.w = v2x_relu_bf16x2_floor(p.w, relu_floor);
                    hb[s] = __builtin_bit_cast(bf16x8_t, p);
                }
                const int y = ty * TH + 2 * wave + (f >> 1), x = tx * TW + (f & 1) * 16 + fj;
                const size_t pix = (size_t)(n * a.H + y) * a.W + x;
#pragma unroll
                for (int i2 = 0; i2 < TCO2; ++i2) {
                    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < COUT / 32; ++s)
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2][s], hb[s], d, 0, 0, 0);
                    const int co = i2 * 16 + fq * 4;
                    if (co >= a.cout2_real) continue;
                    const float4 s2 = s2v[i2], t2 = t2v[i2];
                    float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                    if (a.relu2) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                        v2 = fmaxf(v2, 0.f);
                        v3 = fmaxf(v3, 0.f);
                    }
                    if constexpr (EPI2 == 2) {
                        const bool second = a.split > 0 && co >= a.split;
                        float *dst = second ? reinterpret_cast<float *>(a.out2) + pix * a.out2_cstride + (co - a.split)
                                            : reinterpret_cast<float *>(a.out) + pix * a.out_cstride + a.out_coff + co;
                        *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                    } else {
                        uint2 o;
                        o.x = pack_bf16x2(v0, v1);
                        o.y = pack_bf16x2(v2, v3);
                        uint16_t *dst = reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + co;
                        *reinterpret_cast<uint2 *>(dst) = o;
                    }
                }
            }
        }
        if constexpr (DB) {
            cur ^= 1;
        } else {
            // everyone is done READING the patch (fragment reads were consumed by MFMAs); the output stores stay
            // in flight across this raw barrier and drain under the next tile's patch load
            __builtin_amdgcn_s_barrier();
        }
    }
}

// (the chained 64 -> 64 -> 64 instance -- the odd-tile-count fallback of the ping-pong kernel -- holds 158 KiB of LDS, one workgroup per
// CU: it may use the registers of a whole SIMD, where 256 left it 48 bytes of scratch)
template <int C0, int C1, int COUT, int COUT2, int EPI2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((C0 == 0 && C1 == 64 && COUT2 == 64) ? 1 : 2, 2))) void conv3x3_halo_kernel(const HaloArgs a) {
    conv3x3_halo_body<C0, C1, COUT, COUT2, EPI2, true>(a);
}

// single-buffer form: up to 4 workgroups (16 waves) per CU
template <int C0, int C1, int COUT, int COUT2, int EPI2, bool BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3x3_halo_sb_kernel(const HaloArgs a) {
    conv3x3_halo_body<C0, C1, COUT, COUT2, EPI2, false, BITS>(a);
}

// ---- 8-wave "ping-pong" form (conv8_1) -----------------------------------------------------------------
// conv8_1 (cat(up(x_7) [64 ch], x [32 ch]) -> 32) keeps 54 KiB of weights resident; with a double-buffered 36-KiB patch
// that is 126 KiB, i.e. ONE 4-wave workgroup per CU = one wave per SIMD.  Per tile such a wave spends ~3.5k cycles issuing
// MFMAs and as much again on everything that cannot overlap them when nobody else is resident: the fragment reads'
// issue slots, 8 output stores with their conversions, ~10 LDS-DMA instructions (60-185 issue cycles each next to
// MFMAs), counted waits and the barrier (SQ counters of the 4-wave form: MfmaUtil 41.6 %, 4.0 waves/CU).
// Here ONE 512-thread workgroup per CU holds ONE copy of the weights and two SINGLE-buffered patches, one per
// 4-wave group; the groups own consecutive tiles (x-neighbours) and run half a period out of phase:
//
//      interval   2k                          2k+1
//      group 0    C(k)  MFMAs of its tile     E(k): stores of tile k;  L(k+1): patch DMAs of its next tile, wait
//      group 1    E(k-1), L(k), wait          C(k)
//
// with one s_barrier between intervals, so that on every SIMD the MFMA stream of one wave runs beside the store /
// DMA / wait phase of the other.  Hazards: a group's patch is overwritten (L) only in the interval after its own
// C finished reading it (barrier in between); C(k) of a group starts after the barrier that closes its own L(k), before
// which every wave waited for its own DMA pieces (vmcnt(N_STORES): the DMAs are issued BEFORE the stores of E, so the
// stores may stay in flight).  The accumulators live across the barrier from C to E.  K order, fragment mapping and
// epilogue arithmetic are those of conv3x3_halo_kernel: results are bit-identical.
template <int C0, int C1, int COUT, int COUT2 = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_pp_kernel(const HaloArgs a) {
    constexpr int SPP0 = C0 / 8, SPP1 = C1 / 8;
    constexpr int NS1 = round64(PH * PW * SPP1);
    constexpr int NS0 = C0 ? round64(PH0 * PW0 * SPP0) : 0;
    constexpr int PATCH_BYTES = (NS0 + NS1) * 16;
    constexpr int KSLOTS = 9 * (SPP0 + SPP1);
    constexpr int W_BYTES = KSLOTS * COUT * 16;
    constexpr int TCO = COUT / 16;
    static_assert(C0 % 32 == 0 && C1 % 32 == 0 && (TCO == 2 || TCO == 4), "ping-pong form: 32 or 64 output channels");
    static_assert(W_BYTES % 1024 == 0, "weights are moved 1 KiB per wave instruction");
    constexpr int TCO2 = COUT2 / 16;
    static_assert(COUT2 == 0 || (COUT2 == 64 && COUT == 64), "chained 1x1: 64 -> 64 (conv1_2 -> conv3d_1)");
    constexpr int N_STORES = (COUT2 ? TCO2 : TCO) * 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;   // v2x_relu_bf16x2_floor: identity when the layer has no ReLU
    char *s_w = smem;
    char *s_patch = smem + W_BYTES;                       // [group][PATCH_BYTES]
    float *s_ss = reinterpret_cast<float *>(smem + W_BYTES + 2 * PATCH_BYTES);
    for (int i = threadIdx.x; i < COUT; i += 512) {
        s_ss[i] = a.scale[i];
        s_ss[COUT + i] = a.shift[i];
    }
    if constexpr (COUT2 > 0) {   // [scale | shift | scale2 | shift2]
        for (int i = threadIdx.x; i < COUT2; i += 512) {
            s_ss[2 * COUT + i] = a.scale2[i];
            s_ss[2 * COUT + COUT2 + i] = a.shift2[i];
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, wv = wave & 3;
    const int fj = lane & 15, fq = lane >> 4;
    char *pb = s_patch + grp * PATCH_BYTES;

    for (int off = wave * 1024; off < W_BYTES; off += 8192)
        glds16h(reinterpret_cast<const char *>(a.w) + off + lane * 16, s_w + off);

    // per-lane DMA tables (see conv3x3_halo_body): piece t of wave wv of a group covers patch slots [wv*64 + t*256, +64)
    // (the chained form recomputes them per piece instead: its store phase has slack next to the other group's MFMA interval,
    // its register budget has none)
    constexpr bool TABLES = (COUT2 == 0);
    constexpr int NP1 = (NS1 + 255) / 256, NP0 = (NS0 + 255) / 256;
    int tb1[TABLES ? NP1 : 1], tb0[NP0 ? NP0 : 1];
#pragma unroll
    for (int t = 0; t < (TABLES ? NP1 : 0); ++t) {
        const int L = wv * 64 + t * 256 + lane;
        const int pix = L / SPP1, phys = L - pix * SPP1;
        const int pr = pix / PW, pc = pix - pr * PW;
        tb1[t] = pix < PH * PW ? (((pr * a.W + pc) * C1 + swz<SPP1>(phys, pc) * 8) | (pr << 20) | (pc << 24)) : -1;
    }
    if constexpr (C0 > 0) {
        constexpr int S0 = SPP0 ? SPP0 : 1;
#pragma unroll
        for (int t = 0; t < NP0; ++t) {
            const int L = wv * 64 + t * 256 + lane;
            const int pix = L / S0, phys = L - pix * S0;
            const int pr = pix / PW0, pc = pix - pr * PW0;
            tb0[t] = pix < PH0 * PW0 ? (((pr * (a.W >> 1) + pc) * C0 + swz<(SPP0 ? SPP0 : 4)>(phys, pc) * 8) | (pr << 20) | (pc << 24)) : -1;
        }
    }

    const int txy = a.tiles_x * a.tiles_y;
    auto coords = [&](int tile, int &n, int &y0, int &x0) {
        n = tile / txy;
        const int r = tile - n * txy;
        const int ty = r / a.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * a.tiles_x) * TW;
    };
    auto load_patch = [&](int tile) {   // this group's patch <- tile; <= NP1 + NP0 DMAs per wave
        int n, y0, x0;
        coords(tile, n, y0, x0);
        if constexpr (TABLES) {
            const unsigned base = ((unsigned)(n * a.H + y0 - 1) * (unsigned)a.W + (unsigned)(x0 - 1)) * (unsigned)C1;
#pragma unroll
            for (int t = 0; t < NP1; ++t) {
                if (wv * 64 + t * 256 >= NS1) break;   // wave-uniform
                const int y = y0 - 1 + ((tb1[t] >> 20) & 15), x = x0 - 1 + ((tb1[t] >> 24) & 63);
                const bool ok = tb1[t] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                glds16h(ok ? (const void *)(a.in1 + (base + (unsigned)(tb1[t] & 0xfffff))) : (const void *)g_zero_page_h,
                        pb + NS0 * 16 + (wv * 64 + t * 256) * 16);
            }
        } else {
            for (int base = wv * 64; base < NS1; base += 256) {
                const int L = base + lane;
                const int pix = L / SPP1, phys = L - pix * SPP1;
                const int pr = pix / PW, pc = pix - pr * PW;
                const int y = y0 - 1 + pr, x = x0 - 1 + pc;
                const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const unsigned off = ((unsigned)(n * a.H + y) * (unsigned)a.W + (unsigned)x) * (unsigned)C1 + swz<SPP1>(phys, pc) * 8;
                glds16h(ok ? (const void *)(a.in1 + off) : (const void *)g_zero_page_h, pb + NS0 * 16 + base * 16);
            }
        }
        if constexpr (C0 > 0) {
            const int Hs = a.H >> 1, Ws = a.W >> 1;
            const unsigned base = ((unsigned)(n * Hs + (y0 >> 1) - 1) * (unsigned)Ws + (unsigned)((x0 >> 1) - 1)) * (unsigned)C0;
#pragma unroll
            for (int t = 0; t < NP0; ++t) {
                if (wv * 64 + t * 256 >= NS0) break;   // wave-uniform
                const int y = (y0 >> 1) - 1 + ((tb0[t] >> 20) & 15), x = (x0 >> 1) - 1 + ((tb0[t] >> 24) & 63);
                const bool ok = tb0[t] >= 0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
                glds16h(ok ? (const void *)(a.in0 + (base + (unsigned)(tb0[t] & 0xfffff))) : (const void *)g_zero_page_h,
                        pb + (wv * 64 + t * 256) * 16);
            }
        }
    };

    f32x4_t acc[TCO][4];
    constexpr int KC0 = C0 / 32, KC1 = C1 / 32, KC = KC0 + KC1, NG = 3 * KC;
    struct Frags {
        bf16x8_t A[3][TCO];
        bf16x8_t B[8];
    };
    auto load_group = [&](int g, Frags &F) {
        const int kx = g / KC, kk = g - kx * KC;
        if (kk < KC0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int kslot = (ky * 3 + kx) * (SPP0 + SPP1) + kk * 4 + fq;
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    F.A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + i * 16 + fj) * 16);
            }
#pragma unroll
            for (int hr = 0; hr < 3; ++hr)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int pr = wv + hr;
                    const int pc = ((ch * 16 + fj + kx - 1) >> 1) + 1;
                    F.B[hr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(pb + ((pr * PW0 + pc) * (SPP0 ? SPP0 : 1) + swz<(SPP0 ? SPP0 : 4)>(kk * 4 + fq, pc)) * 16);
                }
        } else {
            const int kc = kk - KC0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int kslot = (ky * 3 + kx) * (SPP0 + SPP1) + SPP0 + kc * 4 + fq;
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    F.A[ky][i] = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + i * 16 + fj) * 16);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int pr = 2 * wv + rr;
                    const int pc = ch * 16 + fj + kx;
                    F.B[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(pb + NS0 * 16 + ((pr * PW + pc) * SPP1 + swz<SPP1>(kc * 4 + fq, pc)) * 16);
                }
        }
    };
    auto mma_group = [&](int g, const Frags &F) {
        const bool half = (g % KC) < KC0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int r = f >> 1, ch = f & 1;
                    const int row = half ? (((r + ky - 1) >> 1) + 1) : (r + ky);
                    acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.A[ky][i], F.B[row * 2 + ch], acc[i][f], 0, 0, 0);
                }
    };
    auto compute = [&]() {
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (TCO == 2) {   // fragments of group g+1 are read while the MFMAs of group g issue (register double-buffer)
            Frags fr[2];
            load_group(0, fr[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, fr[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(g, fr[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // 64 output channels: the group's 8 pixel fragments in one register set, its weight fragments per tap row ky in
            // two alternating sets of 4 (the reads of row ky+1 are issued before the 16 MFMAs of row ky): 64 fragment
            // registers instead of 80 -- what the chained epilogue needs to stay inside the 256-VGPR budget
            static_assert(C0 == 0 || TCO == 2, "the 64-channel form has one full-resolution source");
            bf16x8_t B[8], A[2][TCO];
            auto load_A = [&](int g, int ky, bf16x8_t (&Ak)[TCO]) {
                const int kx = g / KC, kc = g - kx * KC;
                const int kslot = (ky * 3 + kx) * (SPP0 + SPP1) + SPP0 + kc * 4 + fq;
#pragma unroll
                for (int i = 0; i < TCO; ++i) Ak[i] = *reinterpret_cast<const bf16x8_t *>(s_w + (kslot * COUT + i * 16 + fj) * 16);
            };
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int kx = g / KC, kc = g - kx * KC;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int pr = 2 * wv + rr, pc = ch * 16 + fj + kx;
                        B[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(pb + NS0 * 16 + ((pr * PW + pc) * SPP1 + swz<SPP1>(kc * 4 + fq, pc)) * 16);
                    }
                load_A(g, 0, A[0]);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (ky < 2) load_A(g, ky + 1, A[(ky + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ky & 1][i], B[((f >> 1) + ky) * 2 + (f & 1)], acc[i][f], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    auto epilogue = [&](int tile) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        // opaque zero: the scale/shift reads below are loop-invariant LDS loads; hoisted out of the interval loop they would
        // sit in 32-64 registers across the MFMA phase (the chained form spilled)
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *ss = s_ss + zo;
        if constexpr (COUT2 > 0) {
            // chained 1x1 (rows of the first GEMM in kappa order: a lane's accumulators ARE its B fragment of the second GEMM,
            // see conv3x3_halo_body).  The second GEMM's weights (8 KiB, L1/L2-resident) are read from global memory -- the LDS
            // is full -- and this whole phase runs beside the other group's MFMA interval.
            bf16x8_t w2f[TCO2][COUT / 32];
            int w2off = fj * COUT + fq * 8;
            asm volatile("" : "+v"(w2off));   // opaque: the loads are loop-invariant and would be hoisted across the MFMA interval
                                              // (32 more live registers on top of the 234 of the compute phase: it spilled)
#pragma unroll
            for (int i2 = 0; i2 < TCO2; ++i2)
#pragma unroll
                for (int ks = 0; ks < COUT / 32; ++ks)
                    w2f[i2][ks] = *reinterpret_cast<const bf16x8_t *>(a.w2 + w2off + i2 * 16 * COUT + ks * 32);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // w2 fragments (and the patch DMAs issued before them) have landed
            // one pixel fragment at a time: its hidden activations (8 registers) -> the four output-channel tiles
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                bf16x8_t hb[COUT / 32];
#pragma unroll
                for (int ks = 0; ks < COUT / 32; ++ks) {
                    float h[8];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * ks + hf;
                        const int kappa = 32 * (i >> 1) + 8 * fq + 4 * (i & 1);
                        const float4 sc = *reinterpret_cast<const float4 *>(ss + kappa);
                        const float4 sf = *reinterpret_cast<const float4 *>(ss + COUT + kappa);
                        h[hf * 4 + 0] = acc[i][f][0] * sc.x + sf.x;
                        h[hf * 4 + 1] = acc[i][f][1] * sc.y + sf.y;
                        h[hf * 4 + 2] = acc[i][f][2] * sc.z + sf.z;
                        h[hf * 4 + 3] = acc[i][f][3] * sc.w + sf.w;
                    }
                    uint4 p;
                    p.x = pack_bf16x2(h[0], h[1]);
                    p.y = pack_bf16x2(h[2], h[3]);
                    p.z = pack_bf16x2(h[4], h[5]);
                    p.w = pack_bf16x2(h[6], h[7]);
                    p.x = v2x_relu_bf16x2_floor(p.x, relu_floor);
                    p.y = v2x_relu_bf16x2_floor(p.y, relu_floor);
                    p.z = v2x_relu_bf16x2_floor(p.z, relu_floor);
                    p.w = v2x_relu_bf16x2_floor(p.w, relu_floor);
                    hb[ks] = __builtin_bit_cast(bf16x8_t, p);
                }
                const int y = y0 + 2 * wv + (f >> 1), x = x0 + (f & 1) * 16 + fj;
                uint16_t *prow = reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff;
                if (a.x4) {   // 16-byte stores: output tiles i2, i2 + 1 exchanged between the k-slot quarters (common.h: v2x_store_pair_x4)
                    const uint32_t floor2 = a.relu2 ? 0u : 0x80008000u;
#pragma unroll
                    for (int i2 = 0; i2 + 1 < TCO2; i2 += 2) {
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int co = (i2 + h) * 16 + fq * 4;
                            const float4 s2 = *reinterpret_cast<const float4 *>(ss + 2 * COUT + co);
                            const float4 t2 = *reinterpret_cast<const float4 *>(ss + 2 * COUT + COUT2 + co);
                            f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < COUT / 32; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2 + h][ks], hb[ks], d, 0, 0, 0);
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(d[0] * s2.x + t2.x, d[1] * s2.y + t2.y), floor2);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(d[2] * s2.z + t2.z, d[3] * s2.w + t2.w), floor2);
                        }
                        v2x_store_pair_x4(prow + i2 * 16 + fq * 4, fq, ox[0], oy[0], ox[1], oy[1]);
                    }
                    continue;
                }
#pragma unroll
                for (int i2 = 0; i2 < TCO2; ++i2) {
                    const int co = i2 * 16 + fq * 4;
                    const float4 s2 = *reinterpret_cast<const float4 *>(ss + 2 * COUT + co);
                    const float4 t2 = *reinterpret_cast<const float4 *>(ss + 2 * COUT + COUT2 + co);
                    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < COUT / 32; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2][ks], hb[ks], d, 0, 0, 0);
                    float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                    if (a.relu2) {
                        v0 = fmaxf(v0, 0.f);
                        v1 = fmaxf(v1, 0.f);
                        v2 = fmaxf(v2, 0.f);
                        v3 = fmaxf(v3, 0.f);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    *reinterpret_cast<uint2 *>(prow + co) = o;
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
                    sc[h] = *reinterpret_cast<const float4 *>(ss + (i + h) * 16 + fq * 4);
                    sf[h] = *reinterpret_cast<const float4 *>(ss + COUT + (i + h) * 16 + fq * 4);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int y = y0 + 2 * wv + (f >> 1), x = x0 + (f & 1) * 16 + fj;
                    uint32_t ox[2], oy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), relu_floor);
                        oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), relu_floor);
                    }
                    v2x_store_pair_x4(reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + i * 16 + fq * 4, fq,
                                      ox[0], oy[0], ox[1], oy[1]);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = i * 16 + fq * 4;
            const float4 sc = *reinterpret_cast<const float4 *>(ss + co);
            const float4 sf = *reinterpret_cast<const float4 *>(ss + COUT + co);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int y = y0 + 2 * wv + (f >> 1), x = x0 + (f & 1) * 16 + fj;
                float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                o.x = v2x_relu_bf16x2_floor(o.x, relu_floor);   // ReLU on the packed bf16 pairs; the floor is the identity when the layer has none
                o.y = v2x_relu_bf16x2_floor(o.y, relu_floor);
                uint16_t *dst = reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + co;
                *reinterpret_cast<uint2 *>(dst) = o;
            }
        }
    };

    // pairs of tiles (2p, 2p + 1): group g owns tile 2p + g;  p = blockIdx.x + k * gridDim.x
    const int n_pairs = a.n_tiles >> 1;
    const v2x_tile_walk walk = v2x_xcd_tile_walk(n_pairs, a.xcd_walk);      // (XCD-contiguous pairs: common.h)
    const int K = walk.first < walk.end ? (walk.end - walk.first + walk.step - 1) / walk.step : 0;
    auto tile_of = [&](int k) { return 2 * (walk.first + k * walk.step) + grp; };

    if (grp == 0) load_patch(tile_of(0));
    __syncthreads();   // weights, scale/shift and group 0's first patch have landed

    // interval i: group g computes tile k = i >> 1 when (i & 1) == g; otherwise it stores tile m (computed in the previous
    // interval) and fetches the patch of tile m + 1 (computed in the next one), m = (i - g - 1) / 2.  ONE call site per role:
    // two inlined copies of compute() cost ~45 spilled registers, and scratch traffic would also corrupt the vmcnt counts.
    for (int i = 0; i <= 2 * K; ++i) {
        if ((i & 1) == grp) {
            if ((i >> 1) < K) compute();
        } else {
            const int m = (i - grp - 1) / 2;               // -1 in group 1's first interval (truncating division)
            if (m + 1 < K) load_patch(tile_of(m + 1));     // its previous compute is behind the last barrier
            if (m >= 0) {
                epilogue(tile_of(m));
                // the DMAs are OLDER than the N_STORES stores (half as many in the 16-byte form, a.x4): they have landed, the stores may stay in flight
                if constexpr (N_STORES == 8) {
                    if (a.x4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    if (a.x4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(N_STORES == 8 || N_STORES == 16, "the counted waits above know 8 or 16 stores per tile and wave");
}

template <int C0, int C1, int COUT, int COUT2 = 0>
static int launch_halo_pp(const HaloArgs &a, hipStream_t s) {
    constexpr int NS1 = round64(PH * PW * (C1 / 8));
    constexpr int NS0 = C0 ? round64(PH0 * PW0 * (C0 / 8)) : 0;
    constexpr int smem = 9 * (C0 + C1) / 8 * COUT * 16 + 2 * (NS0 + NS1) * 16 + 2 * COUT * 4 + 2 * COUT2 * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_halo_pp_kernel<C0, C1, COUT, COUT2>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    int grid = 256;                       // one 8-wave workgroup per CU, persistent over tile pairs
    if (grid > a.n_tiles / 2) grid = a.n_tiles / 2;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_halo_pp_kernel");
    return V2X_OK;
}

// ---- parity-class ("sub-pixel") form of the decoder's upsample -> concat -> 3x3 layers (conv8_1) ---------------------------------
// The x2 NEAREST-upsampled half of the operand takes only (H/2) x (W/2) distinct values: for an output pixel (2Y + py, 2X + px) the three
// tap rows ky of the 3x3 read the half-resolution rows {Y-1, Y, Y} (py = 0) or {Y, Y, Y+1} (py = 1), and likewise the columns.  Per output
// PARITY CLASS (py, px) the up-half of the layer is therefore a 2 x 2-tap convolution on the half-resolution map whose weights are sums of
// the 3x3's: W'[py][px][a][b] = sum over ky in G(py, a), kx in G(px, b) of W[ky][kx], G(0,0) = {0}, G(0,1) = {1,2}, G(1,0) = {0,1},
// G(1,1) = {2}  --  C0 x 4 + C1 x 9 instead of (C0 + C1) x 9 MACs per pixel (conv8_1: -37 %), an exact identity in real arithmetic.  The sums
// are formed in fp32 from the fp32 parameters and rounded to bf16 ONCE by the packer (w_layout 3: v2x_pack_conv / packing.pack_conv_halo_parity).
// The oracle is NOT changed: tests compare this kernel with the unmodified fp32 9-tap layer (tolerance = bf16 rounding of the weights, as for
// the 9-tap kernel) and, as a kernel check, with a torch evaluation of the same bf16 operands.
// Tiling: the 8-wave ping-pong kernel above with WAVE = CLASS: wave wv of a group owns class (py, px) = (wv >> 1, wv & 1) of the group's 8 x 32 tile,
// i.e. the 4 x 16 pixels (2Y + py, 2X + px) = 4 fragments of 16 lanes (lane = X), so that a weight fragment still stands in front of 4 MFMAs (split
// the other way -- every wave all four classes -- each would serve ONE).  The full-resolution (skip) half keeps its 9 taps, read with a column stride
// of 2 pixels: its patch is stored as two COLUMN-PARITY planes (the DMA descriptors permute the pixels on their way in), so a fragment is again 16
// consecutive pixels of one plane row and the swizzle of the dense form applies.  Per tile and wave: 136 MFMAs from 81 fragment reads (216 from 114).
constexpr int PWH = PW / 2;   // columns per parity plane of the full-resolution patch

template <int C0, int C1, int COUT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_ppc_kernel(const HaloArgs a) {
    constexpr int SPP0 = C0 / 8, SPP1 = C1 / 8;
    constexpr int NS1 = round64(2 * PH * PWH * SPP1);
    constexpr int NS0 = round64(PH0 * PW0 * SPP0);
    constexpr int PATCH_BYTES = (NS0 + NS1) * 16;
    constexpr int W_UP_BYTES = 16 * SPP0 * COUT * 16;      // [class][tap a*2+b][k-slot][COUT][8]
    constexpr int W_BYTES = W_UP_BYTES + 9 * SPP1 * COUT * 16;   // + [tap ky*3+kx][k-slot][COUT][8]
    constexpr int TCO = COUT / 16;
    constexpr int KC0 = C0 / 32, KC1 = C1 / 32;
    static_assert(C0 % 32 == 0 && C1 % 32 == 0 && C0 > 0 && C1 > 0 && TCO == 2, "parity-class form: two sources, 32 output channels");
    static_assert(W_BYTES % 1024 == 0, "weights are moved 1 KiB per wave instruction");
    constexpr int N_STORES = TCO * 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;
    char *s_w = smem;
    char *s_patch = smem + W_BYTES;                       // [group][PATCH_BYTES]
    float *s_ss = reinterpret_cast<float *>(smem + W_BYTES + 2 * PATCH_BYTES);
    for (int i = threadIdx.x; i < COUT; i += 512) {
        s_ss[i] = a.scale[i];
        s_ss[COUT + i] = a.shift[i];
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, wv = wave & 3;
    const int py = wv >> 1, px = wv & 1;                          // the wave's parity class
    const int fj = lane & 15, fq = lane >> 4;
    char *pb = s_patch + grp * PATCH_BYTES;

    for (int off = wave * 1024; off < W_BYTES; off += 8192)
        glds16h(reinterpret_cast<const char *>(a.w) + off + lane * 16, s_w + off);

    // per-lane DMA tables (see conv3x3_halo_body).  Full-resolution patch: LDS slot L -> (plane q, row pr, plane column cc, physical slot);
    // the pixel it holds is patch column pc = 2 cc + q.
    constexpr int NP1 = (NS1 + 255) / 256, NP0 = (NS0 + 255) / 256;
    int tb1[NP1], tb0[NP0];
#pragma unroll
    for (int t = 0; t < NP1; ++t) {
        const int L = wv * 64 + t * 256 + lane;
        const int pix = L / SPP1, phys = L - pix * SPP1;
        const int q = pix / (PH * PWH), rem = pix - q * (PH * PWH);
        const int pr = rem / PWH, cc = rem - pr * PWH;
        const int pc = 2 * cc + q;
        tb1[t] = pix < 2 * PH * PWH ? (((pr * a.W + pc) * C1 + swz<SPP1>(phys, cc) * 8) | (pr << 20) | (pc << 24)) : -1;
    }
#pragma unroll
    for (int t = 0; t < NP0; ++t) {
        const int L = wv * 64 + t * 256 + lane;
        const int pix = L / SPP0, phys = L - pix * SPP0;
        const int pr = pix / PW0, pc = pix - pr * PW0;
        tb0[t] = pix < PH0 * PW0 ? (((pr * (a.W >> 1) + pc) * C0 + swz<SPP0>(phys, pc) * 8) | (pr << 20) | (pc << 24)) : -1;
    }

    const int txy = a.tiles_x * a.tiles_y;
    auto coords = [&](int tile, int &n, int &y0, int &x0) {
        n = tile / txy;
        const int r = tile - n * txy;
        const int ty = r / a.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * a.tiles_x) * TW;
    };
    auto load_patch = [&](int tile) {   // this group's patch <- tile; NP1 + NP0 DMAs per wave at most
        int n, y0, x0;
        coords(tile, n, y0, x0);
        {
            const unsigned base = ((unsigned)(n * a.H + y0 - 1) * (unsigned)a.W + (unsigned)(x0 - 1)) * (unsigned)C1;
#pragma unroll
            for (int t = 0; t < NP1; ++t) {
                if (wv * 64 + t * 256 >= NS1) break;   // wave-uniform
                const int y = y0 - 1 + ((tb1[t] >> 20) & 15), x = x0 - 1 + ((tb1[t] >> 24) & 63);
                const bool ok = tb1[t] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                glds16h(ok ? (const void *)(a.in1 + (base + (unsigned)(tb1[t] & 0xfffff))) : (const void *)g_zero_page_h,
                        pb + NS0 * 16 + (wv * 64 + t * 256) * 16);
            }
        }
        {
            const int Hs = a.H >> 1, Ws = a.W >> 1;
            const unsigned base = ((unsigned)(n * Hs + (y0 >> 1) - 1) * (unsigned)Ws + (unsigned)((x0 >> 1) - 1)) * (unsigned)C0;
#pragma unroll
            for (int t = 0; t < NP0; ++t) {
                if (wv * 64 + t * 256 >= NS0) break;   // wave-uniform
                const int y = (y0 >> 1) - 1 + ((tb0[t] >> 20) & 15), x = (x0 >> 1) - 1 + ((tb0[t] >> 24) & 63);
                const bool ok = tb0[t] >= 0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
                glds16h(ok ? (const void *)(a.in0 + (base + (unsigned)(tb0[t] & 0xfffff))) : (const void *)g_zero_page_h,
                        pb + (wv * 64 + t * 256) * 16);
            }
        }
    };

    // K walk: groups g < 2 KC0 = (up chunk kc = g >> 1, class tap column b = g & 1): the two class tap rows a share their 5 half-resolution patch
    // rows; then (tap column kx, skip chunk kc): the three tap rows ky share the 9 full-resolution rows py .. py + 8 of one column-parity plane.
    f32x4_t acc[TCO][4];
    constexpr int NG = 2 * KC0 + 3 * KC1;
    struct Frags {
        bf16x8_t A[3 * TCO];
        bf16x8_t B[9];
    };
    auto load_group = [&](int g, Frags &F) {
        if (g < 2 * KC0) {
            const int kc = g >> 1, b = g & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    F.A[t * TCO + i] = *reinterpret_cast<const bf16x8_t *>(s_w + ((((wv * 4 + t * 2 + b) * SPP0 + kc * 4 + fq) * COUT) + i * 16 + fj) * 16);
            const int pc = fj + b + px;
#pragma unroll
            for (int r = 0; r < 5; ++r)
                F.B[r] = *reinterpret_cast<const bf16x8_t *>(pb + (((py + r) * PW0 + pc) * SPP0 + swz<SPP0>(kc * 4 + fq, pc)) * 16);
        } else {
            const int h = g - 2 * KC0, kx = h / KC1, kc = h - kx * KC1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    F.A[ky * TCO + i] = *reinterpret_cast<const bf16x8_t *>(s_w + W_UP_BYTES + ((((ky * 3 + kx) * SPP1 + kc * 4 + fq) * COUT) + i * 16 + fj) * 16);
            const int q = (px + kx) & 1, cc = fj + ((px + kx) >> 1);
#pragma unroll
            for (int r = 0; r < 9; ++r)
                F.B[r] = *reinterpret_cast<const bf16x8_t *>(pb + NS0 * 16 + ((((q * PH) + py + r) * PWH + cc) * SPP1 + swz<SPP1>(kc * 4 + fq, cc)) * 16);
        }
    };
    auto mma_group = [&](int g, const Frags &F) {
        if (g < 2 * KC0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.A[t * TCO + i], F.B[f + t], acc[i][f], 0, 0, 0);
        } else {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.A[ky * TCO + i], F.B[2 * f + ky], acc[i][f], 0, 0, 0);
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        Frags fr[2];   // fragments of group g+1 are read while the MFMAs of group g issue (register double-buffer)
        load_group(0, fr[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1, fr[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mma_group(g, fr[g & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto epilogue = [&](int tile) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        int zo = 0;
        asm volatile("" : "+v"(zo));   // opaque zero: keeps the loop-invariant scale/shift reads out of the MFMA interval (see the kernel above)
        const float *ss = s_ss + zo;
        float4 sc[2], sf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sc[h] = *reinterpret_cast<const float4 *>(ss + h * 16 + fq * 4);
            sf[h] = *reinterpret_cast<const float4 *>(ss + COUT + h * 16 + fq * 4);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int y = y0 + 2 * f + py, x = x0 + 2 * fj + px;
            uint32_t ox[2], oy[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[h][f][0] * sc[h].x + sf[h].x, acc[h][f][1] * sc[h].y + sf[h].y), relu_floor);
                oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[h][f][2] * sc[h].z + sf[h].z, acc[h][f][3] * sc[h].w + sf[h].w), relu_floor);
            }
            uint16_t *prow = reinterpret_cast<uint16_t *>(a.out) + ((size_t)(n * a.H + y) * a.W + x) * a.out_cstride + a.out_coff + fq * 4;
            if (a.x4) {
                v2x_store_pair_x4(prow, fq, ox[0], oy[0], ox[1], oy[1]);
            } else {
                *reinterpret_cast<uint2 *>(prow) = make_uint2(ox[0], oy[0]);
                *reinterpret_cast<uint2 *>(prow + 16) = make_uint2(ox[1], oy[1]);
            }
        }
    };

    // pairs of tiles (2p, 2p + 1): group g owns tile 2p + g (an odd tile count leaves group 1 without a tile in the last pair)
    const int n_pairs = (a.n_tiles + 1) >> 1;
    const v2x_tile_walk walk = v2x_xcd_tile_walk(n_pairs, a.xcd_walk);
    const int K = walk.first < walk.end ? (walk.end - walk.first + walk.step - 1) / walk.step : 0;
    auto tile_of = [&](int k) { return 2 * (walk.first + k * walk.step) + grp; };
    auto has = [&](int k) { return k < K && tile_of(k) < a.n_tiles; };

    if (grp == 0 && has(0)) load_patch(tile_of(0));
    __syncthreads();   // weights, scale/shift and group 0's first patch have landed

    for (int i = 0; i <= 2 * K; ++i) {
        if ((i & 1) == grp) {
            if (has(i >> 1)) compute();
        } else {
            const int m = (i - grp - 1) / 2;               // -1 in group 1's first interval (truncating division)
            if (has(m + 1)) load_patch(tile_of(m + 1));
            if (m >= 0 && has(m)) {
                epilogue(tile_of(m));
                // the DMAs are OLDER than the stores: they have landed, the stores may stay in flight
                if (a.x4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(N_STORES == 8, "the counted waits above know 8 (4 in the 16-byte form) stores per tile and wave");
}

template <int C0, int C1, int COUT>
static int launch_halo_ppc(const HaloArgs &a, hipStream_t s) {
    constexpr int NS1 = round64(2 * PH * PWH * (C1 / 8));
    constexpr int NS0 = round64(PH0 * PW0 * (C0 / 8));
    constexpr int smem = (16 * C0 + 9 * C1) / 8 * COUT * 16 + 2 * (NS0 + NS1) * 16 + 2 * COUT * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_halo_ppc_kernel<C0, C1, COUT>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    int grid = 256;                       // one 8-wave workgroup per CU, persistent over tile pairs
    if (grid > (a.n_tiles + 1) / 2) grid = (a.n_tiles + 1) / 2;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_halo_ppc_kernel");
    return V2X_OK;
}

// ---- host side ---------------------------------------------------------------------------------
template <int C0, int C1, int COUT, int COUT2, int EPI2, bool BITS>
static int launch_halo_sb(const HaloArgs &a, hipStream_t s) {
    static_assert(C0 == 0, "single-buffer form: one source");
    // bit-grid input: the patch is written with ds_write for exactly PH*PW pixels, so it is allocated WITHOUT the 768-B
    // wave padding the LDS-DMA fill needs -- 39.25 KiB, which lets a 4th workgroup fit beside the slack rule below
    // (measured per launch at 320 maps: 2 / 3 / 4 workgroups per CU = 684 / 568 / 520 us).  The bf16-input layers get
    // SLOWER with a 4th workgroup (628 / 577 / 649 us) and keep the padded 40-KiB allocation = 3 per CU.
    constexpr int smem = 9 * C1 / 8 * COUT * 16 + (BITS ? PH * PW * (C1 / 8) : round64(PH * PW * (C1 / 8))) * 16 +
                         (COUT2 == 0 ? 2 * COUT * 4 : 0);   // + scale/shift
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_halo_sb_kernel<C0, C1, COUT, COUT2, EPI2, BITS>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    int per_cu = (160 * 1024 - 2048) / smem;  // leave a little LDS slack: exactly-full allocations may not co-reside
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = 256 * per_cu;
    if (grid > a.n_tiles) grid = a.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_halo_sb_kernel");
    return V2X_OK;
}

template <int C0, int C1, int COUT, int COUT2, int EPI2>
static int launch_halo(const HaloArgs &a, hipStream_t s) {
    constexpr int NS1 = round64(PH * PW * (C1 / 8));
    constexpr int NS0 = C0 ? round64(PH0 * PW0 * (C0 / 8)) : 0;
    constexpr int smem = 9 * (C0 + C1) / 8 * COUT * 16 + 2 * (NS0 + NS1) * 16 + (COUT2 == 0 ? 2 * COUT * 4 : 0);
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_halo_kernel<C0, C1, COUT, COUT2, EPI2>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    const int per_cu = (160 * 1024) / smem >= 2 ? 2 : 1;
    int grid = 256 * per_cu;
    if (grid > a.n_tiles) grid = a.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_halo_kernel");
    return V2X_OK;
}

// Returns V2X_OK if handled, 1 if the shape is not one the halo kernel covers (caller falls back
// to the gather kernel -- which needs the row-major weight layout, so callers decide at pack time).
int v2x_conv_halo_dispatch(const v2x_conv_desc *d, hipStream_t s) {
    HaloArgs a;
    a.xcd_walk = v2x_tune(V2X_TUNE_HALO_XCD);
    // 16-byte stores for the bf16 outputs (plain layers and the bf16 chain; the fp32 heads already store 16 bytes per lane)
    a.x4 = (d->epilogue == V2X_EPI_BF16) ? v2x_x4_ok(d->out, d->out_cstride, d->out_coff, d->Cout2 > 0 ? d->Cout2 : d->Cout) : 0;
    a.in0 = d->C1 ? d->in0 : nullptr;
    a.in1 = d->C1 ? d->in1 : d->in0;
    a.bits = d->in_format == 1 ? reinterpret_cast<const uint32_t *>(d->in0) : nullptr;
    a.zbits = d->in_zbits;
    const int C0 = d->C1 ? d->C0 : 0, C1 = d->C1 ? d->C1 : d->C0;
    a.N = d->N;
    a.H = d->H;
    a.W = d->W;
    a.w = d->weight;
    a.scale = d->scale;
    a.shift = d->shift;
    a.relu = d->relu;
    a.out = d->out;
    a.out_cstride = d->out_cstride;
    a.out_coff = d->out_coff;
    a.w2 = d->weight2;
    a.scale2 = d->scale2;
    a.shift2 = d->shift2;
    a.relu2 = d->relu2;
    a.cout2_real = d->Cout2;
    a.split = d->split;
    a.out2 = d->out2;
    a.out2_cstride = d->out2_cstride;
    a.det_counts = d->det_counts;
    a.det_thr = d->det_thr;
    {
        const double t = d->det_thr;
        // The exact test is the fp32 softmax fg >= thr.  One ulp of fg (6e-8) moves the logit by 6e-8 / (thr (1 - thr)): the slack grows with it
        // near 1 (a fixed 1e-3 dropped candidates the logits path keeps once thr > 0.9999), and fg is exactly 1.0f from c1 - c0 ~ 16.7
        // on, whatever thr <= 1 says -- so the pre-test never asks for more than 16.  thr <= 0: every anchor; thr > 1: nothing passes the exact test.
        if (t <= 0.0) a.det_margin = -3.0e38f;
        else if (t >= 1.0) a.det_margin = 16.0f;
        else {
            const double slack = fmax(1e-3, 4e-7 / (t * (1.0 - t)));
            a.det_margin = (float)fmin(log(t / (1.0 - t)) - slack, 16.0);
        }
    }
    a.det_cap = d->det_cap;
    {   // the packed DMA tables hold the lane's element offset inside the patch in 20 bits
        const int cmax = d->C1 ? (d->C0 > d->C1 ? d->C0 : d->C1) : d->C0;
        if ((long long)(PH * d->W + PW) * cmax >= (1 << 20)) return 1;
    }
    a.tiles_x = d->W / TW;
    a.tiles_y = d->H / TH;
    a.n_tiles = d->N * a.tiles_x * a.tiles_y;
    const int co2 = d->Cout2 > 0 ? (d->Cout2 + 15) / 16 * 16 : 0;
    const int e2 = d->Cout2 > 0 ? (d->epilogue == V2X_EPI_DET ? 3 : (d->epilogue == V2X_EPI_F32 ? 2 : 1)) : 0;
#define HALO_CASE(c0, c1, co, c2, ep) \
    if (C0 == c0 && C1 == c1 && d->Cout == co && co2 == c2 && e2 == ep) return launch_halo<c0, c1, co, c2, ep>(a, s);
    // HBM-bound 32 -> 32 layers (conv_pre_1 (13 -> 32 padded), conv_pre_2, conv8_2): single-buffer form
    if (C0 == 0 && C1 == 32 && d->Cout == 32 && co2 == 0 && e2 == 0)
        return d->in_format == 1 ? launch_halo_sb<0, 32, 32, 0, 0, true>(a, s) : launch_halo_sb<0, 32, 32, 0, 0, false>(a, s);
    if (d->in_format == 1) return 1;  // bit-grid input exists for the 32 -> 32 first layer only
    if (d->w_layout == 3) {   // parity-class packing (pre-summed 2x2-tap weights for the upsampled source): its own kernel, whatever the tile count
        if (C0 == 64 && C1 == 32 && d->Cout == 32 && co2 == 0 && e2 == 0) return launch_halo_ppc<64, 32, 32>(a, s);
        return 1;
    }
    if (C0 == 64 && C1 == 32 && d->Cout == 32 && co2 == 0 && e2 == 0 && a.n_tiles >= 2 && a.n_tiles % 2 == 0) {
        // conv8_1: 8-wave ping-pong form (tuning switch HALO_PP = 0 keeps the 4-wave kernel: A/B runs and the bitwise-equality test)
        if (v2x_tune(V2X_TUNE_HALO_PP) != 0) return launch_halo_pp<64, 32, 32>(a, s);
    }
    HALO_CASE(64, 32, 32, 0, 0)   // conv8_1: cat(up(x_7), x)
    if (C0 == 0 && C1 == 64 && d->Cout == 64 && co2 == 0 && e2 == 0 && a.n_tiles >= 2 && a.n_tiles % 2 == 0) {
        // conv7_2: resident 72-KiB weights + two single-buffered 43-KiB patches = 158.5 KiB, 8-wave ping-pong form
        if (v2x_tune(V2X_TUNE_HALO_PP) != 0) return launch_halo_pp<0, 64, 64>(a, s);
    }
    HALO_CASE(0, 64, 64, 0, 0)    // conv7_2
    if (C0 == 0 && C1 == 64 && d->Cout == 64 && co2 == 64 && e2 == 1 && d->Cout2 == 64 && a.n_tiles >= 2 && a.n_tiles % 2 == 0 &&
        d->split == 0)
        return launch_halo_pp<0, 64, 64, 64>(a, s);   // conv1_2 -> conv3d_1 chained: ping-pong form
    // ... and its 4-wave form for an ODD number of tiles (odd batch x odd tiles per map, e.g. a 24 x 32 map): the ping-pong kernel pairs
    // tiles.  Same K order and epilogue -> the same bits (tests/test_gpu_stages.py::test_halo_chain_odd_tile_count), so the choice
    // may depend on the batch without breaking the R-rank == 1-rank equality.
    HALO_CASE(0, 64, 64, 64, 1)
    // row f-3: the data gradient of conv1_1 (64 -> 32 over the zero-inserted dy at 256 x 256) had only the gather kernel.  (Its two companions
    // there, conv8_1 on the concatenated 96-channel map and conv8_1's data gradient 32 -> 96, do not fit: <0, 32, 96> spills 232 bytes beside 24
    // accumulator tiles, the single-buffer <0, 96, 32> 192 bytes under its 128-register budget -- and a spill breaks the counted waits.)
    HALO_CASE(0, 64, 32, 0, 0)
    // round 6: conv8_1's data gradient in TWO launches into one 96-channel map -- 32 -> 64 (the x2-upsampled source's channels; this case) and 32 -> 32 (the skip's;
    // the single-buffer form above) -- from row slices of the transposed weights (v2x_pack_spec.src_rows / src_row0) instead of the gather kernel's 32 -> 96
    HALO_CASE(0, 32, 64, 0, 0)
    HALO_CASE(0, 32, 64, 48, 2)   // det heads: (cls | reg) hidden -> 12 + 36 logits
    HALO_CASE(0, 32, 32, 16, 2)   // seg: conv8_2 chained with the 1x1 class head (<= 16 classes, fp32 logits)
    if (e2 == 3) {                // det heads with the score threshold in the epilogue: candidates instead of logits
        if (!(C0 == 0 && C1 == 32 && d->Cout == 64 && d->Cout2 == 64 && d->det_counts && d->out && d->out2 && d->det_cap > 0 &&
              d->det_cap <= 4096 && (long long)d->H * d->W * 6 < (1 << 20)))
            return 1;
        return launch_halo<0, 32, 64, 64, 3>(a, s);
    }
#undef HALO_CASE
    return 1;
}
