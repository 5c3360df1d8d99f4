// The decoder's last layer and the detection heads as ONE kernel: conv8_2 (3x3, 32 -> 32, +BN+ReLU) -> [ClassificationHead |
// SingleRegressionHead] (fused hidden 3x3 32 -> 64 +BN+ReLU, block-diagonal 1x1 64 -> 12 + 36, fp32 logits) of upstream
// coperception/models/det/backbone/Backbone.py::LidarDecoder + base/DetModelBase.py (code absent from /root/reference, see
// include/v2x_amd.h).
//
// STATUS: correct (bit-identical to the two launches) but NOT the default -- measured 1.93-1.96 ms per 320 maps against
// 0.55 + 1.13 ms for the two launches in same-box A/B runs, with and without the operand pipelining below: per tile the two
// 4-wave groups together issue ~190 ds_read_b128 per wave (layer A in linear fragments cannot share pixel operands between tap
// rows: 72 reads for 108 MFMAs, against 42 for 72 in the stand-alone layer) plus the stores, conversions and DMAs of three
// stages, and one 8-wave workgroup per CU is bound by LDS bandwidth and issue slots before HBM.  Enabled by V2X_CONV_TAIL=1.
//
// Why: both launches run at the HBM copy ceiling (0.56 + 1.18 ms per 320 maps): conv8_2 WRITES the 32-channel map (4.2 MB per
// map), the heads read it back.  Fused, the map lives in LDS only: 16.8 instead of 25.2 MB per map cross HBM (-2.7 GB per
// 320-map launch), and the chip being power-limited bytes not moved are the gains that survive.
//
// Form: the 8-wave ping-pong scheme of conv3x3_halo_pp_kernel with one more stage.  One 512-thread workgroup per CU keeps both
// weight tensors resident (18 + 36 KiB) and gives each 4-wave group its own input window (12 x 36 pixels, 27 KiB, LDS-DMA) and
// intermediate patch (10 x 34 pixels, 21.3 KiB); a group's tile goes through four sub-intervals separated by workgroup barriers,
// the two groups two sub-intervals apart:
//
//      sub-interval     group 0                                   group 1
//      4k               A(k)  conv8_2 on the 10x34 halo region     S(k-1) + start DMA of window k
//      4k+1             B(k)  heads hidden 3x3 from the patch      wait for the window
//      4k+2             S(k)  chained 1x1, 12 fp32 stores; DMA k+1 A(k)
//      4k+3             wait                                       B(k)
//
// so one group's MFMA phases (A: 108, B: 144 MFMAs per wave) always run beside the other's store / DMA / wait phases.
// A(k): the region's 340 pixels as 22 linear 16-pixel fragments (pair-kernel style: a fragment may wrap to the next region
// row, every lane addresses its own pixel), scale/shift/ReLU, rounded to bf16 exactly as the stand-alone layer stores it,
// ZEROED outside the image (= the heads' zero padding), written to the patch in conv_halo.hip's swizzled pixel-major layout.
// B(k), S(k): conv3x3_halo_body's <0, 32, 64, 48, 2> arithmetic (hidden rows in kappa order: a lane's accumulators are its B
// fragment of the 1x1).  Tap order (kx outer, ky inner) and epilogues are those of the stand-alone kernels: the logits are
// bit-identical to v2x_conv2d(conv8_2) followed by v2x_conv2d(heads) (tests/test_gpu_stages.py::test_tail_equals_two_launches).
#include "common.h"
#include <cstdlib>

static __device__ __attribute__((aligned(64))) unsigned int g_zero_page_t[16];

typedef const __attribute__((address_space(1))) void *gptr_tl_t;
typedef __attribute__((address_space(3))) void *lptr_tl_t;

struct TailArgs {
    const uint16_t *in;    // [N][H][W][32] bf16
    int N, H, W;
    const uint16_t *wA;    // conv8_2: k-slot-major [36][32][8]
    const float *scA, *shA;
    int reluA;
    const uint16_t *wB;    // heads hidden: k-slot-major [36][64][8], rows in chain (kappa) order
    const float *scB, *shB;   // [64] natural channel order
    int reluB;
    const uint16_t *w2;    // [48][64] row-major bf16 (K in kappa order)
    const float *sc2, *sh2;   // [48]
    int relu2, split;
    float *out, *out2;     // fp32 NHWC: channels < split -> out (+out_coff), the rest -> out2
    int out_cstride, out_coff, out2_cstride;
    int tiles_x, tiles_y, n_tiles;
};

namespace tail {
constexpr int TH = 8, TW = 32;
constexpr int MH = TH + 2, MW = TW + 2;      // layer-A region = the heads' input patch
constexpr int IH = TH + 4, IW = TW + 4;      // input window
constexpr int NMID = MH * MW;                // 340
constexpr int NFRAG = (NMID + 15) / 16;      // 22
constexpr int FPW = (NFRAG + 3) / 4;         // 6 fragments per wave (waves 2, 3: 5 real ones)
constexpr int WA_BYTES = 36 * 32 * 16;       // 18 432
constexpr int WB_BYTES = 36 * 64 * 16;       // 36 864
constexpr int IN_SLOTS = IH * IW * 4;        // 1 728 = 27 wave pieces exactly
constexpr int IN_BYTES = IN_SLOTS * 16;      // 27 648
constexpr int MID_BYTES = NMID * 64;         // 21 760
constexpr int SS_FLOATS = 32 + 32 + 64 + 64 + 48 + 48;
constexpr int SMEM = WA_BYTES + WB_BYTES + 2 * (IN_BYTES + MID_BYTES) + SS_FLOATS * 4;
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(IN_SLOTS % 64 == 0, "whole DMA pieces");
#ifndef V2X_HALO_PSWZ_BUILD
#define V2X_HALO_PSWZ_BUILD 1
#endif
// 1: (x >> 1) & 3 (round 1);  2: (x >> 2) & 3, whose fragment reads time 30 % faster in isolation and change nothing here (conv_halo.hip)
__device__ __forceinline__ int swz4(int slot, int x) { return slot ^ ((x >> V2X_HALO_PSWZ_BUILD) & 3); }
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t x) {   // max(x, 0) on two packed bf16 (see conv_halo_pair.hip)
    const s16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, x), (s16x2_t){0, 0});
    return __builtin_bit_cast(uint32_t, r);
}
// Lane id produced by VOLATILE asm: it cannot be hoisted or CSE'd, so every phase derives its lane constants (fragment
// coordinates, ~40 LDS addresses) afresh instead of the compiler keeping -- and spilling -- them across the interval loop.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
}  // namespace tail

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_tail_kernel(const TailArgs a) {
    using namespace tail;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_wA = smem, *s_wB = smem + WA_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wv = wave & 3;
    char *s_in = smem + WA_BYTES + WB_BYTES + grp * (IN_BYTES + MID_BYTES);
    char *s_mid = s_in + IN_BYTES;
    float *s_ss = reinterpret_cast<float *>(smem + WA_BYTES + WB_BYTES + 2 * (IN_BYTES + MID_BYTES));
    // [scA 32 | shA 32 | scB 64 | shB 64 | sc2 48 | sh2 48]
    for (int i = tid; i < 32; i += 512) {
        s_ss[i] = a.scA[i];
        s_ss[32 + i] = a.shA[i];
    }
    for (int i = tid; i < 64; i += 512) {
        s_ss[64 + i] = a.scB[i];
        s_ss[128 + i] = a.shB[i];
    }
    for (int i = tid; i < 48; i += 512) {
        s_ss[192 + i] = a.sc2[i];
        s_ss[240 + i] = a.sh2[i];
    }
    // weights: linear LDS-DMA copies, resident for the whole kernel
    for (int off = wave * 1024; off < WA_BYTES + WB_BYTES; off += 8192) {
        const char *src = off < WA_BYTES ? reinterpret_cast<const char *>(a.wA) + off : reinterpret_cast<const char *>(a.wB) + (off - WA_BYTES);
        __builtin_amdgcn_global_load_lds((gptr_tl_t)(src + lane * 16), (lptr_tl_t)(smem + off), 16, 0, 0);
    }
    const int txy = a.tiles_x * a.tiles_y;
    auto coords = [&](int tile, int &n, int &y0, int &x0) {
        n = tile / txy;
        const int r = tile - n * txy;
        const int ty = r / a.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * a.tiles_x) * TW;
    };
    // pairs of tiles (2p, 2p + 1): group g owns tile 2p + g; p = blockIdx.x + k * gridDim.x
    const int n_pairs = a.n_tiles >> 1;
    const int K = (n_pairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int k) { return 2 * ((int)blockIdx.x + k * (int)gridDim.x) + grp; };

    auto dma_window = [&](int tile) {   // 27 pieces over the group's 4 waves: the 12 x 36 x 32-channel input window
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const int lane_o = fresh_lane();   // everything below is tile-invariant per lane: derived afresh, never kept across phases
        // exactly 7 DMA instructions per wave (wave 3 owns 6 pieces and repeats its last one): the S phase counts on it
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int piece = min(wv + 4 * it, IN_SLOTS / 64 - 4 + wv);
            const int L = piece * 64 + lane_o;
            const int pix = L >> 2, phys = L & 3;
            const int pr = pix / IW, pc = pix - pr * IW;
            const int y = y0 - 2 + pr, x = x0 - 2 + pc;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ((unsigned)(n * a.H + y) * (unsigned)a.W + (unsigned)x) * 32u + (unsigned)(swz4(phys, pc) * 8);
            __builtin_amdgcn_global_load_lds((gptr_tl_t)(ok ? (const void *)(a.in + off) : (const void *)g_zero_page_t),
                                             (lptr_tl_t)(s_in + piece * 1024), 16, 0, 0);
        }
    };

    auto layer_a = [&](int tile) {   // conv8_2 on the 10 x 34 region -> s_mid (bf16, zero outside the image)
        int n, y0, x0;
        coords(tile, n, y0, x0);
        f32x4_t acc[FPW][2];
#pragma unroll
        for (int t = 0; t < FPW; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[t][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int lane_a = fresh_lane();
        const int fjo = lane_a & 15, fqo = lane_a >> 4;
        int rco[FPW];   // region row | column << 8 | valid << 16 of the lane's pixel in fragment t, recomputed per phase
#pragma unroll
        for (int t = 0; t < FPW; ++t) {
            const int p = (wv + 4 * t) * 16 + fjo;
            const int pcl = p < NMID ? p : NMID - 1;
            const int r = pcl / MW, c = pcl - r * MW;
            rco[t] = r | (c << 8) | ((p < NMID ? 1 : 0) << 16);
        }
        // nine k-steps (kx outer, ky inner = the stand-alone kernel's accumulation order).  With one group in its MFMA phase at a
        // time nobody else hides the fragment reads' latency, so the operands of step s+1 are read into a second register set
        // before the 12 MFMAs of step s issue (first version without this: 1.9 ms per 320 maps, slower than the two launches).
        bf16x8_t A[2][2], B[2][FPW];
        auto load_step = [&](int st, bf16x8_t (&Ak)[2], bf16x8_t (&Bk)[FPW]) {
            const int kx = st / 3, ky = st - kx * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) Ak[i] = *reinterpret_cast<const bf16x8_t *>(s_wA + (((ky * 3 + kx) * 4 + fqo) * 32 + i * 16 + fjo) * 16);
#pragma unroll
            for (int t = 0; t < FPW; ++t) {
                const int r = rco[t] & 0xff, c = ((rco[t] >> 8) & 0xff) + kx;
                Bk[t] = *reinterpret_cast<const bf16x8_t *>(s_in + ((r * IW + c) * 4 + swz4(fqo, c)) * 16 + ky * (IW * 64));
            }
        };
        load_step(0, A[0], B[0]);
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            if (st + 1 < 9) load_step(st + 1, A[(st + 1) & 1], B[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < FPW; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[st & 1][i], B[st & 1][t], acc[t][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        int zo = 0;
        asm volatile("" : "+v"(zo));   // opaque: keeps the (loop-invariant) scale/shift reads inside this phase
        const float *ss = s_ss + zo;
#pragma unroll
        for (int t = 0; t < FPW; ++t) {
            const int r = rco[t] & 0xff, c = (rco[t] >> 8) & 0xff;
            const int y = y0 - 1 + r, x = x0 - 1 + c;
            const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            if (!(rco[t] >> 16)) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = *reinterpret_cast<const float4 *>(ss + i * 16 + fqo * 4);
                const float4 sf = *reinterpret_cast<const float4 *>(ss + 32 + i * 16 + fqo * 4);
                float v0 = acc[t][i][0] * sc.x + sf.x, v1 = acc[t][i][1] * sc.y + sf.y;
                float v2 = acc[t][i][2] * sc.z + sf.z, v3 = acc[t][i][3] * sc.w + sf.w;
                uint2 o;   // ReLU on the packed pair (one v_pk_max_i16 for two values; the host wrapper guarantees relu = 1)
                o.x = relu_bf16x2(pack_bf16x2(v0, v1));
                o.y = relu_bf16x2(pack_bf16x2(v2, v3));
                o.x = inside ? o.x : 0u;
                o.y = inside ? o.y : 0u;
                *reinterpret_cast<uint2 *>(s_mid + ((r * MW + c) * 4 + swz4(i * 2 + (fqo >> 1), c)) * 16 + (fqo & 1) * 8) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is written before the closing barrier
    };

    f32x4_t hacc[4][4];   // heads hidden accumulators: live from B(k) to S(k)
    auto layer_b = [&]() {   // heads hidden 3x3 32 -> 64 on s_mid (conv3x3_halo_body<0, 32, 64, ...>'s walk: kx groups, ky inner)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) hacc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t B[2][8], A[2][4];   // pixel fragments of tap column kx / weight fragments of one tap, each in two alternating sets
        // fresh lane coordinates: the ~40 LDS addresses below are lane constants; hoisted out of the interval loop they were
        // spilled in the prologue and reloaded here -- scratch loads that would also corrupt the vmcnt bookkeeping
        const int lane_b = fresh_lane();
        const int fjo = lane_b & 15, fqo = lane_b >> 4;
        auto load_A = [&](int st, bf16x8_t (&Ak)[4]) {
            const int kx = st / 3, ky = st - kx * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) Ak[i] = *reinterpret_cast<const bf16x8_t *>(s_wB + (((ky * 3 + kx) * 4 + fqo) * 64 + i * 16 + fjo) * 16);
        };
        auto load_B = [&](int kx, bf16x8_t (&Bk)[8]) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int pr = 2 * wv + rr, pc = ch * 16 + fjo + kx;
                    Bk[rr * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(s_mid + ((pr * MW + pc) * 4 + swz4(fqo, pc)) * 16);
                }
        };
        load_B(0, B[0]);
        load_A(0, A[0]);
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int kx = st / 3, ky = st - kx * 3;
            if (st + 1 < 9) load_A(st + 1, A[(st + 1) & 1]);
            if (ky == 0 && kx < 2) load_B(kx + 1, B[(kx + 1) & 1]);   // lands under the 48 MFMAs of this tap column
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    hacc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[st & 1][i], B[kx & 1][((f >> 1) + ky) * 2 + (f & 1)], hacc[i][f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto load_w2 = [&](bf16x8_t (&w2f)[3][2]) {   // the chained 1x1's weight fragments (6 KiB, L1/L2-resident): 6 loads
        const int lane_w = fresh_lane();   // loop-invariant loads would be hoisted across the MFMA phases (24 registers)
        const int w2off = (lane_w & 15) * 64 + (lane_w >> 4) * 8;
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) w2f[i2][ks] = *reinterpret_cast<const bf16x8_t *>(a.w2 + w2off + i2 * 16 * 64 + ks * 32);
    };
    auto chain_store = [&](int tile, const bf16x8_t (&w2f)[3][2], bool dma_behind) {   // hidden -> bf16 -> 1x1 (64 -> 48) -> fp32 logits, cls | loc
        // (the host wrapper guarantees ReLU on the hidden layer, none on the logits, and a split output: no uniform branches here)
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const int lane_s = fresh_lane();
        const int fjs = lane_s & 15, fqs = lane_s >> 4;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            // opaque per fragment: the scale/shift reads are the same for all four fragments; read once they would sit in
            // 56 registers next to the 64 accumulators and the 24 weight registers (this phase spilled)
            int zo = 0;
            asm volatile("" : "+v"(zo));
            const float *ss = s_ss + zo;
            bf16x8_t hb[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float h[8];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int i = 2 * ks + hf;
                    const int kappa = 32 * (i >> 1) + 8 * fqs + 4 * (i & 1);
                    const float4 sc = *reinterpret_cast<const float4 *>(ss + 64 + kappa);
                    const float4 sf = *reinterpret_cast<const float4 *>(ss + 128 + kappa);
                    h[hf * 4 + 0] = fmaxf(hacc[i][f][0] * sc.x + sf.x, 0.f);
                    h[hf * 4 + 1] = fmaxf(hacc[i][f][1] * sc.y + sf.y, 0.f);
                    h[hf * 4 + 2] = fmaxf(hacc[i][f][2] * sc.z + sf.z, 0.f);
                    h[hf * 4 + 3] = fmaxf(hacc[i][f][3] * sc.w + sf.w, 0.f);
                }
                uint4 p;
                p.x = pack_bf16x2(h[0], h[1]);
                p.y = pack_bf16x2(h[2], h[3]);
                p.z = pack_bf16x2(h[4], h[5]);
                p.w = pack_bf16x2(h[6], h[7]);
                hb[ks] = __builtin_bit_cast(bf16x8_t, p);
            }
            const int y = y0 + 2 * wv + (f >> 1), x = x0 + (f & 1) * 16 + fjs;
            const size_t pix = (size_t)(n * a.H + y) * a.W + x;
            if (f == 0) {   // first use of w2f: its 6 loads are OLDER than the 7 window DMAs issued after them
                if (dma_behind) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) {
                f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[i2][ks], hb[ks], d, 0, 0, 0);
                const int co = i2 * 16 + fqs * 4;
                const float4 s2 = *reinterpret_cast<const float4 *>(ss + 192 + co);
                const float4 t2 = *reinterpret_cast<const float4 *>(ss + 240 + co);
                const float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                float *dst = co >= a.split ? a.out2 + pix * a.out2_cstride + (co - a.split) : a.out + pix * a.out_cstride + a.out_coff + co;
                *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
            }
        }
    };

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // weights and scale/shift tables are in LDS

    // sub-interval j; group g runs u = j - 2g:  u = -2: DMA of its first window, -1: wait;  u >= 0: tile k = u / 4, phase u % 4 =
    // 0: A(k)   1: B(k)   2: DMA of window k+1, then S(k)   3: wait (DMAs landed; the 12 stores of S(k) may stay in flight)
    for (int j = -2; j <= 4 * K; ++j) {
        const int u = j - 2 * grp;
        if (u >= -2 && u <= 4 * (K - 1) + 3) {
            const int ph = (u + 4) & 3, k = (u + 4) / 4 - 1;
            if (ph == 0) {
                layer_a(tile_of(k));
            } else if (ph == 1) {
                layer_b();
            } else if (ph == 2) {
                bf16x8_t w2f[3][2];
                if (k >= 0) load_w2(w2f);
                if (k + 1 < K) dma_window(tile_of(k + 1));
                if (k >= 0) chain_store(tile_of(k), w2f, k + 1 < K);
            } else {
                if (k >= 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

// first: conv8_2 (3x3 s1 p1, w_layout 1, 32 -> 32, bf16 epilogue);  second: the fused det heads (3x3 s1 p1, w_layout 1, 32 -> 64,
// chained 1x1 to Cout2 = 48, V2X_EPI_F32, split outputs).  first->out is ignored: the 32-channel map never reaches memory.
extern "C" int v2x_conv2d_tail(const v2x_conv_desc *first, const v2x_conv_desc *second, v2x_stream_t stream) {
    V2X_REQUIRE(first && second, "v2x_conv2d_tail: null descriptor");
    V2X_REQUIRE(first->ksize == 3 && first->stride == 1 && first->pad == 1 && first->w_layout == 1 && first->C0 == 32 && first->C1 == 0 &&
                    first->Cout == 32 && first->Cout2 == 0 && first->epilogue == V2X_EPI_BF16 && first->up0 == 0 && first->in_format == 0,
                "v2x_conv2d_tail: the first layer must be a halo-packed 3x3 stride-1 32 -> 32 bf16 layer");
    V2X_REQUIRE(second->ksize == 3 && second->stride == 1 && second->pad == 1 && second->w_layout == 1 && second->C0 == 32 &&
                    second->C1 == 0 && second->Cout == 64 && second->Cout2 == 48 && second->epilogue == V2X_EPI_F32 && second->up0 == 0,
                "v2x_conv2d_tail: the second layer must be the halo-packed heads layer 32 -> 64 chained to 48 fp32 channels");
    V2X_REQUIRE(first->in0 && first->weight && first->scale && first->shift && second->weight && second->scale && second->shift &&
                    second->weight2 && second->scale2 && second->shift2 && second->out,
                "v2x_conv2d_tail: null parameter / buffer");
    V2X_REQUIRE(first->relu == 1 && second->relu == 1 && second->relu2 == 0 && second->split > 0,
                "v2x_conv2d_tail: expects ReLU after conv8_2 and the hidden heads layer, raw logits, and split outputs");
    V2X_REQUIRE(first->N == second->N && first->H == second->H && first->W == second->W, "v2x_conv2d_tail: extents differ");
    V2X_REQUIRE(first->H % tail::TH == 0 && first->W % tail::TW == 0, "v2x_conv2d_tail: H %% 8 == 0 and W %% 32 == 0 required");
    V2X_REQUIRE((long long)first->N * first->H * first->W < (1ll << 26), "v2x_conv2d_tail: N*H*W must stay below 2^26 (32-bit offsets)");
    if (second->split > 0) {
        V2X_REQUIRE(second->out2 && second->split % 4 == 0 && second->split < 48 && second->out_cstride >= second->out_coff + second->split &&
                        second->out2_cstride >= 48 - second->split,
                    "v2x_conv2d_tail: bad split output windows");
    } else {
        V2X_REQUIRE(second->out_cstride >= second->out_coff + 48, "v2x_conv2d_tail: bad output channel window");
    }
    if (first->N == 0) return V2X_OK;
    TailArgs a;
    a.in = first->in0;
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
    V2X_REQUIRE(a.n_tiles % 2 == 0, "v2x_conv2d_tail: an even number of 8x32 tiles is required (tile pairs)");
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tail::SMEM);
    }
    int grid = 256;
    if (grid > a.n_tiles / 2) grid = a.n_tiles / 2;
    hipLaunchKernelGGL(conv3x3_tail_kernel, dim3(grid), dim3(512), tail::SMEM, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("conv3x3_tail_kernel");
    return V2X_OK;
}
