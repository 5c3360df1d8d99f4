// a2 / a4 / a6 / a7 / a8 -- implicit-GEMM convolution on the gfx950 matrix cores.
//
// One kernel family replaces every conv + BN + ReLU of upstream
// coperception/models/det/backbone/Backbone.py (LidarEncoder/LidarDecoder, incl. the
// F.interpolate(x2)+torch.cat feeding conv5_1..conv8_1), the Conv2dGRU cell step of
// V2VNet.py (convolutional_rnn), the heads of DetModelBase.py and the PolicyNet4 /
// KmGenerator layers of When2com.py (code absent from /root/reference, see v2x_amd.h).
//
// GEMM view:  D[co][px] = sum_k W[co][k] * X[k][px],  k = (ky*ks + kx)*Cin + c.
//   * MFMA "A" operand (rows i) = packed weights, K-contiguous  [w_rows][w_kpad] bf16.
//   * MFMA "B" operand (cols j) = activations gathered on the fly from NHWC bf16: the 8
//     bf16 a lane feeds are 8 consecutive channels of one input pixel = one 16-B load,
//     so im2col never exists in memory.  Zero padding, stride, nearest-x2 upsample of
//     source 0 and the channel concat of two sources are all address arithmetic here.
//   * v_mfma_f32_16x16x32_bf16, fp32 accumulate.  D layout: lane holds pixel (lane&15)
//     and 4 consecutive output channels ((lane>>4)*4+r) -> one 8-B NHWC store per tile.
//   * K is walked in chunks of 64; each 16-B slot of a chunk carries its own (tap, c),
//     so Cin only has to be a multiple of 8 (first layer: 13 -> 16 channels).
//   * LDS tiles are [row][64 bf16] = 128-B rows with the 16-B slot index XOR (row&7):
//     conflict-free for both the ds_write_b128 fill and the ds_read_b128 fragment reads
//     (MI355X_MICROARCH.md LDS lane groups).  Tiles arrive by LDS-DMA (global_load_lds_dwordx4):
//     no staging VGPRs, no ds_write pass; the swizzle is applied to the per-lane SOURCE address
//     and padding taps read a zero page.  Two stages, one barrier per chunk; the DMA of chunk
//     t+1 is issued before the MFMAs of chunk t.
#include "common.h"

struct ConvArgs {
    const uint16_t *in0, *in1;
    int C0, C1, Cin, up0;
    int N, H, W, Ho, Wo;
    int ks, stride, pad, ntaps;
    int M;  // N*Ho*Wo output pixels
    int Cout, w_rows, w_kpad;
    const uint16_t *w;
    const float *scale, *shift;
    int relu;
    void *out;
    int out_cstride, out_coff;
    void *out2;
    int split, out2_cstride;
};

enum { EPI_BF16 = V2X_EPI_BF16, EPI_F32 = V2X_EPI_F32, EPI_GRU = V2X_EPI_GRU };

// 64 B of zeros: padding taps / K-tail slots / out-of-range pixels point their LDS-DMA source here.
static __device__ __attribute__((aligned(64))) unsigned int g_zero_page[16];

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// 16 B per lane, global -> LDS without a VGPR round trip (global_load_lds_dwordx4).  The LDS
// destination is wave-uniform base + lane*16, so the XOR swizzle lives in the per-lane SOURCE
// address (guide rule 21): lane l fills physical slot (l&7) of row (l>>3) with logical slot
// (l&7)^((l>>3)&7).
__device__ __forceinline__ void glds16(const void *g, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int BCO, int BPX, int WCO, int WPX, int EPI>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int TCO = BCO / WCO / 16;  // MFMA tiles per wave along output channels
    constexpr int TPX = BPX / WPX / 16;  // ... along pixels
    static_assert(WCO * WPX == 4, "4 waves per workgroup");
    static_assert(BCO % (WCO * 16) == 0 && BPX % (WPX * 16) == 0, "tile shape");
    static_assert(BPX % 32 == 0 && BCO % 8 == 0, "tile rows are moved 8 per wave instruction");
    constexpr int A_IT = (BCO + 31) / 32;  // wave instructions (8 rows each) per wave for the weight tile
    constexpr int B_IT = BPX / 32;
    constexpr int STAGE_BYTES = (BCO + BPX) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX;
    const int wpx = wave % WPX;

    // XCD-aware tile order (guide T1): workgroup b runs on XCD b%8; give every XCD a contiguous run
    // of tiles so the channel tiles of one pixel tile (same activations) and neighbouring pixel
    // tiles (shared halos) hit the same L2.  Bijective for any grid size.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int n_co_tiles = a.w_rows / BCO;
    const int co_tile = bid % n_co_tiles;
    const int px_tile = bid / n_co_tiles;
    const int co0 = co_tile * BCO;
    const int px0 = px_tile * BPX;

    // ---- per-thread gather bookkeeping (fixed for the whole K loop) -------------------
    const int lrow = lane >> 3;                 // row inside the 8-row group one wave instruction moves
    const int lslot = (lane & 7) ^ (lrow & 7);  // logical 16-B slot (8 channels) this lane fetches
    int iy0[B_IT], ix0[B_IT], nbase[B_IT];      // nbase = image index (or -1 for rows past M)
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int p = px0 + (wave + 4 * it) * 8 + lrow;
        if (p < a.M) {
            const int hw = a.Ho * a.Wo;
            const int n = p / hw;
            const int rem = p - n * hw;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            nbase[it] = n;
            iy0[it] = oy * a.stride - a.pad;
            ix0[it] = ox * a.stride - a.pad;
        } else {
            nbase[it] = -1;
            iy0[it] = 0;
            ix0[it] = 0;
        }
    }
    // weight rows: this lane's element offset inside the packed matrix for chunk 0
    unsigned woff[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (wave + 4 * it) * 8 + lrow;
        woff[it] = (unsigned)(co0 + row) * (unsigned)a.w_kpad + lslot * 8;
    }
    // (tap, c) of this lane's slot, advanced by 64 channels per chunk
    int k_tap = (lslot * 8) / a.Cin;
    int k_c = lslot * 8 - k_tap * a.Cin;

    auto stage_chunk = [&](int t, int stage) {
        char *sa = smem + stage * STAGE_BYTES;
        char *sb = sa + BCO * 128;
        // weights
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            if ((wave + 4 * it) * 8 < BCO)  // wave-uniform (only BCO=48 skips)
                glds16(a.w + woff[it] + (unsigned)t * 64u, sa + (wave + 4 * it) * 1024);
        }
        // activations: implicit im2col
        const int tap = k_tap;
        const bool tap_ok = tap < a.ntaps;
        const int ky = (a.ks == 1) ? 0 : ((tap * 11) >> 5);
        const int kx = tap - ky * a.ks;
        const uint16_t *src = a.in0;
        int c = k_c, cs = a.C0, sh = a.up0;
        if (c >= a.C0) {
            src = a.in1;
            c -= a.C0;
            cs = a.C1;
            sh = 0;
        }
        const int Hs = a.H >> sh, Ws = a.W >> sh;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int iy = iy0[it] + ky;
            const int ix = ix0[it] + kx;
            const bool ok = tap_ok && nbase[it] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            // 24-bit multiplies: every factor is < 2^24 (host checks N*H*W < 2^24 rows), product < 2^32
            const unsigned pix = __umul24(__umul24((unsigned)nbase[it], (unsigned)Hs) + (unsigned)(iy >> sh), (unsigned)Ws) +
                                 (unsigned)(ix >> sh);
            const unsigned off = pix * (unsigned)cs + (unsigned)c;
            const void *g = ok ? (const void *)(src + off) : (const void *)g_zero_page;
            glds16(g, sb + (wave + 4 * it) * 1024);
        }
        // advance this lane's (tap, c) to the next chunk
        k_c += 64;
        while (k_c >= a.Cin) {
            k_c -= a.Cin;
            ++k_tap;
        }
    };

    f32x4_t acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.w_kpad >> 6;
    const int frow = lane & 15;  // row inside a 16-row fragment
    const int fk = lane >> 4;    // 8-element k group inside a 32-deep MFMA step

    stage_chunk(0, 0);
    __syncthreads();  // (the compiler drains vmcnt before the barrier while LDS-DMA is in flight)

    for (int t = 0; t < nchunks; ++t) {
        if (t + 1 < nchunks) stage_chunk(t + 1, (t + 1) & 1);  // next tile's DMA overlaps this tile's MFMAs

        const char *sa = smem + (t & 1) * STAGE_BYTES;
        const char *sb = sa + BCO * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[TCO], fb[TPX];
            const int ls = kk * 4 + fk;
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int row = (wco * TCO + i) * 16 + frow;
                fa[i] = *reinterpret_cast<const bf16x8_t *>(sa + row * 128 + ((ls ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int row = (wpx * TPX + j) * 16 + frow;
                fb[j] = *reinterpret_cast<const bf16x8_t *>(sb + row * 128 + ((ls ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------
    // lane holds pixel (lane&15) of pixel-tile j and rows (lane>>4)*4 + r of channel-tile i
    if constexpr (EPI == EPI_GRU) {
        static_assert(EPI != EPI_GRU || TCO == 3, "GRU tiles are (r,z,n) triples");
        // packed row = g*48 + gate*16 + e  <->  hidden channel g*16 + e
        const int g = co_tile * WCO + wco;
        const int hc = g * 16 + fk * 4;
        if (hc < a.Cout) {
            float4 bias[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = reinterpret_cast<const float4 *>(a.scale)[hc + r];
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int p = px0 + (wpx * TPX + j) * 16 + frow;
                if (p >= a.M) continue;
                float h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = v2x_gru_h0(acc[0][j][r], acc[1][j][r], acc[2][j][r], bias[r]);  // h0 = 0:  n + z*(h0 - n)
                }
                uint2 o;
                o.x = pack_bf16x2(h[0], h[1]);
                o.y = pack_bf16x2(h[2], h[3]);
                uint16_t *dst = reinterpret_cast<uint16_t *>(a.out) + (size_t)p * a.out_cstride + a.out_coff + hc;
                *reinterpret_cast<uint2 *>(dst) = o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = co0 + (wco * TCO + i) * 16 + fk * 4;
            if (co >= a.Cout) continue;
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + co);
            const float4 sf = *reinterpret_cast<const float4 *>(a.shift + co);
            const bool full = (co + 4 <= a.Cout);
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int p = px0 + (wpx * TPX + j) * 16 + frow;
                if (p >= a.M) continue;
                float y[4];
                y[0] = acc[i][j][0] * sc.x + sf.x;
                y[1] = acc[i][j][1] * sc.y + sf.y;
                y[2] = acc[i][j][2] * sc.z + sf.z;
                y[3] = acc[i][j][3] * sc.w + sf.w;
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
                }
                const bool second = a.split > 0 && co >= a.split;
                void *obase = second ? a.out2 : a.out;
                const size_t o = second ? ((size_t)p * a.out2_cstride + (co - a.split))
                                        : ((size_t)p * a.out_cstride + a.out_coff + co);
                if constexpr (EPI == EPI_F32) {
                    float *dst = reinterpret_cast<float *>(obase) + o;
                    if (full && ((o & 3) == 0)) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(y[0], y[1], y[2], y[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.Cout) dst[r] = y[r];
                    }
                } else {
                    uint16_t *dst = reinterpret_cast<uint16_t *>(obase) + o;
                    if (full && ((o & 3) == 0)) {
                        uint2 v;
                        v.x = pack_bf16x2(y[0], y[1]);
                        v.y = pack_bf16x2(y[2], y[3]);
                        *reinterpret_cast<uint2 *>(dst) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.Cout) dst[r] = f32_to_bf16_rne(y[r]);
                    }
                }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------
int v2x_conv1x1_dispatch(const v2x_conv_desc *d, hipStream_t s);   // conv1x1.hip: V2X_OK / error, or 1 = not a shape of the streaming 1x1 kernel
extern "C" int v2x_conv_tile_rows(int Cout, int epilogue) {
    if (epilogue == V2X_EPI_GRU) return 96;
    if (Cout <= 32) return 32;
    if (Cout <= 48) return 48;
    if (Cout <= 64) return 64;
    return 128;
}

template <int BCO, int BPX, int WCO, int WPX, int EPI>
static int launch_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int smem = (BCO + BPX) * 128 * 2;
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_kernel<BCO, BPX, WCO, WPX, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    const int n_co = a.w_rows / BCO;
    const int n_px = (a.M + BPX - 1) / BPX;
    hipLaunchKernelGGL((conv_igemm_kernel<BCO, BPX, WCO, WPX, EPI>), dim3(n_co * n_px), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv_igemm_kernel");
    return V2X_OK;
}

template <int EPI>
static int dispatch_rows(const ConvArgs &a, int rows, hipStream_t s) {
    switch (rows) {
        case 32: return launch_cfg<32, 256, 1, 4, EPI>(a, s);
        case 48: return launch_cfg<48, 256, 1, 4, EPI>(a, s);
        case 64: return launch_cfg<64, 128, 2, 2, EPI>(a, s);
        default: return launch_cfg<128, 128, 2, 2, EPI>(a, s);
    }
}

int v2x_conv_halo_dispatch(const v2x_conv_desc *d, hipStream_t s);    // conv_halo.hip
int v2x_conv_stream_dispatch(const v2x_conv_desc *d, hipStream_t s);  // conv_stream.hip
int v2x_conv_stream_s2_dispatch(const v2x_conv_desc *d, hipStream_t s);  // conv_stream_s2.hip

extern "C" int v2x_conv2d(const v2x_conv_desc *d, v2x_stream_t stream) {
    V2X_REQUIRE(d, "v2x_conv2d: null descriptor");
    if (d->w_layout == 1 || d->w_layout == 3) {
        V2X_REQUIRE(d->in0 && d->weight && d->scale && d->shift && d->out, "v2x_conv2d(halo): null tensor pointer");
        V2X_REQUIRE(d->w_layout == 1 || (d->C1 > 0 && d->up0 == 1 && d->Cout2 == 0 && d->in_format == 0 && d->epilogue == V2X_EPI_BF16 &&
                                         d->w_kpad == 16 * d->C0 + 9 * d->C1),
                    "v2x_conv2d(halo, parity-class weights): needs the upsampled + skip source pair, a plain bf16 epilogue and w_kpad = 16 C0 + 9 C1");
        V2X_REQUIRE(d->ksize == 3 && d->stride == 1 && d->pad == 1, "v2x_conv2d(halo): 3x3 stride 1 pad 1 only");
        V2X_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->H % 8 == 0 && d->W % 32 == 0,
                    "v2x_conv2d(halo): H=%d must be a multiple of 8 and W=%d of 32", d->H, d->W);
        V2X_REQUIRE((d->C1 == 0 && d->up0 == 0) || (d->C1 > 0 && d->up0 == 1 && d->in1),
                    "v2x_conv2d(halo): two sources means in0 is the x2-upsampled one");
        V2X_REQUIRE((long long)d->N * d->H * d->W * (d->C0 > d->C1 ? d->C0 : d->C1) < (1ll << 32),
                    "v2x_conv2d(halo): tensor exceeds 32-bit element offsets");
        V2X_REQUIRE(d->epilogue == V2X_EPI_BF16 || d->epilogue == V2X_EPI_F32 || d->epilogue == V2X_EPI_DET, "v2x_conv2d(halo): bad epilogue");
        if (d->epilogue == V2X_EPI_DET) {   // detection heads with the threshold in the epilogue: candidates, not logits
            V2X_REQUIRE(d->C0 == 32 && d->C1 == 0 && d->Cout == 64 && d->Cout2 == 64 && d->weight2 && d->scale2 && d->shift2 && d->in_format == 0,
                        "v2x_conv2d(det heads): expects 32 -> 64 hidden chained with 64 rows in det order");
            V2X_REQUIRE(d->out2 && d->det_counts && d->det_cap >= 64 && d->det_cap <= 4096 && (d->det_cap & (d->det_cap - 1)) == 0,
                        "v2x_conv2d(det heads): needs keys (out), codes (out2), det_counts and a power-of-two det_cap in [64, 4096]");
            V2X_REQUIRE((long long)d->H * d->W * 6 < (1ll << 20), "v2x_conv2d(det heads): H*W*6 must stay below 2^20 (anchor index and slot share a word)");
            const int rcd = v2x_conv_halo_dispatch(d, (hipStream_t)stream);
            V2X_REQUIRE(rcd != 1, "v2x_conv2d(det heads): no kernel for this shape");
            return rcd;
        }
        V2X_REQUIRE(d->in_format == 0 || (d->in_format == 1 && d->C0 == 32 && d->C1 == 0 && d->in_zbits >= 1 && d->in_zbits <= 32),
                    "v2x_conv2d(halo): bit-grid input needs C0 == 32, C1 == 0 and 1 <= in_zbits <= 32");
        if (d->Cout2 > 0) {
            V2X_REQUIRE(d->weight2 && d->scale2 && d->shift2, "v2x_conv2d(halo): chained 1x1 needs weight2/scale2/shift2");
            const int cfin = d->Cout2;
            if (d->split > 0)
                V2X_REQUIRE(d->out2 && d->split % 4 == 0 && d->split < cfin && d->out_cstride >= d->out_coff + d->split &&
                            d->out2_cstride >= cfin - d->split, "v2x_conv2d(halo): bad split windows");
            else
                V2X_REQUIRE(d->out_cstride >= d->out_coff + cfin, "v2x_conv2d(halo): bad output channel window");
            V2X_REQUIRE(cfin % 4 == 0, "v2x_conv2d(halo): chained Cout2 must be a multiple of 4");
        } else {
            V2X_REQUIRE(d->epilogue == V2X_EPI_BF16 && d->split == 0 && d->out_cstride >= d->out_coff + d->Cout,
                        "v2x_conv2d(halo): plain epilogue is bf16, unsplit");
        }
        const int rc = v2x_conv_halo_dispatch(d, (hipStream_t)stream);
        V2X_REQUIRE(rc != 1, "v2x_conv2d(halo): no halo kernel for C0=%d C1=%d Cout=%d Cout2=%d epilogue=%d", d->C0, d->C1,
                    d->Cout, d->Cout2, d->epilogue);
        return rc;
    }
    V2X_REQUIRE(d->Cout2 == 0 || d->w_layout == 2, "v2x_conv2d: chained 1x1 needs the halo or streamed layout (w_layout 1/2)");
    V2X_REQUIRE(d->in_format == 0, "v2x_conv2d: bit-grid input needs the halo layout (w_layout=1)");
    if (d->w_layout == 4)
        V2X_REQUIRE(d->C0 > 0 && d->C1 > 0 && d->up0 == 1 && d->stride == 1 && (d->Cout % 128 == 0 || d->Cout == 64) && d->Cout2 == 0 && d->epilogue == V2X_EPI_BF16 &&
                    d->H % 16 == 0 && d->W % (d->Cout == 64 ? 64 : 32) == 0 && d->w_kpad == 16 * d->C0 + 9 * d->C1,
                    "v2x_conv2d(stream, parity-class weights): needs the upsampled + skip source pair, stride 1, Cout %% 128 == 0 (or 64), a plain bf16 epilogue, H %% 16 == 0, W %% 32 == 0 (64 at Cout 64) and w_kpad = 16 C0 + 9 C1");
    if (d->w_layout == 2 || d->w_layout == 4) {
        V2X_REQUIRE(d->in0 && d->weight && d->scale && d->out, "v2x_conv2d(stream): null tensor pointer");
        V2X_REQUIRE(d->epilogue == V2X_EPI_GRU || d->shift, "v2x_conv2d(stream): null shift");
        V2X_REQUIRE(d->ksize == 3 && (d->stride == 1 || d->stride == 2) && d->pad == 1, "v2x_conv2d(stream): 3x3 stride 1/2 pad 1 only");
        V2X_REQUIRE(d->C0 > 0 && d->C0 % 32 == 0 && d->C1 >= 0 && d->C1 % 32 == 0 && (d->C1 == 0 || d->in1),
                    "v2x_conv2d(stream): C0=%d, C1=%d must be multiples of 32", d->C0, d->C1);
        V2X_REQUIRE(d->up0 == 0 || (d->up0 == 1 && d->H % 2 == 0 && d->W % 2 == 0), "v2x_conv2d(stream): bad up0");
        V2X_REQUIRE(d->N > 0 && (long long)d->N * d->H * d->W * (d->C0 > d->C1 ? d->C0 : d->C1) < (1ll << 32),
                    "v2x_conv2d(stream): tensor exceeds 32-bit element offsets");
        const int rows = d->w_layout == 4 ? (d->Cout == 64 ? 64 : 128) : v2x_conv_stream_tile_rows(d->Cout, d->epilogue);
        const int need = d->epilogue == V2X_EPI_GRU ? 3 * d->Cout : d->Cout;
        V2X_REQUIRE(rows > 0 && d->w_rows == need && d->w_rows % rows == 0 && d->split == 0 &&
                    d->out_cstride >= d->out_coff + d->Cout && d->out_coff >= 0,
                    "v2x_conv2d(stream): unsupported Cout=%d / w_rows=%d / output window", d->Cout, d->w_rows);
        if (d->stride == 2) {
            const int rc2 = v2x_conv_stream_s2_dispatch(d, (hipStream_t)stream);
            V2X_REQUIRE(rc2 != 1, "v2x_conv2d(stream, stride 2): needs one bf16 source, H %% 8 == 0, W %% 64 == 0 (got %dx%d)", d->H, d->W);
            return rc2;
        }
        const int rc = v2x_conv_stream_dispatch(d, (hipStream_t)stream);
        V2X_REQUIRE(rc != 1, "v2x_conv2d(stream): extent %dx%d not tileable (8x32 or 16x16 tiles)", d->H, d->W);
        return rc;
    }
    V2X_REQUIRE(d->in0 && d->weight && d->scale && d->out, "v2x_conv2d: null tensor pointer");
    V2X_REQUIRE(d->C0 > 0 && d->C0 % 8 == 0 && d->C1 >= 0 && d->C1 % 8 == 0, "v2x_conv2d: C0=%d C1=%d must be multiples of 8", d->C0, d->C1);
    V2X_REQUIRE(d->C1 == 0 || d->in1, "v2x_conv2d: C1 > 0 needs in1");
    V2X_REQUIRE(d->up0 == 0 || d->up0 == 1, "v2x_conv2d: up0 must be 0 or 1");
    V2X_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "v2x_conv2d: bad input extent");
    V2X_REQUIRE(d->up0 == 0 || (d->H % 2 == 0 && d->W % 2 == 0), "v2x_conv2d: upsampled source needs even H, W");
    V2X_REQUIRE((d->ksize == 1 && d->pad == 0) || (d->ksize == 3 && d->pad == 1), "v2x_conv2d: ksize/pad must be 1/0 or 3/1");
    V2X_REQUIRE(d->stride == 1 || d->stride == 2, "v2x_conv2d: stride must be 1 or 2");
    V2X_REQUIRE(d->epilogue >= V2X_EPI_BF16 && d->epilogue <= V2X_EPI_GRU, "v2x_conv2d: bad epilogue");
    V2X_REQUIRE(d->epilogue == V2X_EPI_GRU || d->shift, "v2x_conv2d: null shift");
    const int Cin = d->C0 + d->C1;
    const int rows = v2x_conv_tile_rows(d->Cout, d->epilogue);
    const int need_rows = (d->epilogue == V2X_EPI_GRU) ? 3 * d->Cout : d->Cout;
    V2X_REQUIRE(d->Cout > 0 && d->w_rows >= need_rows && d->w_rows % rows == 0,
                "v2x_conv2d: w_rows=%d must be a multiple of %d and >= %d", d->w_rows, rows, need_rows);
    V2X_REQUIRE(d->epilogue != V2X_EPI_GRU || d->Cout % 32 == 0, "v2x_conv2d: GRU hidden size must be a multiple of 32");
    V2X_REQUIRE(d->w_kpad % 64 == 0 && d->w_kpad >= d->ksize * d->ksize * Cin, "v2x_conv2d: w_kpad=%d invalid for K=%d", d->w_kpad, d->ksize * d->ksize * Cin);
    if (d->split > 0) {
        V2X_REQUIRE(d->epilogue != V2X_EPI_GRU && d->out2 && d->split % 4 == 0 && d->split < d->Cout,
                    "v2x_conv2d: split=%d must be a multiple of 4 below Cout with out2 set", d->split);
        V2X_REQUIRE(d->out_cstride >= d->out_coff + d->split && d->out_coff >= 0 && d->out2_cstride >= d->Cout - d->split,
                    "v2x_conv2d: bad split output channel windows");
    } else {
        V2X_REQUIRE(d->split == 0, "v2x_conv2d: negative split");
        V2X_REQUIRE(d->out_cstride >= d->out_coff + d->Cout && d->out_coff >= 0, "v2x_conv2d: bad output channel window");
    }

    if (d->ksize == 1 && v2x_tune(V2X_TUNE_CONV1X1) != 0) {     // 1x1 layers with <= 128 channels either side: the streaming kernel (conv1x1.hip), same bits
        const int rc1 = v2x_conv1x1_dispatch(d, (hipStream_t)stream);
        if (rc1 != 1) return rc1;
    }
    ConvArgs a;
    a.in0 = d->in0;
    a.in1 = d->in1;
    a.C0 = d->C0;
    a.C1 = d->C1;
    a.Cin = Cin;
    a.up0 = d->up0;
    a.N = d->N;
    a.H = d->H;
    a.W = d->W;
    a.ks = d->ksize;
    a.stride = d->stride;
    a.pad = d->pad;
    a.ntaps = d->ksize * d->ksize;
    a.Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1;
    a.Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
    const long long M = (long long)d->N * a.Ho * a.Wo;
    V2X_REQUIRE(M > 0 && M < (1ll << 31) - 512, "v2x_conv2d: too many output pixels");
    V2X_REQUIRE((long long)d->N * d->H * d->W < (1ll << 24), "v2x_conv2d: N*H*W must be < 2^24 (24-bit row arithmetic)");
    V2X_REQUIRE((long long)d->N * d->H * d->W * (d->C0 > d->C1 ? d->C0 : d->C1) < (1ll << 32) &&
                (long long)d->w_rows * d->w_kpad < (1ll << 32), "v2x_conv2d: tensor exceeds 32-bit element offsets");
    a.M = (int)M;
    a.Cout = d->Cout;
    a.w_rows = d->w_rows;
    a.w_kpad = d->w_kpad;
    a.w = d->weight;
    a.scale = d->scale;
    a.shift = d->shift;
    a.relu = d->relu;
    a.out = d->out;
    a.out_cstride = d->out_cstride;
    a.out_coff = d->out_coff;
    a.out2 = d->out2;
    a.split = d->split;
    a.out2_cstride = d->out2_cstride;
    hipStream_t s = (hipStream_t)stream;
    switch (d->epilogue) {
        case V2X_EPI_GRU: return launch_cfg<96, 128, 2, 2, EPI_GRU>(a, s);
        case V2X_EPI_F32: return dispatch_rows<EPI_F32>(a, rows, s);
        default: return dispatch_rows<EPI_BF16>(a, rows, s);
    }
}
