// 1x1 layers as a STREAMING kernel (round 6): upstream's Conv3D 1x1x1 layers (Backbone.py::LidarEncoder.conv3d_1 / conv3d_2), the
// detection heads' last layers (DetModelBase.py: ClassificationHead.conv2, SingleRegressionHead's last conv) and, in the training graph, their data gradients
// dx = dy . W (train/hip_graph.py::_Conv1x1) -- code absent from /root/reference, see include/v2x_amd.h.
//
// Why: in INFERENCE these layers ride in the epilogue of the 3x3 layer in front of them.  In TRAINING a batch-statistics BatchNorm sits between the two, so
// each is a launch of its own, and until round 6 that launch was the gather kernel (conv_igemm.hip): an implicit-GEMM built for 3x3 taps -- LDS staging,
// a K loop with barriers, 256-pixel tiles -- around a contraction of 32 ... 128 channels.  Measured at 40 maps (profiles/r04_train_step_profile.txt):
// 170-230 us per launch for 60-170 MB of traffic, i.e. 0.3-1.0 TB/s; ten such launches were 13 % of a FaFNet training step.
//
// A 1x1 layer is a pure stream: out[p][:] = W x[p][:] + b, 64-512 bytes in and out per pixel, the whole weight matrix <= 32 KiB.  So:
//   * the weights live in REGISTERS as MFMA A fragments for the whole kernel (KS x CT fragments of 4 registers: 128 at 128 x 128, 4 at 32 x 12);
//   * a wave owns runs of PF 16-pixel fragments: it loads their channels straight from HBM into B fragments (16 bytes per lane, one pixel's
//     32 * KS channels are contiguous: a fragment is one 512-B ... 2-KiB contiguous read), multiplies, and stores -- no LDS, no barrier,
//     nothing shared between waves; PF * KS loads per lane are in flight per iteration and the grid is sized for ~16 waves per CU;
//   * K order = ascending 32-channel chunks, one v_mfma_f32_16x16x32_bf16 each, epilogue acc * scale + shift (+ ReLU), the gather kernel's arithmetic:
//     results are bit-identical to it (tests/test_gpu_train_kernels.py), so the switch CONV1X1 (default 1) changes time only.
// Roof: HBM.  Algorithmic bytes per pixel = 2 Cin + (2 | 4) Cout.
#include "common.h"

struct C1Args {
    const uint16_t *in;    // [M][C0] bf16
    const uint16_t *w;     // [w_rows][w_kpad] bf16, row-major (gather layout, w_layout 0)
    const float *scale, *shift;
    void *out;             // [M][out_cstride] (+ out_coff), bf16 or fp32
    int M, C0, Cout, w_kpad, out_cstride, out_coff, relu;
    int x4;                // bf16 output, Cout % 32 == 0, 16-byte aligned rows: 16-byte stores (same bytes)
};

template <int KS, int CT, bool F32, int PF>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const C1Args a) {
    const int lane = threadIdx.x & 63;
    const int fj = lane & 15, fq = lane >> 4;
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
    const int n_waves = (int)(gridDim.x * 4u);

    bf16x8_t A[CT][KS];
    float4 sc[CT], sf[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            A[i][ks] = *reinterpret_cast<const bf16x8_t *>(a.w + (size_t)(i * 16 + fj) * a.w_kpad + ks * 32 + fq * 8);   // (rows beyond Cout are zero rows of the packing)
        const int co = i * 16 + fq * 4;
        // scale / shift hold w_rows >= 16 CT entries (packing pads them with 1 / 0)
        sc[i] = *reinterpret_cast<const float4 *>(a.scale + co);
        sf[i] = *reinterpret_cast<const float4 *>(a.shift + co);
    }
    const int n_frag = (a.M + 15) >> 4;
    for (int f0 = wave * PF; f0 < n_frag; f0 += n_waves * PF) {
        bf16x8_t B[PF][KS];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            int p = (f0 + f) * 16 + fj;
            p = p < a.M ? p : a.M - 1;                           // clamped: the lanes behind the end load a valid pixel and store nothing
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) B[f][ks] = *reinterpret_cast<const bf16x8_t *>(a.in + (size_t)p * a.C0 + ks * 32 + fq * 8);
        }
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int p = (f0 + f) * 16 + fj;
            f32x4_t acc[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][ks], B[f][ks], acc[i], 0, 0, 0);
            }
            if (p >= a.M) continue;
            if constexpr (!F32 && CT % 2 == 0) {
                if (a.x4) {     // 16-byte stores: two channel tiles exchanged between the k-slot quarters (common.h: v2x_store_pair_x4); Cout % 32 == 0 here
                    const uint32_t floor_bits = a.relu ? 0u : 0x80008000u;
#pragma unroll
                    for (int i = 0; i < CT; i += 2) {
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][0] * sc[i + h].x + sf[i + h].x, acc[i + h][1] * sc[i + h].y + sf[i + h].y), floor_bits);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][2] * sc[i + h].z + sf[i + h].z, acc[i + h][3] * sc[i + h].w + sf[i + h].w), floor_bits);
                        }
                        v2x_store_pair_x4(reinterpret_cast<uint16_t *>(a.out) + (size_t)p * a.out_cstride + a.out_coff + i * 16 + fq * 4, fq, ox[0], oy[0], ox[1], oy[1]);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int co = i * 16 + fq * 4;
                if (co >= a.Cout) continue;
                float y[4];
                y[0] = acc[i][0] * sc[i].x + sf[i].x;
                y[1] = acc[i][1] * sc[i].y + sf[i].y;
                y[2] = acc[i][2] * sc[i].z + sf[i].z;
                y[3] = acc[i][3] * sc[i].w + sf[i].w;
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
                }
                // Cout, the channel stride and the channel offset are multiples of 4 (the dispatch sends anything else to the gather kernel): a lane's four
                // channels are all inside the layer and its store is aligned
                const size_t o = (size_t)p * a.out_cstride + a.out_coff + co;
                if constexpr (F32) {
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.out) + o) = make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    uint2 v;
                    v.x = pack_bf16x2(y[0], y[1]);
                    v.y = pack_bf16x2(y[2], y[3]);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.out) + o) = v;
                }
            }
        }
    }
}

int v2x_num_cus();   // conv_stream.hip

template <int KS, int CT, bool F32>
static int launch_c1(const C1Args &a, hipStream_t s) {
    constexpr int PF = (KS * CT >= 16) ? 2 : 4;                  // 128 x 128: 128 weight registers -- two fragments in flight; small layers four
    const int n_frag = (a.M + 15) / 16;
    int grid = (n_frag + 4 * PF - 1) / (4 * PF);                 // one run of PF fragments per wave ...
    const int cap = v2x_num_cus() * 8;                           // ... up to 8 workgroups (32 waves) per CU, then the waves loop
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((conv1x1_stream_kernel<KS, CT, F32, PF>), dim3(grid), dim3(256), 0, s, a);
    V2X_CHECK_LAUNCH("conv1x1_stream_kernel");
    return V2X_OK;
}

template <int KS, bool F32>
static int dispatch_ct(const C1Args &a, int ct, hipStream_t s) {
    switch (ct) {
        case 1: return launch_c1<KS, 1, F32>(a, s);
        case 2: return launch_c1<KS, 2, F32>(a, s);
        case 3: return launch_c1<KS, 3, F32>(a, s);
        case 4: return launch_c1<KS, 4, F32>(a, s);
        case 6: return launch_c1<KS, 6, F32>(a, s);
        case 8: return launch_c1<KS, 8, F32>(a, s);
        default: return 1;
    }
}

// Called by v2x_conv2d (conv_igemm.hip) for a gather-layout 1x1 layer.  -> V2X_OK, an error code, or 1 = "not mine" (the gather kernel takes it).
int v2x_conv1x1_dispatch(const v2x_conv_desc *d, hipStream_t s) {
    if (!(d->ksize == 1 && d->stride == 1 && d->pad == 0 && d->C1 == 0 && d->up0 == 0 && d->split == 0 && d->Cout2 == 0 && d->in_format == 0 &&
          (d->epilogue == V2X_EPI_BF16 || d->epilogue == V2X_EPI_F32) && d->splitk <= 1))
        return 1;
    if (d->C0 % 32 != 0 || d->C0 > 128 || d->Cout < 4 || d->Cout > 128 || d->Cout % 4 != 0 || d->w_kpad < d->C0 || d->w_kpad % 8 != 0) return 1;
    if (d->out_cstride % 4 != 0 || d->out_coff % 4 != 0 || d->out_coff < 0 || d->out_cstride < d->out_coff + d->Cout || !d->out ||
        (reinterpret_cast<uintptr_t>(d->out) & 15))
        return 1;
    const int ct = (d->Cout + 15) / 16;
    if (d->w_rows < ct * 16) return 1;                           // the fragments read whole 16-row tiles (the packing pads rows with zeros, scale 1, shift 0)
    if ((reinterpret_cast<uintptr_t>(d->in0) & 15) || (reinterpret_cast<uintptr_t>(d->weight) & 15) || (reinterpret_cast<uintptr_t>(d->scale) & 15) ||
        (reinterpret_cast<uintptr_t>(d->shift) & 15))
        return 1;
    const long long M = (long long)d->N * d->H * d->W;
    if (M <= 0 || M >= (1ll << 31) - 64) return 1;
    C1Args a;
    a.in = d->in0;
    a.w = d->weight;
    a.scale = d->scale;
    a.shift = d->shift;
    a.out = d->out;
    a.M = (int)M;
    a.C0 = d->C0;
    a.Cout = d->Cout;
    a.w_kpad = d->w_kpad;
    a.out_cstride = d->out_cstride;
    a.out_coff = d->out_coff;
    a.relu = d->relu;
    a.x4 = d->epilogue == V2X_EPI_BF16 ? v2x_x4_ok(d->out, d->out_cstride, d->out_coff, d->Cout) : 0;
    const bool f32 = d->epilogue == V2X_EPI_F32;
    switch (d->C0 / 32) {
        case 1: return f32 ? dispatch_ct<1, true>(a, ct, s) : dispatch_ct<1, false>(a, ct, s);
        case 2: return f32 ? dispatch_ct<2, true>(a, ct, s) : dispatch_ct<2, false>(a, ct, s);
        case 4: return f32 ? dispatch_ct<4, true>(a, ct, s) : dispatch_ct<4, false>(a, ct, s);
        default: return 1;
    }
}
