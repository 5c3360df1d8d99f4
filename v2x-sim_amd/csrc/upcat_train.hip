// cat(nearest x2 upsample(lo), skip) along the channels of bf16 NHWC maps, and its backward -- the decoder's up + concat of the TRAINING
// graph (row f-3).  Upstream: coperception/models/det/backbone/Backbone.py::LidarDecoder: torch.cat((F.interpolate(x, scale_factor=(2, 2)),
// skip), dim=1) in front of conv5_1 ... conv8_1 (code absent from /root/reference, see include/v2x_amd.h).  The inference kernels fold this into
// their patch loaders and never materialise it; the training graph needs the concatenated map as the weight-gradient kernel's input, and as
// PyTorch ops it was an expand-copy + a cat forward and two slice copies + a bf16 reduction backward per decoder level (~20 launches, 0.4 ms of a
// 10-map step).  One launch each way here; the backward's 2x2 sum is four bf16 values added in fp32 in a fixed order (row-major) and rounded
// once -- what torch's bf16 sum does -- no atomics.
#include "common.h"

namespace {
struct UpcatArgs {
    const uint16_t *a, *b;   // forward: lo [N][H][W][C0], skip [N][2H][2W][C1];  backward: dcat [N][2H][2W][C0+C1], unused
    uint16_t *o0, *o1;       // forward: out [N][2H][2W][C0+C1], unused;            backward: d_lo [N][H][W][C0], d_skip [N][2H][2W][C1]
    int N, H, W, C0, C1;     // H, W: the LOW-resolution extent
};

__global__ __launch_bounds__(256) void upcat_fwd_kernel(const UpcatArgs a) {
    const int v0 = a.C0 >> 3, vc = (a.C0 + a.C1) >> 3;
    const long long total = (long long)a.N * 2 * a.H * 2 * a.W * vc;
    const uint4 *lo = reinterpret_cast<const uint4 *>(a.a), *sk = reinterpret_cast<const uint4 *>(a.b);
    uint4 *out = reinterpret_cast<uint4 *>(a.o0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % vc);
        const long long pix = i / vc;                       // (n, y, x) of the full-resolution map
        if (cv < v0) {
            const int x = (int)(pix % (2 * a.W));
            const long long ny = pix / (2 * a.W);
            const int y = (int)(ny % (2 * a.H));
            const long long n = ny / (2 * a.H);
            out[i] = lo[((n * a.H + (y >> 1)) * a.W + (x >> 1)) * v0 + cv];
        } else {
            out[i] = sk[pix * (a.C1 >> 3) + (cv - v0)];
        }
    }
}

__device__ __forceinline__ void upcat_acc8(float (&s)[8], const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s[2 * k] += __uint_as_float(w[k] << 16);
        s[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
    }
}

__global__ __launch_bounds__(256) void upcat_bwd_kernel(const UpcatArgs a) {
    const int v0 = a.C0 >> 3, v1 = a.C1 >> 3, vc = v0 + v1;
    const long long n_lo = (long long)a.N * a.H * a.W * v0, n_sk = (long long)a.N * 2 * a.H * 2 * a.W * v1;
    const uint4 *dc = reinterpret_cast<const uint4 *>(a.a);
    uint4 *dlo = reinterpret_cast<uint4 *>(a.o0), *dsk = reinterpret_cast<uint4 *>(a.o1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_lo + n_sk; i += (long long)gridDim.x * 256) {
        if (i < n_lo) {
            const int cv = (int)(i % v0);
            const long long pix = i / v0;
            const int x = (int)(pix % a.W);
            const long long ny = pix / a.W;
            const int y = (int)(ny % a.H);
            const long long n = ny / a.H;
            const long long row0 = ((n * 2 * a.H + 2 * y) * 2 * a.W + 2 * x) * vc + cv;     // (2y, 2x)
            const long long row1 = row0 + (long long)2 * a.W * vc;                           // (2y + 1, 2x)
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            upcat_acc8(s, dc[row0]);
            upcat_acc8(s, dc[row0 + vc]);
            upcat_acc8(s, dc[row1]);
            upcat_acc8(s, dc[row1 + vc]);
            dlo[i] = make_uint4(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]), pack_bf16x2(s[4], s[5]), pack_bf16x2(s[6], s[7]));
        } else {
            const long long j = i - n_lo;
            const int cv = (int)(j % v1);
            dsk[j] = dc[(j / v1) * vc + v0 + cv];
        }
    }
}

int upcat_check(const char *who, const void *p0, const void *p1, const void *p2, int N, int H, int W, int C0, int C1) {
    V2X_REQUIRE(p0 && p1 && p2, "%s: null pointer", who);
    V2X_REQUIRE(N >= 0 && H > 0 && W > 0 && C0 > 0 && C1 > 0 && C0 % 8 == 0 && C1 % 8 == 0, "%s: bad extent (channels in multiples of 8)", who);
    V2X_REQUIRE((long long)N * 4 * H * W * (C0 + C1) < (1ll << 40), "%s: map too large", who);
    return V2X_OK;
}
int upcat_grid(long long vectors) {
    long long b = (vectors + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
}  // namespace

extern "C" int v2x_upcat_bf16(const uint16_t *lo, const uint16_t *skip, int N, int H, int W, int C0, int C1, uint16_t *out, v2x_stream_t stream) {
    if (int rc = upcat_check("v2x_upcat_bf16", lo, skip, out, N, H, W, C0, C1)) return rc;
    if (N == 0) return V2X_OK;
    UpcatArgs a{lo, skip, out, nullptr, N, H, W, C0, C1};
    hipLaunchKernelGGL(upcat_fwd_kernel, dim3(upcat_grid((long long)N * 4 * H * W * ((C0 + C1) / 8))), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("v2x_upcat_bf16");
    return V2X_OK;
}

extern "C" int v2x_upcat_bwd_bf16(const uint16_t *dcat, int N, int H, int W, int C0, int C1, uint16_t *d_lo, uint16_t *d_skip, v2x_stream_t stream) {
    if (int rc = upcat_check("v2x_upcat_bwd_bf16", dcat, d_lo, d_skip, N, H, W, C0, C1)) return rc;
    if (N == 0) return V2X_OK;
    UpcatArgs a{dcat, nullptr, d_lo, d_skip, N, H, W, C0, C1};
    hipLaunchKernelGGL(upcat_bwd_kernel, dim3(upcat_grid((long long)N * H * W * (C0 / 8) + (long long)N * 4 * H * W * (C1 / 8))), dim3(256), 0,
                       (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("v2x_upcat_bwd_bf16");
    return V2X_OK;
}

// ---- zero insertion: the gradient of a stride-2 convolution's output, spread onto the input grid ------------------------------------------------
// The data gradient of a 3x3 stride-2 layer runs as a stride-1 convolution (flipped, transposed weights) over dy with zeros between its
// pixels: out[n][2y][2x] = dy[n][y][x], everything else 0.  As torch ops that is a fill of the 4x tensor plus a strided copy into it (two
// launches, the big tensor written twice: 4 layers per FaFNet step); here one pass writes every 16-byte vector once.
__global__ __launch_bounds__(256) void zero_insert_kernel(const uint16_t *__restrict__ dy, uint16_t *__restrict__ out, int N, int Ho, int Wo, int C8) {
    const long long total = (long long)N * 2 * Ho * 2 * Wo * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long long p = i / C8;
        const int x = (int)(p % (2 * Wo));
        p /= 2 * Wo;
        const int y = (int)(p % (2 * Ho));
        const int n = (int)(p / (2 * Ho));
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (((x | y) & 1) == 0) v = reinterpret_cast<const uint4 *>(dy)[(((long long)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C8 + c];
        reinterpret_cast<uint4 *>(out)[i] = v;
    }
}

extern "C" int v2x_zero_insert_bf16(const uint16_t *dy, int N, int Ho, int Wo, int C, uint16_t *out, v2x_stream_t stream) {
    V2X_REQUIRE(dy && out, "v2x_zero_insert_bf16: null pointer");
    V2X_REQUIRE(N >= 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "v2x_zero_insert_bf16: needs C %% 8 == 0 and positive extents");
    if (N == 0) return V2X_OK;
    hipLaunchKernelGGL(zero_insert_kernel, dim3(upcat_grid((long long)N * 4 * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream, dy, out, N, Ho, Wo, C / 8);
    V2X_CHECK_LAUNCH("v2x_zero_insert_bf16");
    return V2X_OK;
}
