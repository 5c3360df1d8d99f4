// a3 -- cross-agent spatial feature warp, fused with the reduction that consumes it.
//
// Replaces upstream coperception/models/det/base/DetModelBase.py::feature_transformation
// (two affine_grid + two grid_sample launches per neighbour, issued from an O(B*A^2)
// python loop) together with torch.mean(torch.stack(...)) of V2VNet.py and the
// attention-weighted sum of When2com.py (code absent from /root/reference; v2x_amd.h).
//
// The reference resamples twice -- rotate about the map centre, then translate -- and
// the intermediate image matters (its zero padding and its bilinear smoothing are not
// the same as one composed affine resample).  We keep those semantics exactly but never
// materialise the intermediate: for an output pixel the translate step needs 4 taps of
// the rotated image, each of which is 4 taps of the neighbour map -> 16 gathered 16-B
// loads per 8 channels, all L2-resident (a 256x32x32 bf16 map is 512 KiB), accumulated
// in fp32 in the reference's tap order (nw, ne, sw, se).  One launch handles every
// (ego, neighbour) pair of every frame: no per-pair launches, no stack/mean temporaries.
//
// Coordinate conventions = PyTorch 1.8 defaults (README.md:88-95): align_corners=False,
//   base grid x_k = (2k+1)/W - 1,  unnormalise f = (g+1)*W/2 - 0.5,  zeros padding.
#include "common.h"

struct Bilin {
    int x0, y0;
    float nw, ne, sw, se;
};

__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int W, int H) {
    // explicit fused forms: every kernel variant must round these identically (with implicit contraction the compiler is
    // free to fuse a different product in each instantiation, which made the variants differ in the last bit)
    const float fx = __fmaf_rn(gx + 1.0f, 0.5f * (float)W, -0.5f);
    const float fy = __fmaf_rn(gy + 1.0f, 0.5f * (float)H, -0.5f);
    const float xw = floorf(fx), yn = floorf(fy);
    const float w = fx - xw, e = 1.0f - w;
    const float n = fy - yn, s = 1.0f - n;
    Bilin b;
    // clamp before the int conversion so wild poses cannot overflow
    b.x0 = (int)fminf(fmaxf(xw, -4.0f), (float)W + 4.0f);
    b.y0 = (int)fminf(fmaxf(yn, -4.0f), (float)H + 4.0f);
    b.nw = e * s;
    b.ne = w * s;
    b.sw = e * n;
    b.se = w * n;
    return b;
}

__device__ __forceinline__ void fma8(float (&acc)[8], const uint4 v, float w) {
    acc[0] += __uint_as_float(v.x << 16) * w;
    acc[1] += __uint_as_float(v.x & 0xffff0000u) * w;
    acc[2] += __uint_as_float(v.y << 16) * w;
    acc[3] += __uint_as_float(v.y & 0xffff0000u) * w;
    acc[4] += __uint_as_float(v.z << 16) * w;
    acc[5] += __uint_as_float(v.z & 0xffff0000u) * w;
    acc[6] += __uint_as_float(v.w << 16) * w;
    acc[7] += __uint_as_float(v.w & 0xffff0000u) * w;
}

// value of the ROTATED image of `src` at integer pixel (qx, qy), 8 channels
__device__ __forceinline__ void rot_sample(const uint16_t *__restrict__ src, int H, int W, int C, int cvec, int qx,
                                           int qy, float r00, float r01, float r10, float r11, float (&out)[8]) {
    const float xq = (float)(2 * qx + 1) / (float)W - 1.0f;
    const float yq = (float)(2 * qy + 1) / (float)H - 1.0f;
    const Bilin b = bilin_setup(__fmaf_rn(r00, xq, __fmul_rn(r01, yq)), __fmaf_rn(r10, xq, __fmul_rn(r11, yq)), W, H);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
    const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
    const uint16_t *base = src + (((long long)b.y0 * W + b.x0) * C + cvec * 8);  // only dereferenced in-bounds
    if (yt && xl) fma8(out, *reinterpret_cast<const uint4 *>(base), b.nw);
    if (yt && xr) fma8(out, *reinterpret_cast<const uint4 *>(base + C), b.ne);
    if (yb && xl) fma8(out, *reinterpret_cast<const uint4 *>(base + (long long)W * C), b.sw);
    if (yb && xr) fma8(out, *reinterpret_cast<const uint4 *>(base + (long long)W * C + C), b.se);
}

// The coordinate work of a tap (rotation, two bilinear set-ups, clamps, address) is the same for every channel of a pixel;
// with one thread per 8-channel vector it was ~60 % of the instruction stream of a VALU-bound kernel (~2400 VALU
// instructions per thread; 16 per gathered 16-B load are the inherent unpack + FMA).  Here a thread owns CV consecutive
// 8-channel vectors of its pixel, so the set-up is paid once per CV*8 channels; per pixel the lanes still read one
// contiguous C*2-byte run.  Measured at 160 output maps: CV=1 336 us, CV=2 257 us, CV=4 290 us (166 VGPRs) per launch.
template <int CV>
__device__ __forceinline__ void rot_sample_cv(const uint16_t *__restrict__ src, int H, int W, int C, int c0, int qx, int qy,
                                              float r00, float r01, float r10, float r11, float (&out)[CV][8]) {
    const float xq = (float)(2 * qx + 1) / (float)W - 1.0f;
    const float yq = (float)(2 * qy + 1) / (float)H - 1.0f;
    const Bilin b = bilin_setup(__fmaf_rn(r00, xq, __fmul_rn(r01, yq)), __fmaf_rn(r10, xq, __fmul_rn(r11, yq)), W, H);
#pragma unroll
    for (int v = 0; v < CV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[v][e] = 0.f;
    const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
    const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
    const uint16_t *base = src + (((long long)b.y0 * W + b.x0) * C + c0);  // only dereferenced in-bounds
    if (yt && xl) {
#pragma unroll
        for (int v = 0; v < CV; ++v) fma8(out[v], *reinterpret_cast<const uint4 *>(base + v * 8), b.nw);
    }
    if (yt && xr) {
#pragma unroll
        for (int v = 0; v < CV; ++v) fma8(out[v], *reinterpret_cast<const uint4 *>(base + C + v * 8), b.ne);
    }
    if (yb && xl) {
#pragma unroll
        for (int v = 0; v < CV; ++v) fma8(out[v], *reinterpret_cast<const uint4 *>(base + (long long)W * C + v * 8), b.sw);
    }
    if (yb && xr) {
#pragma unroll
        for (int v = 0; v < CV; ++v) fma8(out[v], *reinterpret_cast<const uint4 *>(base + (long long)W * C + C + v * 8), b.se);
    }
}

template <int CV>
__global__ __launch_bounds__(256) void warp_fuse_kernel(const uint16_t *__restrict__ feat, int A, int Bt, int H, int W,
                                                        int C, const float *__restrict__ trans,
                                                        const int32_t *__restrict__ items,
                                                        const float *__restrict__ coef, int mode,
                                                        uint16_t *__restrict__ out) {
    const int m = blockIdx.y;
    const int ego = items[2 * m + 0];
    const int f = items[2 * m + 1];
    const int cgroups = C / (8 * CV);
    const int total = H * W * cgroups;
    const size_t map_elems = (size_t)H * W * C;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int pix = idx / cgroups;
        const int c0 = (idx - pix * cgroups) * (8 * CV);
        const int h = pix / W, w = pix - h * W;
        float acc[CV][8];
#pragma unroll
        for (int v = 0; v < CV; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[v][e] = 0.f;
        int count = 0;
        for (int j = 0; j < A; ++j) {
            const float cj = coef[m * A + j];
            if (cj == 0.0f) continue;  // block-uniform
            const uint16_t *src = feat + ((size_t)j * Bt + f) * map_elems;
            ++count;
            const float wj = (mode == V2X_FUSE_MEAN) ? 1.0f : cj;
            if (j == ego) {
#pragma unroll
                for (int v = 0; v < CV; ++v) {
                    if (mode == V2X_FUSE_MAX) {
                        float ev[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) ev[e] = 0.f;
                        fma8(ev, *reinterpret_cast<const uint4 *>(src + (size_t)pix * C + c0 + v * 8), 1.0f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[v][e] = (count == 1) ? ev[e] : fmaxf(acc[v][e], ev[e]);
                    } else {
                        fma8(acc[v], *reinterpret_cast<const uint4 *>(src + (size_t)pix * C + c0 + v * 8), wj);
                    }
                }
                continue;
            }
            const float *T = trans + (((size_t)f * A + ego) * A + j) * 16;
            const float r00 = T[0], r01 = T[1], r10 = T[4], r11 = T[5];
            const float tx = (4.0f * T[3]) / 128.0f;
            const float ty = -((4.0f * T[7]) / 128.0f);
            const float x = (float)(2 * w + 1) / (float)W - 1.0f;
            const float y = (float)(2 * h + 1) / (float)H - 1.0f;
            const Bilin b = bilin_setup(x + tx, y + ty, W, H);
            float vv[CV][8], r[CV][8];
#pragma unroll
            for (int v = 0; v < CV; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[v][e] = 0.f;
            const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
            const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
#define V2X_TAP(COND, QX, QY, WT)                                                   \
    if (COND) {                                                                     \
        rot_sample_cv<CV>(src, H, W, C, c0, QX, QY, r00, r01, r10, r11, r);         \
        _Pragma("unroll") for (int v = 0; v < CV; ++v)                              \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) vv[v][e] += r[v][e] * WT; \
    }
            V2X_TAP(yt && xl, b.x0, b.y0, b.nw)
            V2X_TAP(yt && xr, b.x0 + 1, b.y0, b.ne)
            V2X_TAP(yb && xl, b.x0, b.y0 + 1, b.sw)
            V2X_TAP(yb && xr, b.x0 + 1, b.y0 + 1, b.se)
#undef V2X_TAP
#pragma unroll
            for (int v = 0; v < CV; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (mode == V2X_FUSE_MAX) acc[v][e] = (count == 1) ? vv[v][e] : fmaxf(acc[v][e], vv[v][e]);
                    else acc[v][e] += vv[v][e] * wj;
                }
        }
#pragma unroll
        for (int v = 0; v < CV; ++v) {
            if (mode == V2X_FUSE_MEAN && count > 0) {
                const float d = (float)count;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[v][e] = acc[v][e] / d;
            }
            uint4 o;
            o.x = pack_bf16x2(acc[v][0], acc[v][1]);
            o.y = pack_bf16x2(acc[v][2], acc[v][3]);
            o.z = pack_bf16x2(acc[v][4], acc[v][5]);
            o.w = pack_bf16x2(acc[v][6], acc[v][7]);
            *reinterpret_cast<uint4 *>(out + (size_t)m * map_elems + (size_t)pix * C + c0 + v * 8) = o;
        }
    }
}

extern "C" int v2x_warp_fuse(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans,
                             const int32_t *items, int n_out, const float *coef, int mode, uint16_t *out,
                             v2x_stream_t stream) {
    V2X_REQUIRE(feat && trans && items && coef && out, "v2x_warp_fuse: null pointer");
    V2X_REQUIRE(A > 0 && A <= 32 && Bt > 0 && H > 0 && W > 0, "v2x_warp_fuse: bad dims");
    V2X_REQUIRE(C > 0 && C % 8 == 0, "v2x_warp_fuse: C=%d must be a multiple of 8", C);
    V2X_REQUIRE(mode == V2X_FUSE_WSUM || mode == V2X_FUSE_MEAN || mode == V2X_FUSE_MAX, "v2x_warp_fuse: bad mode");
    V2X_REQUIRE(n_out >= 0 && n_out <= 65535, "v2x_warp_fuse: n_out out of range");
    if (n_out == 0) return V2X_OK;
    const int cv = C % 16 == 0 ? 2 : 1;
    const int total = H * W * (C / (8 * cv));
    int gx = (total + 255) / 256;
    if (gx > 1024) gx = 1024;
    auto kern = cv == 2 ? warp_fuse_kernel<2> : warp_fuse_kernel<1>;
    hipLaunchKernelGGL(kern, dim3(gx, n_out), dim3(256), 0, (hipStream_t)stream, feat, A, Bt, H, W, C,
                       trans, items, coef, mode, out);
    V2X_CHECK_LAUNCH("warp_fuse_kernel");
    return V2X_OK;
}
