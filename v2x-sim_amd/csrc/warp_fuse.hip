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
    // explicit fma: every kernel form must round identically (see bilin_setup)
    acc[0] = __fmaf_rn(__uint_as_float(v.x << 16), w, acc[0]);
    acc[1] = __fmaf_rn(__uint_as_float(v.x & 0xffff0000u), w, acc[1]);
    acc[2] = __fmaf_rn(__uint_as_float(v.y << 16), w, acc[2]);
    acc[3] = __fmaf_rn(__uint_as_float(v.y & 0xffff0000u), w, acc[3]);
    acc[4] = __fmaf_rn(__uint_as_float(v.z << 16), w, acc[4]);
    acc[5] = __fmaf_rn(__uint_as_float(v.z & 0xffff0000u), w, acc[5]);
    acc[6] = __fmaf_rn(__uint_as_float(v.w << 16), w, acc[6]);
    acc[7] = __fmaf_rn(__uint_as_float(v.w & 0xffff0000u), w, acc[7]);
}

// The coordinate work of a tap (rotation, two bilinear set-ups, clamps, address) is the same for every channel of a pixel;
// with one thread per 8-channel vector it was ~60 % of the instruction stream of a VALU-bound kernel (~2400 VALU
// instructions per thread; 16 per gathered 16-B load are the inherent unpack + FMA).  Here a thread owns CV consecutive
// 8-channel vectors of its pixel, so the set-up is paid once per CV*8 channels; per pixel the lanes still read one
// contiguous C*2-byte run.  Measured at 160 output maps: CV=1 336 us, CV=2 257 us, CV=4 290 us (166 VGPRs) per launch.
template <int CV>
__device__ __forceinline__ void rot_sample_cv(const uint16_t *__restrict__ src, int H, int W, int C, int c0, int qx, int qy,
                                              float r00, float r01, float r10, float r11, float (&out)[CV][8],
                                              int vstride = 8) {
    const float xq = (float)(2 * qx + 1) / (float)W - 1.0f;
    const float yq = (float)(2 * qy + 1) / (float)H - 1.0f;
    const Bilin b = bilin_setup(__fmaf_rn(r00, xq, __fmul_rn(r01, yq)), __fmaf_rn(r10, xq, __fmul_rn(r11, yq)), W, H);
#pragma unroll
    for (int v = 0; v < CV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[v][e] = 0.f;
    const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
    const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
    if (!((xl || xr) && (yt || yb))) return;   // the whole footprint is zero padding
    // All four taps are loaded unconditionally from CLAMPED coordinates and a tap that is zero padding gets weight 0: a
    // load under its own branch is followed by a full s_waitcnt (16 serialised L2 round trips per output and neighbour --
    // the kernel sat at 54 % wave-wait), here the 4*CV loads of a sample are in flight together.  fma(x, 0, acc) == acc
    // exactly for the finite x of a real map (acc is never -0), so the result is unchanged.
    const int cx0 = min(max(b.x0, 0), W - 1), cx1 = min(max(b.x0 + 1, 0), W - 1);
    const int cy0 = min(max(b.y0, 0), H - 1), cy1 = min(max(b.y0 + 1, 0), H - 1);
    const uint16_t *row0 = src + ((size_t)cy0 * W) * C + c0, *row1 = src + ((size_t)cy1 * W) * C + c0;
    uint4 t[4][CV];
#pragma unroll
    for (int v = 0; v < CV; ++v) {
        t[0][v] = *reinterpret_cast<const uint4 *>(row0 + (size_t)cx0 * C + v * vstride);
        t[1][v] = *reinterpret_cast<const uint4 *>(row0 + (size_t)cx1 * C + v * vstride);
        t[2][v] = *reinterpret_cast<const uint4 *>(row1 + (size_t)cx0 * C + v * vstride);
        t[3][v] = *reinterpret_cast<const uint4 *>(row1 + (size_t)cx1 * C + v * vstride);
    }
    const float wt[4] = {(yt && xl) ? b.nw : 0.f, (yt && xr) ? b.ne : 0.f, (yb && xl) ? b.sw : 0.f, (yb && xr) ? b.se : 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < CV; ++v) fma8(out[v], t[k][v], wt[k]);
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
            _Pragma("unroll") for (int e = 0; e < 8; ++e) vv[v][e] = __fmaf_rn(r[v][e], WT, vv[v][e]); \
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
                    else acc[v][e] = __fmaf_rn(vv[v][e], wj, acc[v][e]);
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

// ---- LDS-staged form -----------------------------------------------------------------------------------------------
// The translate step's 4 taps of neighbouring output pixels overlap: an 8x8 output tile touches only a ~9x9 window of
// the ROTATED image (the translation is a constant shift), so the direct form above evaluates every rotated sample ~3x.
// Here a workgroup owns one 8x8 tile x 128 channels of one output map and, per neighbour,
//   phase 1: evaluates the rotated image on the 9x9 window anchored at the tap origin of the tile's first pixel (same
//            rot_sample_cv arithmetic, fp32) and parks it in LDS (81 positions x 128 ch x 4 B, padded: 51 KiB, 3 WG/CU);
//   phase 2: every output pixel combines its 4 window entries with the translate weights, in the reference tap order.
// A tap that falls outside the window (possible only through float rounding of the tap origin) is evaluated directly,
// so the result is BIT-IDENTICAL to the direct form (tests/test_gpu_stages.py compares them); 324 instead of 1024
// gathered rotated taps per tile and channel vector.
constexpr int WL_T = 8;            // tile edge
constexpr int WL_R = WL_T + 1;     // window edge
constexpr int WL_CW = 128;         // channels per workgroup
constexpr int WL_G = WL_CW / 16;   // items per position (an item = 2 channel vectors: g and g + WL_G)
constexpr int WL_POS4 = 4 * WL_G + 8;   // float4 stride of a window position (+8: adjacent positions land on the other 32 banks)

__global__ __launch_bounds__(256) void warp_fuse_lds_kernel(const uint16_t *__restrict__ feat, int A, int Bt, int H, int W,
                                                            int C, const float *__restrict__ trans,
                                                            const int32_t *__restrict__ items,
                                                            const float *__restrict__ coef, int mode,
                                                            uint16_t *__restrict__ out, const int32_t *__restrict__ order, int order_stride,
                                                            int n_out_total) {
    __shared__ float4 win[WL_R * WL_R * WL_POS4];
    int m = blockIdx.y, tile = blockIdx.x, cbi = blockIdx.z;
    if (order) {
        // 1-D grid, frame-major per XCD: the workgroups of XCD x (linear id % 8) walk the output maps order[k], k = x, x + 8, ... is NOT enough --
        // the maps of one FRAME must share an XCD (each source map is read by every other agent of its frame): `order` lists the output maps
        // sorted by frame, `per_frame` of them per frame slot; XCD x takes the frame slots x, x + 8, ...
        const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
        const int tiles = (H / WL_T) * (W / WL_T), cbs = C / WL_CW;
        const int per_slot = order_stride * tiles * cbs;
        const int slot = (i / per_slot) * 8 + xcd, rest = i % per_slot;
        const int e = rest / (tiles * cbs), r2 = rest % (tiles * cbs);
        if (slot * order_stride + e >= n_out_total) return;
        m = order[slot * order_stride + e];
        if (m < 0) return;
        cbi = r2 / tiles;
        tile = r2 % tiles;
    }
    const int ego = items[2 * m + 0];
    const int f = items[2 * m + 1];
    const int tiles_x = W / WL_T;
    const int h0 = (tile / tiles_x) * WL_T, w0 = (tile % tiles_x) * WL_T;
    const int cb = cbi * WL_CW;
    const int tid = threadIdx.x;
    const size_t map_elems = (size_t)H * W * C;
    // phase-2 items of this thread: i = tid + 256 * s, pixel = i / WL_G, g = i % WL_G
    constexpr int NI = WL_T * WL_T * WL_G / 256;
    float acc[NI][2][8];
#pragma unroll
    for (int s = 0; s < NI; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[s][v][e] = 0.f;
    int count = 0;
    for (int j = 0; j < A; ++j) {
        const float cj = coef[m * A + j];
        if (cj == 0.0f) continue;  // block-uniform
        const uint16_t *src = feat + ((size_t)j * Bt + f) * map_elems;
        ++count;
        const float wj = (mode == V2X_FUSE_MEAN) ? 1.0f : cj;
        if (j == ego) {
#pragma unroll
            for (int s = 0; s < NI; ++s) {
                const int i = tid + 256 * s;
                const int pixel = i / WL_G, g = i % WL_G;
                const int pix = (h0 + pixel / WL_T) * W + w0 + pixel % WL_T;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(src + (size_t)pix * C + cb + (g + v * WL_G) * 8);
                    if (mode == V2X_FUSE_MAX) {
                        float ev[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) ev[e] = 0.f;
                        fma8(ev, raw, 1.0f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[s][v][e] = (count == 1) ? ev[e] : fmaxf(acc[s][v][e], ev[e]);
                    } else {
                        fma8(acc[s][v], raw, wj);
                    }
                }
            }
            continue;
        }
        const float *T = trans + (((size_t)f * A + ego) * A + j) * 16;
        const float r00 = T[0], r01 = T[1], r10 = T[4], r11 = T[5];
        const float tx = (4.0f * T[3]) / 128.0f;
        const float ty = -((4.0f * T[7]) / 128.0f);
        // window origin = tap origin of the tile's first pixel (workgroup-uniform)
        const Bilin b0 = bilin_setup(((float)(2 * w0 + 1) / (float)W - 1.0f) + tx, ((float)(2 * h0 + 1) / (float)H - 1.0f) + ty, W, H);
        const int qx_lo = b0.x0, qy_lo = b0.y0;
        __syncthreads();   // the previous neighbour's phase 2 is done with the window
        for (int i = tid; i < WL_R * WL_R * WL_G; i += 256) {
            const int pos = i / WL_G, g = i % WL_G;
            const int qx = qx_lo + pos % WL_R, qy = qy_lo + pos / WL_R;
            if ((unsigned)qx >= (unsigned)W || (unsigned)qy >= (unsigned)H) continue;   // never read: the tap is zero padding
            float r[2][8];
            rot_sample_cv<2>(src, H, W, C, cb + g * 8, qx, qy, r00, r01, r10, r11, r, WL_G * 8);
            float4 *dst = win + pos * WL_POS4 + g;
            dst[0 * WL_G] = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]);
            dst[1 * WL_G] = make_float4(r[0][4], r[0][5], r[0][6], r[0][7]);
            dst[2 * WL_G] = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]);
            dst[3 * WL_G] = make_float4(r[1][4], r[1][5], r[1][6], r[1][7]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            const int i = tid + 256 * s;
            const int pixel = i / WL_G, g = i % WL_G;
            const int h = h0 + pixel / WL_T, w = w0 + pixel % WL_T;
            const float x = (float)(2 * w + 1) / (float)W - 1.0f;
            const float y = (float)(2 * h + 1) / (float)H - 1.0f;
            const Bilin b = bilin_setup(x + tx, y + ty, W, H);
            float vv[2][8], r[2][8];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[v][e] = 0.f;
            const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
            const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
#define V2X_TAP(COND, QX, QY, WT)                                                                                 \
    if (COND) {                                                                                                   \
        const int dx = (QX)-qx_lo, dy = (QY)-qy_lo;                                                               \
        if ((unsigned)dx < (unsigned)WL_R && (unsigned)dy < (unsigned)WL_R) {                                     \
            const float4 *p = win + (dy * WL_R + dx) * WL_POS4 + g;                                               \
            const float4 a0 = p[0], a1 = p[WL_G], a2 = p[2 * WL_G], a3 = p[3 * WL_G];                             \
            r[0][0] = a0.x; r[0][1] = a0.y; r[0][2] = a0.z; r[0][3] = a0.w;                                       \
            r[0][4] = a1.x; r[0][5] = a1.y; r[0][6] = a1.z; r[0][7] = a1.w;                                       \
            r[1][0] = a2.x; r[1][1] = a2.y; r[1][2] = a2.z; r[1][3] = a2.w;                                       \
            r[1][4] = a3.x; r[1][5] = a3.y; r[1][6] = a3.z; r[1][7] = a3.w;                                       \
        } else {                                                                                                  \
            rot_sample_cv<2>(src, H, W, C, cb + g * 8, QX, QY, r00, r01, r10, r11, r, WL_G * 8);                  \
        }                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < 2; ++v)                                                             \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) vv[v][e] = __fmaf_rn(r[v][e], WT, vv[v][e]);                               \
    }
            V2X_TAP(yt && xl, b.x0, b.y0, b.nw)
            V2X_TAP(yt && xr, b.x0 + 1, b.y0, b.ne)
            V2X_TAP(yb && xl, b.x0, b.y0 + 1, b.sw)
            V2X_TAP(yb && xr, b.x0 + 1, b.y0 + 1, b.se)
#undef V2X_TAP
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (mode == V2X_FUSE_MAX) acc[s][v][e] = (count == 1) ? vv[v][e] : fmaxf(acc[s][v][e], vv[v][e]);
                    else acc[s][v][e] = __fmaf_rn(vv[v][e], wj, acc[s][v][e]);
                }
        }
    }
#pragma unroll
    for (int s = 0; s < NI; ++s) {
        const int i = tid + 256 * s;
        const int pixel = i / WL_G, g = i % WL_G;
        const int pix = (h0 + pixel / WL_T) * W + w0 + pixel % WL_T;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (mode == V2X_FUSE_MEAN && count > 0) {
                const float d = (float)count;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[s][v][e] = acc[s][v][e] / d;
            }
            uint4 o;
            o.x = pack_bf16x2(acc[s][v][0], acc[s][v][1]);
            o.y = pack_bf16x2(acc[s][v][2], acc[s][v][3]);
            o.z = pack_bf16x2(acc[s][v][4], acc[s][v][5]);
            o.w = pack_bf16x2(acc[s][v][6], acc[s][v][7]);
            *reinterpret_cast<uint4 *>(out + (size_t)m * map_elems + (size_t)pix * C + cb + (g + v * WL_G) * 8) = o;
        }
    }
}

// ---- LDS-staged form with SHARED set-ups (round 5) ------------------------------------------------------------------------------------
// In the kernel above a thread owns 16 channels of a window position (phase 1) or of an output pixel (phase 2), so the coordinate work of a
// position -- rotation, bilinear set-up, clamps, two IEEE divisions -- is issued once per 16 channels: 648 items per tile and neighbour walk
// the SAME 81 set-ups eight times each, ~55 of a phase-1 item's ~170 instructions.  Here the 81 rotate set-ups are evaluated ONCE per
// workgroup and neighbour (threads 0..80, before the barrier phase 1 starts behind anyway) into a 2.5-KiB LDS table {4 clamped source pixel
// offsets, 4 tap weights}; a phase-1 item reads its entry (two broadcast ds_read_b128) and goes straight to its 8 gathers.  The translate
// step's per-pixel normalised coordinates (2 w + 1) / W - 1 (two more divisions per phase-2 item) come from a 16-float table built once per
// workgroup.  Same arithmetic in the same order as rot_sample_cv / the kernel above: BIT-IDENTICAL results (tests compare the three forms).
// LDS: window 50.5 KiB + tables 2.0 KiB = 53 720 B: still three workgroups per CU (the entry is split into an int2 {clamped source pixel index of
// the nw tap or -1 = position outside the map / -2 = footprint outside, (dx, dy) steps to the other taps} and a float4 of weights).  MODE is a template parameter (the max / mean / weighted-sum
// selects and the ego branch's extra code left the inner loops).

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void warp_fuse_lds2_kernel(const uint16_t *__restrict__ feat, int A, int Bt, int H, int W, int C,
                                                             const float *__restrict__ trans, const int32_t *__restrict__ items,
                                                             const float *__restrict__ coef, uint16_t *__restrict__ out,
                                                             const int32_t *__restrict__ order, int order_stride, int n_out_total) {
    __shared__ float4 win[WL_R * WL_R * WL_POS4 - (WL_POS4 - 4 * WL_G)];   // (the last position needs no padding behind it)
    __shared__ float4 rwt[WL_R * WL_R];      // tap weights nw, ne, sw, se of a window position, 0 where the tap is zero padding
    __shared__ int2 roff[WL_R * WL_R];       // x: source pixel index cy0 * W + cx0 (clamped into the map) | -1 | -2;  y: (cx1 - cx0) | (cy1 - cy0) << 1
    __shared__ float base_xy[2 * WL_T];      // (2 w + 1) / W - 1 for the tile's columns, (2 h + 1) / H - 1 for its rows
    int m = blockIdx.y, tile = blockIdx.x, cbi = blockIdx.z;
    if (order) {   // frame-major walk per XCD (see warp_fuse_lds_kernel)
        const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
        const int tiles = (H / WL_T) * (W / WL_T), cbs = C / WL_CW;
        const int per_slot = order_stride * tiles * cbs;
        const int slot = (i / per_slot) * 8 + xcd, rest = i % per_slot;
        const int e = rest / (tiles * cbs), r2 = rest % (tiles * cbs);
        if (slot * order_stride + e >= n_out_total) return;
        m = order[slot * order_stride + e];
        if (m < 0) return;
        cbi = r2 / tiles;
        tile = r2 % tiles;
    }
    const int ego = items[2 * m + 0];
    const int f = items[2 * m + 1];
    const int tiles_x = W / WL_T;
    const int h0 = (tile / tiles_x) * WL_T, w0 = (tile % tiles_x) * WL_T;
    const int cb = cbi * WL_CW;
    const int tid = threadIdx.x;
    const size_t map_elems = (size_t)H * W * C;
    if (tid < 2 * WL_T) {
        const int k = tid & (WL_T - 1);
        base_xy[tid] = tid < WL_T ? (float)(2 * (w0 + k) + 1) / (float)W - 1.0f : (float)(2 * (h0 + k) + 1) / (float)H - 1.0f;
    }
    constexpr int NI = WL_T * WL_T * WL_G / 256;
    float acc[NI][2][8];
#pragma unroll
    for (int s = 0; s < NI; ++s)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[s][v][e] = 0.f;
    int count = 0;
    for (int j = 0; j < A; ++j) {
        const float cj = coef[m * A + j];
        if (cj == 0.0f) continue;  // block-uniform
        const uint16_t *src = feat + ((size_t)j * Bt + f) * map_elems;
        ++count;
        const float wj = (MODE == V2X_FUSE_MEAN) ? 1.0f : cj;
        if (j == ego) {
#pragma unroll
            for (int s = 0; s < NI; ++s) {
                const int i = tid + 256 * s;
                const int pixel = i / WL_G, g = i % WL_G;
                const int pix = (h0 + pixel / WL_T) * W + w0 + pixel % WL_T;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(src + (size_t)pix * C + cb + (g + v * WL_G) * 8);
                    if (MODE == V2X_FUSE_MAX) {
                        float ev[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) ev[e] = 0.f;
                        fma8(ev, raw, 1.0f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[s][v][e] = (count == 1) ? ev[e] : fmaxf(acc[s][v][e], ev[e]);
                    } else {
                        fma8(acc[s][v], raw, wj);
                    }
                }
            }
            continue;
        }
        const float *T = trans + (((size_t)f * A + ego) * A + j) * 16;
        const float r00 = T[0], r01 = T[1], r10 = T[4], r11 = T[5];
        const float tx = (4.0f * T[3]) / 128.0f;
        const float ty = -((4.0f * T[7]) / 128.0f);
        const Bilin b0 = bilin_setup(((float)(2 * w0 + 1) / (float)W - 1.0f) + tx, ((float)(2 * h0 + 1) / (float)H - 1.0f) + ty, W, H);
        const int qx_lo = b0.x0, qy_lo = b0.y0;
        // the 81 rotate set-ups of this neighbour, once (rot_sample_cv's coordinate arithmetic, value for value).  The table was last read in the
        // previous neighbour's phase 1, which every thread left before the barrier that closed it.
        if (tid < WL_R * WL_R) {
            const int qx = qx_lo + tid % WL_R, qy = qy_lo + tid / WL_R;
            int2 eo = make_int2(-1, 0);
            float4 ew = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)qx < (unsigned)W && (unsigned)qy < (unsigned)H) {   // else never read: the translate tap is zero padding
                const float xq = (float)(2 * qx + 1) / (float)W - 1.0f;
                const float yq = (float)(2 * qy + 1) / (float)H - 1.0f;
                const Bilin b = bilin_setup(__fmaf_rn(r00, xq, __fmul_rn(r01, yq)), __fmaf_rn(r10, xq, __fmul_rn(r11, yq)), W, H);
                const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
                const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
                if ((xl || xr) && (yt || yb)) {
                    const int cx0 = min(max(b.x0, 0), W - 1), cx1 = min(max(b.x0 + 1, 0), W - 1);
                    const int cy0 = min(max(b.y0, 0), H - 1), cy1 = min(max(b.y0 + 1, 0), H - 1);
                    eo = make_int2(cy0 * W + cx0, (cx1 - cx0) | ((cy1 - cy0) << 1));
                    ew = make_float4((yt && xl) ? b.nw : 0.f, (yt && xr) ? b.ne : 0.f, (yb && xl) ? b.sw : 0.f, (yb && xr) ? b.se : 0.f);
                } else {
                    eo.x = -2;     // inside the map, whole footprint outside: the rotated sample is exactly zero
                }
            }
            roff[tid] = eo;
            rwt[tid] = ew;
        }
        __syncthreads();   // table written; the previous neighbour's phase 2 is done with the window
        for (int i = tid; i < WL_R * WL_R * WL_G; i += 256) {
            const int pos = i / WL_G, g = i % WL_G;
            const int2 eo = roff[pos];
            if (eo.x == -1) continue;
            float r[2][8];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int q = 0; q < 8; ++q) r[v][q] = 0.f;
            if (eo.x >= 0) {
                const float4 ew = rwt[pos];
                const uint16_t *p00 = src + (size_t)eo.x * C + cb + g * 8;
                const uint16_t *p10 = p00 + (size_t)((eo.y >> 1) * W) * C;
                const int dxc = (eo.y & 1) * C;
                uint4 t[4][2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    t[0][v] = *reinterpret_cast<const uint4 *>(p00 + v * (WL_G * 8));
                    t[1][v] = *reinterpret_cast<const uint4 *>(p00 + dxc + v * (WL_G * 8));
                    t[2][v] = *reinterpret_cast<const uint4 *>(p10 + v * (WL_G * 8));
                    t[3][v] = *reinterpret_cast<const uint4 *>(p10 + dxc + v * (WL_G * 8));
                }
                const float wt[4] = {ew.x, ew.y, ew.z, ew.w};
                // (per channel vector the taps are added in the order nw, ne, sw, se, as rot_sample_cv does: the vectors are independent sums)
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int k = 0; k < 4; ++k) fma8(r[v], t[k][v], wt[k]);
            }
            float4 *dst = win + pos * WL_POS4 + g;
            dst[0 * WL_G] = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]);
            dst[1 * WL_G] = make_float4(r[0][4], r[0][5], r[0][6], r[0][7]);
            dst[2 * WL_G] = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]);
            dst[3 * WL_G] = make_float4(r[1][4], r[1][5], r[1][6], r[1][7]);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            const int i = tid + 256 * s;
            const int pixel = i / WL_G, g = i % WL_G;
            const Bilin b = bilin_setup(base_xy[pixel % WL_T] + tx, base_xy[WL_T + pixel / WL_T] + ty, W, H);
            float vv[2][8], r[2][8];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[v][e] = 0.f;
            const bool xl = (unsigned)b.x0 < (unsigned)W, xr = (unsigned)(b.x0 + 1) < (unsigned)W;
            const bool yt = (unsigned)b.y0 < (unsigned)H, yb = (unsigned)(b.y0 + 1) < (unsigned)H;
#define V2X_TAP(COND, QX, QY, WT)                                                                                 \
    if (COND) {                                                                                                   \
        const int dx = (QX)-qx_lo, dy = (QY)-qy_lo;                                                               \
        if ((unsigned)dx < (unsigned)WL_R && (unsigned)dy < (unsigned)WL_R) {                                     \
            const float4 *p = win + (dy * WL_R + dx) * WL_POS4 + g;                                               \
            const float4 a0 = p[0], a1 = p[WL_G], a2 = p[2 * WL_G], a3 = p[3 * WL_G];                             \
            r[0][0] = a0.x; r[0][1] = a0.y; r[0][2] = a0.z; r[0][3] = a0.w;                                       \
            r[0][4] = a1.x; r[0][5] = a1.y; r[0][6] = a1.z; r[0][7] = a1.w;                                       \
            r[1][0] = a2.x; r[1][1] = a2.y; r[1][2] = a2.z; r[1][3] = a2.w;                                       \
            r[1][4] = a3.x; r[1][5] = a3.y; r[1][6] = a3.z; r[1][7] = a3.w;                                       \
        } else {                                                                                                  \
            rot_sample_cv<2>(src, H, W, C, cb + g * 8, QX, QY, r00, r01, r10, r11, r, WL_G * 8);                  \
        }                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < 2; ++v)                                                             \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) vv[v][e] = __fmaf_rn(r[v][e], WT, vv[v][e]);                               \
    }
            V2X_TAP(yt && xl, b.x0, b.y0, b.nw)
            V2X_TAP(yt && xr, b.x0 + 1, b.y0, b.ne)
            V2X_TAP(yb && xl, b.x0, b.y0 + 1, b.sw)
            V2X_TAP(yb && xr, b.x0 + 1, b.y0 + 1, b.se)
#undef V2X_TAP
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (MODE == V2X_FUSE_MAX) acc[s][v][e] = (count == 1) ? vv[v][e] : fmaxf(acc[s][v][e], vv[v][e]);
                    else acc[s][v][e] = __fmaf_rn(vv[v][e], wj, acc[s][v][e]);
                }
        }
    }
#pragma unroll
    for (int s = 0; s < NI; ++s) {
        const int i = tid + 256 * s;
        const int pixel = i / WL_G, g = i % WL_G;
        const int pix = (h0 + pixel / WL_T) * W + w0 + pixel % WL_T;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (MODE == V2X_FUSE_MEAN && count > 0) {
                const float d = (float)count;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[s][v][e] = acc[s][v][e] / d;
            }
            uint4 o;
            o.x = pack_bf16x2(acc[s][v][0], acc[s][v][1]);
            o.y = pack_bf16x2(acc[s][v][2], acc[s][v][3]);
            o.z = pack_bf16x2(acc[s][v][4], acc[s][v][5]);
            o.w = pack_bf16x2(acc[s][v][6], acc[s][v][7]);
            *reinterpret_cast<uint4 *>(out + (size_t)m * map_elems + (size_t)pix * C + cb + (g + v * WL_G) * 8) = o;
            __builtin_amdgcn_sched_barrier(0);   // one vector at a time: eight IEEE divisions' temporaries on top of 32 live accumulators spilled otherwise
        }
    }
}

static int warp_fuse_impl(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans,
                          const int32_t *items, int n_out, const float *coef, int mode, uint16_t *out,
                          const int32_t *order, int order_stride, int order_len, v2x_stream_t stream) {
    V2X_REQUIRE(feat && trans && items && coef && out, "v2x_warp_fuse: null pointer");
    V2X_REQUIRE(A > 0 && A <= 32 && Bt > 0 && H > 0 && W > 0, "v2x_warp_fuse: bad dims");
    V2X_REQUIRE(C > 0 && C % 8 == 0, "v2x_warp_fuse: C=%d must be a multiple of 8", C);
    V2X_REQUIRE(mode == V2X_FUSE_WSUM || mode == V2X_FUSE_MEAN || mode == V2X_FUSE_MAX, "v2x_warp_fuse: bad mode");
    V2X_REQUIRE(n_out >= 0 && n_out <= 65535, "v2x_warp_fuse: n_out out of range");
    if (n_out == 0) return V2X_OK;
    if (v2x_tune(V2X_TUNE_WARP_LDS) >= 2 && H % WL_T == 0 && W % WL_T == 0 && C % WL_CW == 0 && (long long)H * W < (1ll << 30)) {
        // the shared-set-up form (default); WARP_LDS = 1 keeps the per-item set-ups (A/B runs, bitwise test), 0 the direct form
        dim3 grid((H / WL_T) * (W / WL_T), n_out, C / WL_CW);
        const int32_t *ord = nullptr;
        int ostride = 0, olen = 0;
        if (order && order_stride > 0 && order_len > 0) {
            const int slots = (order_len + order_stride - 1) / order_stride;
            const int per_slot = order_stride * (H / WL_T) * (W / WL_T) * (C / WL_CW);
            const long long blocks = (long long)((slots + 7) / 8) * per_slot * 8;
            V2X_REQUIRE(blocks < (1ll << 31), "v2x_warp_fuse: grid too large");
            grid = dim3((unsigned)blocks);
            ord = order;
            ostride = order_stride;
            olen = order_len;
        }
        hipStream_t st = (hipStream_t)stream;
        if (mode == V2X_FUSE_MEAN) hipLaunchKernelGGL(warp_fuse_lds2_kernel<V2X_FUSE_MEAN>, grid, dim3(256), 0, st, feat, A, Bt, H, W, C, trans, items, coef, out, ord, ostride, olen);
        else if (mode == V2X_FUSE_MAX) hipLaunchKernelGGL(warp_fuse_lds2_kernel<V2X_FUSE_MAX>, grid, dim3(256), 0, st, feat, A, Bt, H, W, C, trans, items, coef, out, ord, ostride, olen);
        else hipLaunchKernelGGL(warp_fuse_lds2_kernel<V2X_FUSE_WSUM>, grid, dim3(256), 0, st, feat, A, Bt, H, W, C, trans, items, coef, out, ord, ostride, olen);
        V2X_CHECK_LAUNCH("warp_fuse_lds2_kernel");
        return V2X_OK;
    }
    if (v2x_tune(V2X_TUNE_WARP_LDS) != 0 && H % WL_T == 0 && W % WL_T == 0 && C % WL_CW == 0) {
        if (order && order_stride > 0 && order_len > 0) {
            // frame slots of order_stride output maps each (padded with -1); XCD x = linear workgroup id % 8 takes the slots x, x + 8, ...
            const int slots = (order_len + order_stride - 1) / order_stride;
            const int per_slot = order_stride * (H / WL_T) * (W / WL_T) * (C / WL_CW);
            const long long blocks = (long long)((slots + 7) / 8) * per_slot * 8;
            V2X_REQUIRE(blocks < (1ll << 31), "v2x_warp_fuse: grid too large");
            hipLaunchKernelGGL(warp_fuse_lds_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feat, A, Bt, H, W, C, trans, items, coef,
                               mode, out, order, order_stride, order_len);
        } else {
            hipLaunchKernelGGL(warp_fuse_lds_kernel, dim3((H / WL_T) * (W / WL_T), n_out, C / WL_CW), dim3(256), 0,
                               (hipStream_t)stream, feat, A, Bt, H, W, C, trans, items, coef, mode, out, nullptr, 0, 0);
        }
        V2X_CHECK_LAUNCH("warp_fuse_lds_kernel");
        return V2X_OK;
    }
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

extern "C" int v2x_warp_fuse(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans,
                             const int32_t *items, int n_out, const float *coef, int mode, uint16_t *out,
                             v2x_stream_t stream) {
    return warp_fuse_impl(feat, A, Bt, H, W, C, trans, items, n_out, coef, mode, out, nullptr, 0, 0, stream);
}

extern "C" int v2x_warp_fuse_ordered(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans,
                                     const int32_t *items, int n_out, const float *coef, int mode, uint16_t *out,
                                     const int32_t *order, int order_stride, int order_len, v2x_stream_t stream) {
    V2X_REQUIRE(order && order_stride > 0 && order_len >= n_out, "v2x_warp_fuse_ordered: bad order table");
    return warp_fuse_impl(feat, A, Bt, H, W, C, trans, items, n_out, coef, mode, out, order, order_stride, order_len, stream);
}
