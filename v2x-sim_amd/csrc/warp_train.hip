// Affine bilinear resampling of fp32 NCHW maps and its EXACT transpose -- the cross-agent warp of the training graph (row f-3).
// Upstream: coperception/models/det/base/IntermediateModelBase.py::feature_transformation = F.affine_grid + F.grid_sample
// (bilinear, zeros padding, align_corners=False), applied twice (rotation about the map centre, then translation); code absent from
// /root/reference, see include/v2x_amd.h.  In the PyTorch graph the BACKWARD of grid_sample is a scatter with fp32 atomics
// (grid_sampler_2d_backward_kernel: 0.9 ms per call on 40 x 256 x 32 x 32 maps = 17 % of a V2VNet training step, and not
// bit-reproducible).  Here the data gradient is a GATHER: for an affine map the output pixels q whose sample point falls within one pixel
// of an input pixel p lie in a small parallelogram around M^-1 (p - t) (<= 3 x 3 for a rotation, 2 x 2 for a translation); every thread
// owns one input pixel, walks those candidates in a fixed order, recomputes each candidate's sample position with the SAME device function
// as the forward kernel and adds w(q, p) * dout[q] -- the exact transpose of the forward operator, deterministic, no atomics.
// A singular or strongly shrinking map (|det M| small: many output pixels per input pixel) widens the candidate box up to the whole map --
// still exact, only slower.
#include "common.h"

namespace {
constexpr int WT_CCH = 16;   // channels per thread (the taps / candidates of a pixel are computed once per chunk)

struct WarpTrainArgs {
    const float *src;    // forward: input maps; backward: output gradient   [P][C][H][W]
    const float *theta;  // [P][2][3]
    float *dst;          // forward: output maps; backward: input gradient   [P][C][H][W]
    int P, C, H, W;
};

// sample position (input pixel units) of output pixel (j = column, i = row) under theta: F.affine_grid + the unnormalisation of
// F.grid_sample with align_corners = False
__device__ __forceinline__ void warp_sample_pos(const float th[6], int j, int i, int H, int W, float &ix, float &iy) {
    const float xn = (2.0f * (float)j + 1.0f) / (float)W - 1.0f;
    const float yn = (2.0f * (float)i + 1.0f) / (float)H - 1.0f;
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
    iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
}

__global__ __launch_bounds__(256) void warp_affine_fwd_kernel(const WarpTrainArgs a) {
    const int p = blockIdx.z, c0 = blockIdx.y * WT_CCH;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;
    if (pix >= HW) return;
    float th[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) th[k] = a.theta[p * 6 + k];
    const int i = pix / a.W, j = pix - i * a.W;
    float ix, iy;
    warp_sample_pos(th, j, i, a.H, a.W, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    // out-of-range sample positions (also inf / nan) contribute nothing: compare in float before converting
    const bool any = fx >= -1.0f && fx < (float)a.W && fy >= -1.0f && fy < (float)a.H;
    const int x0 = any ? (int)fx : 0, y0 = any ? (int)fy : 0;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool vx0 = any && x0 >= 0, vx1 = any && x0 + 1 < a.W, vy0 = any && y0 >= 0, vy1 = any && y0 + 1 < a.H;
    const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    const float w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    const int o00 = (vx0 && vy0) ? y0 * a.W + x0 : 0, o01 = (vx1 && vy0) ? y0 * a.W + x0 + 1 : 0;
    const int o10 = (vx0 && vy1) ? (y0 + 1) * a.W + x0 : 0, o11 = (vx1 && vy1) ? (y0 + 1) * a.W + x0 + 1 : 0;
    const int nc = min(WT_CCH, a.C - c0);
    for (int c = 0; c < nc; ++c) {
        const float *s = a.src + ((size_t)p * a.C + c0 + c) * HW;
        // the same order of additions as at::native's grid_sampler_2d kernel: nw, ne, sw, se
        float v = s[o00] * w00;
        v += s[o01] * w01;
        v += s[o10] * w10;
        v += s[o11] * w11;
        a.dst[((size_t)p * a.C + c0 + c) * HW + pix] = v;
    }
}

__global__ __launch_bounds__(256) void warp_affine_bwd_kernel(const WarpTrainArgs a) {
    const int p = blockIdx.z, c0 = blockIdx.y * WT_CCH;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = a.H * a.W;
    if (pix >= HW) return;
    float th[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) th[k] = a.theta[p * 6 + k];
    const int y = pix / a.W, x = pix - y * a.W;
    // sample position as an affine function of the output pixel: (ix, iy) = M (j, i) + t
    const float fw = (float)a.W, fh = (float)a.H;
    const float m00 = th[0], m01 = th[1] * fw / fh, m10 = th[3] * fh / fw, m11 = th[4];
    float t0, t1;
    warp_sample_pos(th, 0, 0, a.H, a.W, t0, t1);
    const float det = m00 * m11 - m01 * m10;
    int jlo = 0, jhi = a.W - 1, ilo = 0, ihi = a.H - 1;
    if (fabsf(det) > 1e-6f && isfinite(det) && isfinite(t0) && isfinite(t1)) {
        const float r00 = m11 / det, r01 = -m01 / det, r10 = -m10 / det, r11 = m00 / det;
        const float qj = r00 * ((float)x - t0) + r01 * ((float)y - t1), qi = r10 * ((float)x - t0) + r11 * ((float)y - t1);
        // |ix - x| < 1 and |iy - y| < 1  <=>  q in q0 + M^-1 (-1, 1)^2; the slack covers the rounding of the two evaluations
        const float ej = fabsf(r00) + fabsf(r01) + 1e-2f, ei = fabsf(r10) + fabsf(r11) + 1e-2f;
        const float a0 = ceilf(qj - ej), a1 = floorf(qj + ej), b0 = ceilf(qi - ei), b1 = floorf(qi + ei);
        jlo = (int)fmaxf(a0, 0.f);
        jhi = (int)fminf(a1, fw - 1.f);
        ilo = (int)fmaxf(b0, 0.f);
        ihi = (int)fminf(b1, fh - 1.f);
    }
    const int nc = min(WT_CCH, a.C - c0);
    float acc[WT_CCH];
#pragma unroll
    for (int c = 0; c < WT_CCH; ++c) acc[c] = 0.f;
    const float *s = a.src + ((size_t)p * a.C + c0) * HW;
    for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
            float ix, iy;
            warp_sample_pos(th, j, i, a.H, a.W, ix, iy);
            const float fx = floorf(ix), fy = floorf(iy);
            // the forward kernel's weights of this output pixel on input pixel (x, y)
            const float wx = ((float)x == fx) ? 1.0f - (ix - fx) : (((float)x == fx + 1.0f) ? ix - fx : 0.f);
            const float wy = ((float)y == fy) ? 1.0f - (iy - fy) : (((float)y == fy + 1.0f) ? iy - fy : 0.f);
            const float w = wx * wy;
            if (w == 0.f) continue;
            const int q = i * a.W + j;
#pragma unroll
            for (int c = 0; c < WT_CCH; ++c)
                if (c < nc) acc[c] += w * s[(size_t)c * HW + q];
        }
    for (int c = 0; c < nc; ++c) a.dst[((size_t)p * a.C + c0 + c) * HW + pix] = acc[c];
}

int warp_train_launch(bool bwd, const float *src, const float *theta, float *dst, int P, int C, int H, int W, hipStream_t s, const char *who) {
    V2X_REQUIRE(src && theta && dst, "%s: null pointer", who);
    V2X_REQUIRE(P >= 0 && C > 0 && H > 0 && W > 0 && (long long)H * W <= (1 << 24) && (long long)P * C * H * W < (1ll << 40), "%s: bad extent", who);
    if (P == 0) return V2X_OK;
    V2X_REQUIRE(P <= 65535 && (C + WT_CCH - 1) / WT_CCH <= 65535, "%s: more than 65535 maps or channel chunks per launch", who);
    WarpTrainArgs a{src, theta, dst, P, C, H, W};
    const dim3 grid((H * W + 255) / 256, (C + WT_CCH - 1) / WT_CCH, P);
    if (bwd) hipLaunchKernelGGL(warp_affine_bwd_kernel, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(warp_affine_fwd_kernel, grid, dim3(256), 0, s, a);
    V2X_CHECK_LAUNCH(who);
    return V2X_OK;
}
}  // namespace

extern "C" int v2x_warp_affine_f32(const float *in, const float *theta, int P, int C, int H, int W, float *out, v2x_stream_t stream) {
    return warp_train_launch(false, in, theta, out, P, C, H, W, (hipStream_t)stream, "v2x_warp_affine_f32");
}

extern "C" int v2x_warp_affine_bwd_f32(const float *dout, const float *theta, int P, int C, int H, int W, float *din, v2x_stream_t stream) {
    return warp_train_launch(true, dout, theta, din, P, C, H, W, (hipStream_t)stream, "v2x_warp_affine_bwd_f32");
}
