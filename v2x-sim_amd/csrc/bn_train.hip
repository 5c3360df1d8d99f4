// Batch-statistics BatchNorm + ReLU for TRAINING on bf16 NHWC maps.
//
// Upstream trains through nn.BatchNorm2d / nn.BatchNorm3d in train mode followed by F.relu (Backbone.py's
// `F.relu(self.bnK_J(self.convK_J(x)))`); on ROCm that is MIOpen's BN kernels over NCHW fp32.  Here a map is [M][C] bf16 with
// M = N*H*W pixels and the channel index fastest, the way every conv kernel of this library reads and writes it, so BN is four
// streaming passes -- all HBM-bound, 16-B accesses, no layout change:
//
//   forward   (1) bn_partial_kernel<STATS>   per-channel sum x, sum x^2           read x
//             (2) bn_finish_stats_kernel     mean, biased var -> 1/sqrt(var+eps); running statistics as nn.BatchNorm does
//             (3) bn_apply_kernel            y = relu(x * a + b) -> bf16          read x, write y     (a = gamma/std, b = beta - mean a)
//   backward  (1) bn_partial_kernel<GRADS>   sum g, sum g xhat,  g = dy * [y > 0] read x, dy          (y recomputed, never stored twice)
//             (2) bn_finish_grads_kernel     dbeta, dgamma
//             (3) bn_bwd_apply_kernel        dx = a (g - dbeta/M - xhat dgamma/M) read x, dy, write dx
//
// Determinism: a workgroup owns a contiguous run of pixels and writes ONE partial per channel; the finish kernels add the partials
// in a fixed order in fp64.  No atomics anywhere, so two runs give the same bits (MIOpen's BN backward does not promise that).
// var = E[x^2] - mean^2 is formed in fp64 from the fp32 partials (a thread adds at most M * C / (8 * 256 * n_blocks) values per channel).
#include "common.h"

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 2048;

struct BnArgs {
    const uint16_t *x;     // [M][C] bf16: the convolution's output
    const uint16_t *dy;    // [M][C] bf16 (backward)
    uint16_t *out;         // y (forward) or dx (backward)
    const float *gamma, *beta;
    float *mean, *invstd;  // [C] saved statistics
    float *running_mean, *running_var;   // may be null
    float *dgamma, *dbeta;
    float *partial;        // [n_blocks][2][C]
    long long M;
    int C, G;              // G = C / 8 channel groups
    int n_blocks;
    long long vec_per_block;   // 16-B vectors per workgroup (a multiple of BN_THREADS)
    float eps, momentum;
    int relu;
    int partial_t;         // layout of `partial`: 1 = [kind][channel][workgroup] (round 6), 0 = [workgroup][kind][channel]
};

__device__ __forceinline__ void unpack8(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

// MODE 0: sum x, sum x^2.   MODE 1: sum g, sum g * xhat with g = dy masked by the recomputed ReLU.
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_kernel(const BnArgs a) {
    __shared__ float red[BN_THREADS][17];
    const int t = threadIdx.x;
    const int g = t % a.G;                 // BN_THREADS % G == 0: a thread keeps its channel group over the whole run
    const long long total = a.M * a.G;
    const long long v0 = (long long)blockIdx.x * a.vec_per_block;
    long long v1 = v0 + a.vec_per_block;
    if (v1 > total) v1 = total;
    float s0[8], s1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s0[i] = s1[i] = 0.f;
    float mu[8], is[8], ga[8], be[8];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = g * 8 + i;
            mu[i] = a.mean[c];
            is[i] = a.invstd[c];
            ga[i] = a.gamma[c];
            be[i] = a.beta[c];
        }
    }
    const uint4 *xv = reinterpret_cast<const uint4 *>(a.x);
    const uint4 *dv = reinterpret_cast<const uint4 *>(a.dy);
    for (long long v = v0 + t; v < v1; v += BN_THREADS) {
        float x[8];
        unpack8(xv[v], x);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s0[i] += x[i];
                s1[i] = fmaf(x[i], x[i], s1[i]);
            }
        } else {
            float d[8];
            unpack8(dv[v], d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (x[i] - mu[i]) * is[i];
                const float y = fmaf(xh, ga[i], be[i]);
                const float gr = (a.relu && !(y > 0.f)) ? 0.f : d[i];
                s0[i] += gr;
                s1[i] = fmaf(gr, xh, s1[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[t][i] = s0[i];
        red[t][8 + i] = s1[i];
    }
    __syncthreads();
    // thread (g, i) for g < G, i < 16 adds the BN_THREADS / G rows of its channel group in row order
    for (int o = t; o < a.G * 16; o += BN_THREADS) {
        const int gg = o >> 4, i = o & 15;
        float s = 0.f;
        for (int r = gg; r < BN_THREADS; r += a.G) s += red[r][i];
        // [kind][channel][workgroup]: the finish kernel (one workgroup per channel) then reads its channel's partials as contiguous floats -- in the
        // [workgroup][kind][channel] layout of rounds 3-5 every one of its loads touched its own 64-byte line, which 16 workgroups re-read (134 MB of L2
        // traffic per finish launch at 512 channels x 2048 partials)
        if (a.partial_t) a.partial[((size_t)(i >> 3) * a.C + gg * 8 + (i & 7)) * a.n_blocks + blockIdx.x] = s;
        else a.partial[((size_t)blockIdx.x * 2 + (i >> 3)) * a.C + gg * 8 + (i & 7)] = s;
    }
}

// One workgroup per channel: thread t adds partials t, t + 256, ... in fp64 (coalesced: the channel's partials are contiguous), then a fixed-shape
// reduction -- xor shuffles inside a wave, the four wave sums added in wave order by every thread: the same association every run, one barrier
// (rounds 3-5: an eight-level LDS tree, eight barriers).  A single thread walking all <= 2048 partials was 0.3 ms of dependent loads per launch.
__device__ __forceinline__ void bn_sum_partials(const BnArgs &a, int c, double &s_out, double &q_out) {
    __shared__ double rs[BN_THREADS / 64], rq[BN_THREADS / 64];
    const int t = threadIdx.x;
    double s = 0.0, q = 0.0;
    // all of the thread's (at most 8) loads first, then the additions in the same order as a plain loop: one memory latency instead of eight
    constexpr int NP = BN_MAX_BLOCKS / BN_THREADS;
    float vs[NP], vq[NP];
    const float *ps = a.partial + (size_t)c * a.n_blocks, *pq = a.partial + ((size_t)a.C + c) * a.n_blocks;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int b = t + k * BN_THREADS;
        const bool ok = b < a.n_blocks;
        vs[k] = ok ? (a.partial_t ? ps[b] : a.partial[((size_t)b * 2) * a.C + c]) : 0.f;
        vq[k] = ok ? (a.partial_t ? pq[b] : a.partial[((size_t)b * 2 + 1) * a.C + c]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        s += (double)vs[k];
        q += (double)vq[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
    }
    if ((t & 63) == 0) {
        rs[t >> 6] = s;
        rq[t >> 6] = q;
    }
    __syncthreads();
    s = rs[0];
    q = rq[0];
#pragma unroll
    for (int w = 1; w < BN_THREADS / 64; ++w) {
        s += rs[w];
        q += rq[w];
    }
    s_out = s;
    q_out = q;
}

__global__ __launch_bounds__(BN_THREADS) void bn_finish_stats_kernel(const BnArgs a) {
    const int c = blockIdx.x;
    double s, q;
    bn_sum_partials(a, c, s, q);
    if (threadIdx.x != 0) return;
    const double m = s / (double)a.M;
    double var = q / (double)a.M - m * m;
    if (var < 0.0) var = 0.0;
    a.mean[c] = (float)m;
    a.invstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
    if (a.running_var) {
        const double unbiased = a.M > 1 ? var * (double)a.M / (double)(a.M - 1) : var;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_finish_grads_kernel(const BnArgs a) {
    const int c = blockIdx.x;
    double s, q;
    bn_sum_partials(a, c, s, q);
    if (threadIdx.x != 0) return;
    a.dbeta[c] = (float)s;
    a.dgamma[c] = (float)q;
}

__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const BnArgs a) {
    const int t = threadIdx.x;
    const int g = t % a.G;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = g * 8 + i;
        sc[i] = a.gamma[c] * a.invstd[c];
        sh[i] = fmaf(-a.mean[c], sc[i], a.beta[c]);
    }
    const long long total = a.M * a.G;
    const uint4 *xv = reinterpret_cast<const uint4 *>(a.x);
    uint4 *yv = reinterpret_cast<uint4 *>(a.out);
    for (long long v = (long long)blockIdx.x * BN_THREADS + t; v < total; v += (long long)gridDim.x * BN_THREADS) {
        float x[8];
        unpack8(xv[v], x);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] = fmaf(x[i], sc[i], sh[i]);
            if (a.relu) x[i] = fmaxf(x[i], 0.f);
        }
        yv[v] = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
    }
}

// SUM: also the per-channel sum of the dx values AS STORED (bf16-rounded) -- the bias gradient of the convolution that produced x
// (db = sum over pixels of dx; upstream: autograd's reduction in nn.Conv2d.backward), one partial per workgroup and channel in
// a.partial[blockIdx.x][C] (the finish kernel of the channel sum adds them in workgroup order): saves the separate read of dx
template <bool SUM>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const BnArgs a) {
    __shared__ float red[SUM ? BN_THREADS : 1][9];
    const int t = threadIdx.x;
    const int g = t % a.G;
    float mu[8], is[8], ga[8], be[8], k0[8], k1[8], acc[8];
    const float inv_m = 1.f / (float)a.M;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = g * 8 + i;
        mu[i] = a.mean[c];
        is[i] = a.invstd[c];
        ga[i] = a.gamma[c];
        be[i] = a.beta[c];
        k0[i] = a.dbeta[c] * inv_m;
        k1[i] = a.dgamma[c] * inv_m;
        acc[i] = 0.f;
    }
    const long long total = a.M * a.G;
    const uint4 *xv = reinterpret_cast<const uint4 *>(a.x);
    const uint4 *dv = reinterpret_cast<const uint4 *>(a.dy);
    uint4 *ov = reinterpret_cast<uint4 *>(a.out);
    for (long long v = (long long)blockIdx.x * BN_THREADS + t; v < total; v += (long long)gridDim.x * BN_THREADS) {
        float x[8], d[8];
        unpack8(xv[v], x);
        unpack8(dv[v], d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (x[i] - mu[i]) * is[i];
            const float y = fmaf(xh, ga[i], be[i]);
            const float gr = (a.relu && !(y > 0.f)) ? 0.f : d[i];
            x[i] = ga[i] * is[i] * (gr - k0[i] - xh * k1[i]);
        }
        const uint4 o = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
        ov[v] = o;
        if (SUM) {
            float r[8];
            unpack8(o, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += r[i];
        }
    }
    if (SUM) {
        // (the grid stride is a multiple of BN_THREADS and BN_THREADS % G == 0: a thread keeps its channel group.)  Lanes l, l + G, l + 2G, ...
        // of a wave share a group: butterfly over those offsets, then the (at most four) waves' results of a group through LDS, in wave order.
        for (int off = a.G; off < 64; off <<= 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off);
        }
        const int lane = t & 63, gw = a.G < 64 ? a.G : 64;      // lanes 0 .. gw-1 of a wave hold its sums
        if (lane < gw) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[t][i] = acc[i];
        }
        __syncthreads();
        if (t < a.G) {
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < BN_THREADS / 64; ++w) {
                const int l = ((t - w * 64) % a.G + a.G) % a.G;                // the lane of wave w whose group is t
                if (l < gw) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum[i] += red[w * 64 + l][i];
                }
            }
            // channel-major partials [C][gridDim.x]: the finish kernel reads a channel's partials as one contiguous run
#pragma unroll
            for (int i = 0; i < 8; ++i) a.partial[(size_t)(t * 8 + i) * gridDim.x + blockIdx.x] = sum[i];
        }
    }
}

// one wave per channel over the channel-major partials of bn_bwd_apply_kernel<true>: lane l adds entries l, l + 64, ... (fp64), fixed butterfly
__global__ __launch_bounds__(256) void bn_dxsum_finish_kernel(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    const float *p = part + (size_t)c * nblk;
    double s = 0.0;
    for (int b0 = lane; b0 < nblk; b0 += 64 * 8) {        // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (b0 + 64 * k < nblk) ? p[b0 + 64 * k] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[c] = (float)s;
}

static int bn_plan(long long M, int C, BnArgs &a) {
    a.M = M;
    a.C = C;
    a.G = C / 8;
    const long long total = M * a.G;
    // <= 4096 pixels' worth per partial would need too many blocks for big maps; bound the block count instead and keep the
    // per-thread run short enough for fp32 (a thread adds total / (n_blocks * 256) values per channel)
    long long per = (total + BN_MAX_BLOCKS - 1) / BN_MAX_BLOCKS;
    per = (per + BN_THREADS - 1) / BN_THREADS * BN_THREADS;
    if (per < BN_THREADS) per = BN_THREADS;
    a.vec_per_block = per;
    a.n_blocks = (int)((total + per - 1) / per);
    a.partial_t = v2x_tune(V2X_TUNE_BN_PARTIAL_T) != 0;
    return a.n_blocks;
}

static bool bn_shape_ok(long long M, int C) { return M > 0 && C >= 8 && C % 8 == 0 && BN_THREADS % (C / 8) == 0; }

extern "C" long long v2x_bn_train_workspace_size(long long M, int C) {
    if (!bn_shape_ok(M, C)) return 0;
    BnArgs a;
    return (long long)bn_plan(M, C, a) * 2 * C * (long long)sizeof(float);
}

extern "C" int v2x_bn_train_forward(const uint16_t *x, long long M, int C, const float *gamma, const float *beta, float eps,
                                    float momentum, float *running_mean, float *running_var, int relu, uint16_t *y,
                                    float *save_mean, float *save_invstd, float *workspace, v2x_stream_t stream) {
    V2X_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "v2x_bn_train_forward: null pointer");
    V2X_REQUIRE(bn_shape_ok(M, C), "v2x_bn_train_forward: needs M > 0 and C in {8, 16, 32, ..., 2048} (C / 8 divides 256), got M=%lld C=%d", M, C);
    V2X_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "v2x_bn_train_forward: running_mean and running_var go together");
    BnArgs a = {};
    bn_plan(M, C, a);
    a.x = x;
    a.out = y;
    a.gamma = gamma;
    a.beta = beta;
    a.mean = save_mean;
    a.invstd = save_invstd;
    a.running_mean = running_mean;
    a.running_var = running_var;
    a.partial = workspace;
    a.eps = eps;
    a.momentum = momentum;
    a.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(a.n_blocks), dim3(BN_THREADS), 0, s, a);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(C), dim3(BN_THREADS), 0, s, a);
    const long long total = M * a.G;
    long long blocks = (total + BN_THREADS - 1) / BN_THREADS;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, a);
    V2X_CHECK_LAUNCH("bn_train_forward");
    return V2X_OK;
}

constexpr int BN_DXSUM_MAX_BLOCKS = 2048;

static int bn_train_backward_impl(const uint16_t *x, const uint16_t *dy, long long M, int C, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, int relu, uint16_t *dx, float *dgamma,
                                  float *dbeta, float *workspace, float *dx_sum, float *sum_workspace, v2x_stream_t stream) {
    V2X_REQUIRE(x && dy && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "v2x_bn_train_backward: null pointer");
    V2X_REQUIRE(bn_shape_ok(M, C), "v2x_bn_train_backward: needs M > 0 and C in {8, 16, 32, ..., 2048} (C / 8 divides 256), got M=%lld C=%d", M, C);
    BnArgs a = {};
    bn_plan(M, C, a);
    a.x = x;
    a.dy = dy;
    a.out = dx;
    a.gamma = gamma;
    a.beta = beta;
    a.mean = const_cast<float *>(save_mean);
    a.invstd = const_cast<float *>(save_invstd);
    a.dgamma = dgamma;
    a.dbeta = dbeta;
    a.partial = workspace;
    a.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(a.n_blocks), dim3(BN_THREADS), 0, s, a);
    hipLaunchKernelGGL(bn_finish_grads_kernel, dim3(C), dim3(BN_THREADS), 0, s, a);
    const long long total = M * a.G;
    long long blocks = (total + BN_THREADS - 1) / BN_THREADS;
    if (dx_sum) {
        if (blocks > BN_DXSUM_MAX_BLOCKS) blocks = BN_DXSUM_MAX_BLOCKS;
        a.partial = sum_workspace;                 // the statistics' partials (workspace) are consumed by the finish kernel above
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, a);
        hipLaunchKernelGGL(bn_dxsum_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, sum_workspace, (int)blocks, C, dx_sum);
    } else {
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, a);
    }
    V2X_CHECK_LAUNCH("bn_train_backward");
    return V2X_OK;
}

extern "C" int v2x_bn_train_backward(const uint16_t *x, const uint16_t *dy, long long M, int C, const float *gamma, const float *beta,
                                     const float *save_mean, const float *save_invstd, int relu, uint16_t *dx, float *dgamma,
                                     float *dbeta, float *workspace, v2x_stream_t stream) {
    return bn_train_backward_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, workspace, nullptr, nullptr, stream);
}

extern "C" long long v2x_bn_dxsum_workspace_size(long long M, int C) {
    if (!bn_shape_ok(M, C)) return 0;
    long long blocks = (M * (C / 8) + BN_THREADS - 1) / BN_THREADS;
    if (blocks > BN_DXSUM_MAX_BLOCKS) blocks = BN_DXSUM_MAX_BLOCKS;
    return blocks * C * (long long)sizeof(float);
}

extern "C" int v2x_bn_train_backward_dxsum(const uint16_t *x, const uint16_t *dy, long long M, int C, const float *gamma, const float *beta,
                                           const float *save_mean, const float *save_invstd, int relu, uint16_t *dx, float *dgamma,
                                           float *dbeta, float *dx_sum, float *workspace, float *sum_workspace, v2x_stream_t stream) {
    V2X_REQUIRE(dx_sum && sum_workspace, "v2x_bn_train_backward_dxsum: null pointer");
    return bn_train_backward_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, workspace, dx_sum, sum_workspace, stream);
}


// ---- per-channel sum of a bf16 [M][C] map (row f-3: the bias gradient of a convolution, db[c] = sum over pixels of dy[..][c]) --------
// torch's `dy.float().sum((0, 1, 2))` was a cast kernel + a reduction per layer (24 + 22 launches, 0.76 ms of a 10-map FaFNet step).  Two
// launches here: per-workgroup partials (a thread keeps its 8 channels over its rows; the workgroup's threads of one channel group are added
// in thread order through LDS), then one thread per channel adds the partials in workgroup order in fp64 -- fixed order, bit-reproducible.
constexpr int CS_MAX_BLOCKS = 512;
static int cs_blocks(long long M, int C) {
    const int rows_per_pass = 256 / (C / 8);
    long long b = (M + (long long)rows_per_pass * 16 - 1) / ((long long)rows_per_pass * 16);   // >= 16 passes per workgroup
    if (b < 1) b = 1;
    return (int)(b < CS_MAX_BLOCKS ? b : CS_MAX_BLOCKS);
}

__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const uint16_t *__restrict__ x, long long M, int C, float *__restrict__ part) {
    __shared__ float red[256][8];
    const int groups = C / 8, rpp = 256 / groups;
    const int cg = threadIdx.x % groups, r0 = threadIdx.x / groups;
    const long long per = (M + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long r = lo + r0; r < hi; r += rpp) {
        const uint4 v = *reinterpret_cast<const uint4 *>(x + r * C + cg * 8);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32((uint16_t)(wds[j >> 1] >> ((j & 1) * 16)));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (r0 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
            for (int k = 0; k < rpp; ++k) s += red[k * groups + cg][j];
            part[(size_t)blockIdx.x * C + cg * 8 + j] = s;
        }
    }
}

// one wave per channel: lane l adds the partials l, l + 64, ... in order (fp64), then a fixed butterfly over the 64 lanes -- a fixed tree,
// bit-reproducible (one THREAD per channel walking up to 1 024 partials was 72 us per call: dependent loads)
__global__ __launch_bounds__(256) void channel_sum_finish_kernel(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int b0 = lane; b0 < nblk; b0 += 64 * 8) {        // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (b0 + 64 * k < nblk) ? part[(size_t)(b0 + 64 * k) * C + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[c] = (float)s;
}

// fp32 [M][C] -> bf16 [M][Cp] (zero-padded channels) + per-workgroup partial channel sums of the fp32 values: the logit gradients of a 1x1 head on their way
// into the data- / weight-gradient kernels (include/v2x_amd.h: v2x_cast_pad_chsum_f32).  A thread owns one 8-channel group of the padded row (one 16-B store).
__global__ __launch_bounds__(256) void cast_pad_chsum_kernel(const float *__restrict__ x, long long M, int C, int Cp, uint16_t *__restrict__ out, float *__restrict__ part) {
    __shared__ float red[256][8];
    const int groups = Cp / 8, rpp = 256 / groups;
    const int cg = threadIdx.x % groups, r0 = threadIdx.x / groups;
    const long long per = (M + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
    const bool q0 = cg * 8 + 4 <= C, q1 = cg * 8 + 8 <= C;      // which of the group's two float4 exist (C % 4 == 0)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long r = lo + r0; r < hi; r += rpp) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (q0) a = *reinterpret_cast<const float4 *>(x + r * C + cg * 8);
        if (q1) b = *reinterpret_cast<const float4 *>(x + r * C + cg * 8 + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        uint4 v;
        v.x = pack_bf16x2(a.x, a.y);
        v.y = pack_bf16x2(a.z, a.w);
        v.z = pack_bf16x2(b.x, b.y);
        v.w = pack_bf16x2(b.z, b.w);
        *reinterpret_cast<uint4 *>(out + r * Cp + cg * 8) = v;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (r0 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
            for (int k = 0; k < rpp; ++k) s += red[k * groups + cg][j];
            part[(size_t)blockIdx.x * Cp + cg * 8 + j] = s;
        }
    }
}

extern "C" long long v2x_channel_sum_workspace_size(long long M, int C) {
    if (!bn_shape_ok(M, C)) return 0;
    return (long long)cs_blocks(M, C) * C * (long long)sizeof(float);
}

extern "C" int v2x_channel_sum_bf16(const uint16_t *x, long long M, int C, float *out, float *workspace, v2x_stream_t stream) {
    V2X_REQUIRE(x && out && workspace, "v2x_channel_sum_bf16: null pointer");
    V2X_REQUIRE(bn_shape_ok(M, C), "v2x_channel_sum_bf16: needs M > 0 and C in {8, 16, 32, ..., 2048} (C / 8 divides 256), got M=%lld C=%d", M, C);
    const int nblk = cs_blocks(M, C);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, M, C, workspace);
    hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, workspace, nblk, C, out);
    V2X_CHECK_LAUNCH("channel_sum kernels");
    return V2X_OK;
}

extern "C" long long v2x_cast_pad_chsum_workspace_size(long long M, int Cp) { return v2x_channel_sum_workspace_size(M, Cp); }

extern "C" int v2x_cast_pad_chsum_f32(const float *x, long long M, int C, int Cp, uint16_t *out, float *sums, float *workspace, v2x_stream_t stream) {
    V2X_REQUIRE(x && out && sums && workspace, "v2x_cast_pad_chsum_f32: null pointer");
    V2X_REQUIRE(bn_shape_ok(M, Cp) && C > 0 && C % 4 == 0 && C <= Cp, "v2x_cast_pad_chsum_f32: needs M > 0, C %% 4 == 0, C <= Cp, Cp in {8, 16, 32, ...} (Cp / 8 divides 256), got M=%lld C=%d Cp=%d", M, C, Cp);
    V2X_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "v2x_cast_pad_chsum_f32: x and out must be 16-byte aligned");
    const int nblk = cs_blocks(M, Cp);
    hipLaunchKernelGGL(cast_pad_chsum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, M, C, Cp, out, workspace);
    hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((Cp + 3) / 4), dim3(256), 0, (hipStream_t)stream, workspace, nblk, Cp, sums);
    V2X_CHECK_LAUNCH("cast_pad_chsum kernels");
    return V2X_OK;
}
