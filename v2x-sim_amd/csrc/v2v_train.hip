// V2VNet's message-passing round in the TRAINING graph on bf16 NHWC maps, forward and backward.
//
// Upstream: coperception/models/det/V2VNet.py::forward -- per ego agent: every neighbour's fusion-layer map warped into the ego frame
// (IntermediateModelBase.py::feature_transformation = two F.affine_grid + F.grid_sample passes: rotation about the map centre, then translation), the mean
// over the neighbours, cat([ego map, mean]) into convolutional_rnn.Conv2dGRU (hidden = None); code absent from /root/reference (include/v2x_amd.h), restated
// in v2x_sim_amd/train/graph.py::v2v_fuse.  Until round 6 that stage ran on the fp32 NCHW PyTorch graph around three kernels (warp, input convolution,
// gates): 89 PyTorch-op launches of a 10-map step (layout / precision copies both ways, index_select, mean, cat, sum, div, ...: 0.84 ms of 5.5,
// profiles/r06_train_op_census_before.txt).  Here the stage is four launches forward and four backward, no layout change:
//   v2v_message_kernel        cur / base bf16 [N][H][W][C] -> conv_in bf16 [M][H][W][2C] = [ego map | mean over the K neighbours of warp2(base[src])]
//   (the input convolution 2C -> 3C on the library's 3x3 kernels)
//   gru_gates_nhwc_kernel     gi bf16 [M][HW][3C], bias_hh -> h bf16 [M][HW][C]
//   gru_gates_nhwc_bwd_kernel dh -> dgi bf16 [M][HW][3C] + per-workgroup channel sums (-> d bias_ih, d bias_hh by sum_finish_kernel, fixed order)
//   v2v_message_bwd_kernel    d conv_in -> d base (+ the ego half: d cur) bf16 [N][H][W][C]
// The two resampling passes are evaluated in ONE kernel without the intermediate map: an output pixel's translation pass reads four pixels of the rotated map,
// each of which is recomputed from its four source pixels with the arithmetic (and the order of additions) of warp_train.hip's forward kernel; the backward is
// the exact transpose, a gather over the candidate pixels of both passes in a fixed order (no atomics; as warp_train.hip's backward, whose candidate boxes
// and weight recomputation it repeats).  The pose matrices are read directly: rot = [T00 T01 0; T10 T11 0], tr = [1 0 T03 / 32; 0 1 -T13 / 32]
// (graph.py::warp_batch's 4 T / 128, exact in binary).
#include "common.h"

namespace {

__device__ __forceinline__ void vt_unpack8(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 vt_pack8(const float f[8]) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// (as warp_train.hip) sample position, in input pixel units, of output pixel (j = column, i = row) under theta
__device__ __forceinline__ void vt_sample_pos(const float th[6], int j, int i, int H, int W, float &ix, float &iy) {
    const float xn = (2.0f * (float)j + 1.0f) / (float)W - 1.0f;
    const float yn = (2.0f * (float)i + 1.0f) / (float)H - 1.0f;
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
    iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
}

struct VtTaps {
    float w[4];   // nw, ne, sw, se (at::native grid_sampler_2d's order of additions); 0 for a tap outside the map
    int o[4];     // pixel index of the tap (0 where the weight is 0)
};

__device__ __forceinline__ VtTaps vt_taps(const float th[6], int j, int i, int H, int W) {
    float ix, iy;
    vt_sample_pos(th, j, i, H, W, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const bool any = fx >= -1.0f && fx < (float)W && fy >= -1.0f && fy < (float)H;     // (inf / nan positions contribute nothing)
    const int x0 = any ? (int)fx : 0, y0 = any ? (int)fy : 0;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool vx0 = any && x0 >= 0, vx1 = any && x0 + 1 < W, vy0 = any && y0 >= 0, vy1 = any && y0 + 1 < H;
    VtTaps t;
    t.w[0] = (vx0 && vy0) ? wx0 * wy0 : 0.f;
    t.w[1] = (vx1 && vy0) ? wx1 * wy0 : 0.f;
    t.w[2] = (vx0 && vy1) ? wx0 * wy1 : 0.f;
    t.w[3] = (vx1 && vy1) ? wx1 * wy1 : 0.f;
    t.o[0] = (vx0 && vy0) ? y0 * W + x0 : 0;
    t.o[1] = (vx1 && vy0) ? y0 * W + x0 + 1 : 0;
    t.o[2] = (vx0 && vy1) ? (y0 + 1) * W + x0 : 0;
    t.o[3] = (vx1 && vy1) ? (y0 + 1) * W + x0 + 1 : 0;
    return t;
}

struct VtBox {
    int jlo, jhi, ilo, ihi;
};

// the output pixels whose sample point can fall within one pixel of input pixel (x, y) (warp_train.hip's backward: a parallelogram around M^-1 (p - t),
// the whole map for a singular or non-finite theta -- still exact, only slower)
__device__ __forceinline__ VtBox vt_candidates(const float th[6], int x, int y, int H, int W) {
    const float fw = (float)W, fh = (float)H;
    const float m00 = th[0], m01 = th[1] * fw / fh, m10 = th[3] * fh / fw, m11 = th[4];
    float t0, t1;
    vt_sample_pos(th, 0, 0, H, W, t0, t1);
    const float det = m00 * m11 - m01 * m10;
    VtBox b = {0, W - 1, 0, H - 1};
    if (fabsf(det) > 1e-6f && isfinite(det) && isfinite(t0) && isfinite(t1)) {
        const float r00 = m11 / det, r01 = -m01 / det, r10 = -m10 / det, r11 = m00 / det;
        const float qj = r00 * ((float)x - t0) + r01 * ((float)y - t1), qi = r10 * ((float)x - t0) + r11 * ((float)y - t1);
        const float ej = fabsf(r00) + fabsf(r01) + 1e-2f, ei = fabsf(r10) + fabsf(r11) + 1e-2f;
        b.jlo = (int)fmaxf(ceilf(qj - ej), 0.f);
        b.jhi = (int)fminf(floorf(qj + ej), fw - 1.f);
        b.ilo = (int)fmaxf(ceilf(qi - ei), 0.f);
        b.ihi = (int)fminf(floorf(qi + ei), fh - 1.f);
    }
    return b;
}

// the forward kernel's weight of output pixel (j, i) on input pixel (x, y)
__device__ __forceinline__ float vt_weight(const float th[6], int j, int i, int x, int y, int H, int W) {
    float ix, iy;
    vt_sample_pos(th, j, i, H, W, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ((float)x == fx) ? 1.0f - (ix - fx) : (((float)x == fx + 1.0f) ? ix - fx : 0.f);
    const float wy = ((float)y == fy) ? 1.0f - (iy - fy) : (((float)y == fy + 1.0f) ? iy - fy : 0.f);
    return wx * wy;
}

__device__ __forceinline__ void vt_thetas(const float *T, float rot[6], float tr[6]) {
    rot[0] = T[0]; rot[1] = T[1]; rot[2] = 0.f;
    rot[3] = T[4]; rot[4] = T[5]; rot[5] = 0.f;
    tr[0] = 1.f; tr[1] = 0.f; tr[2] = 4.0f * T[3] / 128.0f;
    tr[3] = 0.f; tr[4] = 1.f; tr[5] = -4.0f * T[7] / 128.0f;
}

struct V2vMsgArgs {
    const uint16_t *cur;      // [N][HW][C]   forward: the ego maps (first half of conv_in)
    const uint16_t *base;     // [N][HW][C]   forward: the maps the neighbours send (== cur unless neighbor_source = 'feat' in a later round)
    const float *trans;       // [*][4][4]    pose matrices
    const int *src;           // [M * K]      forward: row of base a pair reads
    const int *tsel;          // [M * K]      a pair's matrix in trans
    const int *rows;          // [M]          forward: row of cur of item m
    const int *inv;           // [N][K]       backward: the pairs that read row r, in pair order
    const int *item_of_row;   // [N]          backward: the item whose ego map row r is (-1: none)
    uint16_t *conv_in;        // [M][HW][2C]  forward output
    const uint16_t *dconv_in; // [M][HW][2C]  backward input
    uint16_t *dbase;          // [N][HW][C]   backward output (add_ego: + the ego half)
    uint16_t *dcur;           // [N][HW][C]   backward output of the ego half when it is not added into dbase (rows that are no item: zeros)
    int M, K, N, C, H, W, add_ego;
};

// one thread = one output pixel x 8 channels (16 bytes)
__global__ __launch_bounds__(256) void v2v_message_kernel(const V2vMsgArgs a) {
    const int G = a.C >> 3, HW = a.H * a.W;
    const long long total = (long long)a.M * HW * G;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int g = (int)(id % G);
    const long long mp = id / G;
    const int pix = (int)(mp % HW), m = (int)(mp / HW);
    const int i = pix / a.W, j = pix - i * a.W;
    const uint4 *cur = reinterpret_cast<const uint4 *>(a.cur), *base = reinterpret_cast<const uint4 *>(a.base);
    uint4 *out = reinterpret_cast<uint4 *>(a.conv_in);
    out[((size_t)m * HW + pix) * (2 * G) + g] = cur[((size_t)a.rows[m] * HW + pix) * G + g];
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int k = 0; k < a.K; ++k) {
        const int pi = m * a.K + k;
        float rot[6], tr[6];
        vt_thetas(a.trans + (size_t)a.tsel[pi] * 16, rot, tr);
        const uint4 *s = base + (size_t)a.src[pi] * HW * G + g;
        const VtTaps t2 = vt_taps(tr, j, i, a.H, a.W);
        float o2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o2[c] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t2.w[u] == 0.f) continue;                       // (a zero-weight tap adds +-0 in the two-pass form)
            const int q = t2.o[u];
            const int qi = q / a.W, qj = q - qi * a.W;
            const VtTaps t1 = vt_taps(rot, qj, qi, a.H, a.W);
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (t1.w[w] == 0.f) continue;
                float x[8];
                vt_unpack8(s[(size_t)t1.o[w] * G], x);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += x[c] * t1.w[w];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) o2[c] += v[c] * t2.w[u];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += o2[c];
    }
    const float kf = (float)a.K;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = acc[c] / kf;
    out[((size_t)m * HW + pix) * (2 * G) + G + g] = vt_pack8(acc);
}

// The transpose of the above.  A workgroup owns VB_PIX pixels of one row r of base; the geometry -- which pixels of the rotated map read a pixel (<= 3 x 3
// candidates, weights w1), and which output pixels read each of those (<= 3 x 3, weights w2) -- depends on (pixel, pair) only, not on the channel, so it is
// computed ONCE per workgroup into LDS (one thread per (pixel, pair, candidate)) and the 32 channel lanes of a pixel walk the tables: in the first form of this
// kernel every thread recomputed ~340 sample positions for its 64 loads (156 us per 10-map step; this form: see profiles/r06_train_switch_ab_4.txt).  A
// singular or strongly shrinking pose (a candidate box wider than 3) sends the whole workgroup down the direct loops (vb_direct) -- exact all the same.
// The order of additions is the direct form's: candidates row-major, the inner sum before the outer product.
constexpr int VB_PIX = 8, VB_KMAX = 8;

template <int KMAX>
struct VbTables {
    float w1[VB_PIX][KMAX][9];      // weight of outer candidate c1 (0: none)
    int n2[VB_PIX][KMAX][9];        // its inner candidates with a non-zero weight, in row-major order, padded with (pixel 0, weight 0) to a multiple of 4
    float w2[VB_PIX][KMAX][9][12];
    int q2[VB_PIX][KMAX][9][12];
    float th[KMAX][12];             // rot | tr of pair k of this row, loaded once per workgroup
    int item[KMAX];                 // the item whose message pair k feeds
    int irregular;
};

__device__ __forceinline__ void vb_direct(const V2vMsgArgs &a, const float rot[6], const float tr[6], const uint4 *dm, int G, int x, int y, float ak[8]) {
    const VtBox b1 = vt_candidates(rot, x, y, a.H, a.W);            // pixels of the rotated map that read (x, y)
    for (int i1 = b1.ilo; i1 <= b1.ihi; ++i1)
        for (int j1 = b1.jlo; j1 <= b1.jhi; ++j1) {
            const float w1 = vt_weight(rot, j1, i1, x, y, a.H, a.W);
            if (w1 == 0.f) continue;
            float d1[8];                                            // gradient of the rotated map at (j1, i1)
#pragma unroll
            for (int c = 0; c < 8; ++c) d1[c] = 0.f;
            const VtBox b2 = vt_candidates(tr, j1, i1, a.H, a.W);   // output pixels that read (j1, i1)
            for (int i2 = b2.ilo; i2 <= b2.ihi; ++i2)
                for (int j2 = b2.jlo; j2 <= b2.jhi; ++j2) {
                    const float w2 = vt_weight(tr, j2, i2, j1, i1, a.H, a.W);
                    if (w2 == 0.f) continue;
                    float gq[8];
                    vt_unpack8(dm[(size_t)(i2 * a.W + j2) * (2 * G)], gq);
#pragma unroll
                    for (int c = 0; c < 8; ++c) d1[c] += w2 * gq[c];
                }
#pragma unroll
            for (int c = 0; c < 8; ++c) ak[c] += w1 * d1[c];
        }
}

// KMAX = 4 (up to five agents: 28 KiB of tables, five workgroups per CU) or 8 (57 KiB, two)
template <int KMAX>
__global__ __launch_bounds__(256) void v2v_message_bwd_kernel(const V2vMsgArgs a) {
    __shared__ VbTables<KMAX> tb;
    const int G = a.C >> 3, HW = a.H * a.W;
    const int chunks = (HW + VB_PIX - 1) / VB_PIX;
    const int r = blockIdx.x / chunks, p0 = (blockIdx.x - r * chunks) * VB_PIX;
    const int t = threadIdx.x;
    if (t == 0) tb.irregular = a.K > KMAX ? 1 : 0;
    if (t < a.K && t < KMAX) {          // phase 0: the row's K poses
        const int pi = a.inv[r * a.K + t];
        float rot[6], tr[6];
        vt_thetas(a.trans + (size_t)a.tsel[pi] * 16, rot, tr);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            tb.th[t][c] = rot[c];
            tb.th[t][6 + c] = tr[c];
        }
        tb.item[t] = pi / a.K;
    }
    __syncthreads();
    // phase 1: one thread per (pixel, pair, outer candidate)
    if (a.K <= KMAX) {
        for (int it = t; it < VB_PIX * a.K * 9; it += 256) {
            const int c1 = it % 9, k = (it / 9) % a.K, pl = it / (9 * a.K);
            const int pix = p0 + pl;
            float w1 = 0.f;
            int n2 = 0;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                tb.w2[pl][k][c1][c] = 0.f;
                tb.q2[pl][k][c1][c] = 0;
            }
            if (pix < HW) {
                const int y = pix / a.W, x = pix - y * a.W;
                float rot[6], tr[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    rot[c] = tb.th[k][c];
                    tr[c] = tb.th[k][6 + c];
                }
                const VtBox b1 = vt_candidates(rot, x, y, a.H, a.W);
                if (b1.jhi - b1.jlo > 2 || b1.ihi - b1.ilo > 2) tb.irregular = 1;
                const int i1 = b1.ilo + c1 / 3, j1 = b1.jlo + c1 % 3;
                if (i1 <= b1.ihi && j1 <= b1.jhi) {
                    w1 = vt_weight(rot, j1, i1, x, y, a.H, a.W);
                    if (w1 != 0.f) {
                        const VtBox b2 = vt_candidates(tr, j1, i1, a.H, a.W);
                        if (b2.jhi - b2.jlo > 2 || b2.ihi - b2.ilo > 2) tb.irregular = 1;
                        for (int c = 0; c < 9; ++c) {
                            const int i2 = b2.ilo + c / 3, j2 = b2.jlo + c % 3;
                            if (i2 > b2.ihi || j2 > b2.jhi) continue;
                            const float w2 = vt_weight(tr, j2, i2, j1, i1, a.H, a.W);
                            if (w2 == 0.f) continue;
                            tb.w2[pl][k][c1][n2] = w2;
                            tb.q2[pl][k][c1][n2] = i2 * a.W + j2;
                            ++n2;
                        }
                    }
                }
            }
            tb.w1[pl][k][c1] = w1;
            tb.n2[pl][k][c1] = n2;
        }
    }
    __syncthreads();
    const bool direct = tb.irregular != 0;
    // phase 2: 32 channel lanes per pixel
    const int pl = t >> 5, lane = t & 31;
    const int pix = p0 + pl;
    if (pix >= HW) return;
    const int y = pix / a.W, x = pix - y * a.W;
    const uint4 *d = reinterpret_cast<const uint4 *>(a.dconv_in);
    const int me = a.item_of_row[r];
    const float kf = (float)a.K;
    for (int g = lane; g < G; g += 32) {
        float acc[8], ego[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = ego[c] = 0.f;
        if (me >= 0) vt_unpack8(d[((size_t)me * HW + pix) * (2 * G) + g], ego);
        for (int k = 0; k < a.K; ++k) {
            float ak[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) ak[c] = 0.f;
            if (direct) {
                const int pi = a.inv[r * a.K + k];
                const uint4 *dm = d + (size_t)(pi / a.K) * HW * (2 * G) + G + g;
                float rot[6], tr[6];
                vt_thetas(a.trans + (size_t)a.tsel[pi] * 16, rot, tr);
                vb_direct(a, rot, tr, dm, G, x, y, ak);
            } else {
                const uint4 *dm = d + (size_t)tb.item[k] * HW * (2 * G) + G + g;       // the message half of that item's gradient
                for (int c1 = 0; c1 < 9; ++c1) {
                    const float w1 = tb.w1[pl][k][c1];
                    if (w1 == 0.f) continue;
                    const int n2 = tb.n2[pl][k][c1];
                    float d1[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) d1[c] = 0.f;
                    for (int e0 = 0; e0 < n2; e0 += 4) {          // four loads in flight (a padded entry reads pixel 0 with weight 0)
                        uint4 raw[4];
                        float w2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            w2[e] = tb.w2[pl][k][c1][e0 + e];
                            raw[e] = dm[(size_t)tb.q2[pl][k][c1][e0 + e] * (2 * G)];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float gq[8];
                            vt_unpack8(raw[e], gq);
#pragma unroll
                            for (int c = 0; c < 8; ++c) d1[c] += w2[e] * gq[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) ak[c] += w1 * d1[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += ak[c];
        }
        // mean's backward (the gradient / K) applied ONCE, to the sum: an IEEE division per tap and channel was most of this kernel's instructions (106 -> 46 us per 10-map step)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = acc[c] / kf;
        if (a.add_ego) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += ego[c];
        } else {
            reinterpret_cast<uint4 *>(a.dcur)[((size_t)r * HW + pix) * G + g] = vt_pack8(ego);
        }
        reinterpret_cast<uint4 *>(a.dbase)[((size_t)r * HW + pix) * G + g] = vt_pack8(acc);
    }
}

// ---------------------------------------------------------------------------------------------- gates (gru_train.hip's arithmetic on bf16 NHWC)
struct GatesNhwcArgs {
    const uint16_t *gi;    // [P][3C]  P = maps x pixels
    const float *bhh;      // [3C]
    const uint16_t *dh;    // [P][C]   (backward)
    uint16_t *h;           // [P][C]   (forward)
    uint16_t *dgi;         // [P][3C]  (backward)
    float *partial;        // [blocks][6C] (backward): sums of dgi AS STORED (r, z, n: d bias_ih) | the same r, z | sums of dpre_n * r (d bias_hh)
    long long P;
    int C, rows_per_block;
};

__device__ __forceinline__ float gn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool BWD>
__global__ __launch_bounds__(256) void gru_gates_nhwc_kernel(const GatesNhwcArgs a) {
    __shared__ float red[BWD ? 256 : 1][BWD ? 33 : 1];
    const int G = a.C >> 3;
    const int t = threadIdx.x, g = t % G, sub = t / G, nsub = 256 / G;
    float br[8], bz[8], bn[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        br[c] = a.bhh[g * 8 + c];
        bz[c] = a.bhh[a.C + g * 8 + c];
        bn[c] = a.bhh[2 * a.C + g * 8 + c];
    }
    float s[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) s[c] = 0.f;
    const long long p0 = (long long)blockIdx.x * a.rows_per_block;
    long long p1 = p0 + a.rows_per_block;
    if (p1 > a.P) p1 = a.P;
    const uint4 *gi = reinterpret_cast<const uint4 *>(a.gi);
    for (long long p = p0 + sub; p < p1; p += nsub) {
        float vr[8], vz[8], vn[8];
        vt_unpack8(gi[(size_t)p * 3 * G + g], vr);
        vt_unpack8(gi[(size_t)p * 3 * G + G + g], vz);
        vt_unpack8(gi[(size_t)p * 3 * G + 2 * G + g], vn);
        float dh[8];
        if (BWD) vt_unpack8(reinterpret_cast<const uint4 *>(a.dh)[(size_t)p * G + g], dh);
        float o0[8], o1[8], o2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float r = gn_sigmoid(vr[c] + br[c]), z = gn_sigmoid(vz[c] + bz[c]);
            const float n = tanhf(vn[c] + r * bn[c]);
            if (!BWD) {
                o0[c] = n - z * n;
            } else {
                const float dn = dh[c] * (1.0f - z), dz = -dh[c] * n;
                const float dpn = dn * (1.0f - n * n);
                const float dr = dpn * bn[c];
                o0[c] = dr * r * (1.0f - r);
                o1[c] = dz * z * (1.0f - z);
                o2[c] = dpn;
                s[24 + c] += dpn * r;
            }
        }
        if (!BWD) {
            reinterpret_cast<uint4 *>(a.h)[(size_t)p * G + g] = vt_pack8(o0);
        } else {
            const uint4 q0 = vt_pack8(o0), q1 = vt_pack8(o1), q2 = vt_pack8(o2);
            uint4 *dg = reinterpret_cast<uint4 *>(a.dgi);
            dg[(size_t)p * 3 * G + g] = q0;
            dg[(size_t)p * 3 * G + G + g] = q1;
            dg[(size_t)p * 3 * G + 2 * G + g] = q2;
            float f0[8], f1[8], f2[8];       // the sums are over the values AS STORED: what the convolution's own gradients see
            vt_unpack8(q0, f0);
            vt_unpack8(q1, f1);
            vt_unpack8(q2, f2);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s[c] += f0[c];
                s[8 + c] += f1[c];
                s[16 + c] += f2[c];
            }
        }
    }
    if (BWD) {
#pragma unroll
        for (int c = 0; c < 32; ++c) red[t][c] = s[c];
        __syncthreads();
        // thread (g, c) adds the nsub rows of its channel in row order
        for (int o = t; o < G * 32; o += 256) {
            const int gg = o >> 5, c = o & 31;
            float v = 0.f;
            for (int rr = 0; rr < nsub; ++rr) v += red[rr * G + gg][c];
            const int kind = c >> 3, ch = gg * 8 + (c & 7);
            float *row = a.partial + (size_t)blockIdx.x * 6 * a.C;
            if (kind < 3) row[kind * a.C + ch] = v;
            if (kind < 2) row[(3 + kind) * a.C + ch] = v;
            if (kind == 3) row[5 * a.C + ch] = v;
        }
    }
}

// out[c] = sum over the blocks' partials in block order (fp64).  A workgroup owns 32 columns: thread (rg, col) adds rows rg, rg + 8, ... (eight loads in flight,
// 128 contiguous bytes per row), then the eight row groups are added in order -- the same association every run.
__global__ __launch_bounds__(256) void vt_sum_finish_kernel(const float *partial, int n_blocks, int n_cols, float *out) {
    __shared__ double red[8][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    double s = 0.0;
    if (c < n_cols) {
        for (int b0 = rg; b0 < n_blocks; b0 += 64) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = b0 + 8 * k < n_blocks ? partial[(size_t)(b0 + 8 * k) * n_cols + c] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
    }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && c < n_cols) {
        double t = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][col];
        out[c] = (float)t;
    }
}

constexpr int GATES_MAX_BLOCKS = 256;

bool gates_shape_ok(long long P, int C) { return P > 0 && C >= 8 && C % 8 == 0 && 256 % (C / 8) == 0; }

int gates_plan(long long P, int C, int &rows_per_block) {
    const int nsub = 256 / (C / 8);
    long long rpb = (P + GATES_MAX_BLOCKS - 1) / GATES_MAX_BLOCKS;
    rpb = (rpb + nsub - 1) / nsub * nsub;
    rows_per_block = (int)rpb;
    return (int)((P + rpb - 1) / rpb);
}

bool msg_shape_ok(int M, int K, int N, int C, int H, int W) {
    return M > 0 && K > 0 && N > 0 && C >= 8 && C % 8 == 0 && H > 0 && W > 0 && (long long)H * W <= (1 << 24) && (long long)N * H * W * C < (1ll << 40) &&
           (long long)M * H * W * C < (1ll << 39);
}
}  // namespace

extern "C" int v2x_v2v_message_bf16(const uint16_t *cur, const uint16_t *base, const float *trans, const int *src, const int *tsel, const int *rows, int M, int K,
                                    int N, int C, int H, int W, uint16_t *conv_in, v2x_stream_t stream) {
    V2X_REQUIRE(cur && base && trans && src && tsel && rows && conv_in, "v2x_v2v_message_bf16: null pointer");
    V2X_REQUIRE(msg_shape_ok(M, K, N, C, H, W), "v2x_v2v_message_bf16: needs M, K, N, H, W > 0 and C %% 8 == 0, got M=%d K=%d N=%d C=%d H=%d W=%d", M, K, N, C, H, W);
    V2vMsgArgs a = {};
    a.cur = cur;
    a.base = base;
    a.trans = trans;
    a.src = src;
    a.tsel = tsel;
    a.rows = rows;
    a.conv_in = conv_in;
    a.M = M; a.K = K; a.N = N; a.C = C; a.H = H; a.W = W;
    const long long total = (long long)M * H * W * (C / 8);
    hipLaunchKernelGGL(v2v_message_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("v2v_message_kernel");
    return V2X_OK;
}

extern "C" int v2x_v2v_message_bwd_bf16(const uint16_t *dconv_in, const float *trans, const int *inv, const int *tsel, const int *item_of_row, int M, int K, int N,
                                        int C, int H, int W, uint16_t *dbase, uint16_t *dcur, v2x_stream_t stream) {
    V2X_REQUIRE(dconv_in && trans && inv && tsel && item_of_row && dbase, "v2x_v2v_message_bwd_bf16: null pointer");
    V2X_REQUIRE(msg_shape_ok(M, K, N, C, H, W), "v2x_v2v_message_bwd_bf16: needs M, K, N, H, W > 0 and C %% 8 == 0, got M=%d K=%d N=%d C=%d H=%d W=%d", M, K, N, C, H, W);
    V2vMsgArgs a = {};
    a.dconv_in = dconv_in;
    a.trans = trans;
    a.inv = inv;
    a.tsel = tsel;
    a.item_of_row = item_of_row;
    a.dbase = dbase;
    a.dcur = dcur;
    a.add_ego = dcur == nullptr;
    a.M = M; a.K = K; a.N = N; a.C = C; a.H = H; a.W = W;
    const long long blocks = (long long)N * (((long long)H * W + VB_PIX - 1) / VB_PIX);
    V2X_REQUIRE(blocks < (1ll << 31), "v2x_v2v_message_bwd_bf16: too many pixels for one launch");
    if (K <= 4) hipLaunchKernelGGL(v2v_message_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(v2v_message_bwd_kernel<VB_KMAX>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("v2v_message_bwd_kernel");
    return V2X_OK;
}

extern "C" int v2x_gru_gates_nhwc_bf16(const uint16_t *gi, const float *bias_hh, long long P, int C, uint16_t *h, v2x_stream_t stream) {
    V2X_REQUIRE(gi && bias_hh && h, "v2x_gru_gates_nhwc_bf16: null pointer");
    V2X_REQUIRE(gates_shape_ok(P, C), "v2x_gru_gates_nhwc_bf16: needs P > 0 and C in {8, 16, 32, ..., 2048} (C / 8 divides 256), got P=%lld C=%d", P, C);
    GatesNhwcArgs a = {};
    a.gi = gi;
    a.bhh = bias_hh;
    a.h = h;
    a.P = P;
    a.C = C;
    // forward: plenty of small workgroups (no partials to keep few)
    const int nsub = 256 / (C / 8);
    a.rows_per_block = nsub * 4;
    const long long blocks = (P + a.rows_per_block - 1) / a.rows_per_block;
    hipLaunchKernelGGL(gru_gates_nhwc_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("gru_gates_nhwc_kernel");
    return V2X_OK;
}

extern "C" long long v2x_gru_gates_nhwc_workspace_size(long long P, int C) {
    if (!gates_shape_ok(P, C)) return 0;
    int rpb;
    return (long long)gates_plan(P, C, rpb) * 6 * C * (long long)sizeof(float);
}

extern "C" int v2x_gru_gates_nhwc_bwd_bf16(const uint16_t *gi, const float *bias_hh, const uint16_t *dh, long long P, int C, uint16_t *dgi, float *sums6c,
                                           float *workspace, v2x_stream_t stream) {
    V2X_REQUIRE(gi && bias_hh && dh && dgi && sums6c && workspace, "v2x_gru_gates_nhwc_bwd_bf16: null pointer");
    V2X_REQUIRE(gates_shape_ok(P, C), "v2x_gru_gates_nhwc_bwd_bf16: needs P > 0 and C in {8, 16, 32, ..., 2048} (C / 8 divides 256), got P=%lld C=%d", P, C);
    GatesNhwcArgs a = {};
    a.gi = gi;
    a.bhh = bias_hh;
    a.dh = dh;
    a.dgi = dgi;
    a.partial = workspace;
    a.P = P;
    a.C = C;
    const int nblk = gates_plan(P, C, a.rows_per_block);
    hipLaunchKernelGGL(gru_gates_nhwc_kernel<true>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(vt_sum_finish_kernel, dim3((6 * C + 31) / 32), dim3(256), 0, (hipStream_t)stream, workspace, nblk, 6 * C, sums6c);
    V2X_CHECK_LAUNCH("gru_gates_nhwc_kernel<bwd>");
    return V2X_OK;
}
