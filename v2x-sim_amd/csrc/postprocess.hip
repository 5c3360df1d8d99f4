// f-1 -- detection post-processing on the device: foreground score, score threshold, 'faf' anchor decode and greedy NMS
// on the axis-aligned "stand-up" boxes of the rotated boxes.
//
// Replaces (per agent and frame) upstream coperception/utils/postprocess.py::apply_nms_det -- host numpy over
// 393 216 anchors per map, which at ~4 500 frames/s (22 000 maps/s) would be ~50x slower than the network that feeds it
// (code absent from /root/reference; README.md:36 names the detection task, README.md:101 the scripts calling it).
// Arithmetic = the build-owned spec of v2x_sim_amd/utils/postprocess.py, in fp32 like numpy:
//     score = softmax(cls)[1];  keep score >= thr;
//     x = xa+dx, y = ya+dy, w = wa*exp(clip(dw,-4,4)), h = ha*exp(clip(dh,-4,4)), yaw = atan2(sa,ca) + atan2(ds,dc);
//     greedy NMS in (score desc, anchor index asc) order, suppress when IoU(stand-up boxes) > nms_thr.
//
// Two launches:
//   1. det_candidates_kernel: one pass over the logits (HBM-bound: 8 B per anchor), candidates compacted per map through
//      an atomic slot counter (slot order is arbitrary -- the sort below makes the result deterministic);
//   2. det_nms_kernel: one 256-thread workgroup per map: bitonic sort of the <= cap (power of two, <= 4096) candidate keys
//      in LDS (key = ~score_bits << 32 | anchor index: ascending key = descending score, ties by ascending index), decode
//      of the sorted candidates into LDS, then the greedy scan -- candidate i is tested against the kept list by all
//      threads in parallel (__syncthreads_or), so the serial depth is the number of candidates, not candidates x kept.
#include "common.h"

constexpr int DET_MAX_CAP = 4096;

// ---- rotated-box IoU (row f-1: upstream's eval_map intersects shapely polygons; here: convex clipping in fp64) ----------
// Corners of (x, y, w, h, yaw), counter-clockwise, first = (+w/2, +h/2) rotated (the order of utils/postprocess.box_corners).
__device__ __forceinline__ void box_corners_d(const float *b, double (&cx)[4], double (&cy)[4]) {
    const double x = b[0], y = b[1], w = b[2], h = b[3], yaw = b[4];
    const double c = cos(yaw), s = sin(yaw);
    const double dx[4] = {0.5 * w, -0.5 * w, -0.5 * w, 0.5 * w}, dy[4] = {0.5 * h, 0.5 * h, -0.5 * h, -0.5 * h};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cx[q] = x + dx[q] * c - dy[q] * s;
        cy[q] = y + dx[q] * s + dy[q] * c;
    }
}

// Area of (quad 1) intersected with (quad 2): Sutherland-Hodgman, quad 1 clipped by the four edges of quad 2.  A convex
// polygon gains at most one vertex per clip: 4 + 4 = 8.
__device__ double quad_intersection_area(const double (&ax)[4], const double (&ay)[4], const double (&bx)[4], const double (&by)[4]) {
    double px[10], py[10], qx[10], qy[10];
    int n = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        px[i] = ax[i];
        py[i] = ay[i];
    }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const double ex0 = bx[e], ey0 = by[e], ex1 = bx[(e + 1) & 3], ey1 = by[(e + 1) & 3];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double sp = (ex1 - ex0) * (py[i] - ey0) - (ey1 - ey0) * (px[i] - ex0);
            const double sq = (ex1 - ex0) * (py[j] - ey0) - (ey1 - ey0) * (px[j] - ex0);
            if (sp >= 0.0) {
                qx[m] = px[i];
                qy[m] = py[i];
                ++m;
            }
            if ((sp >= 0.0) != (sq >= 0.0)) {
                const double t = sp / (sp - sq);
                qx[m] = px[i] + t * (px[j] - px[i]);
                qy[m] = py[i] + t * (py[j] - py[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) {
            px[i] = qx[i];
            py[i] = qy[i];
        }
    }
    if (n < 3) return 0.0;
    double a2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a2 += px[i] * py[j] - px[j] * py[i];
    }
    return 0.5 * fabs(a2);
}

__device__ double rotated_iou_d(const float *a, const float *b) {
    double ax[4], ay[4], bx[4], by[4];
    box_corners_d(a, ax, ay);
    box_corners_d(b, bx, by);
    // stand-up boxes first: disjoint -> 0 without clipping
    double a0 = ax[0], a1 = ax[0], a2 = ay[0], a3 = ay[0], b0 = bx[0], b1 = bx[0], b2 = by[0], b3 = by[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        a0 = fmin(a0, ax[q]); a1 = fmax(a1, ax[q]); a2 = fmin(a2, ay[q]); a3 = fmax(a3, ay[q]);
        b0 = fmin(b0, bx[q]); b1 = fmax(b1, bx[q]); b2 = fmin(b2, by[q]); b3 = fmax(b3, by[q]);
    }
    if (a1 < b0 || b1 < a0 || a3 < b2 || b3 < a2) return 0.0;
    const double inter = quad_intersection_area(ax, ay, bx, by);
    const double uni = (double)a[2] * a[3] + (double)b[2] * b[3] - inter;   // |w*h| of each rectangle
    return uni > 0.0 ? inter / uni : 0.0;
}

__global__ __launch_bounds__(256) void rotated_iou_kernel(const float *__restrict__ a, int na, const float *__restrict__ b, int nb,
                                                          float *__restrict__ iou) {
    const long long total = (long long)na * nb;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(p / nb), j = (int)(p - (long long)i * nb);
        iou[p] = (float)rotated_iou_d(a + (size_t)i * 5, b + (size_t)j * 5);
    }
}

// ---- eval_map's matching step on the device --------------------------------------------------------------------------
// One workgroup per image.  Detections arrive in descending-score order (the order v2x_det_postprocess emits); each in turn
// takes the ground-truth box of highest rotated IoU (lowest index on ties) and is a true positive iff that IoU >= thr and
// the box is still free -- mmdet's tpfp_default as upstream's mean_ap.py uses it; there is no second choice.  The serial
// depth is the number of detections; the IoUs of one detection against all ground truths run in parallel.
__global__ __launch_bounds__(256) void match_detections_kernel(const float *__restrict__ det, const int32_t *__restrict__ det_count,
                                                               int det_cap, const float *__restrict__ gt,
                                                               const int32_t *__restrict__ gt_count, int gt_cap, float thr,
                                                               int32_t *__restrict__ tp, float *__restrict__ best_iou) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *taken = reinterpret_cast<int *>(smem);                 // [gt_cap]
    __shared__ double s_best[256];
    __shared__ int s_arg[256];
    const int img = blockIdx.x, tid = threadIdx.x;
    int nd = det_count[img];
    nd = nd < 0 ? 0 : (nd > det_cap ? det_cap : nd);
    int ng = gt_count[img];
    ng = ng < 0 ? 0 : (ng > gt_cap ? gt_cap : ng);
    for (int g = tid; g < ng; g += 256) taken[g] = 0;
    __syncthreads();
    for (int j = 0; j < nd; ++j) {
        const float *d = det + ((size_t)img * det_cap + j) * 5;
        double best = 0.0;
        int arg = -1;
        for (int g = tid; g < ng; g += 256) {
            const double v = rotated_iou_d(d, gt + ((size_t)img * gt_cap + g) * 5);
            if (v > best) {
                best = v;
                arg = g;
            }
        }
        s_best[tid] = best;
        s_arg[tid] = arg;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const double o = s_best[tid + st];
                const int oa = s_arg[tid + st];
                // larger IoU wins; equal IoU: the lower ground-truth index (a sequential scan keeps the first maximum)
                if (o > s_best[tid] || (o == s_best[tid] && oa >= 0 && (s_arg[tid] < 0 || oa < s_arg[tid]))) {
                    s_best[tid] = o;
                    s_arg[tid] = oa;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int a0 = s_arg[0];
            const bool hit = a0 >= 0 && s_best[0] >= (double)thr && !taken[a0];
            if (hit) taken[a0] = 1;
            tp[(size_t)img * det_cap + j] = hit ? 1 : 0;
            if (best_iou) best_iou[(size_t)img * det_cap + j] = (float)s_best[0];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void det_candidates_kernel(const float *__restrict__ cls, int n, int M, float thr,
                                                             int cap, unsigned long long *__restrict__ keys,
                                                             int32_t *__restrict__ counts) {
    const int map = blockIdx.y;
    const float2 *c2 = reinterpret_cast<const float2 *>(cls) + (size_t)map * M;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const float2 c = c2[m];
        // numpy: z = c - max(c); e = exp(z); e1 / (e0 + e1)
        const float mx = fmaxf(c.x, c.y);
        const float e0 = expf(c.x - mx), e1 = expf(c.y - mx);
        const float fg = e1 / (e0 + e1);
        if (fg >= thr) {
            const int pos = atomicAdd(&counts[map], 1);
            if (pos < cap) keys[(size_t)map * cap + pos] = ((unsigned long long)(~__float_as_uint(fg)) << 32) | (unsigned)m;
        }
    }
}

// SLOTTED (candidates from the fused detection heads, V2X_EPI_DET): the key's low word is (anchor index << 12 | slot) and `loc` holds the
// six codes of each CANDIDATE at [map][slot] instead of every anchor's at [map][anchor] -- same order (score, then anchor index), same
// decode, same suppression.
// Two launches per batch (det_nms_launch): FAST = true handles the maps with at most `lcap` (512) candidates -- the normal case -- with
// 46 KiB of LDS (three workgroups per CU, so that 320 maps run in ONE round) and a PARALLEL suppression: all pairwise overlaps are
// evaluated at once into a bit matrix (row i = the later candidates that i suppresses), then one wave walks the candidates in order,
// OR-ing the rows of the kept ones (the greedy scan of the first version synchronised the workgroup twice per candidate: ~1 us each,
// 430 us for 400 candidates per map).  Maps with more candidates are marked NMS_PENDING and taken by the second launch (FAST = false,
// `pending_only`: the serial scan with LDS for `cap` = up to 4096 candidates), which returns at once for every other map.
// Same order, same overlap arithmetic, same decisions: identical detections.
constexpr int NMS_PENDING = (int)0x80000000;
constexpr int NMS_FAST_CAP = 512;

template <bool ROTATED, bool SLOTTED = false, bool FAST = false>
__global__ __launch_bounds__(256) void det_nms_kernel(const float *__restrict__ loc, const float *__restrict__ anchors,
                                                      int M, int cap, float nms_thr, const unsigned long long *__restrict__ keys,
                                                      const int32_t *__restrict__ counts, float *__restrict__ out_boxes,
                                                      float *__restrict__ out_scores, int32_t *__restrict__ out_index,
                                                      int32_t *__restrict__ out_count, int lcap, int pending_only) {
    // lcap = candidates this launch's LDS holds (FAST: NMS_FAST_CAP, else cap); cap = the row stride of keys / outputs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem);          // [lcap]
    float4 *sbox = reinterpret_cast<float4 *>(smem + (size_t)lcap * 8);                 // [lcap] stand-up x1,y1,x2,y2
    int *skept = reinterpret_cast<int *>(smem + (size_t)lcap * 24);                     // [lcap] sorted positions kept
    unsigned long long *smask = reinterpret_cast<unsigned long long *>(smem + (size_t)lcap * 28);   // FAST: [lcap][lcap / 64]
    __shared__ int s_nkept;

    const int map = blockIdx.x;
    const int tid = threadIdx.x;
    if (pending_only && out_count[map] != NMS_PENDING) return;
    int cnt = counts[map];
    if (cnt > cap) {            // overflow: report -count, the caller decides (raise / host path / larger cap)
        if (tid == 0) out_count[map] = -cnt;
        return;
    }
    if (FAST && cnt > lcap) {   // more candidates than the fast form holds: the second launch takes this map
        if (tid == 0) out_count[map] = NMS_PENDING;
        return;
    }
    // pow2 >= cnt for the sort
    int np2 = 1;
    while (np2 < cnt) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) skey[i] = i < cnt ? keys[(size_t)map * cap + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = skey[i], b = skey[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        skey[i] = b;
                        skey[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // decode the sorted candidates: stand-up boxes to LDS, (x, y, w, h, yaw) to the output staging area (global, sorted
    // order; compacted below)
    float *stage = out_boxes + (size_t)map * cap * 5;   // reused: kept boxes are written in place, front-compacted later
    for (int i = tid; i < cnt; i += 256) {
        const unsigned lowk = (unsigned)(skey[i] & 0xffffffffu);
        const unsigned m = SLOTTED ? (lowk >> 12) : lowk;
        const float *l = SLOTTED ? loc + ((size_t)map * cap + (lowk & 0xfffu)) * 6 : loc + ((size_t)map * M + m) * 6;
        const float *a = anchors + (size_t)m * 6;
        const float x = a[0] + l[0], y = a[1] + l[1];
        const float w = a[2] * expf(fminf(fmaxf(l[2], -4.0f), 4.0f));
        const float h = a[3] * expf(fminf(fmaxf(l[3], -4.0f), 4.0f));
        const float yaw = atan2f(a[4], a[5]) + atan2f(l[4], l[5]);
        const float c = cosf(yaw), s = sinf(yaw);
        // corners (+-w/2, +-h/2) rotated: x extents = |w/2*c| + |h/2*s| in exact arithmetic; numpy takes min/max of the four
        // corners -- do the same so that the roundings agree
        const float dx[4] = {w / 2, -w / 2, -w / 2, w / 2}, dy[4] = {h / 2, h / 2, -h / 2, -h / 2};
        float x1 = 3.0e38f, y1 = 3.0e38f, x2 = -3.0e38f, y2 = -3.0e38f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // separately rounded products and sums, in numpy's order (no FMA contraction)
            const float cx = __fsub_rn(__fadd_rn(x, __fmul_rn(dx[q], c)), __fmul_rn(dy[q], s));
            const float cy = __fadd_rn(__fadd_rn(y, __fmul_rn(dx[q], s)), __fmul_rn(dy[q], c));
            x1 = fminf(x1, cx);
            x2 = fmaxf(x2, cx);
            y1 = fminf(y1, cy);
            y2 = fmaxf(y2, cy);
        }
        sbox[i] = make_float4(x1, y1, x2, y2);
        float *o = stage + (size_t)i * 5;
        o[0] = x;
        o[1] = y;
        o[2] = w;
        o[3] = h;
        o[4] = yaw;
    }
    if (tid == 0) s_nkept = 0;
    __syncthreads();
    if constexpr (FAST) {
        // suppression bit matrix: word (i, w) = the candidates j in [64 w, 64 w + 64), j > i, that candidate i suppresses if it is kept
        const int W = (cnt + 63) >> 6;
        for (int idx = tid; idx < cnt * W; idx += 256) {
            const int i = idx / W, w = idx - i * W;
            unsigned long long bits = 0;
            const int j0 = w << 6;
            if (j0 + 63 > i) {
                const float4 bi = sbox[i];
                const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
                for (int b = 0; b < 64; ++b) {
                    const int j = j0 + b;
                    if (j <= i || j >= cnt) continue;
                    // the serial form's arithmetic with (kept, candidate) = (i, j)
                    const float4 bj = sbox[j];
                    const float iw = fmaxf(0.0f, fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x));
                    const float ih = fmaxf(0.0f, fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y));
                    const float inter = iw * ih;
                    const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
                    bool sup;
                    if constexpr (ROTATED) sup = inter > 0.0f && (float)rotated_iou_d(stage + (size_t)i * 5, stage + (size_t)j * 5) > nms_thr;
                    else sup = inter / (area_i + area_j - inter + 1e-12f) > nms_thr;
                    if (sup) bits |= 1ull << b;
                }
            }
            smask[idx] = bits;
        }
        __syncthreads();
        if (tid < 64) {   // one wave: lane w owns word w of the "removed" set (W <= 8)
            unsigned long long removed = 0;
            int nk = 0;
            for (int i = 0; i < cnt; ++i) {
                const unsigned long long word = __shfl(removed, i >> 6);
                if (!((word >> (i & 63)) & 1ull)) {        // wave-uniform
                    if (tid == 0) skept[nk] = i;
                    ++nk;
                    if (tid < W) removed |= smask[i * W + tid];
                }
            }
            if (tid == 0) s_nkept = nk;
        }
        __syncthreads();
    } else
    // greedy scan
    for (int i = 0; i < cnt; ++i) {
        const float4 bi = sbox[i];
        const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
        const int nk = s_nkept;
        int sup = 0;
        for (int k = tid; k < nk; k += 256) {
            const float4 bk = sbox[skept[k]];
            const float iw = fmaxf(0.0f, fminf(bk.z, bi.z) - fmaxf(bk.x, bi.x));
            const float ih = fmaxf(0.0f, fminf(bk.w, bi.w) - fmaxf(bk.y, bi.y));
            const float inter = iw * ih;
            const float area_k = (bk.z - bk.x) * (bk.w - bk.y);
            const float iou = inter / (area_k + area_i - inter + 1e-12f);
            if constexpr (ROTATED) {
                // rotated mode: the stand-up overlap is only the cheap reject; the decision is the polygon IoU of the decoded
                // boxes (staged in global memory in sorted order)
                if (inter > 0.0f && (float)rotated_iou_d(stage + (size_t)skept[k] * 5, stage + (size_t)i * 5) > nms_thr) sup = 1;
            } else {
                if (iou > nms_thr) sup = 1;
            }
        }
        sup = __syncthreads_or(sup);
        if (!sup && tid == 0) {
            skept[nk] = i;
            s_nkept = nk + 1;
        }
        __syncthreads();
    }
    // compact the kept detections to the front (sorted positions are increasing, so in-place forward copy is safe when done
    // by one wave in order: position k <- skept[k] >= k)
    const int nk = s_nkept;
    for (int k0 = 0; k0 < nk; k0 += 256) {
        const int k = k0 + tid;
        float v[5];
        unsigned long long key = 0;
        if (k < nk) {
            const int src = skept[k];
            for (int e = 0; e < 5; ++e) v[e] = stage[(size_t)src * 5 + e];
            key = skey[src];
        }
        __syncthreads();   // all reads of this batch before its writes (sources are >= destinations, later batches read further right)
        if (k < nk) {
            for (int e = 0; e < 5; ++e) stage[(size_t)k * 5 + e] = v[e];
            out_scores[(size_t)map * cap + k] = __uint_as_float(~(unsigned)(key >> 32));
            out_index[(size_t)map * cap + k] = SLOTTED ? (int)((unsigned)(key & 0xffffffffu) >> 12) : (int)(key & 0xffffffffu);
        }
        __syncthreads();
    }
    if (tid == 0) out_count[map] = nk;
}

template <bool SLOTTED>
static int det_nms_launch(bool rotated, const float *loc_or_codes, const float *anchors, int n, int M, int cap, float nms_thr,
                          const unsigned long long *keys, const int32_t *counts, float *out_boxes, float *out_scores, int32_t *out_index,
                          int32_t *out_count, hipStream_t s) {
    const int smem_fast = NMS_FAST_CAP * 28 + NMS_FAST_CAP * (NMS_FAST_CAP / 64) * 8;   // 46 KiB: three workgroups per CU
    const int smem_full = cap * 28;   // keys 8 B + stand-up boxes 16 B + kept list 4 B per candidate
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(det_nms_kernel<false, SLOTTED, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DET_MAX_CAP * 28);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(det_nms_kernel<true, SLOTTED, false>), hipFuncAttributeMaxDynamicSharedMemorySize, DET_MAX_CAP * 28);
    }
    if (rotated) {
        hipLaunchKernelGGL((det_nms_kernel<true, SLOTTED, true>), dim3(n), dim3(256), smem_fast, s, loc_or_codes, anchors, M, cap, nms_thr, keys, counts,
                           out_boxes, out_scores, out_index, out_count, NMS_FAST_CAP, 0);
        hipLaunchKernelGGL((det_nms_kernel<true, SLOTTED, false>), dim3(n), dim3(256), smem_full, s, loc_or_codes, anchors, M, cap, nms_thr, keys, counts,
                           out_boxes, out_scores, out_index, out_count, cap, 1);
    } else {
        hipLaunchKernelGGL((det_nms_kernel<false, SLOTTED, true>), dim3(n), dim3(256), smem_fast, s, loc_or_codes, anchors, M, cap, nms_thr, keys, counts,
                           out_boxes, out_scores, out_index, out_count, NMS_FAST_CAP, 0);
        hipLaunchKernelGGL((det_nms_kernel<false, SLOTTED, false>), dim3(n), dim3(256), smem_full, s, loc_or_codes, anchors, M, cap, nms_thr, keys, counts,
                           out_boxes, out_scores, out_index, out_count, cap, 1);
    }
    V2X_CHECK_LAUNCH("det_nms_kernel");
    return V2X_OK;
}

static int det_postprocess_impl(bool rotated, const float *cls, const float *loc, const float *anchors, int n, int M,
                                float score_thr, float nms_thr, int cap, float *out_boxes, float *out_scores,
                                int32_t *out_index, int32_t *out_count, unsigned long long *key_scratch,
                                int32_t *count_scratch, v2x_stream_t stream) {
    V2X_REQUIRE(cls && loc && anchors && out_boxes && out_scores && out_index && out_count && key_scratch && count_scratch,
                "v2x_det_postprocess: null pointer");
    V2X_REQUIRE(n >= 0 && M > 0, "v2x_det_postprocess: bad sizes");
    V2X_REQUIRE(cap >= 64 && cap <= DET_MAX_CAP && (cap & (cap - 1)) == 0, "v2x_det_postprocess: cap=%d must be a power of two in [64, %d]", cap, DET_MAX_CAP);
    if (n == 0) return V2X_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count_scratch, 0, (size_t)n * sizeof(int32_t), s) != hipSuccess) {
        v2x_set_error("v2x_det_postprocess: memset failed");
        return V2X_EIO;
    }
    int gx = (M + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(det_candidates_kernel, dim3(gx, n), dim3(256), 0, s, cls, n, M, score_thr, cap, key_scratch, count_scratch);
    V2X_CHECK_LAUNCH("det_candidates_kernel");
    return det_nms_launch<false>(rotated, loc, anchors, n, M, cap, nms_thr, key_scratch, count_scratch, out_boxes, out_scores, out_index, out_count, s);
}

extern "C" int v2x_det_postprocess(const float *cls, const float *loc, const float *anchors, int n, int M,
                                   float score_thr, float nms_thr, int cap, float *out_boxes, float *out_scores,
                                   int32_t *out_index, int32_t *out_count, unsigned long long *key_scratch,
                                   int32_t *count_scratch, v2x_stream_t stream) {
    return det_postprocess_impl(false, cls, loc, anchors, n, M, score_thr, nms_thr, cap, out_boxes, out_scores, out_index, out_count,
                                key_scratch, count_scratch, stream);
}

extern "C" int v2x_det_postprocess_rotated(const float *cls, const float *loc, const float *anchors, int n, int M,
                                           float score_thr, float nms_thr, int cap, float *out_boxes, float *out_scores,
                                           int32_t *out_index, int32_t *out_count, unsigned long long *key_scratch,
                                           int32_t *count_scratch, v2x_stream_t stream) {
    return det_postprocess_impl(true, cls, loc, anchors, n, M, score_thr, nms_thr, cap, out_boxes, out_scores, out_index, out_count,
                                key_scratch, count_scratch, stream);
}

extern "C" int v2x_det_nms_candidates(const unsigned long long *keys, const float *codes, const int32_t *counts, const float *anchors,
                                      int n, int M, int cap, float nms_thr, int rotated, float *out_boxes, float *out_scores,
                                      int32_t *out_index, int32_t *out_count, v2x_stream_t stream) {
    V2X_REQUIRE(keys && codes && counts && anchors && out_boxes && out_scores && out_index && out_count, "v2x_det_nms_candidates: null pointer");
    V2X_REQUIRE(n >= 0 && M > 0 && M < (1 << 20), "v2x_det_nms_candidates: M=%d must be below 2^20 (anchor index and slot share a 32-bit word)", M);
    V2X_REQUIRE(cap >= 64 && cap <= DET_MAX_CAP && (cap & (cap - 1)) == 0, "v2x_det_nms_candidates: cap=%d must be a power of two in [64, %d]", cap, DET_MAX_CAP);
    if (n == 0) return V2X_OK;
    return det_nms_launch<true>(rotated != 0, codes, anchors, n, M, cap, nms_thr, keys, counts, out_boxes, out_scores, out_index, out_count, (hipStream_t)stream);
}

extern "C" int v2x_rotated_iou(const float *boxes_a, int na, const float *boxes_b, int nb, float *iou, v2x_stream_t stream) {
    V2X_REQUIRE(na >= 0 && nb >= 0, "v2x_rotated_iou: negative count");
    if (na == 0 || nb == 0) return V2X_OK;
    V2X_REQUIRE(boxes_a && boxes_b && iou, "v2x_rotated_iou: null pointer");
    const long long total = (long long)na * nb;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(rotated_iou_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, boxes_a, na, boxes_b, nb, iou);
    V2X_CHECK_LAUNCH("rotated_iou_kernel");
    return V2X_OK;
}

extern "C" int v2x_match_detections(const float *det_boxes, const int32_t *det_count, int det_cap, const float *gt_boxes,
                                    const int32_t *gt_count, int gt_cap, int n_img, float iou_thr, int32_t *tp, float *best_iou,
                                    v2x_stream_t stream) {
    V2X_REQUIRE(n_img >= 0 && det_cap > 0 && gt_cap > 0 && gt_cap <= 8192, "v2x_match_detections: bad sizes (gt_cap <= 8192)");
    if (n_img == 0) return V2X_OK;
    V2X_REQUIRE(det_boxes && det_count && gt_boxes && gt_count && tp, "v2x_match_detections: null pointer");
    hipLaunchKernelGGL(match_detections_kernel, dim3(n_img), dim3(256), (size_t)gt_cap * sizeof(int), (hipStream_t)stream, det_boxes,
                       det_count, det_cap, gt_boxes, gt_count, gt_cap, iou_thr, tp, best_iou);
    V2X_CHECK_LAUNCH("match_detections_kernel");
    return V2X_OK;
}
