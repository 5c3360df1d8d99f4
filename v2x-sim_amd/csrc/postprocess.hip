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

__global__ __launch_bounds__(256) void det_nms_kernel(const float *__restrict__ loc, const float *__restrict__ anchors,
                                                      int M, int cap, float nms_thr, const unsigned long long *__restrict__ keys,
                                                      const int32_t *__restrict__ counts, float *__restrict__ out_boxes,
                                                      float *__restrict__ out_scores, int32_t *__restrict__ out_index,
                                                      int32_t *__restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem);          // [cap]
    float4 *sbox = reinterpret_cast<float4 *>(smem + (size_t)cap * 8);                  // [cap] stand-up x1,y1,x2,y2
    int *skept = reinterpret_cast<int *>(smem + (size_t)cap * 24);                      // [cap] sorted positions kept
    __shared__ int s_nkept;

    const int map = blockIdx.x;
    const int tid = threadIdx.x;
    int cnt = counts[map];
    if (cnt > cap) {            // overflow: report -count, the caller decides (raise / host path / larger cap)
        if (tid == 0) out_count[map] = -cnt;
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
        const unsigned m = (unsigned)(skey[i] & 0xffffffffu);
        const float *l = loc + ((size_t)map * M + m) * 6;
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
            if (iou > nms_thr) sup = 1;
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
            out_index[(size_t)map * cap + k] = (int)(key & 0xffffffffu);
        }
        __syncthreads();
    }
    if (tid == 0) out_count[map] = nk;
}

extern "C" int v2x_det_postprocess(const float *cls, const float *loc, const float *anchors, int n, int M,
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
    const int smem = cap * 28;   // keys 8 B + stand-up boxes 16 B + kept list 4 B per candidate
    static v2x_once_per_device attr_once;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(det_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DET_MAX_CAP * 28);
    }
    hipLaunchKernelGGL(det_nms_kernel, dim3(n), dim3(256), smem, s, loc, anchors, M, cap, nms_thr, key_scratch, count_scratch,
                       out_boxes, out_scores, out_index, out_count);
    V2X_CHECK_LAUNCH("det_nms_kernel");
    return V2X_OK;
}
