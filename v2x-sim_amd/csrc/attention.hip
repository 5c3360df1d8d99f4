// a5 -- when2com / who2com attention handshake, and a8 -- seg argmax + confusion matrix.
//
// Replaces upstream coperception/models/det/When2com.py::MIMOGeneralDotProductAttention
// (query projection nn.Linear(query_size, key_size), bmm(key, query^T), softmax over the
// keys) and the inference-time selections activated_select (p * (p > 0.2)) and
// argmax_select (one-hot of the arg-max key) -- code absent from /root/reference, see
// include/v2x_amd.h.  The whole handshake of a frame is a 5x5 problem with 1024-long dot
// products: one workgroup per frame, projected queries kept in LDS, one wave per
// (key, query) pair with a 64-lane shuffle reduction, softmax over <= 32 keys by a single
// thread per query.  fp32 throughout.
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void attn_handshake_kernel(const float *__restrict__ keys,
                                                             const float *__restrict__ querys,
                                                             const float *__restrict__ w_lin,
                                                             const float *__restrict__ b_lin, int A, int Bt,
                                                             int key_size, int query_size, int mode, float thres,
                                                             float *__restrict__ prob, float *__restrict__ coef) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *qp = reinterpret_cast<float *>(smem_raw);  // [A][key_size] projected queries
    float *sc = qp + (size_t)A * key_size;             // [A(k)][A(q)] scores
    const int f = blockIdx.x;
    // 1. query projection: qp[q][d] = b[d] + sum_s w[d][s] * query[q][s].  A thread owns output dimension d for ALL queries: row d of the
    // weight is read once, as independent 16-byte loads (the first form walked it with query_size dependent 4-byte loads per (q, d): 20 rounds
    // of 32 load latencies = the whole 120 us of the launch); the A query vectors are staged in LDS.  Sums in the order t = 0, 1, ...: same bits.
    float *qs = sc + A * A;                            // [A][query_size]
    for (int i = threadIdx.x; i < A * query_size; i += blockDim.x) {
        const int q = i / query_size, t = i - q * query_size;
        qs[i] = querys[((size_t)q * Bt + f) * query_size + t];
    }
    __syncthreads();
    if (query_size == 32 && A <= 8) {
        for (int d = threadIdx.x; d < key_size; d += blockDim.x) {
            float4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = reinterpret_cast<const float4 *>(w_lin + (size_t)d * 32)[j];
            const float b = b_lin[d];
            for (int q = 0; q < A; ++q) {
                const float *qv = qs + q * 32;
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s += w[j].x * qv[4 * j];
                    s += w[j].y * qv[4 * j + 1];
                    s += w[j].z * qv[4 * j + 2];
                    s += w[j].w * qv[4 * j + 3];
                }
                qp[(size_t)q * key_size + d] = s + b;
            }
        }
    } else {
        for (int i = threadIdx.x; i < A * key_size; i += blockDim.x) {
            const int q = i / key_size, d = i - q * key_size;
            const float *qv = qs + q * query_size;
            const float *wr = w_lin + (size_t)d * query_size;
            float s = 0.f;
            for (int t = 0; t < query_size; ++t) s += wr[t] * qv[t];
            qp[i] = s + b_lin[d];
        }
    }
    __syncthreads();
    // 2. scores[k][q] = key_k . qp_q   (one wave per pair)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int pair = wave; pair < A * A; pair += nw) {
        const int k = pair / A, q = pair - k * A;
        const float *kv = keys + ((size_t)k * Bt + f) * key_size;
        const float *qv = qp + (size_t)q * key_size;
        float s = 0.f;
        for (int d = lane; d < key_size; d += 64) s += kv[d] * qv[d];
        s = wave_sum(s);
        if (lane == 0) sc[k * A + q] = s;
    }
    __syncthreads();
    // 3. softmax over keys k for every query q, then the selection
    if (threadIdx.x < A) {
        const int q = threadIdx.x;
        float mx = -INFINITY;
        for (int k = 0; k < A; ++k) mx = fmaxf(mx, sc[k * A + q]);
        float den = 0.f;
        for (int k = 0; k < A; ++k) den += expf(sc[k * A + q] - mx);
        int arg = 0;
        float best = -INFINITY;
        for (int k = 0; k < A; ++k) {
            const float p = expf(sc[k * A + q] - mx) / den;
            prob[((size_t)f * A + k) * A + q] = p;
            if (p > best) {  // first maximum wins, as torch.max
                best = p;
                arg = k;
            }
        }
        for (int k = 0; k < A; ++k) {
            const float p = prob[((size_t)f * A + k) * A + q];
            float c;
            if (mode == 0) c = p;
            else if (mode == 1) c = (p > thres) ? p : 0.f;
            else c = (k == arg) ? 1.f : 0.f;
            coef[((size_t)f * A + k) * A + q] = c;
        }
    }
}

extern "C" int v2x_attn_handshake(const float *keys, const float *querys, const float *w_lin, const float *b_lin,
                                  int A, int Bt, int key_size, int query_size, int mode, float thres, float *prob,
                                  float *coef, v2x_stream_t stream) {
    V2X_REQUIRE(keys && querys && w_lin && b_lin && prob && coef, "v2x_attn_handshake: null pointer");
    V2X_REQUIRE(A > 0 && A <= 32 && Bt > 0 && key_size > 0 && query_size > 0, "v2x_attn_handshake: bad dims");
    V2X_REQUIRE(mode >= 0 && mode <= 2, "v2x_attn_handshake: mode must be 0 (softmax), 1 (activated) or 2 (argmax_test)");
    const size_t smem = ((size_t)A * key_size + (size_t)A * A + (size_t)A * query_size) * sizeof(float);
    V2X_REQUIRE(smem <= 64 * 1024, "v2x_attn_handshake: A*key_size too large for LDS staging");
    hipLaunchKernelGGL(attn_handshake_kernel, dim3(Bt), dim3(256), smem, (hipStream_t)stream, keys, querys, w_lin,
                       b_lin, A, Bt, key_size, query_size, mode, thres, prob, coef);
    V2X_CHECK_LAUNCH("attn_handshake_kernel");
    return V2X_OK;
}

// ---- a8: per-pixel argmax over class logits + LDS-binned integer confusion matrix ----------
__global__ __launch_bounds__(256) void seg_argmax_confusion_kernel(const float *__restrict__ logits,
                                                                   const uint8_t *__restrict__ label, size_t n_pix,
                                                                   int n_cls, uint8_t *__restrict__ pred,
                                                                   unsigned long long *__restrict__ conf) {
    extern __shared__ unsigned int bins[];  // [n_cls*n_cls]
    for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x) bins[i] = 0u;
    __syncthreads();
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += (size_t)gridDim.x * blockDim.x) {
        const float *l = logits + p * n_cls;
        int arg = 0;
        float best = l[0];
        for (int c = 1; c < n_cls; ++c) {
            const float v = l[c];
            if (v > best) {  // first maximum wins, as torch.argmax
                best = v;
                arg = c;
            }
        }
        if (pred) pred[p] = (uint8_t)arg;
        if (label) {
            const int lb = label[p];
            if (lb < n_cls) atomicAdd(&bins[lb * n_cls + arg], 1u);
        }
    }
    __syncthreads();
    if (label)
        for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x)
            if (bins[i]) atomicAdd(&conf[i], (unsigned long long)bins[i]);
}

// The 8-class form (V2X-Sim): a thread owns FOUR consecutive pixels -- 128 contiguous bytes of logits as eight 16-byte loads, the four labels and
// the four predictions as one 32-bit word each (the generic kernel: eight strided 4-byte loads and 1-byte accesses per pixel) -- and the
// workgroup keeps 16 copies of the 64-bin histogram (copy = lane & 15; the single copy ran at 89 % LDS bank conflicts).  Same integers.
__global__ __launch_bounds__(256) void seg_argmax_confusion8_kernel(const float4 *__restrict__ logits, const uint32_t *__restrict__ label4,
                                                                    size_t n_quads, uint32_t *__restrict__ pred4,
                                                                    unsigned long long *__restrict__ conf) {
    __shared__ unsigned int bins[16][64];
    for (int i = threadIdx.x; i < 16 * 64; i += 256) (&bins[0][0])[i] = 0u;
    __syncthreads();
    const int copy = threadIdx.x & 15;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = logits[q * 8 + j];
        const uint32_t lb = label4 ? label4[q] : 0u;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float l[8] = {v[2 * k].x, v[2 * k].y, v[2 * k].z, v[2 * k].w, v[2 * k + 1].x, v[2 * k + 1].y, v[2 * k + 1].z, v[2 * k + 1].w};
            int arg = 0;
            float best = l[0];
#pragma unroll
            for (int c = 1; c < 8; ++c)
                if (l[c] > best) {  // first maximum wins, as torch.argmax
                    best = l[c];
                    arg = c;
                }
            out |= (uint32_t)arg << (8 * k);
            if (label4) {
                const uint32_t b = (lb >> (8 * k)) & 0xffu;
                if (b < 8u) atomicAdd(&bins[copy][b * 8 + arg], 1u);
            }
        }
        if (pred4) pred4[q] = out;
    }
    __syncthreads();
    if (label4 && threadIdx.x < 64) {
        unsigned int t = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) t += bins[c][threadIdx.x];
        if (t) atomicAdd(&conf[threadIdx.x], (unsigned long long)t);
    }
}

extern "C" int v2x_seg_argmax_confusion(const float *logits, const uint8_t *label, int n, int H, int W, int n_cls,
                                        uint8_t *pred, long long *conf, v2x_stream_t stream) {
    V2X_REQUIRE(logits, "v2x_seg_argmax_confusion: null logits");
    V2X_REQUIRE(n >= 0 && H > 0 && W > 0 && n_cls > 0 && n_cls <= 64, "v2x_seg_argmax_confusion: bad dims (n_cls <= 64)");
    V2X_REQUIRE((label == nullptr) == (conf == nullptr), "v2x_seg_argmax_confusion: label and conf go together");
    V2X_REQUIRE(pred || label, "v2x_seg_argmax_confusion: nothing to compute");
    if (n == 0) return V2X_OK;
    const size_t n_pix = (size_t)n * H * W;
    if (n_cls == 8 && n_pix % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(pred) & 3) == 0) {
        const size_t nq = n_pix / 4;
        size_t g8 = (nq + 255) / 256;
        if (g8 > 4096) g8 = 4096;
        hipLaunchKernelGGL(seg_argmax_confusion8_kernel, dim3((unsigned)g8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(logits),
                           reinterpret_cast<const uint32_t *>(label), nq, reinterpret_cast<uint32_t *>(pred), reinterpret_cast<unsigned long long *>(conf));
        V2X_CHECK_LAUNCH("seg_argmax_confusion8_kernel");
        return V2X_OK;
    }
    size_t g = (n_pix + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(seg_argmax_confusion_kernel, dim3((unsigned)g), dim3(256), n_cls * n_cls * sizeof(unsigned int),
                       (hipStream_t)stream, logits, label, n_pix, n_cls, pred,
                       reinterpret_cast<unsigned long long *>(conf));
    V2X_CHECK_LAUNCH("seg_argmax_confusion_kernel");
    return V2X_OK;
}


// ---- f-4 (DiscoNet): per-pixel softmax over the source agents + weighted sum of their warped maps -------
// Replaces the tail of upstream coperception/models/det/DiscoNet.py::fusion: w_k = exp(s_k) / sum_j exp(s_j) per pixel
// (s_k = the 1-channel output of PixelWeightedFusionSoftmax for source k, already ReLU'd), out = sum_k w_k * map_k.
// scores fp32 [n_items][A][HW][score_stride] (channel 0 is read), valid fp32 [n_items][A] (0 = source absent),
// maps bf16 [n_items][A][HW][C] -> out bf16 [n_items][HW][C].  One thread per (pixel, 8 channels); fp32 accumulation in
// source-index order; exp without max-subtraction, exactly as upstream (scores are small non-negative numbers).
__global__ __launch_bounds__(256) void pixel_weighted_fuse_kernel(const float *__restrict__ scores, int score_stride,
                                                                  const float *__restrict__ valid,
                                                                  const uint16_t *__restrict__ maps, int A, int HW, int C,
                                                                  uint16_t *__restrict__ out) {
    const int m = blockIdx.y;
    const int cvecs = C >> 3;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < HW * cvecs; idx += gridDim.x * blockDim.x) {
        const int pix = idx / cvecs, cv = idx - pix * cvecs;
        float e[32];
        float sum = 0.f;
        for (int k = 0; k < A; ++k) {
            const bool ok = valid[m * A + k] != 0.f;
            e[k] = ok ? expf(scores[(((size_t)m * A + k) * HW + pix) * score_stride]) : 0.f;
            sum += e[k];
        }
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int k = 0; k < A; ++k) {
            if (e[k] == 0.f) continue;
            const float w = e[k] / sum;
            const uint4 v = *reinterpret_cast<const uint4 *>(maps + (((size_t)m * A + k) * HW + pix) * C + cv * 8);
            acc[0] += __uint_as_float(v.x << 16) * w;
            acc[1] += __uint_as_float(v.x & 0xffff0000u) * w;
            acc[2] += __uint_as_float(v.y << 16) * w;
            acc[3] += __uint_as_float(v.y & 0xffff0000u) * w;
            acc[4] += __uint_as_float(v.z << 16) * w;
            acc[5] += __uint_as_float(v.z & 0xffff0000u) * w;
            acc[6] += __uint_as_float(v.w << 16) * w;
            acc[7] += __uint_as_float(v.w & 0xffff0000u) * w;
        }
        uint4 o;
        o.x = pack_bf16x2(acc[0], acc[1]);
        o.y = pack_bf16x2(acc[2], acc[3]);
        o.z = pack_bf16x2(acc[4], acc[5]);
        o.w = pack_bf16x2(acc[6], acc[7]);
        *reinterpret_cast<uint4 *>(out + ((size_t)m * HW + pix) * C + cv * 8) = o;
    }
}

extern "C" int v2x_pixel_weighted_fuse(const float *scores, int score_stride, const float *valid, const uint16_t *maps,
                                       int n_items, int A, int H, int W, int C, uint16_t *out, v2x_stream_t stream) {
    V2X_REQUIRE(scores && valid && maps && out, "v2x_pixel_weighted_fuse: null pointer");
    V2X_REQUIRE(n_items >= 0 && n_items <= 65535 && A > 0 && A <= 32 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && score_stride > 0,
                "v2x_pixel_weighted_fuse: bad dims (A <= 32, C %% 8 == 0)");
    if (n_items == 0) return V2X_OK;
    int gx = (H * W * (C / 8) + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(pixel_weighted_fuse_kernel, dim3(gx, n_items), dim3(256), 0, (hipStream_t)stream, scores, score_stride,
                       valid, maps, A, H * W, C, out);
    V2X_CHECK_LAUNCH("pixel_weighted_fuse_kernel");
    return V2X_OK;
}
