// The optimizer step of the TRAINING graph as launches of this library: Adam (Kingma & Ba) with
// torch.optim.Adam's arithmetic -- the scripts /root/reference/README.md:101 names build `optim.Adam(model.parameters(), lr=args.lr)` (code absent from
// /root/reference; restated from the published update rule and torch's documented form):
//     g  = grad (+ weight_decay * p)                 m = m + (1 - beta1) (g - m)             v = beta2 v + (1 - beta2) g g
//     p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// over ALL parameter tensors of a group in one launch per <= V2X_ADAM_MAX_TENSORS tensors: the tensor table travels in the kernel arguments (nothing to
// upload, so the launch can be captured into a hipGraph whatever addresses the gradients have), a workgroup finds its tensor by bisection of the table's
// block prefix.  16-byte accesses when every pointer of a launch is 16-byte aligned (torch allocations are), four of them in flight per array.
// torch's own fused step (multi_tensor_apply, 3 launches for FaFNet's ~100 tensors) streams the same 28 B per parameter at 1.8 TB/s.
#include "common.h"

namespace {
constexpr int ADAM_NT = V2X_ADAM_MAX_TENSORS;
constexpr int ADAM_BLOCK_ELEMS = 256 * 4 * 4;

struct AdamTable {
    float *p[ADAM_NT];
    const float *g[ADAM_NT];
    float *m[ADAM_NT];
    float *v[ADAM_NT];
    const float *step[ADAM_NT];   // device step counters (capturable optimizers), already incremented; NULL: step_host
    int numel[ADAM_NT];
    int start[ADAM_NT + 1];       // first workgroup of tensor i
};

struct AdamScalars {
    const float *lr_dev;          // a device learning rate (torch.optim.Adam(lr=tensor)); NULL: lr
    double beta1, beta2;          // the bias corrections are formed in fp64 once per workgroup: (float)0.999 is 1.3e-5 away from 0.999 in 1 - beta2^t
    float lr, omb1, omb2, eps, weight_decay, step_host;     // omb = 1 - beta, rounded from fp64 (1.0f - 0.999f is off by 4.7e-5)
    int n, vec;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2, float eps, float wd, float step_size, float rsqrt_bc2) {
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    const float denom = sqrtf(v) * rsqrt_bc2 + eps;
    p = p - step_size * m / denom;
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTable t, const AdamScalars s) {
    int lo = 0, hi = s.n;            // largest i with start[i] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const int i = lo;
    const int n = t.numel[i];
    const long long e0 = (long long)((int)blockIdx.x - t.start[i]) * ADAM_BLOCK_ELEMS;
    const float step = t.step[i] ? *t.step[i] : s.step_host;
    const float lr = s.lr_dev ? *s.lr_dev : s.lr;
    const double bc1 = 1.0 - pow(s.beta1, (double)step), bc2 = 1.0 - pow(s.beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float b2 = (float)s.beta2;
    float *p = t.p[i], *m = t.m[i], *v = t.v[i];
    const float *g = t.g[i];
    if (s.vec && e0 + ADAM_BLOCK_ELEMS <= n) {
        float4 pp[4], gg[4], mm[4], vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long q = (e0 >> 2) + k * 256 + threadIdx.x;
            pp[k] = reinterpret_cast<const float4 *>(p)[q];
            gg[k] = reinterpret_cast<const float4 *>(g)[q];
            mm[k] = reinterpret_cast<const float4 *>(m)[q];
            vv[k] = reinterpret_cast<const float4 *>(v)[q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long q = (e0 >> 2) + k * 256 + threadIdx.x;
            adam_one(pp[k].x, gg[k].x, mm[k].x, vv[k].x, s.omb1, b2, s.omb2, s.eps, s.weight_decay, step_size, rsqrt_bc2);
            adam_one(pp[k].y, gg[k].y, mm[k].y, vv[k].y, s.omb1, b2, s.omb2, s.eps, s.weight_decay, step_size, rsqrt_bc2);
            adam_one(pp[k].z, gg[k].z, mm[k].z, vv[k].z, s.omb1, b2, s.omb2, s.eps, s.weight_decay, step_size, rsqrt_bc2);
            adam_one(pp[k].w, gg[k].w, mm[k].w, vv[k].w, s.omb1, b2, s.omb2, s.eps, s.weight_decay, step_size, rsqrt_bc2);
            reinterpret_cast<float4 *>(p)[q] = pp[k];
            reinterpret_cast<float4 *>(m)[q] = mm[k];
            reinterpret_cast<float4 *>(v)[q] = vv[k];
        }
        return;
    }
    for (long long e = e0 + threadIdx.x; e < e0 + ADAM_BLOCK_ELEMS && e < n; e += 256) {     // a tensor's tail, or unaligned pointers
        float pe = p[e], me = m[e], ve = v[e];
        adam_one(pe, g[e], me, ve, s.omb1, b2, s.omb2, s.eps, s.weight_decay, step_size, rsqrt_bc2);
        p[e] = pe;
        m[e] = me;
        v[e] = ve;
    }
}
}  // namespace

extern "C" int v2x_adam_step_f32(const v2x_adam_tensors *tensors, int n_tensors, const float *lr_dev, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, double step_host, v2x_stream_t stream) {
    V2X_REQUIRE(tensors || n_tensors == 0, "v2x_adam_step_f32: null tensor table");
    V2X_REQUIRE(n_tensors >= 0 && n_tensors <= ADAM_NT, "v2x_adam_step_f32: at most %d tensors per call, got %d", ADAM_NT, n_tensors);
    V2X_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "v2x_adam_step_f32: betas in [0, 1), eps >= 0");
    if (n_tensors == 0) return V2X_OK;
    AdamTable t = {};
    AdamScalars s = {lr_dev, beta1, beta2, (float)lr, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)step_host, 0, 1};
    int blocks = 0, n = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const long long ne = tensors->numel[i];
        V2X_REQUIRE(ne >= 0 && ne < (1ll << 31), "v2x_adam_step_f32: tensor %d has %lld elements (one tensor < 2^31)", i, ne);
        if (ne == 0) continue;
        V2X_REQUIRE(tensors->param[i] && tensors->grad[i] && tensors->exp_avg[i] && tensors->exp_avg_sq[i], "v2x_adam_step_f32: tensor %d: null pointer", i);
        V2X_REQUIRE(tensors->step[i] || step_host >= 1.0, "v2x_adam_step_f32: tensor %d has no device step counter and step_host < 1", i);
        t.p[n] = tensors->param[i];
        t.g[n] = tensors->grad[i];
        t.m[n] = tensors->exp_avg[i];
        t.v[n] = tensors->exp_avg_sq[i];
        t.step[n] = tensors->step[i];
        t.numel[n] = (int)ne;
        t.start[n] = blocks;
        const uintptr_t bits = reinterpret_cast<uintptr_t>(t.p[n]) | reinterpret_cast<uintptr_t>(t.g[n]) | reinterpret_cast<uintptr_t>(t.m[n]) | reinterpret_cast<uintptr_t>(t.v[n]);
        if (bits & 15) s.vec = 0;
        blocks += (int)((ne + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS);
        ++n;
    }
    if (n == 0) return V2X_OK;
    t.start[n] = blocks;
    s.n = n;
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, s);
    V2X_CHECK_LAUNCH("adam_step_kernel");
    return V2X_OK;
}
