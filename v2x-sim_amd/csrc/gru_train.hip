// ConvGRU gate arithmetic of the TRAINING graph, forward and backward, as one launch each.
//
// Upstream: convolutional_rnn.Conv2dGRU (V2VNet.py calls convgru(x, None): h0 = 0, so the hidden-to-hidden convolution contributes its bias
// only) -- restated in v2x_sim_amd/train/graph.py::_gru_step:
//     r = sigmoid(gi_r + bh_r),  z = sigmoid(gi_z + bh_z),  n = tanh(gi_n + r bh_n),  h = n - z n
// on the fp32 NCHW pre-activations gi (P, 3C, H, W) of the input convolution (bias_ih included) and bias_hh (3C).  As PyTorch ops that is 8
// elementwise launches forward and ~14 backward; here one each, the backward also producing the tensor whose channel sums are d bias_hh's
// n part (d bias_hh's r and z parts are the channel sums of dgi itself).  Forward arithmetic = expf / tanhf in fp32 (this is the training graph:
// it is compared with torch's own sigmoid / tanh to 1e-6).
#include "common.h"

struct GruGateArgs {
    const float *gi;      // [P][3C][HW]
    const float *bhh;     // [3C]
    const float *dh;      // [P][C][HW] (backward)
    float *h;             // [P][C][HW] (forward)
    float *dgi;           // [P][3C][HW] (backward)
    float *dn_r;          // [P][C][HW] (backward): dpre_n * r, whose channel sums are d bias_hh[2C + c]
    long long P;
    int C, HW4;           // HW / 4
};

__device__ __forceinline__ float gg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool BWD>
__global__ __launch_bounds__(256) void gru_gates_kernel(const GruGateArgs a) {
    const long long total = a.P * a.C * a.HW4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % a.HW4);
        const long long pc = i / a.HW4;
        const int c = (int)(pc % a.C);
        const long long p = pc / a.C;
        const size_t base3 = ((size_t)p * 3 * a.C + c) * a.HW4 + q;       // float4 index of the r plane
        const size_t plane = (size_t)a.C * a.HW4;
        const float4 gr = reinterpret_cast<const float4 *>(a.gi)[base3];
        const float4 gz = reinterpret_cast<const float4 *>(a.gi)[base3 + plane];
        const float4 gn = reinterpret_cast<const float4 *>(a.gi)[base3 + 2 * plane];
        const float br = a.bhh[c], bz = a.bhh[a.C + c], bn = a.bhh[2 * a.C + c];
        const float vr[4] = {gr.x, gr.y, gr.z, gr.w}, vz[4] = {gz.x, gz.y, gz.z, gz.w}, vn[4] = {gn.x, gn.y, gn.z, gn.w};
        float o0[4], o1[4], o2[4], o3[4];
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
        if (BWD) {
            const float4 d = reinterpret_cast<const float4 *>(a.dh)[(size_t)pc * a.HW4 + q];
            dh[0] = d.x; dh[1] = d.y; dh[2] = d.z; dh[3] = d.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float r = gg_sigmoid(vr[k] + br), z = gg_sigmoid(vz[k] + bz);
            const float n = tanhf(vn[k] + r * bn);
            if (!BWD) {
                o0[k] = n - z * n;
            } else {
                const float dn = dh[k] * (1.0f - z), dz = -dh[k] * n;
                const float dpn = dn * (1.0f - n * n);
                const float dr = dpn * bn;
                o0[k] = dr * r * (1.0f - r);      // d gi_r
                o1[k] = dz * z * (1.0f - z);      // d gi_z
                o2[k] = dpn;                      // d gi_n
                o3[k] = dpn * r;                  // contributes to d bias_hh (n part)
            }
        }
        if (!BWD) {
            reinterpret_cast<float4 *>(a.h)[(size_t)pc * a.HW4 + q] = make_float4(o0[0], o0[1], o0[2], o0[3]);
        } else {
            reinterpret_cast<float4 *>(a.dgi)[base3] = make_float4(o0[0], o0[1], o0[2], o0[3]);
            reinterpret_cast<float4 *>(a.dgi)[base3 + plane] = make_float4(o1[0], o1[1], o1[2], o1[3]);
            reinterpret_cast<float4 *>(a.dgi)[base3 + 2 * plane] = make_float4(o2[0], o2[1], o2[2], o2[3]);
            reinterpret_cast<float4 *>(a.dn_r)[(size_t)pc * a.HW4 + q] = make_float4(o3[0], o3[1], o3[2], o3[3]);
        }
    }
}

static unsigned gg_grid(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

extern "C" int v2x_gru_gates_f32(const float *gi, const float *bias_hh, long long P, int C, int HW, float *h, v2x_stream_t stream) {
    V2X_REQUIRE(gi && bias_hh && h, "v2x_gru_gates_f32: null pointer");
    V2X_REQUIRE(P > 0 && C > 0 && HW > 0 && HW % 4 == 0, "v2x_gru_gates_f32: needs P, C > 0 and H * W %% 4 == 0");
    GruGateArgs a = {};
    a.gi = gi;
    a.bhh = bias_hh;
    a.h = h;
    a.P = P;
    a.C = C;
    a.HW4 = HW / 4;
    hipLaunchKernelGGL(gru_gates_kernel<false>, dim3(gg_grid(P * C * a.HW4)), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("gru_gates_kernel");
    return V2X_OK;
}

extern "C" int v2x_gru_gates_bwd_f32(const float *gi, const float *bias_hh, const float *dh, long long P, int C, int HW, float *dgi, float *dn_r,
                                     v2x_stream_t stream) {
    V2X_REQUIRE(gi && bias_hh && dh && dgi && dn_r, "v2x_gru_gates_bwd_f32: null pointer");
    V2X_REQUIRE(P > 0 && C > 0 && HW > 0 && HW % 4 == 0, "v2x_gru_gates_bwd_f32: needs P, C > 0 and H * W %% 4 == 0");
    GruGateArgs a = {};
    a.gi = gi;
    a.bhh = bias_hh;
    a.dh = dh;
    a.dgi = dgi;
    a.dn_r = dn_r;
    a.P = P;
    a.C = C;
    a.HW4 = HW / 4;
    hipLaunchKernelGGL(gru_gates_kernel<true>, dim3(gg_grid(P * C * a.HW4)), dim3(256), 0, (hipStream_t)stream, a);
    V2X_CHECK_LAUNCH("gru_gates_kernel<bwd>");
    return V2X_OK;
}
