// Host-side weight packing of libv2x_amd.so: OIHW fp32 parameters (what a PyTorch checkpoint holds) -> the bf16 device
// layouts v2x_conv2d's kernels consume (include/v2x_amd.h, "weight layouts").  Pure host code, no HIP call: a C, C++ or
// ctypes caller builds every `w_layout` buffer with these entry points, uploads it, and fills v2x_conv_desc from the
// sizes v2x_pack_conv_size reports.  v2x_sim_amd/packing.py produces the same bytes with torch ops;
// tests/test_pack_cpu.py holds the two to bit equality for every layout the models use.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

static inline uint16_t host_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline int ceil_to(int x, int m) { return (x + m - 1) / m * m; }

// packed row rho = 16*i + 4*q + r of a layer with a chained 1x1 computes hidden channel kappa (conv_halo.hip: the lane's
// accumulators of the first GEMM are then exactly its B fragment of the second)
static inline int chain_kappa(int rho) {
    const int i = rho >> 4, q = (rho >> 2) & 3, r = rho & 3;
    return 32 * (i >> 1) + 8 * q + 4 * (i & 1) + r;
}

struct PackGeom {
    int rows_src;   // rows of the source weight tensor (Cout, or 3*hidden for the GRU)
    int cin_p;      // input channels after zero padding
    int K;          // ksize*ksize*cin_p
    int w_rows, w_kpad;
    size_t elems;   // bf16 elements of the packed buffer
    int tile;       // layout 2: rows per channel tile
};

static int pack_geometry(const v2x_pack_spec *p, PackGeom *g) {
    V2X_REQUIRE(p, "v2x_pack_conv: null spec");
    V2X_REQUIRE(p->Cout > 0 && p->Cin > 0 && (p->ksize == 1 || p->ksize == 3), "v2x_pack_conv: Cout, Cin > 0 and ksize in {1, 3}");
    const bool gru = p->epilogue == V2X_EPI_GRU;
    g->cin_p = p->cin_pad > 0 ? p->cin_pad : p->Cin;
    V2X_REQUIRE(g->cin_p >= p->Cin && g->cin_p % 8 == 0, "v2x_pack_conv: cin_pad=%d must be >= Cin=%d and a multiple of 8", g->cin_p, p->Cin);
    V2X_REQUIRE(!gru || p->Cout % 32 == 0, "v2x_pack_conv: GRU hidden size must be a multiple of 32");
    V2X_REQUIRE(!(gru && p->chain), "v2x_pack_conv: the GRU epilogue has no chained layer");
    g->rows_src = gru ? 3 * p->Cout : p->Cout;
    g->K = p->ksize * p->ksize * g->cin_p;
    g->tile = 0;
    switch (p->w_layout) {
    case 0: {
        const int tile = v2x_conv_tile_rows(p->Cout, p->epilogue);
        V2X_REQUIRE(!p->chain, "v2x_pack_conv: layout 0 (gather kernel) has no chained epilogue");
        g->w_rows = gru ? p->Cout / 16 * 48 : ceil_to(p->Cout, tile);
        g->w_kpad = ceil_to(g->K, 64);
        g->elems = (size_t)g->w_rows * g->w_kpad;
        return V2X_OK;
    }
    case 1:
        V2X_REQUIRE(p->ksize == 3 && !gru && g->cin_p % 32 == 0 && p->Cout % 32 == 0,
                    "v2x_pack_conv: layout 1 (halo kernel) needs 3x3, Cin (padded) %% 32 == 0, Cout %% 32 == 0, no GRU");
        g->w_rows = p->Cout;
        g->w_kpad = g->K;
        g->elems = (size_t)g->K * p->Cout;
        return V2X_OK;
    case 2: {
        g->tile = v2x_conv_stream_tile_rows(p->Cout, p->epilogue);
        V2X_REQUIRE(p->ksize == 3 && g->tile > 0 && g->cin_p % 32 == 0,
                    "v2x_pack_conv: layout 2 (streamed kernel) needs 3x3, Cin %% 32 == 0 and Cout %% 64 == 0 (GRU: hidden %% 32 == 0)");
        V2X_REQUIRE(!p->chain || p->Cout == 64 || p->Cout == 128, "v2x_pack_conv: the streamed kernel chains only at Cout 64 / 128");
        g->w_rows = g->rows_src;
        g->w_kpad = g->K;
        g->elems = (size_t)g->rows_src * g->K + 32;   // + 64 B of zeros: the kernel's zero page
        return V2X_OK;
    }
    case 3: {
        const int c_up = p->c_up, c1 = g->cin_p - c_up;
        V2X_REQUIRE(p->ksize == 3 && !gru && !p->chain && g->cin_p == p->Cin && c_up > 0 && c1 > 0 && c_up % 32 == 0 && c1 % 32 == 0 && p->Cout % 32 == 0,
                    "v2x_pack_conv: layout 3 (parity-class halo kernel) needs 3x3, no padding, c_up and Cin - c_up multiples of 32 (both > 0), Cout %% 32 == 0");
        g->w_rows = p->Cout;
        g->w_kpad = 16 * c_up + 9 * c1;
        g->elems = (size_t)g->w_kpad * p->Cout;
        return V2X_OK;
    }
    case 4: {
        const int c_up = p->c_up, c1 = g->cin_p - c_up;
        V2X_REQUIRE(p->ksize == 3 && !gru && !p->chain && g->cin_p == p->Cin && c_up > 0 && c1 > 0 && c_up % 32 == 0 && c1 % 32 == 0 && (p->Cout % 128 == 0 || p->Cout == 64),
                    "v2x_pack_conv: layout 4 (streamed parity-class kernel) needs 3x3, no padding, c_up and Cin - c_up multiples of 32 (both > 0), Cout %% 128 == 0 or Cout == 64");
        g->w_rows = p->Cout;
        g->w_kpad = 16 * c_up + 9 * c1;
        g->tile = p->Cout == 64 ? 64 : 128;
        g->elems = (size_t)g->w_kpad * p->Cout + 32;   // + 64 B of zeros: the kernel's zero page
        return V2X_OK;
    }
    default:
        v2x_set_error("v2x_pack_conv: w_layout=%d unknown", p->w_layout);
        return V2X_EINVAL;
    }
}

extern "C" size_t v2x_pack_conv_size(const v2x_pack_spec *spec, int32_t *w_rows, int32_t *w_kpad) {
    PackGeom g;
    if (pack_geometry(spec, &g) != V2X_OK) return 0;
    if (w_rows) *w_rows = g.w_rows;
    if (w_kpad) *w_kpad = g.w_kpad;
    return g.elems * sizeof(uint16_t);
}

extern "C" int v2x_pack_conv(const v2x_pack_spec *p, const float *w_oihw, uint16_t *dst) {
    PackGeom g;
    const int rc = pack_geometry(p, &g);
    if (rc != V2X_OK) return rc;
    V2X_REQUIRE(w_oihw && dst, "v2x_pack_conv: null pointer");
    V2X_REQUIRE(p->src_rows == 0 && p->src_row0 == 0, "v2x_pack_conv: row slices (src_rows / src_row0) exist for the device packer's transform = 1 only");
    const int ks = p->ksize, taps = ks * ks, cin = p->Cin, cin_p = g.cin_p, K = g.K;
    const bool gru = p->epilogue == V2X_EPI_GRU;
    if (p->w_layout == 4) {   // streamed parity-class form: per 128-row tile [up chunk][class tap][class][k-slot][row][8], then [skip chunk][kx][ky][k-slot][row][8]
#pragma clang fp contract(off)
        static const int G[2][2][2] = {{{0, 0}, {1, 2}}, {{0, 1}, {2, 2}}};
        const int c_up = p->c_up, c1 = cin - c_up, cout = p->Cout, T = g.tile;   // rows per channel tile: 128, or 64 for the 64-channel pair kernel
        size_t o = 0;
        for (int tl = 0; tl < cout / T; ++tl) {
            for (int kc = 0; kc < c_up / 32; ++kc)
                for (int t = 0; t < 4; ++t)
                    for (int cls = 0; cls < 4; ++cls) {
                        const int py = cls >> 1, px = cls & 1, a = t >> 1, b = t & 1;
                        for (int slot = 0; slot < 4; ++slot)
                            for (int r = 0; r < T; ++r)
                                for (int j = 0; j < 8; ++j, ++o) {
                                    const float *ws = w_oihw + ((size_t)(tl * T + r) * cin + kc * 32 + slot * 8 + j) * 9;
                                    float acc = 0.0f;
                                    bool first = true;
                                    for (int ky = G[py][a][0]; ky <= G[py][a][1]; ++ky)
                                        for (int kx = G[px][b][0]; kx <= G[px][b][1]; ++kx) {
                                            acc = first ? ws[ky * 3 + kx] : acc + ws[ky * 3 + kx];
                                            first = false;
                                        }
                                    dst[o] = host_bf16_rne(acc);
                                }
                    }
            for (int kc = 0; kc < c1 / 32; ++kc)
                for (int kx = 0; kx < 3; ++kx)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int slot = 0; slot < 4; ++slot)
                            for (int r = 0; r < T; ++r)
                                for (int j = 0; j < 8; ++j, ++o)
                                    dst[o] = host_bf16_rne(w_oihw[((size_t)(tl * T + r) * cin + c_up + kc * 32 + slot * 8 + j) * 9 + ky * 3 + kx]);
        }
        for (int j = 0; j < 32; ++j) dst[o + j] = 0;
        return V2X_OK;
    }
    if (p->w_layout == 3) {   // parity-class form: pre-summed 2x2-tap weights for the upsampled source (fp32 sums, (ky, kx) ascending, ONE rounding)
#pragma clang fp contract(off)
        static const int G[2][2][2] = {{{0, 0}, {1, 2}}, {{0, 1}, {2, 2}}};   // [parity][class tap] -> first and last 3x3 tap
        const int c_up = p->c_up, c1 = cin - c_up, cout = p->Cout;
        size_t o = 0;
        for (int cls = 0; cls < 4; ++cls)
            for (int t = 0; t < 4; ++t) {
                const int py = cls >> 1, px = cls & 1, a = t >> 1, b = t & 1;
                for (int s = 0; s < c_up / 8; ++s)
                    for (int co = 0; co < cout; ++co)
                        for (int j = 0; j < 8; ++j, ++o) {
                            const float *ws = w_oihw + ((size_t)co * cin + s * 8 + j) * 9;
                            float acc = 0.0f;
                            bool first = true;
                            for (int ky = G[py][a][0]; ky <= G[py][a][1]; ++ky)
                                for (int kx = G[px][b][0]; kx <= G[px][b][1]; ++kx) {
                                    acc = first ? ws[ky * 3 + kx] : acc + ws[ky * 3 + kx];
                                    first = false;
                                }
                            dst[o] = host_bf16_rne(acc);
                        }
            }
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < c1 / 8; ++s)
                for (int co = 0; co < cout; ++co)
                    for (int j = 0; j < 8; ++j, ++o) dst[o] = host_bf16_rne(w_oihw[((size_t)co * cin + c_up + s * 8 + j) * 9 + tap]);
        return V2X_OK;
    }
    const int hid = p->Cout;
    // wk[row][k], k = (ky*ks + kx)*cin_p + c, rows in the order the kernel wants them
    std::vector<uint16_t> wk((size_t)g.rows_src * K, 0);
    for (int row = 0; row < g.rows_src; ++row) {
        int src = row;
        if (gru) {  // packed row grp*48 + gate*16 + e  <-  gate row gate*hid + grp*16 + e  ((r, z, n) triples per 16 channels)
            const int grp = row / 48, gate = (row % 48) / 16, e = row % 16;
            src = gate * hid + grp * 16 + e;
        } else if (p->chain) {
            src = chain_kappa(row);
        }
        const float *ws = w_oihw + (size_t)src * cin * taps;
        uint16_t *wd = wk.data() + (size_t)row * K;
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t) wd[t * cin_p + c] = host_bf16_rne(ws[c * taps + t]);
    }
    memset(dst, 0, g.elems * sizeof(uint16_t));
    if (p->w_layout == 0) {   // [w_rows][w_kpad] row-major, zero padded
        for (int row = 0; row < g.rows_src; ++row) memcpy(dst + (size_t)row * g.w_kpad, wk.data() + (size_t)row * K, (size_t)K * 2);
    } else if (p->w_layout == 1) {   // k-slot-major [K/8][Cout][8]
        for (int s = 0; s < K / 8; ++s)
            for (int co = 0; co < p->Cout; ++co) memcpy(dst + ((size_t)s * p->Cout + co) * 8, wk.data() + (size_t)co * K + s * 8, 16);
    } else {   // streamed slices [co_tile][chunk][tap][slot][row][8], k = tap*cin + chunk*32 + slot*8 + j
        const int n_tiles = g.rows_src / g.tile, n_chunks = cin_p / 32;
        size_t o = 0;
        for (int t = 0; t < n_tiles; ++t)
            for (int ch = 0; ch < n_chunks; ++ch)
                for (int tap = 0; tap < 9; ++tap)
                    for (int slot = 0; slot < 4; ++slot)
                        for (int r = 0; r < g.tile; ++r, o += 8)
                            memcpy(dst + o, wk.data() + (size_t)(t * g.tile + r) * K + tap * cin_p + ch * 32 + slot * 8, 16);
    }
    return V2X_OK;
}

// ---- the same packing ON THE DEVICE (row f-3: training re-packs every layer after every optimizer step) ------------------------
// One launch per layer: a thread builds one 16-byte group (8 consecutive k of one packed row) of the destination straight from the fp32
// OIHW parameter -- the torch-op form of this (permute, pad, reshape, cast, scatter into a zeroed buffer) was ~10 small kernels per
// layer and packing: 336 fills and 181 copies per FaFNet training step.  transform = 1 builds the DATA-GRADIENT layer of a convolution:
// `spec` then describes that layer (Cout = the convolution's Cin, Cin = its Cout) and W'[o][c][ky][kx] = W[c][o][2-ky][2-kx] is read
// from the convolution's own weight tensor.  Plain layers only (no GRU row regrouping, no chain order).
// the geometry of one device packing IS the public job struct (include/v2x_amd.h: v2x_pack_job); w / dst / block_begin are used by the batched form
typedef v2x_pack_job DevPackGeom;

// 16-byte group `grp` of one packing
__device__ __forceinline__ void pack_group(const float *__restrict__ w, uint16_t *__restrict__ dst, const DevPackGeom &g, long long grp) {
    int row = -1, k0 = 0;
    if (g.layout == 0) {
        const int per = g.w_kpad / 8;
        row = (int)(grp / per);
        k0 = (int)(grp - (long long)row * per) * 8;
        if (row >= g.rows_src || k0 >= g.K) row = -1;
    } else if (g.layout == 1) {
        const int s = (int)(grp / g.cout);
        row = (int)(grp - (long long)s * g.cout);
        k0 = 8 * s;
    } else if (grp < g.data_groups) {
        long long o = grp;
        const int r = (int)(o % g.tile);
        o /= g.tile;
        const int slot = (int)(o & 3);
        o >>= 2;
        const int tap = (int)(o % 9);
        o /= 9;
        const int n_chunks = g.cin_p / 32;
        const int ch = (int)(o % n_chunks);
        const int t = (int)(o / n_chunks);
        row = t * g.tile + r;
        k0 = tap * g.cin_p + ch * 32 + slot * 8;
    }
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row >= 0) {
        const int tap = k0 / g.cin_p, c0 = k0 - tap * g.cin_p;   // cin_p % 8 == 0: the 8 k of a group share their tap
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            float x = 0.0f;
            if (c < g.cin)
                x = g.transform ? w[((size_t)c * g.src_stride + row) * g.taps + (g.taps - 1 - tap)] : w[((size_t)row * g.cin + c) * g.taps + tap];
            f[j] = x;
        }
        v = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
    *reinterpret_cast<uint4 *>(dst + grp * 8) = v;
}

__global__ __launch_bounds__(256) void pack_conv_device_kernel(const float *__restrict__ w, uint16_t *__restrict__ dst, const DevPackGeom g) {
    for (long long grp = (long long)blockIdx.x * 256 + threadIdx.x; grp < g.groups; grp += (long long)gridDim.x * 256) pack_group(w, dst, g, grp);
}

// many packings, one launch: workgroup b belongs to the job with the largest block_begin <= b (jobs sorted by block_begin)
__global__ __launch_bounds__(256) void pack_conv_device_batch_kernel(const v2x_pack_job *__restrict__ jobs, int n_jobs) {
    __shared__ v2x_pack_job job;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].block_begin <= (long long)blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        job = jobs[lo];
    }
    __syncthreads();
    const long long grp = ((long long)blockIdx.x - job.block_begin) * 256 + threadIdx.x;
    if (grp < job.groups) pack_group(job.w, job.dst, job, grp);
}

static int fill_pack_job(const v2x_pack_spec *p, const float *w_oihw_dev, int transform, uint16_t *dst_dev, DevPackGeom &d, const char *who) {
    PackGeom g;
    const int rc = pack_geometry(p, &g);
    if (rc != V2X_OK) return rc;
    V2X_REQUIRE(w_oihw_dev && dst_dev, "%s: null pointer", who);
    V2X_REQUIRE(p->epilogue != V2X_EPI_GRU && !p->chain && p->w_layout != 3 && p->w_layout != 4, "%s: plain layers only (no GRU regrouping, no chain order, no parity-class sums)", who);
    V2X_REQUIRE(transform == 0 || transform == 1, "%s: transform must be 0 or 1", who);
    V2X_REQUIRE(g.elems % 8 == 0, "%s: internal: destination not a whole number of 16-byte groups", who);
    V2X_REQUIRE(p->src_rows == 0 || (transform == 1 && p->src_row0 >= 0 && p->src_row0 + g.rows_src <= p->src_rows),
                "%s: a row slice (src_rows = %d, src_row0 = %d) needs transform = 1 and src_row0 + %d <= src_rows", who, p->src_rows, p->src_row0, g.rows_src);
    d.w = w_oihw_dev + (p->src_rows ? (size_t)p->src_row0 * (p->ksize * p->ksize) : 0);
    d.dst = dst_dev;
    d.src_stride = p->src_rows ? p->src_rows : g.rows_src;
    d.reserved0 = 0;
    d.rows_src = g.rows_src;
    d.cin = p->Cin;
    d.cin_p = g.cin_p;
    d.taps = p->ksize * p->ksize;
    d.K = g.K;
    d.w_kpad = g.w_kpad;
    d.tile = g.tile;
    d.cout = p->Cout;
    d.layout = p->w_layout;
    d.transform = transform;
    d.groups = (long long)(g.elems / 8);
    d.data_groups = p->w_layout == 2 ? (long long)g.rows_src * g.K / 8 : d.groups;
    d.block_begin = 0;
    return V2X_OK;
}

extern "C" int v2x_pack_conv_device(const v2x_pack_spec *p, const float *w_oihw_dev, int transform, uint16_t *dst_dev, v2x_stream_t stream) {
    DevPackGeom d;
    const int rc = fill_pack_job(p, w_oihw_dev, transform, dst_dev, d, "v2x_pack_conv_device");
    if (rc != V2X_OK) return rc;
    const long long blocks = (d.groups + 255) / 256;
    hipLaunchKernelGGL(pack_conv_device_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, d.w, d.dst, d);   // (d.w: the source with a row slice's offset applied)
    V2X_CHECK_LAUNCH("pack_conv_device_kernel");
    return V2X_OK;
}

extern "C" int v2x_pack_conv_device_job(const v2x_pack_spec *p, const float *w_oihw_dev, int transform, uint16_t *dst_dev, int64_t block_begin,
                                        v2x_pack_job *job_host, int64_t *n_blocks) {
    V2X_REQUIRE(job_host && n_blocks && block_begin >= 0, "v2x_pack_conv_device_job: bad arguments");
    const int rc = fill_pack_job(p, w_oihw_dev, transform, dst_dev, *job_host, "v2x_pack_conv_device_job");
    if (rc != V2X_OK) return rc;
    job_host->block_begin = block_begin;
    *n_blocks = (job_host->groups + 255) / 256;
    return V2X_OK;
}

extern "C" int v2x_pack_conv_device_batch(const v2x_pack_job *jobs_dev, int32_t n_jobs, int64_t total_blocks, v2x_stream_t stream) {
    V2X_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "v2x_pack_conv_device_batch: bad arguments");
    hipLaunchKernelGGL(pack_conv_device_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, (int)n_jobs);
    V2X_CHECK_LAUNCH("pack_conv_device_batch_kernel");
    return V2X_OK;
}

extern "C" int v2x_pack_chain_1x1(int Cout2, int Cout, const float *w2, const float *scale2, const float *shift2,
                                  uint16_t *dst_w, float *dst_scale, float *dst_shift) {
    V2X_REQUIRE(Cout2 > 0 && Cout > 0 && w2 && dst_w, "v2x_pack_chain_1x1: bad arguments");
    const int rows = ceil_to(Cout2, 16);
    memset(dst_w, 0, (size_t)rows * Cout * 2);
    for (int r = 0; r < Cout2; ++r)
        for (int c = 0; c < Cout; ++c) dst_w[(size_t)r * Cout + c] = host_bf16_rne(w2[(size_t)r * Cout + c]);
    for (int r = 0; r < rows; ++r) {
        if (dst_scale) dst_scale[r] = (r < Cout2 && scale2) ? scale2[r] : (r < Cout2 ? 1.0f : 0.0f);
        if (dst_shift) dst_shift[r] = (r < Cout2 && shift2) ? shift2[r] : 0.0f;
    }
    return V2X_OK;
}

extern "C" int v2x_pack_gru_bias(int hidden, const float *bias_ih, const float *bias_hh, float *dst) {
#pragma clang fp contract(off)
    V2X_REQUIRE(hidden > 0 && bias_ih && bias_hh && dst, "v2x_pack_gru_bias: bad arguments");
    for (int c = 0; c < hidden; ++c) {
        dst[4 * c + 0] = bias_ih[c] + bias_hh[c];
        dst[4 * c + 1] = bias_ih[hidden + c] + bias_hh[hidden + c];
        dst[4 * c + 2] = bias_ih[2 * hidden + c];
        dst[4 * c + 3] = bias_hh[2 * hidden + c];
    }
    return V2X_OK;
}

extern "C" int v2x_fold_bn(int C, int n_out, const float *conv_bias, const float *gamma, const float *beta, const float *mean,
                           const float *var, float eps, float *scale, float *shift) {
#pragma clang fp contract(off)   // torch rounds the product and the sum separately: no FMA here, or the last bit differs
    V2X_REQUIRE(C > 0 && n_out >= C && scale && shift, "v2x_fold_bn: bad arguments");
    const bool bn = gamma && beta && mean && var;
    for (int c = 0; c < n_out; ++c) {
        if (c >= C) {
            scale[c] = shift[c] = 0.0f;
            continue;
        }
        const float cb = conv_bias ? conv_bias[c] : 0.0f;
        if (!bn) {
            scale[c] = 1.0f;
            shift[c] = cb;
        } else {   // the order of torch's fp32 ops in packing.fold_bn: g / sqrt(var + eps);  b + s * (cb - mu)
            const float s = gamma[c] / sqrtf(var[c] + eps);
            scale[c] = s;
            shift[c] = beta[c] + s * (cb - mean[c]);
        }
    }
    return V2X_OK;
}
