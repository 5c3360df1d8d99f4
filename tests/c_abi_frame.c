/* A torch-free C99 host of the WHOLE hot path (VERDICT r4 item 5 / SURVEY.md section 8(b): "one entry per stage ... called from wrappers"):
 *
 *   points --v2x_voxelize_bits--> bit grid --v2x_conv2d_pair / v2x_conv2d x N (encoder)--> pyramid
 *          --v2x_warp_fuse (mean over neighbours) --v2x_conv2d (ConvGRU)--> fused level
 *          --v2x_conv2d x 7 (decoder; the four _1 layers in the parity-class layouts)--> v2x_conv2d_pair (conv8_2 + det heads, one launch) --> cls / loc logits
 *          --v2x_det_postprocess--> boxes, scores, anchor indices, counts
 *
 * driven from raw fp32 CHECKPOINT tensors (OIHW weights, BN statistics, biases) through the C packers only (v2x_fold_bn, v2x_pack_conv,
 * v2x_pack_chain_1x1, v2x_pack_gru_bias) -- the host logic of v2x_sim_amd/packing.py + ops.run_layer + models/det/V2VNet.py restated in C,
 * kernel selection included (same shape rules: the results are BIT-IDENTICAL to the Python host's, which tests/test_c_abi.py asserts, next to
 * the oracle comparison at the end-to-end tolerance).  No torch, no ctypes, no Python at run time:
 *
 *   c_abi_frame <in.bin> <out.bin>
 *
 * in.bin / out.bin: a flat list of named tensors (format below), written / read by tests/test_c_abi.py.
 * Upstream counterpart of the sequence: coperception/models/det/V2VNet.py::forward + utils/postprocess.py::apply_nms_det (not in
 * /root/reference; README.md:101). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "v2x_amd.h"

/* ---------------------------------------------------------------- tensor file: "V2XF", int32 count, then per tensor
 * char name[64]; int32 dtype (0 f32, 1 f64, 2 i32); int32 ndim; int32 dims[6]; raw little-endian data */
typedef struct {
    char name[64];
    int dtype, ndim, dims[6];
    size_t count;
    void *data;
} tensor;
static tensor g_in[256];
static int g_nin = 0;

#define CHECK(cond, ...)                  \
    do {                                  \
        if (!(cond)) {                    \
            fprintf(stderr, __VA_ARGS__); \
            fprintf(stderr, "\n");        \
            exit(1);                      \
        }                                 \
    } while (0)
#define HIPOK(call) CHECK((call) == hipSuccess, "HIP call failed: %s", #call)
#define V2XOK(call)                                                                \
    do {                                                                           \
        const int rc_ = (call);                                                    \
        CHECK(rc_ == V2X_OK, "%s -> %d: %s", #call, rc_, v2x_last_error());        \
    } while (0)

static size_t dsize(int dtype) { return dtype == 1 ? 8 : 4; }

static void read_file(const char *path) {
    FILE *f = fopen(path, "rb");
    CHECK(f, "cannot open %s", path);
    char magic[4];
    int32_t n;
    CHECK(fread(magic, 1, 4, f) == 4 && memcmp(magic, "V2XF", 4) == 0, "bad magic");
    CHECK(fread(&n, 4, 1, f) == 1 && n > 0 && n <= 256, "bad tensor count");
    for (int i = 0; i < n; ++i) {
        tensor *t = &g_in[g_nin++];
        int32_t hdr[8];
        CHECK(fread(t->name, 1, 64, f) == 64 && fread(hdr, 4, 8, f) == 8, "truncated header");
        t->dtype = hdr[0];
        t->ndim = hdr[1];
        t->count = 1;
        for (int d = 0; d < 6; ++d) {
            t->dims[d] = hdr[2 + d];
            if (d < t->ndim) t->count *= (size_t)t->dims[d];
        }
        t->data = malloc(t->count * dsize(t->dtype) + 8);
        CHECK(t->data && fread(t->data, dsize(t->dtype), t->count, f) == t->count, "truncated data of %s", t->name);
    }
    fclose(f);
}
static const tensor *find(const char *name) {
    for (int i = 0; i < g_nin; ++i)
        if (strcmp(g_in[i].name, name) == 0) return &g_in[i];
    CHECK(0, "tensor %s missing from the input file", name);
    return NULL;
}
static const float *fdata(const char *name) {
    const tensor *t = find(name);
    CHECK(t->dtype == 0, "%s is not fp32", name);
    return (const float *)t->data;
}
static const float *fparam(const char *prefix, const char *leaf) {
    char nm[128];
    snprintf(nm, sizeof nm, "%s%s", prefix, leaf);
    return fdata(nm);
}
static FILE *g_out;
static int g_nout = 0;
static void write_tensor(const char *name, int dtype, int ndim, const int *dims, const void *data) {
    char nm[64];
    int32_t hdr[8] = {dtype, ndim, 0, 0, 0, 0, 0, 0};
    size_t count = 1;
    memset(nm, 0, sizeof nm);
    strncpy(nm, name, 63);
    for (int d = 0; d < ndim; ++d) {
        hdr[2 + d] = dims[d];
        count *= (size_t)dims[d];
    }
    fwrite(nm, 1, 64, g_out);
    fwrite(hdr, 4, 8, g_out);
    fwrite(data, dsize(dtype), count, g_out);
    ++g_nout;
}

/* ---------------------------------------------------------------- device helpers */
static void *dalloc(size_t bytes) {
    void *p = NULL;
    HIPOK(hipMalloc(&p, bytes ? bytes : 16));
    return p;
}
static void *upload(const void *host, size_t bytes) {
    void *p = dalloc(bytes);
    HIPOK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    return p;
}
static int ceil_to(int x, int m) { return (x + m - 1) / m * m; }

/* ---------------------------------------------------------------- one packed convolution (what packing.PackedConv holds) */
typedef struct {
    int valid;
    int C0, C1, up0, Cout, ksize, stride, pad, relu, epilogue, w_layout, w_rows, w_kpad;
    uint16_t *weight;
    float *scale, *shift;
    int Cout2, relu2;
    uint16_t *weight2;
    float *scale2, *shift2;
} packed;

/* w fp32 [rows][cin][k][k] on the host -> device buffer in `layout`; scale / shift: host arrays of n_ss floats (GRU: float4 per hidden channel) */
static packed pack(const float *w, int cout, int cin, int ksize, int cin_pad, int layout, int epilogue, int chain, int c_up, const float *scale,
                   const float *shift, int n_ss, int C0, int C1, int up0, int stride, int relu) {
    packed p;
    memset(&p, 0, sizeof p);
    v2x_pack_spec spec = {cout, cin, ksize, cin_pad, layout, epilogue, chain, c_up};
    int32_t rows = 0, kpad = 0;
    const size_t bytes = v2x_pack_conv_size(&spec, &rows, &kpad);
    CHECK(bytes > 0, "v2x_pack_conv_size: %s", v2x_last_error());
    uint16_t *host = (uint16_t *)malloc(bytes);
    V2XOK(v2x_pack_conv(&spec, w, host));
    p.weight = (uint16_t *)upload(host, bytes);
    free(host);
    p.valid = 1;
    p.C0 = C0;
    p.C1 = C1;
    p.up0 = up0;
    p.Cout = cout;
    p.ksize = ksize;
    p.stride = stride;
    p.pad = (ksize - 1) / 2;
    p.relu = relu;
    p.epilogue = epilogue;
    p.w_layout = layout;
    p.w_rows = rows;
    p.w_kpad = kpad;
    p.scale = (float *)upload(scale, (size_t)n_ss * 4);
    p.shift = shift ? (float *)upload(shift, (size_t)n_ss * 4) : NULL;
    return p;
}
/* conv (+ eval BatchNorm) of the checkpoint under `prefix` ("u_encoder.conv1_1" + ".weight" / ".bias", BN under `bn` or NULL) */
typedef struct {
    const float *w, *b, *g, *beta, *mean, *var;
    int cout, cin, k;
} ckpt_conv;
static ckpt_conv get_conv(const char *conv, const char *bn) {
    ckpt_conv c;
    char nm[128];
    memset(&c, 0, sizeof c);
    snprintf(nm, sizeof nm, "%s.weight", conv);
    const tensor *t = find(nm);
    c.w = (const float *)t->data;
    c.cout = t->dims[0];
    c.cin = t->dims[1];
    c.k = t->dims[t->ndim - 1];
    c.b = fparam(conv, ".bias");
    if (bn) {
        c.g = fparam(bn, ".weight");
        c.beta = fparam(bn, ".bias");
        c.mean = fparam(bn, ".running_mean");
        c.var = fparam(bn, ".running_var");
    }
    return c;
}
static void folded(const ckpt_conv *c, int n_out, float *scale, float *shift) {
    V2XOK(v2x_fold_bn(c->cout, n_out, c->b, c->g, c->beta, c->mean, c->var, 1e-5f, scale, shift));
}
/* the gather-kernel packing every layer has (packing.pack_conv_bn) */
static packed pack_gather(const ckpt_conv *c, int cin_pad, int C0, int C1, int up0, int stride, int relu, int epilogue) {
    const int cinp = cin_pad ? cin_pad : c->cin;
    const int rows = ceil_to(c->cout, v2x_conv_tile_rows(c->cout, epilogue));
    float *sc = (float *)malloc((size_t)rows * 4), *sh = (float *)malloc((size_t)rows * 4);
    folded(c, rows, sc, sh);
    packed p = pack(c->w, c->cout, c->cin, c->k, cin_pad, 0, epilogue, 0, 0, sc, sh, rows, C0 ? C0 : cinp, C1, up0, stride, relu);
    free(sc);
    free(sh);
    return p;
}
/* the patch-kernel packing of packing.layer_conv_bn (same shape rules); .valid = 0 when the layer has none */
static packed pack_patch(const ckpt_conv *c, int cin_pad, int C0, int C1, int up0, int stride, int relu) {
    packed none;
    memset(&none, 0, sizeof none);
    const int cinp = cin_pad ? cin_pad : c->cin;
    if (c->k != 3) return none;
    float *sc = (float *)malloc((size_t)c->cout * 4), *sh = (float *)malloc((size_t)c->cout * 4);
    folded(c, c->cout, sc, sh);
    packed p = none;
    if (stride == 1 && c->cout % 32 == 0) {
        const int cin_h = ceil_to(cinp, 32);
        const int k0 = C1 ? C0 : 0, k1 = C1 ? C1 : cin_h;
        if (k0 == 64 && k1 == 32 && c->cout == 32 && up0 == 1)   /* conv8_1: parity-class form (w_layout 3) */
            p = pack(c->w, c->cout, c->cin, 3, 0, 3, V2X_EPI_BF16, 0, C0, sc, sh, c->cout, C0, C1, 1, 1, relu);
        else if ((k0 == 0 && k1 == 32 && c->cout == 32) || (k0 == 0 && k1 == 64 && (c->cout == 32 || c->cout == 64)))
            p = pack(c->w, c->cout, c->cin, 3, cin_h != c->cin ? cin_h : 0, 1, V2X_EPI_BF16, 0, 0, sc, sh, c->cout, cin_h, 0, 0, 1, relu);
        else if (C1 && up0 == 1 && C0 % 32 == 0 && C1 % 32 == 0 && (c->cout % 128 == 0 || c->cout == 64))   /* conv5_1 .. conv7_1: streamed parity-class forms (w_layout 4) */
            p = pack(c->w, c->cout, c->cin, 3, 0, 4, V2X_EPI_BF16, 0, C0, sc, sh, c->cout, C0, C1, 1, 1, relu);
        else if (cinp >= 64 && (C1 ? C0 : cinp) % 32 == 0 && C1 % 32 == 0 && c->cout % 64 == 0)
            p = pack(c->w, c->cout, c->cin, 3, 0, 2, V2X_EPI_BF16, 0, 0, sc, sh, c->cout, C1 ? C0 : cinp, C1, up0, 1, relu);
    } else if (stride == 2 && !C1 && cinp % 32 == 0 && c->cout % 64 == 0) {
        p = pack(c->w, c->cout, c->cin, 3, 0, 2, V2X_EPI_BF16, 0, 0, sc, sh, c->cout, cinp, 0, 0, 2, relu);
    }
    free(sc);
    free(sh);
    return p;
}
/* 3x3 conv + BN + ReLU chained with a 1x1 conv + BN + ReLU (conv1_2 -> conv3d_1 on the halo kernel, conv2_2 -> conv3d_2 on the streamed one) */
static packed pack_chain(const ckpt_conv *c, const ckpt_conv *c3, int layout) {
    float *sc = (float *)malloc((size_t)c->cout * 4), *sh = (float *)malloc((size_t)c->cout * 4);
    folded(c, c->cout, sc, sh);
    packed p = pack(c->w, c->cout, c->cin, 3, 0, layout, V2X_EPI_BF16, 1, 0, sc, sh, c->cout, c->cin, 0, 0, 1, 1);
    const int rows2 = ceil_to(c3->cout, 16);
    float *s2 = (float *)malloc((size_t)c3->cout * 4), *t2 = (float *)malloc((size_t)c3->cout * 4);
    folded(c3, c3->cout, s2, t2);
    uint16_t *w2 = (uint16_t *)malloc((size_t)rows2 * c->cout * 2);
    float *ds = (float *)malloc((size_t)rows2 * 4), *dt = (float *)malloc((size_t)rows2 * 4);
    V2XOK(v2x_pack_chain_1x1(c3->cout, c->cout, c3->w, s2, t2, w2, ds, dt));
    p.Cout2 = c3->cout;
    p.relu2 = 1;
    p.weight2 = (uint16_t *)upload(w2, (size_t)rows2 * c->cout * 2);
    p.scale2 = (float *)upload(ds, (size_t)rows2 * 4);
    p.shift2 = (float *)upload(dt, (size_t)rows2 * 4);
    free(sc); free(sh); free(s2); free(t2); free(w2); free(ds); free(dt);
    return p;
}

/* ---------------------------------------------------------------- launches */
static v2x_conv_desc desc_of(const packed *p, const void *in0, const void *in1, int N, int H, int W, void *out, int out_cstride, int split, void *out2,
                             int in_bits, int zbits) {
    v2x_conv_desc d;
    memset(&d, 0, sizeof d);
    d.in0 = (const uint16_t *)in0;
    d.in1 = (const uint16_t *)in1;
    d.C0 = p->C0; d.C1 = p->C1; d.up0 = p->up0;
    d.N = N; d.H = H; d.W = W;
    d.ksize = p->ksize; d.stride = p->stride; d.pad = p->pad;
    d.Cout = p->Cout; d.w_rows = p->w_rows; d.w_kpad = p->w_kpad;
    d.weight = p->weight; d.scale = p->scale; d.shift = p->shift;
    d.epilogue = p->epilogue; d.relu = p->relu;
    d.out = out; d.out_cstride = out_cstride; d.out_coff = 0;
    if (split) {
        d.out2 = out2;
        d.split = split;
        d.out2_cstride = (p->Cout2 ? p->Cout2 : p->Cout) - split;
        d.out_cstride = split;
    }
    d.w_layout = p->w_layout;
    if (p->Cout2) {
        d.Cout2 = p->Cout2; d.relu2 = p->relu2;
        d.weight2 = p->weight2; d.scale2 = p->scale2; d.shift2 = p->shift2;
    }
    if (in_bits) { d.in_format = 1; d.in_zbits = zbits; }
    return d;
}
static void conv(const packed *p, const void *in0, const void *in1, int N, int H, int W, void *out, int out_cstride, int split, void *out2,
                 int in_bits, int zbits) {
    const v2x_conv_desc d = desc_of(p, in0, in1, N, H, W, out, out_cstride, split, out2, in_bits, zbits);
    V2XOK(v2x_conv2d(&d, NULL));
}
static int halo_eligible(int H, int W, int layout, int cmax, int cout) {
    if (layout == 4) return H % 16 == 0 && W % (cout == 64 ? 64 : 32) == 0;
    if ((layout == 1 || layout == 3) && (long long)(10 * W + 34) * cmax >= (1 << 20)) return 0;
    if (H % 8 == 0 && W % 32 == 0) return 1;
    return layout == 2 && H % 16 == 0 && W % 16 == 0;
}
/* ops.run_layer for a bf16 input: the patch kernel when the extent allows, else the gather kernel(s).  fb2: second gather layer of a chained
 * pair (NULL otherwise).  Output extent = input extent / stride.  Returns the (malloc'ed) device output, channels = *cout. */
static uint16_t *run_layer(const packed *patch, const packed *fb, const packed *fb2, const void *in0, const void *in1, int N, int H, int W, int *cout) {
    const int s = fb->stride, Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
    const int cfin = fb2 ? fb2->Cout : fb->Cout;
    uint16_t *out = (uint16_t *)dalloc((size_t)N * Ho * Wo * cfin * 2);
    *cout = cfin;
    int use = 0;
    if (patch && patch->valid) {
        if (patch->stride == 2) use = (H % 8 == 0 && W % 64 == 0) || (H % 16 == 0 && W % 32 == 0);
        else {
            const int cmax = patch->C0 > patch->C1 ? patch->C0 : patch->C1;
            use = halo_eligible(H, W, patch->w_layout, cmax, patch->Cout) && ((patch->w_layout != 2 && patch->w_layout != 4) || H * W >= 256);
        }
    }
    if (use) {
        conv(patch, in0, in1, N, H, W, out, cfin, 0, NULL, 0, 0);
        return out;
    }
    if (!fb2) {
        conv(fb, in0, in1, N, H, W, out, cfin, 0, NULL, 0, 0);
        return out;
    }
    uint16_t *mid = (uint16_t *)dalloc((size_t)N * Ho * Wo * fb->Cout * 2);
    conv(fb, in0, in1, N, H, W, mid, fb->Cout, 0, NULL, 0, 0);
    conv(fb2, mid, NULL, N, Ho, Wo, out, cfin, 0, NULL, 0, 0);
    HIPOK(hipDeviceSynchronize());
    HIPOK(hipFree(mid));
    return out;
}

int main(int argc, char **argv) {
    CHECK(argc == 3, "usage: c_abi_frame <in.bin> <out.bin>");
    CHECK(v2x_abi_version() == V2X_AMD_ABI_VERSION, "ABI version: library %d, header %d", v2x_abi_version(), V2X_AMD_ABI_VERSION);
    read_file(argv[1]);
    const tensor *cfg = find("config");   /* i32: A, B (frames), X, Y, Z, max_pts, pt_stride, det_cap */
    CHECK(cfg->dtype == 2 && cfg->count >= 8, "config");
    const int32_t *ci = (const int32_t *)cfg->data;
    const int A = ci[0], B = ci[1], X = ci[2], Y = ci[3], Z = ci[4], max_pts = ci[5], pt_stride = ci[6], cap = ci[7];
    const int N = A * B;
    CHECK(A >= 2 && B >= 1 && X % 16 == 0 && Y % 16 == 0 && Z <= 16, "config values");

    /* ---- pack the checkpoint (host) ------------------------------------------------------------------------------------------ */
    char cn[128], bn[128];
    packed enc_fb[10], enc_pt[10];        /* pre_1, pre_2, 1_1, 1_2, 2_1, 2_2, 3_1, 3_2, 4_1, 4_2 */
    packed c3d_fb[2], chain[2];
    const char *enc_names[10] = {"conv_pre_1", "conv_pre_2", "conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2"};
    const char *enc_bn[10] = {"bn_pre_1", "bn_pre_2", "bn1_1", "bn1_2", "bn2_1", "bn2_2", "bn3_1", "bn3_2", "bn4_1", "bn4_2"};
    for (int i = 0; i < 10; ++i) {
        snprintf(cn, sizeof cn, "u_encoder.%s", enc_names[i]);
        snprintf(bn, sizeof bn, "u_encoder.%s", enc_bn[i]);
        const ckpt_conv c = get_conv(cn, bn);
        const int stride = (i >= 2 && i % 2 == 0) ? 2 : 1;
        const int cin_pad = i == 0 ? 32 : 0;
        enc_fb[i] = pack_gather(&c, cin_pad, 0, 0, 0, stride, 1, V2X_EPI_BF16);
        enc_pt[i] = pack_patch(&c, cin_pad, 0, 0, 0, stride, 1);
    }
    for (int l = 0; l < 2; ++l) {   /* the 1x1x1 "Conv3D" layers and their chained forms */
        snprintf(cn, sizeof cn, "u_encoder.conv3d_%d.conv3d", l + 1);
        snprintf(bn, sizeof bn, "u_encoder.conv3d_%d.bn3d", l + 1);
        const ckpt_conv c3 = get_conv(cn, bn);
        c3d_fb[l] = pack_gather(&c3, 0, 0, 0, 0, 1, 1, V2X_EPI_BF16);
        snprintf(cn, sizeof cn, "u_encoder.conv%d_2", l + 1);
        snprintf(bn, sizeof bn, "u_encoder.bn%d_2", l + 1);
        const ckpt_conv c2 = get_conv(cn, bn);
        chain[l] = pack_chain(&c2, &c3, l == 0 ? 1 : 2);   /* conv1_2 -> conv3d_1: halo ping-pong kernel; conv2_2 -> conv3d_2: streamed kernel */
    }
    packed dec_fb[8], dec_pt[8];
    const int dec_up[4] = {512, 256, 128, 64}, dec_skip[4] = {256, 128, 64, 32};
    for (int l = 0; l < 4; ++l)
        for (int j = 0; j < 2; ++j) {
            snprintf(cn, sizeof cn, "decoder.conv%d_%d", 5 + l, j + 1);
            snprintf(bn, sizeof bn, "decoder.bn%d_%d", 5 + l, j + 1);
            const ckpt_conv c = get_conv(cn, bn);
            const int C0 = j == 0 ? dec_up[l] : 0, C1 = j == 0 ? dec_skip[l] : 0, up0 = j == 0;
            dec_fb[2 * l + j] = pack_gather(&c, 0, C0, C1, up0, 1, 1, V2X_EPI_BF16);
            dec_pt[2 * l + j] = pack_patch(&c, 0, C0, C1, up0, 1, 1);
        }
    /* ConvGRU (h0 = 0: only W_ih is multiplied) */
    packed gru_fb, gru_pt;
    {
        const tensor *w = find("convgru.weight_ih_l0");
        const int hid = w->dims[0] / 3, cin = w->dims[1];
        float *b4 = (float *)malloc((size_t)hid * 16);
        V2XOK(v2x_pack_gru_bias(hid, fdata("convgru.bias_ih_l0"), fdata("convgru.bias_hh_l0"), b4));
        gru_fb = pack((const float *)w->data, hid, cin, 3, 0, 0, V2X_EPI_GRU, 0, 0, b4, NULL, hid * 4, cin / 2, cin / 2, 0, 1, 0);
        gru_pt = pack((const float *)w->data, hid, cin, 3, 0, 2, V2X_EPI_GRU, 0, 0, b4, NULL, hid * 4, cin / 2, cin / 2, 0, 1, 0);
        free(b4);
    }
    /* det heads: cls | reg hidden 3x3 (32 -> 64) + block-diagonal 1x1 (64 -> 12 + 36) */
    packed heads_halo, heads_hidden, heads_final;
    int ncls, nreg;
    {
        const ckpt_conv c1 = get_conv("classification.conv1", "classification.bn1"), r1 = get_conv("regression.box_prediction.0", "regression.box_prediction.1");
        const ckpt_conv c2 = get_conv("classification.conv2", NULL), r2 = get_conv("regression.box_prediction.3", NULL);
        ncls = c2.cout;
        nreg = r2.cout;
        const int hc = c1.cout, hr = r1.cout, hid = hc + hr, nout = ncls + nreg;
        float *w1 = (float *)malloc((size_t)hid * 32 * 9 * 4);
        memcpy(w1, c1.w, (size_t)hc * 32 * 9 * 4);
        memcpy(w1 + (size_t)hc * 32 * 9, r1.w, (size_t)hr * 32 * 9 * 4);
        float *sc = (float *)calloc(128, 4), *sh = (float *)calloc(128, 4);
        V2XOK(v2x_fold_bn(hc, hc, c1.b, c1.g, c1.beta, c1.mean, c1.var, 1e-5f, sc, sh));
        V2XOK(v2x_fold_bn(hr, hr, r1.b, r1.g, r1.beta, r1.mean, r1.var, 1e-5f, sc + hc, sh + hc));
        float *w2 = (float *)calloc((size_t)nout * hid, 4), *b2 = (float *)calloc(128, 4), *ones = (float *)calloc(128, 4);
        for (int r = 0; r < ncls; ++r) memcpy(w2 + (size_t)r * hid, c2.w + (size_t)r * hc, (size_t)hc * 4);
        for (int r = 0; r < nreg; ++r) memcpy(w2 + (size_t)(ncls + r) * hid + hc, r2.w + (size_t)r * hr, (size_t)hr * 4);
        memcpy(b2, c2.b, (size_t)ncls * 4);
        memcpy(b2 + ncls, r2.b, (size_t)nreg * 4);
        for (int i = 0; i < nout; ++i) ones[i] = 1.0f;
        /* gather pair: hidden (bf16), final 1x1 (fp32, split) */
        heads_hidden = pack(w1, hid, 32, 3, 0, 0, V2X_EPI_BF16, 0, 0, sc, sh, ceil_to(hid, v2x_conv_tile_rows(hid, V2X_EPI_BF16)), 32, 0, 0, 1, 1);
        heads_final = pack(w2, nout, hid, 1, 0, 0, V2X_EPI_F32, 0, 0, ones, b2, ceil_to(nout, v2x_conv_tile_rows(nout, V2X_EPI_F32)), hid, 0, 0, 1, 0);
        /* halo: hidden rows in chain order + chained 1x1, fp32 split output */
        heads_halo = pack(w1, hid, 32, 3, 0, 1, V2X_EPI_F32, 1, 0, sc, sh, hid, 32, 0, 0, 1, 1);
        const int rows2 = ceil_to(nout, 16);
        uint16_t *pw2 = (uint16_t *)malloc((size_t)rows2 * hid * 2);
        float *ds = (float *)malloc((size_t)rows2 * 4), *dt = (float *)malloc((size_t)rows2 * 4);
        V2XOK(v2x_pack_chain_1x1(nout, hid, w2, ones, b2, pw2, ds, dt));
        heads_halo.Cout2 = nout;
        heads_halo.relu2 = 0;
        heads_halo.weight2 = (uint16_t *)upload(pw2, (size_t)rows2 * hid * 2);
        heads_halo.scale2 = (float *)upload(ds, (size_t)rows2 * 4);
        heads_halo.shift2 = (float *)upload(dt, (size_t)rows2 * 4);
        heads_halo.valid = (hid == 64 && nout == 48);
        free(w1); free(sc); free(sh); free(w2); free(b2); free(ones); free(pw2); free(ds); free(dt);
    }

    /* ---- a1: voxel scatter ------------------------------------------------------------------------------------------------------ */
    const tensor *pts = find("points"), *npts = find("n_pts"), *ext = find("extents"), *vox = find("voxel");
    CHECK(pts->dtype == 0 && npts->dtype == 2 && ext->dtype == 1 && vox->dtype == 1, "input dtypes");
    float *d_pts = (float *)upload(pts->data, pts->count * 4);
    int32_t *d_npts = (int32_t *)upload(npts->data, npts->count * 4);
    uint32_t *d_bits = (uint32_t *)dalloc((size_t)N * X * Y * 4);
    const int32_t dims_xyz[3] = {X, Y, Z};
    V2XOK(v2x_voxelize_bits(d_pts, d_npts, N, max_pts, pt_stride, (const double *)ext->data, (const double *)vox->data, dims_xyz, d_bits, NULL));

    /* ---- a2: encoder ------------------------------------------------------------------------------------------------------------- */
    uint16_t *feat[5];
    int fc[5], fh[5], fw[5];
    {
        uint16_t *x;
        int c = 32;
        if (X % 8 == 0 && Y % 32 == 0 && enc_pt[0].valid && enc_pt[1].valid) {   /* conv_pre_1 -> conv_pre_2 in one launch from the bit grid */
            v2x_conv_desc da, db;
            memset(&da, 0, sizeof da);
            memset(&db, 0, sizeof db);
            const packed *pp[2] = {&enc_pt[0], &enc_pt[1]};
            v2x_conv_desc *dd[2] = {&da, &db};
            for (int k = 0; k < 2; ++k) {
                dd[k]->C0 = pp[k]->C0; dd[k]->N = N; dd[k]->H = X; dd[k]->W = Y;
                dd[k]->ksize = 3; dd[k]->stride = 1; dd[k]->pad = 1;
                dd[k]->Cout = pp[k]->Cout; dd[k]->w_rows = pp[k]->w_rows; dd[k]->w_kpad = pp[k]->w_kpad;
                dd[k]->weight = pp[k]->weight; dd[k]->scale = pp[k]->scale; dd[k]->shift = pp[k]->shift;
                dd[k]->epilogue = V2X_EPI_BF16; dd[k]->relu = 1; dd[k]->w_layout = 1;
            }
            da.in0 = (const uint16_t *)d_bits;
            da.in_format = 1;
            da.in_zbits = Z;
            x = (uint16_t *)dalloc((size_t)N * X * Y * 32 * 2);
            db.out = x;
            db.out_cstride = 32;
            V2XOK(v2x_conv2d_pair(&da, &db, NULL));
        } else {   /* odd extent: expand the bits, then the two layers on the gather kernel */
            uint16_t *x0 = (uint16_t *)dalloc((size_t)N * X * Y * 32 * 2);
            V2XOK(v2x_bits_to_nhwc_bf16(d_bits, N, X, Y, Z, 32, x0, NULL));
            uint16_t *x1 = run_layer(NULL, &enc_fb[0], NULL, x0, NULL, N, X, Y, &c);
            x = run_layer(NULL, &enc_fb[1], NULL, x1, NULL, N, X, Y, &c);
        }
        feat[0] = x; fc[0] = 32; fh[0] = X; fw[0] = Y;
        int H = X, W = Y;
        for (int l = 1; l <= 4; ++l) {
            uint16_t *y = run_layer(&enc_pt[2 * l], &enc_fb[2 * l], NULL, x, NULL, N, H, W, &c);   /* stride 2 */
            H /= 2;
            W /= 2;
            if (l <= 2) x = run_layer(&chain[l - 1], &enc_fb[2 * l + 1], &c3d_fb[l - 1], y, NULL, N, H, W, &c);
            else x = run_layer(&enc_pt[2 * l + 1], &enc_fb[2 * l + 1], NULL, y, NULL, N, H, W, &c);
            feat[l] = x; fc[l] = c; fh[l] = H; fw[l] = W;
        }
    }

    uint16_t *enc_feat[5];
    memcpy(enc_feat, feat, sizeof feat);
    /* ---- a3 + a4: warp + mean over the neighbours, ConvGRU (layer 3, every frame holds all A agents) ------------------------------ */
    {
        const int L = 3, H = fh[L], W = fw[L], C = fc[L];
        int32_t *items = (int32_t *)malloc((size_t)N * 8);
        float *coef = (float *)calloc((size_t)N * A, 4);
        for (int a = 0, m = 0; a < A; ++a)
            for (int f = 0; f < B; ++f, ++m) {
                items[2 * m] = a;
                items[2 * m + 1] = f;
                for (int j = 0; j < A; ++j) coef[m * A + j] = j == a ? 0.0f : 1.0f;
            }
        const tensor *tr = find("trans");
        CHECK(tr->dtype == 0 && tr->count == (size_t)B * A * A * 16, "trans shape");
        float *d_tr = (float *)upload(tr->data, tr->count * 4), *d_coef = (float *)upload(coef, (size_t)N * A * 4);
        int32_t *d_items = (int32_t *)upload(items, (size_t)N * 8);
        uint16_t *mean = (uint16_t *)dalloc((size_t)N * H * W * C * 2);
        V2XOK(v2x_warp_fuse(feat[L], A, B, H, W, C, d_tr, d_items, N, d_coef, V2X_FUSE_MEAN, mean, NULL));
        int c;
        feat[L] = run_layer(&gru_pt, &gru_fb, NULL, feat[L], mean, N, H, W, &c);   /* cat(ego, mean) through the two-source loader */
        free(items);
        free(coef);
    }

    /* ---- a6: decoder, a7: heads ------------------------------------------------------------------------------------------------------ */
    uint16_t *y = feat[4];
    int c = 0;
    /* conv8_2 and the heads go out as ONE launch when both have their halo packings and the extent allows (v2x_conv2d_pair's second form,
     * conv_tail.hip: conv8_2's output is never stored; bit-identical to the two launches) */
    const int tail = heads_halo.valid && dec_pt[7].valid && dec_pt[7].w_layout == 1 && dec_pt[7].C0 == 32 && dec_pt[7].Cout == 32 && halo_eligible(X, Y, 1, 32, 0) &&
                     (long long)N * X * Y < (1ll << 27) && (ncls == 4 || ncls == 8 || ncls == 12);
    for (int l = 0; l < 4; ++l) {
        const int H = fh[3 - l], W = fw[3 - l];
        y = run_layer(&dec_pt[2 * l], &dec_fb[2 * l], NULL, y, feat[3 - l], N, H, W, &c);
        if (l == 3 && tail) break;
        y = run_layer(&dec_pt[2 * l + 1], &dec_fb[2 * l + 1], NULL, y, NULL, N, H, W, &c);
    }
    float *d_cls = (float *)dalloc((size_t)N * X * Y * ncls * 4), *d_loc = (float *)dalloc((size_t)N * X * Y * nreg * 4);
    if (tail) {
        const v2x_conv_desc d8 = desc_of(&dec_pt[7], y, NULL, N, X, Y, NULL, 32, 0, NULL, 0, 0);
        const v2x_conv_desc dh = desc_of(&heads_halo, NULL, NULL, N, X, Y, d_cls, ncls, ncls, d_loc, 0, 0);
        V2XOK(v2x_conv2d_pair(&d8, &dh, NULL));
    } else if (heads_halo.valid && halo_eligible(X, Y, 1, 32, 0)) {
        conv(&heads_halo, y, NULL, N, X, Y, d_cls, ncls, ncls, d_loc, 0, 0);
    } else {
        uint16_t *hid = (uint16_t *)dalloc((size_t)N * X * Y * heads_hidden.Cout * 2);
        conv(&heads_hidden, y, NULL, N, X, Y, hid, heads_hidden.Cout, 0, NULL, 0, 0);
        conv(&heads_final, hid, NULL, N, X, Y, d_cls, ncls, ncls, d_loc, 0, 0);
    }

    /* ---- f-1: post-processing ---------------------------------------------------------------------------------------------------------- */
    const tensor *anc = find("anchors");
    const int M = X * Y * (ncls / 2);
    CHECK(anc->dtype == 0 && anc->count == (size_t)M * 6, "anchors shape");
    float *d_anc = (float *)upload(anc->data, anc->count * 4);
    float *d_boxes = (float *)dalloc((size_t)N * cap * 20), *d_scores = (float *)dalloc((size_t)N * cap * 4);
    int32_t *d_index = (int32_t *)dalloc((size_t)N * cap * 4), *d_count = (int32_t *)dalloc((size_t)N * 4), *d_cnt = (int32_t *)dalloc((size_t)N * 4);
    unsigned long long *d_keys = (unsigned long long *)dalloc((size_t)N * cap * 8);
    const float score_thr = fdata("thresholds")[0], nms_thr = fdata("thresholds")[1];
    V2XOK(v2x_det_postprocess(d_cls, d_loc, d_anc, N, M, score_thr, nms_thr, cap, d_boxes, d_scores, d_index, d_count, d_keys, d_cnt, NULL));
    HIPOK(hipDeviceSynchronize());

    /* ---- results -> out.bin ---------------------------------------------------------------------------------------------------------- */
    g_out = fopen(argv[2], "wb");
    CHECK(g_out, "cannot create %s", argv[2]);
    const int dump = getenv("V2X_FRAME_DUMP") != NULL;   /* + the intermediate maps (bf16 bit patterns as int32 words), for locating a divergence */
    const int32_t n_out = 7 + (dump ? 13 : 0);
    fwrite("V2XF", 1, 4, g_out);
    fwrite(&n_out, 4, 1, g_out);
#define DOWNLOAD_AND_WRITE(name, dptr, dtype, bytes, nd, ...)                 \
    do {                                                                      \
        void *h_ = malloc(bytes);                                             \
        const int dims_[] = {__VA_ARGS__};                                    \
        HIPOK(hipMemcpy(h_, dptr, bytes, hipMemcpyDeviceToHost));             \
        write_tensor(name, dtype, nd, dims_, h_);                             \
        free(h_);                                                             \
    } while (0)
    DOWNLOAD_AND_WRITE("bits", d_bits, 2, (size_t)N * X * Y * 4, 3, N, X, Y);
    DOWNLOAD_AND_WRITE("cls", d_cls, 0, (size_t)N * X * Y * ncls * 4, 4, N, X, Y, ncls);
    DOWNLOAD_AND_WRITE("loc", d_loc, 0, (size_t)N * X * Y * nreg * 4, 4, N, X, Y, nreg);
    DOWNLOAD_AND_WRITE("boxes", d_boxes, 0, (size_t)N * cap * 20, 3, N, cap, 5);
    DOWNLOAD_AND_WRITE("scores", d_scores, 0, (size_t)N * cap * 4, 2, N, cap);
    DOWNLOAD_AND_WRITE("index", d_index, 2, (size_t)N * cap * 4, 2, N, cap);
    DOWNLOAD_AND_WRITE("count", d_count, 2, (size_t)N * 4, 1, N);
    if (dump) {
        char nm[32];
        for (int l = 0; l < 5; ++l) {
            snprintf(nm, sizeof nm, "feat%d", l);
            DOWNLOAD_AND_WRITE(nm, enc_feat[l], 2, (size_t)N * fh[l] * fw[l] * fc[l] * 2, 4, N, fh[l], fw[l], fc[l] / 2);
        }
        DOWNLOAD_AND_WRITE("fused", feat[3], 2, (size_t)N * fh[3] * fw[3] * fc[3] * 2, 4, N, fh[3], fw[3], fc[3] / 2);
        DOWNLOAD_AND_WRITE("dec", y, 2, (size_t)N * X * Y * 32 * 2, 4, N, X, Y, 16);
        for (int k = 0; k < 2; ++k) {
            snprintf(nm, sizeof nm, "pre%d_scale", k + 1);
            DOWNLOAD_AND_WRITE(nm, enc_pt[k].scale, 0, 32 * 4, 1, 32);
            snprintf(nm, sizeof nm, "pre%d_shift", k + 1);
            DOWNLOAD_AND_WRITE(nm, enc_pt[k].shift, 0, 32 * 4, 1, 32);
            snprintf(nm, sizeof nm, "pre%d_w", k + 1);
            DOWNLOAD_AND_WRITE(nm, enc_pt[k].weight, 2, 288 * 32 * 2, 1, 288 * 16);
        }
    }
    fclose(g_out);
    CHECK(g_nout == n_out, "internal: wrote %d tensors", g_nout);
    printf("C ABI frame OK: %d agents x %d frame(s), %d x %d x %d grid, points -> logits -> detections through %s\n", A, B, X, Y, Z,
           "v2x_voxelize_bits, v2x_conv2d_pair, v2x_conv2d, v2x_warp_fuse, v2x_det_postprocess");
    return 0;
}
