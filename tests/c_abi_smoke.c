/* A non-Python caller of the C ABI: plain C99, no torch, no ctypes.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi_smoke.c \
 *       -Lv2x-sim_amd/v2x_sim_amd/lib -lv2x_amd -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_smoke
 *   ./c_abi_smoke --pack-only      host part only (no GPU): packs one layer into w_layout 0 / 1 / 2 and checks every element
 *                                  against the index formulas documented in include/v2x_amd.h ("weight layouts")
 *   ./c_abi_smoke                  + uploads them, runs v2x_conv2d through the gather, halo and streamed kernels and
 *                                  compares the three results with each other (<= 1 bf16 ulp: the K walks differ) and with a
 *                                  float loop on the host; then v2x_bn_train_forward / _backward on that output against a
 *                                  double loop on the host (statistics, running averages, y, dgamma, dbeta, dx)
 *
 * The layer: 3x3 stride-1 conv 64 -> 64 + folded BN + ReLU on 2 x 32 x 64 NHWC bf16 maps (all three kernels cover it).
 * Driven by tests/test_c_abi.py (build + --pack-only on the CPU box, full run under -m gpu). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "v2x_amd.h"

#define NB 2 /* maps */
#define HH 32
#define WW 64
#define CIN 64
#define COUT 64
#define K (9 * CIN)

static uint32_t rng_state = 12345u;
static float frand(void) { /* xorshift in [-1, 1) */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return (float)(rng_state & 0xffffff) / 8388608.0f - 1.0f;
}
static uint16_t to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float from_bf16(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CHECK(cond, ...)                  \
    do {                                  \
        if (!(cond)) {                    \
            fprintf(stderr, __VA_ARGS__); \
            fprintf(stderr, "\n");        \
            return 1;                     \
        }                                 \
    } while (0)
#define HIPOK(call) CHECK((call) == hipSuccess, "HIP call failed: %s", #call)

int main(int argc, char **argv) {
    const int pack_only = argc > 1 && strcmp(argv[1], "--pack-only") == 0;
    CHECK(v2x_abi_version() == V2X_AMD_ABI_VERSION, "ABI version: library %d, header %d", v2x_abi_version(), V2X_AMD_ABI_VERSION);

    /* ---- parameters as a checkpoint holds them: OIHW fp32 weight, conv bias, BN gamma/beta/mean/var */
    float *w = (float *)malloc(sizeof(float) * COUT * CIN * 9);
    float bias[COUT], gamma[COUT], beta[COUT], mean[COUT], var[COUT], scale[COUT], shift[COUT];
    for (int i = 0; i < COUT * CIN * 9; ++i) w[i] = 0.06f * frand();
    for (int c = 0; c < COUT; ++c) {
        bias[c] = 0.1f * frand();
        gamma[c] = 1.0f + 0.3f * frand();
        beta[c] = 0.2f * frand();
        mean[c] = 0.1f * frand();
        var[c] = 0.5f + 0.4f * (frand() + 1.0f);
    }
    CHECK(v2x_fold_bn(COUT, COUT, bias, gamma, beta, mean, var, 1e-5f, scale, shift) == V2X_OK, "fold_bn: %s", v2x_last_error());

    /* ---- pack into the three layouts */
    uint16_t *packed[3];
    size_t bytes[3];
    int32_t rows[3], kpad[3];
    for (int layout = 0; layout < 3; ++layout) {
        v2x_pack_spec spec = {COUT, CIN, 3, 0, layout, V2X_EPI_BF16, 0, 0};
        bytes[layout] = v2x_pack_conv_size(&spec, &rows[layout], &kpad[layout]);
        CHECK(bytes[layout] > 0, "pack_conv_size(layout %d): %s", layout, v2x_last_error());
        packed[layout] = (uint16_t *)malloc(bytes[layout]);
        CHECK(v2x_pack_conv(&spec, w, packed[layout]) == V2X_OK, "pack_conv(layout %d): %s", layout, v2x_last_error());
    }
    CHECK(rows[0] == 64 && kpad[0] == 576 && rows[1] == 64 && kpad[1] == K && rows[2] == 64 && kpad[2] == K, "unexpected w_rows / w_kpad");
    CHECK(bytes[2] == (size_t)COUT * K * 2 + 64, "layout 2 must end with a 64-byte zero page");
    /* every element, through the formulas of the header:  kk = (ky*3 + kx)*Cin + c */
    const int tile = v2x_conv_stream_tile_rows(COUT, V2X_EPI_BF16);
    CHECK(tile == 64, "stream tile rows %d", tile);
    for (int co = 0; co < COUT; ++co)
        for (int t = 0; t < 9; ++t)
            for (int c = 0; c < CIN; ++c) {
                const int kk = t * CIN + c;
                const uint16_t want = to_bf16(w[(co * CIN + c) * 9 + t]);
                const uint16_t l0 = packed[0][(size_t)co * kpad[0] + kk];
                const uint16_t l1 = packed[1][((size_t)(kk / 8) * COUT + co) * 8 + kk % 8];
                const int ch = c / 32, slot = (c % 32) / 8, j = c % 8, tl = co / tile, r = co % tile;
                const uint16_t l2 = packed[2][(((((size_t)tl * (CIN / 32) + ch) * 9 + t) * 4 + slot) * tile + r) * 8 + j];
                CHECK(l0 == want && l1 == want && l2 == want, "layout mismatch at co=%d tap=%d c=%d: %04x %04x %04x want %04x", co, t, c, l0, l1, l2, want);
            }
    for (int i = 0; i < 32; ++i) CHECK(packed[2][(size_t)COUT * K + i] == 0, "zero page not zero");
    printf("pack OK: layouts 0 / 1 / 2 agree with the header's index formulas (%zu / %zu / %zu bytes)\n", bytes[0], bytes[1], bytes[2]);
    if (pack_only) return 0;

    /* ---- device run */
    const size_t n_in = (size_t)NB * HH * WW * CIN, n_out = (size_t)NB * HH * WW * COUT;
    uint16_t *x = (uint16_t *)malloc(n_in * 2);
    for (size_t i = 0; i < n_in; ++i) x[i] = to_bf16(frand());
    void *d_x, *d_scale, *d_shift, *d_w[3], *d_y[3];
    HIPOK(hipMalloc(&d_x, n_in * 2));
    HIPOK(hipMalloc(&d_scale, sizeof(scale)));
    HIPOK(hipMalloc(&d_shift, sizeof(shift)));
    HIPOK(hipMemcpy(d_x, x, n_in * 2, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_scale, scale, sizeof(scale), hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_shift, shift, sizeof(shift), hipMemcpyHostToDevice));
    hipStream_t stream;
    HIPOK(hipStreamCreate(&stream));
    uint16_t *y[3];
    for (int layout = 0; layout < 3; ++layout) {
        HIPOK(hipMalloc(&d_w[layout], bytes[layout]));
        HIPOK(hipMalloc(&d_y[layout], n_out * 2));
        HIPOK(hipMemcpy(d_w[layout], packed[layout], bytes[layout], hipMemcpyHostToDevice));
        HIPOK(hipMemset(d_y[layout], 0xff, n_out * 2));
        v2x_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.in0 = (const uint16_t *)d_x;
        d.C0 = CIN;
        d.N = NB, d.H = HH, d.W = WW;
        d.ksize = 3, d.stride = 1, d.pad = 1;
        d.Cout = COUT, d.w_rows = rows[layout], d.w_kpad = kpad[layout];
        d.weight = (const uint16_t *)d_w[layout];
        d.scale = (const float *)d_scale, d.shift = (const float *)d_shift;
        d.epilogue = V2X_EPI_BF16, d.relu = 1;
        d.out = d_y[layout], d.out_cstride = COUT;
        d.w_layout = layout;
        CHECK(v2x_conv2d(&d, stream) == V2X_OK, "v2x_conv2d(layout %d): %s", layout, v2x_last_error());
        HIPOK(hipStreamSynchronize(stream));
        y[layout] = (uint16_t *)malloc(n_out * 2);
        HIPOK(hipMemcpy(y[layout], d_y[layout], n_out * 2, hipMemcpyDeviceToHost));
    }
    /* host reference: fp32 accumulation over the same bf16 operands, then scale/shift, ReLU */
    double worst[3] = {0, 0, 0};
    size_t ulp_diff_12 = 0, diff_01 = 0, ulp_diff_01 = 0;
    for (int n = 0; n < NB; ++n)
        for (int yy = 0; yy < HH; ++yy)
            for (int xx = 0; xx < WW; ++xx)
                for (int co = 0; co < COUT; ++co) {
                    float acc = 0.0f;
                    for (int t = 0; t < 9; ++t) {
                        const int iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
                        if (iy < 0 || iy >= HH || ix < 0 || ix >= WW) continue;
                        const uint16_t *px = x + (((size_t)n * HH + iy) * WW + ix) * CIN;
                        const uint16_t *wr = packed[0] + (size_t)co * kpad[0] + t * CIN;
                        for (int c = 0; c < CIN; ++c) acc += from_bf16(px[c]) * from_bf16(wr[c]);
                    }
                    float ref = acc * scale[co] + shift[co];
                    if (ref < 0.0f) ref = 0.0f;
                    const size_t o = (((size_t)n * HH + yy) * WW + xx) * COUT + co;
                    for (int l = 0; l < 3; ++l) {
                        const double e = fabs((double)from_bf16(y[l][o]) - ref) / (fabs(ref) + 1e-2);
                        if (e > worst[l]) worst[l] = e;
                    }
                    diff_01 += y[0][o] != y[1][o];
                    /* "one bf16 rounding apart": |a - b| <= 2^-7 * max(|a|, |b|) + 1e-3 (the absolute term covers outputs that ReLU
                     * clamps to 0 in one kernel and leaves at +1e-5 in the other) */
                    const float v0 = from_bf16(y[0][o]), v1 = from_bf16(y[1][o]), v2 = from_bf16(y[2][o]);
                    ulp_diff_01 += fabsf(v0 - v1) > 0.0078125f * fmaxf(fabsf(v0), fabsf(v1)) + 1e-3f;
                    ulp_diff_12 += fabsf(v1 - v2) > 0.0078125f * fmaxf(fabsf(v1), fabsf(v2)) + 1e-3f;
                }
    printf("gather / halo / streamed kernel vs host float loop: worst relative error %.2e / %.2e / %.2e (bf16 ulp = 7.8e-3)\n", worst[0], worst[1], worst[2]);
    printf("gather vs halo: %zu of %zu outputs differ;  halo vs streamed: %zu differ by more than one bf16 ulp\n", diff_01, n_out, ulp_diff_12);
    CHECK(worst[0] < 1.2e-2 && worst[1] < 1.2e-2 && worst[2] < 1.2e-2, "a kernel is further than one bf16 rounding from the host reference");
    /* the kernels walk K in different orders: fp32 sums may differ in the last bit and flip a bf16 rounding now and then */
    CHECK(ulp_diff_01 == 0 && diff_01 * 100 < n_out, "gather and halo kernels differ by more than rounding flips (%zu > 1 ulp, %zu differ)", ulp_diff_01, diff_01);
    CHECK(ulp_diff_12 == 0, "streamed kernel differs from the halo kernel by more than one bf16 ulp");
    /* ---- row f-3 from C: batch-statistics BN + ReLU forward / backward on the halo kernel's output (M = NB*HH*WW pixels x COUT) */
    {
        const long long M = (long long)NB * HH * WW;
        const long long ws_bytes = v2x_bn_train_workspace_size(M, COUT);
        CHECK(ws_bytes > 0, "v2x_bn_train_workspace_size rejected M=%lld C=%d", M, COUT);
        CHECK(v2x_bn_train_workspace_size(M, 24) == 0, "C = 24 (C / 8 does not divide 256) must be reported as unsupported");
        float *gamma = (float *)malloc(COUT * 4), *beta = (float *)malloc(COUT * 4), *rm = (float *)malloc(COUT * 4), *rv = (float *)malloc(COUT * 4);
        uint16_t *dy = (uint16_t *)malloc(n_out * 2), *bn_y = (uint16_t *)malloc(n_out * 2), *bn_dx = (uint16_t *)malloc(n_out * 2);
        for (int c = 0; c < COUT; ++c) {
            gamma[c] = 1.0f + 0.5f * frand();
            beta[c] = 0.3f * frand();
            rm[c] = 0.0f;
            rv[c] = 1.0f;
        }
        for (size_t i = 0; i < n_out; ++i) dy[i] = to_bf16(frand());
        void *d_g, *d_b, *d_rm, *d_rv, *d_dy, *d_by, *d_dx, *d_mean, *d_istd, *d_dg, *d_db, *d_ws;
        HIPOK(hipMalloc(&d_g, COUT * 4)); HIPOK(hipMalloc(&d_b, COUT * 4)); HIPOK(hipMalloc(&d_rm, COUT * 4)); HIPOK(hipMalloc(&d_rv, COUT * 4));
        HIPOK(hipMalloc(&d_mean, COUT * 4)); HIPOK(hipMalloc(&d_istd, COUT * 4)); HIPOK(hipMalloc(&d_dg, COUT * 4)); HIPOK(hipMalloc(&d_db, COUT * 4));
        HIPOK(hipMalloc(&d_dy, n_out * 2)); HIPOK(hipMalloc(&d_by, n_out * 2)); HIPOK(hipMalloc(&d_dx, n_out * 2)); HIPOK(hipMalloc(&d_ws, (size_t)ws_bytes));
        HIPOK(hipMemcpy(d_g, gamma, COUT * 4, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(d_b, beta, COUT * 4, hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(d_rm, rm, COUT * 4, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(d_rv, rv, COUT * 4, hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(d_dy, dy, n_out * 2, hipMemcpyHostToDevice));
        CHECK(v2x_bn_train_forward((const uint16_t *)d_y[1], M, COUT, (const float *)d_g, (const float *)d_b, 1e-5f, 0.1f, (float *)d_rm, (float *)d_rv, 1,
                                   (uint16_t *)d_by, (float *)d_mean, (float *)d_istd, (float *)d_ws, NULL) == V2X_OK, "v2x_bn_train_forward: %s", v2x_last_error());
        CHECK(v2x_bn_train_backward((const uint16_t *)d_y[1], (const uint16_t *)d_dy, M, COUT, (const float *)d_g, (const float *)d_b, (const float *)d_mean,
                                    (const float *)d_istd, 1, (uint16_t *)d_dx, (float *)d_dg, (float *)d_db, (float *)d_ws, NULL) == V2X_OK,
              "v2x_bn_train_backward: %s", v2x_last_error());
        CHECK(v2x_bn_train_forward((const uint16_t *)d_y[1], M, COUT, (const float *)d_g, (const float *)d_b, 1e-5f, 0.1f, (float *)d_rm, NULL, 1, (uint16_t *)d_by,
                                   (float *)d_mean, (float *)d_istd, (float *)d_ws, NULL) == V2X_EINVAL, "running_mean without running_var must be rejected");
        HIPOK(hipDeviceSynchronize());
        float mean[COUT], istd[COUT], dg[COUT], db[COUT];
        HIPOK(hipMemcpy(mean, d_mean, COUT * 4, hipMemcpyDeviceToHost)); HIPOK(hipMemcpy(istd, d_istd, COUT * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(dg, d_dg, COUT * 4, hipMemcpyDeviceToHost)); HIPOK(hipMemcpy(db, d_db, COUT * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(rm, d_rm, COUT * 4, hipMemcpyDeviceToHost)); HIPOK(hipMemcpy(rv, d_rv, COUT * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(bn_y, d_by, n_out * 2, hipMemcpyDeviceToHost)); HIPOK(hipMemcpy(bn_dx, d_dx, n_out * 2, hipMemcpyDeviceToHost));
        double worst_stat = 0, worst_y = 0, worst_dx = 0, worst_g = 0;
        for (int c = 0; c < COUT; ++c) {
            double s1 = 0, s2 = 0;
            for (long long m = 0; m < M; ++m) {
                const double v = from_bf16(y[1][m * COUT + c]);
                s1 += v;
                s2 += v * v;
            }
            const double mu = s1 / M, var = s2 / M - mu * mu, is = 1.0 / sqrt(var + 1e-5);
            double e = fabs(mean[c] - mu) / (fabs(mu) + 1e-3);
            if (e > worst_stat) worst_stat = e;
            e = fabs(istd[c] - is) / is;
            if (e > worst_stat) worst_stat = e;
            e = fabs(rm[c] - 0.1 * mu) / (fabs(0.1 * mu) + 1e-3);
            if (e > worst_stat) worst_stat = e;
            e = fabs(rv[c] - (0.9 + 0.1 * var * M / (M - 1))) / rv[c];
            if (e > worst_stat) worst_stat = e;
            double sg = 0, sgx = 0;
            for (long long m = 0; m < M; ++m) {
                const double xh = (from_bf16(y[1][m * COUT + c]) - mu) * is, yv = xh * gamma[c] + beta[c];
                const double g = yv > 0 ? from_bf16(dy[m * COUT + c]) : 0.0;
                sg += g;
                sgx += g * xh;
                const double yr = yv > 0 ? yv : 0.0;
                e = fabs(from_bf16(bn_y[m * COUT + c]) - yr) / (fabs(yr) + 1e-2);
                if (e > worst_y) worst_y = e;
            }
            e = fabs(db[c] - sg) / (fabs(sg) + 1.0);
            if (e > worst_g) worst_g = e;
            e = fabs(dg[c] - sgx) / (fabs(sgx) + 1.0);
            if (e > worst_g) worst_g = e;
            for (long long m = 0; m < M; ++m) {
                const double xh = (from_bf16(y[1][m * COUT + c]) - mu) * is, yv = xh * gamma[c] + beta[c];
                if (fabs(yv) < 1e-4) continue; /* the ReLU decision of a value this close to 0 may differ in the last bit */
                const double g = yv > 0 ? from_bf16(dy[m * COUT + c]) : 0.0;
                const double dxr = gamma[c] * is * (g - sg / M - xh * sgx / M);
                e = fabs(from_bf16(bn_dx[m * COUT + c]) - dxr) / (fabs(dxr) + 1e-2);
                if (e > worst_dx) worst_dx = e;
            }
        }
        printf("BN train kernels vs host double loop: statistics %.1e, dgamma/dbeta %.1e, y %.1e, dx %.1e (bf16 ulp = 7.8e-3)\n", worst_stat, worst_g, worst_y, worst_dx);
        CHECK(worst_stat < 1e-4 && worst_g < 1e-3 && worst_y < 1.2e-2 && worst_dx < 1.2e-2, "BN training kernels disagree with the host reference");
    }
    printf("C ABI smoke OK\n");
    return 0;
}
