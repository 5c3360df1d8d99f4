// Calibration probes behind v2x_calib_stream / v2x_calib_mfma (include/v2x_amd.h): what THIS box sustains on a pure streaming kernel at a
// given read : write mix, and on a register-resident MFMA loop with random operands -- the measured ceilings the roofline fractions of bench.py
// and DESIGN.md section 6 are graded against next to the 8 TB/s / 2.5 PFLOP/s datasheet peaks, and the per-box normalisation of the bench line
// (box-to-box spread is +-2.5 %).  No upstream counterpart.  tools/hbm_mix_probe.hip drives the same kernels stand-alone.
#include "common.h"

// NR read streams and NW write streams of `units` 16-byte elements each, every stream fully coalesced (lane = 16 B, wave = 1 KiB), the layout
// of the HBM-bound layers: the heads read x once and write cls + loc (1 : 3 in bytes), conv8_2 reads and writes one map (1 : 1), conv1_1 reads
// four times what it writes.  Stores depend on the loads (xor) so that nothing is dead; NT = non-temporal loads and stores.
template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void calib_stream_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long units) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += stride) {
        uint4 v = make_uint4((uint32_t)u, 0u, 0u, 0u);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint4 *p = src + (long long)r * units + u;
            uint4 x;
            if (NT) {
                x.x = __builtin_nontemporal_load(&p->x);
                x.y = __builtin_nontemporal_load(&p->y);
                x.z = __builtin_nontemporal_load(&p->z);
                x.w = __builtin_nontemporal_load(&p->w);
            } else {
                x = *p;
            }
            v.x ^= x.x; v.y ^= x.y; v.z ^= x.z; v.w ^= x.w;
        }
        if (NW == 0) {
            if (v.x == 0x9e3779b9u && v.y == 0x7f4a7c15u && v.z == 1u) dst[0] = v;     // (never: keeps the loads alive)
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint4 *q = dst + (long long)w * units + u;
            uint4 y = make_uint4(v.x + w, v.y, v.z, v.w);
            if (NT) {
                __builtin_nontemporal_store(y.x, &q->x);
                __builtin_nontemporal_store(y.y, &q->y);
                __builtin_nontemporal_store(y.z, &q->z);
                __builtin_nontemporal_store(y.w, &q->w);
            } else {
                *q = y;
            }
        }
    }
}

template <int NR, int NW>
static int launch_stream(const void *src, void *dst, long long units, int nt, int wg_per_cu, hipStream_t s) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long grid = (long long)cus * (wg_per_cu > 0 ? wg_per_cu : 8);
    const long long need = (units + 255) / 256;
    if (grid > need) grid = need;
    if (nt)
        hipLaunchKernelGGL((calib_stream_kernel<NR, NW, true>), dim3((unsigned)grid), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst, units);
    else
        hipLaunchKernelGGL((calib_stream_kernel<NR, NW, false>), dim3((unsigned)grid), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst, units);
    V2X_CHECK_LAUNCH("calib_stream_kernel");
    return V2X_OK;
}

extern "C" int v2x_calib_stream(const void *src, void *dst, int64_t units, int n_read, int n_write, int nontemporal, int wg_per_cu,
                                v2x_stream_t stream) {
    V2X_REQUIRE(src && dst && units > 0, "v2x_calib_stream: null buffer or no units");
    V2X_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "v2x_calib_stream: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int key = n_read * 10 + n_write;
    switch (key) {
        case 11: return launch_stream<1, 1>(src, dst, units, nontemporal, wg_per_cu, s);
        case 13: return launch_stream<1, 3>(src, dst, units, nontemporal, wg_per_cu, s);
        case 21: return launch_stream<2, 1>(src, dst, units, nontemporal, wg_per_cu, s);
        case 41: return launch_stream<4, 1>(src, dst, units, nontemporal, wg_per_cu, s);
        case 10: return launch_stream<1, 0>(src, dst, units, nontemporal, wg_per_cu, s);
        case 1: return launch_stream<0, 1>(src, dst, units, nontemporal, wg_per_cu, s);
        default: break;
    }
    v2x_set_error("v2x_calib_stream: read:write mix %d:%d is not built (1:1, 1:3, 2:1, 4:1, 1:0, 0:1)", n_read, n_write);
    return V2X_EINVAL;
}

// Every wave issues `iters` x 16 back-to-back independent MFMAs on operands that never leave registers (no LDS, no memory): 2 waves per SIMD,
// one workgroup of 512 lanes per CU.  Random operand bits (seed != 0) toggle the datapath the way real activations do -- the chip then runs
// power-limited (2.13-2.22 PFLOP/s measured in round 3); constant operands (seed == 0) reach the 2.5 PFLOP/s datasheet figure.
// clocks[0..1] = s_memtime (shader cycles) and s_memrealtime (100 MHz) elapsed over the loop of workgroup 0's first wave: the sustained shader clock.
template <bool M32>
__global__ __launch_bounds__(512) void calib_mfma_kernel(float *out, int iters, uint32_t seed, unsigned long long *clocks) {
    uint32_t r = seed ? (seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x) : 0x3f803f80u;
    uint4 ua = make_uint4(r, r * 3u + 1u, r * 5u + 2u, r * 7u + 3u), ub = make_uint4(r ^ 0x5555u, r * 11u, r * 13u, r * 17u);
    if (!seed) { ua = make_uint4(r, r, r, r); ub = ua; }
    ua.x &= 0x3fff3fffu; ua.y &= 0x3fff3fffu; ua.z &= 0x3fff3fffu; ua.w &= 0x3fff3fffu;     // small exponents: nothing overflows
    ub.x &= 0x3fff3fffu; ub.y &= 0x3fff3fffu; ub.z &= 0x3fff3fffu; ub.w &= 0x3fff3fffu;
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (M32) {
        f32x16_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = (f32x16_t)((float)i); asm volatile("" : "+v"(acc[i])); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    } else {
        f32x4_t acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = (f32x4_t){(float)i, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[i])); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    if (clocks && blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = w1 - w0;
    }
    if (sum == 12345.678f) out[threadIdx.x] = sum;
}

extern "C" int v2x_calib_mfma(float *scratch, int iters, uint32_t seed, int shape32, uint64_t *clocks, double *flops, v2x_stream_t stream) {
    V2X_REQUIRE(scratch && iters > 0, "v2x_calib_mfma: scratch (>= 512 floats, device) and iters > 0 needed");
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipStream_t s = (hipStream_t)stream;
    if (shape32)
        hipLaunchKernelGGL(calib_mfma_kernel<true>, dim3(cus), dim3(512), 0, s, scratch, iters, seed, (unsigned long long *)clocks);
    else
        hipLaunchKernelGGL(calib_mfma_kernel<false>, dim3(cus), dim3(512), 0, s, scratch, iters, seed, (unsigned long long *)clocks);
    V2X_CHECK_LAUNCH("calib_mfma_kernel");
    // 16 MFMAs of 16x16x32 (16 384 FLOP) or 8 of 32x32x16 (32 768 FLOP) per iteration and wave, 8 waves per workgroup
    if (flops) *flops = (double)cus * 8.0 * (double)iters * 16.0 * 16384.0;
    return V2X_OK;
}
