// What does the MI355X sustain on PURE register-resident bf16 MFMA work?  (The roofline fractions in bench.py are priced against
// the 2.5 PFLOP/s dense peak at 2.4 GHz; the chip runs the step power-limited.)  Every wave issues back-to-back independent
// v_mfma_f32_16x16x32_bf16 on operands that never leave registers: no LDS, no memory.  Variants: 1 or 2 waves per SIMD, constant or
// random operand bits (toggle rate -> power).   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float *out, int iters, uint32_t seed) {
    f32x4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = (f32x4_t){(float)i, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[i])); }   // opaque: no CSE across tiles
    uint32_t r = seed ? (seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x) : 0x3f803f80u;
    uint4 ua, ub;
    ua.x = r; ua.y = r * 3u + 1u; ua.z = r * 5u + 2u; ua.w = r * 7u + 3u;
    ub.x = r ^ 0x5555u; ub.y = r * 11u; ub.z = r * 13u; ub.w = r * 17u;
    if (!seed) { ua = make_uint4(r, r, r, r); ub = ua; }
    // keep the exponents small so that nothing overflows: clear the top exponent bits of every bf16
    ua.x &= 0x3fff3fffu; ua.y &= 0x3fff3fffu; ua.z &= 0x3fff3fffu; ua.w &= 0x3fff3fffu;
    ub.x &= 0x3fff3fffu; ub.y &= 0x3fff3fffu; ub.z &= 0x3fff3fffu; ub.w &= 0x3fff3fffu;
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
}

// The conv kernels' MFMA block: acc[i][f] += A[i] * B[f] with 4 (or 8) different A fragments and 4 different B fragments in registers.
// ORDER 0: i outer, f inner (A reused by 4 consecutive MFMAs);  1: f outer, i inner;  NA = number of A fragments (acc tiles = NA x 4).
template <int NA, int ORDER>
__global__ __launch_bounds__(512) void mfma_block(float *out, int iters, uint32_t seed) {
    f32x4_t acc[NA][4];
    bf16x8_t A[NA], B[4];
    uint32_t r = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint4 u = make_uint4((r * (i + 3)) & 0x3fff3fffu, (r * (i + 5)) & 0x3fff3fffu, (r * (i + 7)) & 0x3fff3fffu, (r * (i + 11)) & 0x3fff3fffu);
        A[i] = __builtin_bit_cast(bf16x8_t, u);
        asm volatile("" : "+v"(A[i]));
#pragma unroll
        for (int f = 0; f < 4; ++f) { acc[i][f] = (f32x4_t){(float)(i * 4 + f), 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[i][f])); }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        uint4 u = make_uint4((r * (f + 13)) & 0x3fff3fffu, (r * (f + 17)) & 0x3fff3fffu, (r * (f + 19)) & 0x3fff3fffu, (r * (f + 23)) & 0x3fff3fffu);
        B[f] = __builtin_bit_cast(bf16x8_t, u);
        asm volatile("" : "+v"(B[f]));
    }
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][f]) : "v"(A[i]), "v"(B[f]));
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][f]) : "v"(A[i]), "v"(B[f]));
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) sum += acc[i][f][0] + acc[i][f][1] + acc[i][f][2] + acc[i][f][3];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
}

// The 32x32x16 form of the same block (round 3, conv3x3_stream8g_kernel<..., M32>): acc[mt][r] += A[mt] * B[r], NM x NR accumulators of 16
// registers, the issue order of the kernel (fragment outer, pixel row inner).
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NM, int NR>
__global__ __launch_bounds__(512) void mfma_block32(float *out, int iters, uint32_t seed) {
    f32x16_t acc[NM][NR];
    bf16x8_t A[NM], B[NR];
    uint32_t r = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        uint4 u = make_uint4((r * (i + 3)) & 0x3fff3fffu, (r * (i + 5)) & 0x3fff3fffu, (r * (i + 7)) & 0x3fff3fffu, (r * (i + 11)) & 0x3fff3fffu);
        if (!seed) u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        A[i] = __builtin_bit_cast(bf16x8_t, u);
        asm volatile("" : "+v"(A[i]));
#pragma unroll
        for (int f = 0; f < NR; ++f) { acc[i][f] = (f32x16_t)((float)(i * NR + f)); asm volatile("" : "+v"(acc[i][f])); }
    }
#pragma unroll
    for (int f = 0; f < NR; ++f) {
        uint4 u = make_uint4((r * (f + 13)) & 0x3fff3fffu, (r * (f + 17)) & 0x3fff3fffu, (r * (f + 19)) & 0x3fff3fffu, (r * (f + 23)) & 0x3fff3fffu);
        if (!seed) u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        B[f] = __builtin_bit_cast(bf16x8_t, u);
        asm volatile("" : "+v"(B[f]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int f = 0; f < NR; ++f) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][f]) : "v"(A[i]), "v"(B[f]));
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NM; ++i)
#pragma unroll
        for (int f = 0; f < NR; ++f) sum += acc[i][f][0] + acc[i][f][5] + acc[i][f][10] + acc[i][f][15];
    if (sum == 12345.678f) out[threadIdx.x] = sum;
}

template <int NM, int NR>
static void run_block32(float *out, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 10000;
    for (int random = 0; random < 2; ++random)
        for (int wps = 1; wps <= 2; ++wps) {
            const int threads = 256 * wps, grid = 256;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL((mfma_block32<NM, NR>), dim3(grid), dim3(threads), 0, 0, out, iters, random ? 99u + rep : 0u);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double n = (double)iters * NM * NR;
            printf("32x32x16 block acc[%d][%d] += A[mt]*B[r], %s operands, %d wave(s)/SIMD: %7.1f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", NM, NR,
                   random ? "random  " : "constant", wps, (double)grid * (threads / 64) * n * 32768.0 / best / 1e9, best * 1e-3 * 2.4e9 / (n * wps));
        }
}

template <int NA, int ORDER>
static void run_block(float *out, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, grid = 256;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((mfma_block<NA, ORDER>), dim3(grid), dim3(threads), 0, 0, out, iters, 99u + rep);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double n = (double)iters * NA * 4;
        printf("block acc[%d][4] += A[i]*B[f], %s, %d wave(s)/SIMD: %7.1f TFLOP/s\n", NA, ORDER ? "f outer" : "i outer", wps,
               (double)grid * (threads / 64) * n * 16384.0 / best / 1e9);
    }
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int random = 0; random < 2; ++random)
        for (int wps = 1; wps <= 2; ++wps) {
            const int threads = 256 * wps, grid = 256;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(threads), 0, 0, out, iters, random ? 12345u + rep : 0u);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)grid * (threads / 64) * iters * 16.0 * 16384.0;
                // cycles per MFMA per SIMD if issue were back to back: time * clock / (MFMAs per SIMD)
                const double mfma_per_simd = (double)wps * iters * 16.0;
                printf("%s operands, %d wave(s)/SIMD: %8.3f ms  %7.1f TFLOP/s  => %.2f GHz if one MFMA per 16 cycles\n", random ? "random  " : "constant",
                       wps, ms, flop / ms / 1e9, mfma_per_simd * 16.0 / (ms * 1e-3) / 1e9);
            }
        }
    run_block<4, 0>(out, e0, e1);
    run_block<4, 1>(out, e0, e1);
    run_block<8, 0>(out, e0, e1);
    run_block<8, 1>(out, e0, e1);
    run_block32<2, 4>(out, e0, e1);   // the 128-row layers' wave tile (2 row tiles x 4 pixel rows)
    run_block32<3, 2>(out, e0, e1);   // the ConvGRU's (3 gates x 2 pixel rows)
    // sustained: the same kernel back to back for ~4 s (the power manager needs far longer than one 5-ms burst to settle)
    for (int random = 0; random < 2; ++random) {
        const int threads = 512, grid = 256;
        const double flop = (double)grid * (threads / 64) * iters * 16.0 * 16384.0;
        for (int sec = 0; sec < 4; ++sec) {
            (void)hipEventRecord(e0, 0);
            const int launches = 200;
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(threads), 0, 0, out, iters, random ? 777u + l : 0u);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("sustained, %s operands, 2 waves/SIMD, window %d (%.0f ms): %7.1f TFLOP/s\n", random ? "random  " : "constant", sec, ms, flop * launches / ms / 1e9);
        }
    }
    return 0;
}
