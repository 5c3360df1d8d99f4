// What does a second wave on the same SIMD cost a wave that issues back-to-back MFMAs?  (The ping-pong conv kernels pair an MFMA-phase wave
// with a load-phase wave on every SIMD; the step timeline shows the 96-MFMA phase taking 1.6x its back-to-back time.)
// Waves 0-3 of a 512-thread workgroup run the MFMA loop and time themselves (s_memrealtime, 100 MHz); waves 4-7 run a companion loop:
//   0 none (exit)   1 dense VALU (8 independent v_add per iteration)   2 VALU at ~25 % duty   3 ds_read_b128 stream   4 LDS-DMA stream
//   5 mixed: what a load phase issues per step (~200 VALU, 14 ds_read_b128, 6 DMA)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// M32: the MFMA waves issue v_mfma_f32_32x32x16_bf16 (acc[2][4] of 16 registers: 8 MFMAs per iteration = the FLOPs of the 16 16x16x32 ones)
template <int MODE, bool MF, bool M32 = false>
__global__ __launch_bounds__(512) void probe(float *out, unsigned *times, const uint4 *src, int iters, int n2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (M32 && wave < 4) {
        f32x16_t acc[2][4];
        bf16x8_t A[2], B[4];
        uint32_t r = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 u = make_uint4((r * (i + 3)) & 0x3fff3fffu, (r * (i + 5)) & 0x3fff3fffu, (r * (i + 7)) & 0x3fff3fffu, (r * (i + 11)) & 0x3fff3fffu);
            if (i < 2) A[i] = __builtin_bit_cast(bf16x8_t, u);
            u.x ^= 0x01010101u;
            B[i] = __builtin_bit_cast(bf16x8_t, u);
            asm volatile("" : "+v"(B[i]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asm volatile("" : "+v"(A[i]));
#pragma unroll
            for (int f = 0; f < 4; ++f) { acc[i][f] = (f32x16_t)((float)(i + f)); asm volatile("" : "+v"(acc[i][f])); }
        }
        __syncthreads();
        if (!MF) return;
        const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][f]) : "v"(A[i]), "v"(B[f]));
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        const unsigned t1 = (unsigned)__builtin_amdgcn_s_memrealtime();
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) sum += acc[i][f][0] + acc[i][f][7] + acc[i][f][15];
        if (sum == 12345.678f) out[threadIdx.x] = sum;
        if (lane == 0) times[blockIdx.x * 4 + wave] = t1 - t0;
    } else if (wave < 4) {
        f32x4_t acc[4][4];
        bf16x8_t A[4], B[4];
        uint32_t r = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 u = make_uint4((r * (i + 3)) & 0x3fff3fffu, (r * (i + 5)) & 0x3fff3fffu, (r * (i + 7)) & 0x3fff3fffu, (r * (i + 11)) & 0x3fff3fffu);
            A[i] = __builtin_bit_cast(bf16x8_t, u);
            u.x ^= 0x01010101u;
            B[i] = __builtin_bit_cast(bf16x8_t, u);
            asm volatile("" : "+v"(A[i]), "+v"(B[i]));
#pragma unroll
            for (int f = 0; f < 4; ++f) { acc[i][f] = (f32x4_t){(float)(i + f), 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[i][f])); }
        }
        __syncthreads();
        if (!MF) return;
        const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][f]) : "v"(A[i]), "v"(B[f]));
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        const unsigned t1 = (unsigned)__builtin_amdgcn_s_memrealtime();
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) sum += acc[i][f][0] + acc[i][f][3];
        if (sum == 12345.678f) out[threadIdx.x] = sum;
        if (lane == 0) times[blockIdx.x * 4 + wave] = t1 - t0;
    } else {
        __syncthreads();
#ifdef COMP_PRIO
        __builtin_amdgcn_s_setprio(COMP_PRIO);   // -DCOMP_PRIO=n: the companion at a raised priority (does its LDS-DMA get through beside 32x32x16 MFMAs then?)
#endif
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lane + u;
        uint4 q = make_uint4(0, 0, 0, 0);
        const uint32_t la = (uint32_t)(uintptr_t)(lptr_t)(smem + lane * 16 + (wave - 4) * 4096);
        const unsigned c0 = (unsigned)__builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < n2; ++it) {
            if (MODE == 1 || MODE == 2 || MODE == 5) {
                const int reps = MODE == 5 ? 25 : 1;
                for (int rr = 0; rr < reps; ++rr) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[u]) : "v"(lane));
                }
                if (MODE == 2) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            }
            if (MODE == 3 || MODE == 5) {
                const int reps = MODE == 5 ? 14 : 4;
                for (int rr = 0; rr < reps; ++rr) {
                    uint4 t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    q.x ^= t.x; q.y ^= t.y;
                }
            }
            if (MODE == 4 || MODE == 5) {
                const int reps = MODE == 5 ? 6 : 4;
                for (int rr = 0; rr < reps; ++rr)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((it * 8 + rr) & 1023) * 64 + lane), (lptr_t)(smem + 32768 + (wave - 4) * 4096 + rr * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE == 5) asm volatile("s_sleep 4" ::: "memory");
        }
        const unsigned c1 = (unsigned)__builtin_amdgcn_s_memrealtime();
        if (lane == 0) times[1024 + blockIdx.x * 4 + wave - 4] = c1 - c0;
        uint32_t s = q.x ^ q.y;
#pragma unroll
        for (int u = 0; u < 8; ++u) s ^= v[u];
        if (s == 0x12345678u) out[threadIdx.x] = (float)s;
    }
}

template <int MODE, bool M32 = false>
static void run(const char *name, float *out, unsigned *d_times, const uint4 *src) {
    const int grid = 256;
    static unsigned h[2048];
    double mf_ns[2] = {0, 0}, co_ns[2] = {0, 0};
    // (a) MFMA waves long, companion short: how fast does the COMPANION progress beside MFMAs?  (b) companion alone.  (c) MFMA waves
    // short, companion long: how fast do the MFMAs go beside the companion?  (d) MFMA waves alone (MODE 0).
    const int co_iters = MODE == 5 ? 300 : 3000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, true, M32>), dim3(grid), dim3(512), 65536, 0, out, d_times, src, 40000, co_iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_times, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; ++i) co_ns[0] += h[1024 + i] * 10.0 / 1024;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, false, M32>), dim3(grid), dim3(512), 65536, 0, out, d_times, src, 0, co_iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_times, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; ++i) co_ns[1] += h[1024 + i] * 10.0 / 1024;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, true, M32>), dim3(grid), dim3(512), 65536, 0, out, d_times, src, 4000, co_iters * 40);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_times, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; ++i) mf_ns[0] += h[i] * 10.0 / 1024;
    printf("%s %-52s: MFMA %.2f ns per 16x16x32-equivalent beside it;  companion iteration %.0f ns beside MFMAs vs %.0f ns alone (%.2fx)\n", M32 ? "[32x32x16]" : "[16x16x32]", name, mf_ns[0] / (4000 * 16.0),
           co_ns[0] / co_iters, co_ns[1] / co_iters, co_ns[0] / co_ns[1]);
}

int main() {
    float *out;
    unsigned *d_times;
    uint4 *src;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&d_times, 8192);
    (void)hipMalloc(&src, 1024 * 64 * 16);
    (void)hipMemset(src, 0, 1024 * 64 * 16);
    (void)hipMemset(d_times, 0, 8192);
    run<1>("companion: dense VALU (8 v_add per iteration)", out, d_times, src);
    run<2>("companion: VALU ~25 % duty", out, d_times, src);
    run<3>("companion: 4 dependent ds_read_b128", out, d_times, src);
    run<4>("companion: LDS-DMA (4 x 1 KiB, drained)", out, d_times, src);
    run<5>("companion: load-phase mix (200 VALU, 14 reads, 6 DMA)", out, d_times, src);
    run<1, true>("companion: dense VALU (8 v_add per iteration)", out, d_times, src);
    run<2, true>("companion: VALU ~25 % duty", out, d_times, src);
    run<3, true>("companion: 4 dependent ds_read_b128", out, d_times, src);
    run<4, true>("companion: LDS-DMA (4 x 1 KiB, drained)", out, d_times, src);
    run<5, true>("companion: load-phase mix (200 VALU, 14 reads, 6 DMA)", out, d_times, src);
    return 0;
}
