// Which per-lane address patterns does a ds_read_b128 serve without bank conflicts?  Times a loop of reads whose 64 byte offsets come
// from a table, for the pixel-fragment reads of the streamed conv kernels: lane (fj = lane & 15, fq = lane >> 4) reads 16 bytes of pixel
// pc(fj) at channel slot fq, stored at (pc * 4 + (fq ^ g(pc))) * 16.  Full-resolution source: pc = c + fj; half-resolution source (nearest
// x2 upsample): pc = ((c + fj) >> 1) + 1.  c = 16 * ch + kx - 1 takes every alignment.  g = the swizzle under test.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_conflict_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void read_loop(const int *table, int iters, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t *>(smem)[i] = i;
    __syncthreads();
    const int off = table[threadIdx.x & 63];
    const char *p = smem + off + (threadIdx.x >> 6) * 8192;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"((uint32_t)(uintptr_t)p));   // 8 reads in flight: throughput
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc.y + acc.z + acc.w;
}

static int g_eval(int g, int pc) {
    switch (g) {
        case 0: return (pc >> 1) & 3;                 // round-1 choice for 4 slots per pixel
        case 1: return (pc >> 2) & 3;
        case 2: return 0;
        case 3: return pc & 3;
        case 4: return pc & 7;                        // round-1 choice for 8 slots per pixel
        case 5: return (pc >> 1) & 7;
        case 6: return ((pc >> 1) & 3) | ((pc & 1) << 2);
        case 7: return ((pc >> 2) & 3) | ((pc & 1) << 2);
        case 8: return ((pc >> 1) & 3) << 1 | (pc & 1);
        case 9: return (pc >> 1) & 1 ? 4 ^ ((pc >> 2) & 3) : (pc >> 2) & 3;
    }
    return 0;
}

// `lds_probe counters`: one dispatch per (swizzle, resolution, alignment) of the 4-slot patch, in a fixed order, so that a
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -- /tmp/lds_probe counters
// pass gives the COUNTERS of exactly the patterns whose TIME the default mode prints (tools/lds_probe_counters.sh joins the two): does
// a counted conflict cost time?
static int counters_mode() {
    int *d_table;
    uint32_t *d_out;
    (void)hipMalloc(&d_table, 64 * 4);
    (void)hipMalloc(&d_out, 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int g = 0; g < 2; ++g)
        for (int half = 0; half < 2; ++half)
            for (int ch = 0; ch < 2; ++ch)
                for (int kx = 0; kx < 3; ++kx) {
                    int table[64];
                    for (int lane = 0; lane < 64; ++lane) {
                        const int fj = lane & 15, fq = lane >> 4;
                        const int c = 16 * ch + kx;
                        const int pc = half ? ((c + fj - 1) >> 1) + 1 : c + fj;
                        table[lane] = (pc * 4 + ((fq ^ g_eval(g, pc)) & 3)) * 16;
                    }
                    (void)hipMemcpy(d_table, table, sizeof(table), hipMemcpyHostToDevice);
                    (void)hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(read_loop, dim3(256), dim3(256), 65536, 0, d_table, iters, d_out);
                    (void)hipEventRecord(e1, 0);
                    (void)hipEventSynchronize(e1);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    printf("dispatch g=%s %s ch=%d kx=%d : %.1f ns per read\n", g ? "(pc>>2)&3" : "(pc>>1)&3", half ? "half-res" : "full-res", ch, kx, ms * 1e6 / (iters * 8.0));
                }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'c') return counters_mode();
    int *d_table;
    uint32_t *d_out;
    (void)hipMalloc(&d_table, 64 * 4);
    (void)hipMalloc(&d_out, 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    const char *gname[] = {"(pc>>1)&3", "(pc>>2)&3", "none", "pc&3", "pc&7", "(pc>>1)&7", "b21|b0<<2", "b32|b0<<2", "b21<<1|b0", "b1?4^b32:b32"};
    // spp = 16-byte slots per pixel (channels / 8); the read takes slots kc*4 + fq, kc = 0 .. spp/4 - 1
    for (int spp = 4; spp <= 12; spp += 4)
        for (int half = 0; half < 2; ++half)
            for (int g = 0; g < 10; ++g) {
                if (spp == 4 && g >= 4) continue;
                printf("%2d slots/pixel, %s source, g = %-13s:", spp, half ? "half-res" : "full-res", gname[g]);
                double worst = 0, sum = 0;
                int cnt = 0;
                for (int kc = 0; kc < spp / 4; ++kc)
                    for (int ch = 0; ch < 2; ++ch)
                        for (int kx = 0; kx < 3; ++kx) {
                            int table[64];
                            for (int lane = 0; lane < 64; ++lane) {
                                const int fj = lane & 15, fq = lane >> 4;
                                const int c = 16 * ch + kx;
                                const int pc = half ? ((c + fj - 1) >> 1) + 1 : c + fj;
                                const int slot = kc * 4 + fq;
                                const int sw = spp == 12 ? (slot & ~3) | ((slot & 3) ^ (g_eval(g, pc) & 3)) : (slot ^ g_eval(g, pc)) & (spp - 1);
                                table[lane] = (pc * spp + sw) * 16;
                            }
                            (void)hipMemcpy(d_table, table, sizeof(table), hipMemcpyHostToDevice);
                            float best = 1e30f;
                            for (int rep = 0; rep < 3; ++rep) {
                                (void)hipEventRecord(e0, 0);
                                hipLaunchKernelGGL(read_loop, dim3(256), dim3(256), 65536, 0, d_table, iters, d_out);
                                (void)hipEventRecord(e1, 0);
                                (void)hipEventSynchronize(e1);
                                float ms = 0;
                                (void)hipEventElapsedTime(&ms, e0, e1);
                                if (ms < best) best = ms;
                            }
                            const double ns_per_read = best * 1e6 / (iters * 8.0);
                            sum += ns_per_read;
                            ++cnt;
                            if (ns_per_read > worst) worst = ns_per_read;
                        }
                printf(" mean %.1f worst %.1f ns per read (8 in flight, 4 waves per CU; over kc x ch x kx)\n", sum / cnt, worst);
            }
    return 0;
}
