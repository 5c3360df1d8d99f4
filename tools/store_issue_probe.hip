// What does a vector STORE instruction cost the memory pipeline of a CU, by width and address pattern?  (Round 4: the bf16 epilogues went from
// dwordx2 to dwordx4 stores -- v_permlane16_swap_b32 -- and every layer got 2-13 % faster: store ISSUE, not bandwidth, was the per-tile time.)
// One 512-lane workgroup per CU on 32 CUs (far below the HBM write rate), every wave issues `n` stores back to back; cycles per store instruction
// and CU from s_memtime.  Patterns (lane = 16 fq + fj as in the 16x16x32 result layout, pixel rows of `row_bytes`):
//   0  dwordx2, lane (fj, fq) -> pixel fj, bytes 8 fq .. 8 fq + 7 of a 32-byte segment          (the old epilogues: 16 segments of 32 B)
//   1  dwordx4, lane (fj, fq) -> pixel fj, bytes 16 fq .. 16 fq + 15 of a 64-byte segment        (the new epilogues: 16 segments of 64 B)
//   3  dwordx4, 4 pixels x 256 contiguous bytes = ONE contiguous KiB                              (what an LDS transpose of the tile would give)
//   2  the same addresses written as `base + 16 * lane`: hipcc cannot prove the 16-byte alignment there and emits FOUR global_store_dword per
//      lane -- listed because it shows what a dword store costs (~16 cycles each)
// Measured (profiles/r04_store_issue_probe.txt): 31.4 cycles per dwordx2 store, 31.6 per dwordx4 store of the same 16-segment pattern (the cost is per
// instruction and segment count, not per byte: why the 16-byte epilogues won), 8.6 for a fully contiguous KiB.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_issue_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int PAT>
__global__ __launch_bounds__(512) void store_probe(char *out, int n, int row_bytes, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    char *base = out + ((size_t)blockIdx.x * 8 + wave) * ((size_t)n * 16 * row_bytes + 4096);
    size_t off;
    if (PAT == 0) off = (size_t)fj * row_bytes + fq * 8;
    else if (PAT == 1) off = (size_t)fj * row_bytes + fq * 16;
    else if (PAT == 2) off = (size_t)lane * 16;
    else off = (size_t)(lane >> 4) * row_bytes + (lane & 15) * 16;
    const size_t step = PAT == 2 ? 1024 : (PAT == 3 ? 4 * (size_t)row_bytes : 16 * (size_t)row_bytes);
    const uint4 v = make_uint4(lane, wave, blockIdx.x, n);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        char *p = base + off + (size_t)i * step;
        if (PAT == 0) *reinterpret_cast<uint2 *>(p) = make_uint2(v.x + i, v.y);
        else *reinterpret_cast<uint4 *>(p) = make_uint4(v.x + i, v.y, v.z, v.w);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();          // all stores ISSUED (the wave got them into the pipeline)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();          // ... and acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        cyc[blockIdx.x * 2] = t1 - t0;
        cyc[blockIdx.x * 2 + 1] = t2 - t0;
    }
}

template <int PAT>
static void run(char *out0, unsigned long long *cyc, const char *what, int bytes_per_store, int region) {
    const int n = 64, grid = 32, row_bytes = 256;
    char *out = out0 + (size_t)region * (192u << 20);      // a region of its own per run: no line is written by two patterns
    unsigned long long h[64];
    double issue = 0, done = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(store_probe<PAT>, dim3(grid), dim3(512), 0, 0, out, n, row_bytes, cyc);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        if (rep == 3)
            for (int i = 0; i < grid; ++i) { issue += (double)h[2 * i]; done += (double)h[2 * i + 1]; }
    }
    issue /= grid; done /= grid;
    // 8 waves x n stores per CU
    printf("%-78s issue %6.1f cycles per store and CU, drained %6.1f  (%5.1f B/clk/CU)\n", what, issue / (8.0 * n), done / (8.0 * n), bytes_per_store * 8.0 * n / done);
}

int main() {
    char *out;
    unsigned long long *cyc;
    if (hipMalloc(&out, (size_t)2 << 30) != hipSuccess) return 1;
    (void)hipMemset(out, 0, (size_t)2 << 30);
    (void)hipMalloc(&cyc, 64 * 8);
    run<0>(out, cyc, "dwordx2, 16 pixels x 32-byte segments (8-byte epilogues)", 512, 0);
    run<1>(out, cyc, "dwordx4, 16 pixels x 64-byte segments (16-byte epilogues, v_permlane16_swap)", 1024, 1);
    run<3>(out, cyc, "dwordx4, 4 pixels x 256 contiguous bytes", 1024, 2);
    run<2>(out, cyc, "dwordx4, one contiguous KiB (= the same addresses, lane for lane)", 1024, 3);
    run<2>(out, cyc, "  again, the other order: one contiguous KiB", 1024, 4);
    run<3>(out, cyc, "  4 pixels x 256 contiguous bytes", 1024, 5);
    run<1>(out, cyc, "  16 pixels x 64-byte segments", 1024, 6);
    run<0>(out, cyc, "  dwordx2, 16 pixels x 32-byte segments", 512, 7);
    return 0;
}
