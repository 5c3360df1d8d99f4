// What exactly does ds_read_b64_tr_b16 return?  LDS holds a [rows][64] bf16 image whose element (r, c) has the value 64 r + c (as raw 16-bit
// integers); every lane reads with the address pattern the wgrad kernel wants to use -- within a 16-lane group lane i points at the 4
// contiguous elements (row r0 + (i >> 2), columns c0 + 4 (i & 3) ...) -- and the four 16-bit results per lane are printed.  Expectation
// (cdna_hip_programming.md T10): lane i receives column c0 + i of rows r0 .. r0 + 3.      hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t *out, int row_stride_elems) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    for (int i = threadIdx.x; i < 64 * row_stride_elems; i += 64) lds[i] = (uint16_t)((i / row_stride_elems) * 64 + (i % row_stride_elems));
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, grp = lane >> 4;
    // group g reads rows 8 g .. 8 g + 3 (the first half of k-slot g), columns 16 .. 31
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + ((8 * grp + (i >> 2)) * row_stride_elems + 16 + 4 * (i & 3)) * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(v.x & 0xffff);
    out[lane * 4 + 1] = (uint16_t)(v.x >> 16);
    out[lane * 4 + 2] = (uint16_t)(v.y & 0xffff);
    out[lane * 4 + 3] = (uint16_t)(v.y >> 16);
}
int main() {
    uint16_t *d, h[256];
    (void)hipMalloc(&d, sizeof(h));
    for (int stride : {64, 32}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 64 * 64 * 2, 0, d, stride);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements; lane: (row, col) of its 4 results\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
            printf("%s", (l & 3) == 3 ? "\n" : "   ");
        }
    }
    return 0;
}
