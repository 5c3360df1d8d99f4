// What does rocprofv3's FETCH_SIZE report for the access pattern of the STREAMED convolution kernels' patch fills -- 64-byte segments (one 32-channel chunk of a pixel: four
// lanes x 16 B) of pixels whose records are 128 ... 512 bytes apart -- as opposed to the wide contiguous reads the guide calibrated (MI355X_MICROARCH.md, HBM: "FETCH_SIZE
// reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern")?
// Round 6: bench.py's roofline.traffic and tools/pmc_traffic.py double FETCH_SIZE for EVERY kernel; profiles/r06_s2g_traffic.txt shows a layer whose doubled count is 1.95 x its
// input.  Each kernel below reads a 1-GiB buffer (4 x the Infinity Cache) EXACTLY ONCE through global_load_lds (16 B per lane, as the kernels do), in a different order:
//   wide      : a wave instruction = 1 KiB contiguous
//   seg<S>    : a wave instruction = 16 segments of 64 B, segment s of pixel p at p * S + 64 * c (S = pixel stride in bytes); a workgroup walks a tile of 1 024 pixels chunk by
//               chunk (c = 0 .. S/64 - 1) -- the other 64-byte half of every 128-byte line is touched S/64 steps later by the same workgroup (an L2 hit if the line stayed)
//   segfar<S> : the same, but ALL pixels' chunk c before any pixel's chunk c + 1 (the second half of a line is touched after the whole GiB has gone by: a miss again)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fetch_calib_probe.hip -o /tmp/fetch_calib && rocprofv3 --pmc FETCH_SIZE -d out -o p --output-format csv -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
constexpr size_t TOTAL = (size_t)1 << 30;

__global__ __launch_bounds__(256) void calib_wide(const char *src, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_pieces = TOTAL / 1024;
    for (size_t p = (size_t)blockIdx.x * 4 + wave; p < n_pieces; p += (size_t)gridDim.x * 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024 + lane * 16), (lptr_t)(lds + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned *>(lds);
}

template <int S, bool FAR>
__global__ __launch_bounds__(256) void calib_seg(const char *src, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CH = S / 64;                       // chunks per pixel
    const size_t n_pix = TOTAL / S;
    const size_t n_groups = n_pix / 16;              // a wave instruction covers 16 pixels' segments of one chunk
    if (FAR) {
        for (int c = 0; c < CH; ++c)
            for (size_t g = (size_t)blockIdx.x * 4 + wave; g < n_groups; g += (size_t)gridDim.x * 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (g * 16 + (lane >> 2)) * S + c * 64 + (lane & 3) * 16), (lptr_t)(lds + wave * 1024), 16, 0, 0);
    } else {
        constexpr int TILE = 1024 / 16;              // groups per tile of 1 024 pixels
        const size_t n_tiles = n_groups / TILE;
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x)
            for (int c = 0; c < CH; ++c)
                for (int g = wave; g < TILE; g += 4)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((t * TILE + g) * 16 + (lane >> 2)) * S + c * 64 + (lane & 3) * 16), (lptr_t)(lds + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned *>(lds);
}

int main() {
    char *src;
    unsigned *sink;
    if (hipMalloc(&src, TOTAL) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) return 1;
    (void)hipMemset(src, 1, TOTAL);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_wide, dim3(grid), dim3(256), 0, 0, src, sink);
        hipLaunchKernelGGL((calib_seg<128, false>), dim3(grid), dim3(256), 0, 0, src, sink);
        hipLaunchKernelGGL((calib_seg<256, false>), dim3(grid), dim3(256), 0, 0, src, sink);
        hipLaunchKernelGGL((calib_seg<512, false>), dim3(grid), dim3(256), 0, 0, src, sink);
        hipLaunchKernelGGL((calib_seg<128, true>), dim3(grid), dim3(256), 0, 0, src, sink);
        hipLaunchKernelGGL((calib_seg<512, true>), dim3(grid), dim3(256), 0, 0, src, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("every kernel read %zu bytes exactly once\n", TOTAL);
    return 0;
}
