// a1 -- LiDAR point cloud -> BEV occupancy.
//
// Replaces upstream coperception/utils/data_util.py::voxelize_occupy and the densify
// scatter of coperception/datasets/V2XSimDet.py::__getitem__ (code absent from
// /root/reference; see include/v2x_amd.h).  The numpy lexsort+unique formulation is
// replaced by an idempotent bit scatter: one uint32 word per BEV pixel, bit z = occupied.
// 13 z-bins fit one word (16 bits in LDS), so the whole 256x256x13 grid of one agent is 128 KiB and is binned inside
// ONE CU's LDS while its 64k points stream through (voxelize_lds_kernel); duplicates collapse for free and the result is
// order-independent (bit-exact, deterministic).  Grids too large for the LDS scatter with device-scope atomicOr into
// the L2-resident global grid instead (voxelize_scatter_kernel, same arithmetic).  The dense layouts the network / the
// reference API want are then produced by fully coalesced expansion kernels (HBM-bound streaming writes).
//
// Numerics: strict  lo < p < hi  in fp64 on the promoted fp32
// coordinate; idx = floor(fp64(p) / fp64(voxel)) - floor(lo / voxel).  IEEE fp64
// division (hipcc default, no fast-math) -- never a reciprocal multiply.
#include "common.h"

struct VoxParams {
    double lo[3], hi[3], vs[3], mn[3];
    double inv[3];   // 1 / vs where vs is a power of two (exact)
    int pow2[3];
    int X, Y, Z;
};

// fp64(p) / fp64(voxel) of the spec.  An IEEE division costs ~40 instructions per coordinate (three of them were half of the LDS kernel's time);
// when the voxel size is a power of two (0.25 m in x and y by default) the product with its EXACT reciprocal is the same correctly rounded
// number -- bit-identical, one instruction.  The choice is per axis and uniform over the launch.
__device__ __forceinline__ double vox_quot(double p, const VoxParams &vp, int k) { return vp.pow2[k] ? p * vp.inv[k] : p / vp.vs[k]; }

static void vox_reciprocal(VoxParams &vp, int a) {
    int e = 0;
    const double m = frexp(vp.vs[a], &e);
    vp.pow2[a] = (m == 0.5 && e > -1000 && e < 1000) ? 1 : 0;     // a power of two whose reciprocal is a normal number
    vp.inv[a] = vp.pow2[a] ? 1.0 / vp.vs[a] : 0.0;
}

__global__ __launch_bounds__(256) void voxelize_scatter_kernel(const float *__restrict__ pts,
                                                               const int32_t *__restrict__ n_pts, int max_pts,
                                                               int pt_stride, VoxParams vp,
                                                               uint32_t *__restrict__ bits) {
    const int cloud = blockIdx.y;
    const int n = min(n_pts[cloud], max_pts);   // a count beyond the cloud's capacity must not read the next cloud (same clamp as the LDS form)
    const float *base = pts + (size_t)cloud * max_pts * pt_stride;
    uint32_t *grid = bits + (size_t)cloud * vp.X * vp.Y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = (double)base[(size_t)i * pt_stride + 0];
        const double y = (double)base[(size_t)i * pt_stride + 1];
        const double z = (double)base[(size_t)i * pt_stride + 2];
        const bool keep = (vp.lo[0] < x) && (x < vp.hi[0]) && (vp.lo[1] < y) && (y < vp.hi[1]) &&
                          (vp.lo[2] < z) && (z < vp.hi[2]);
        if (!keep) continue;
        const int ix = (int)(floor(vox_quot(x, vp, 0)) - vp.mn[0]);
        const int iy = (int)(floor(vox_quot(y, vp, 1)) - vp.mn[1]);
        const int iz = (int)(floor(vox_quot(z, vp, 2)) - vp.mn[2]);
        // extents that are not voxel multiples could produce an edge index == dim; the
        // reference would raise IndexError there, we drop the point instead of corrupting memory.
        if ((unsigned)ix >= (unsigned)vp.X || (unsigned)iy >= (unsigned)vp.Y || (unsigned)iz >= (unsigned)vp.Z)
            continue;
        atomicOr(&grid[(size_t)ix * vp.Y + iy], 1u << iz);
    }
}

// LDS-binned form of the scatter (the default when the grid of one cloud fits the LDS): ONE workgroup per cloud keeps
// the whole occupancy grid in LDS as 16-bit words (Z <= 16; 256 x 256 x 2 B = 128 KiB of the CU's 160 KiB), so every
// atomic is a ds_or_b32 instead of a device-scope L2 atomic, and the grid leaves the CU exactly once, as linear 16-B
// stores of the expanded 32-bit words -- no memset pass, no HBM read-modify-write.  Arithmetic per point is the scatter
// kernel's, so the two forms are bit-identical (tests/test_gpu_stages.py runs both).  The loads are issued UNR deep
// per thread before the fp64 work so that a single CU keeps ~64 KiB of the cloud in flight.
constexpr int VOX_LDS_THREADS = 1024;
constexpr int VOX_LDS_UNR = 4;

__device__ __forceinline__ void vox_lds_point(bool valid, double x, double y, double z, const VoxParams &vp,
                                              uint32_t *sgrid) {
    const bool keep = valid && (vp.lo[0] < x) && (x < vp.hi[0]) && (vp.lo[1] < y) && (y < vp.hi[1]) &&
                      (vp.lo[2] < z) && (z < vp.hi[2]);
    if (!keep) return;
    const int ix = (int)(floor(vox_quot(x, vp, 0)) - vp.mn[0]);
    const int iy = (int)(floor(vox_quot(y, vp, 1)) - vp.mn[1]);
    const int iz = (int)(floor(vox_quot(z, vp, 2)) - vp.mn[2]);
    if ((unsigned)ix >= (unsigned)vp.X || (unsigned)iy >= (unsigned)vp.Y || (unsigned)iz >= (unsigned)vp.Z) return;
    const int pix = ix * vp.Y + iy;
    atomicOr(&sgrid[pix >> 1], (1u << iz) << ((pix & 1) * 16));
}

template <bool VEC4>
__global__ __launch_bounds__(VOX_LDS_THREADS) void voxelize_lds_kernel(const float *__restrict__ pts,
                                                                       const int32_t *__restrict__ n_pts, int max_pts,
                                                                       int pt_stride, VoxParams vp,
                                                                       uint32_t *__restrict__ bits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sgrid[];   // [X*Y/2]: two 16-bit pixels per word
    const int cloud = blockIdx.x;
    const int tid = threadIdx.x;
    const int n_words = (vp.X * vp.Y) >> 1;
    for (int i = tid * 4; i < n_words; i += VOX_LDS_THREADS * 4) *reinterpret_cast<uint4 *>(&sgrid[i]) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int n = min(n_pts[cloud], max_pts);
    const float *base = pts + (size_t)cloud * max_pts * pt_stride;
    // software pipeline: the loads of batch k+1 are in flight while batch k goes through the index math.  Loads
    // are unconditional (index clamped to the last point) so that the compiler keeps all UNR of them in flight; the
    // validity predicate is applied to the scatter only.
    float px[VOX_LDS_UNR], py[VOX_LDS_UNR], pz[VOX_LDS_UNR];
    auto load_batch = [&](int i0, float *ox, float *oy, float *oz) {
#pragma unroll
        for (int u = 0; u < VOX_LDS_UNR; ++u) {
            const int i = min(i0 + u * VOX_LDS_THREADS + tid, n - 1);
            if (VEC4) {
                const float4 p = reinterpret_cast<const float4 *>(base)[i];
                ox[u] = p.x;
                oy[u] = p.y;
                oz[u] = p.z;
            } else {
                ox[u] = base[(size_t)i * pt_stride + 0];
                oy[u] = base[(size_t)i * pt_stride + 1];
                oz[u] = base[(size_t)i * pt_stride + 2];
            }
        }
    };
    constexpr int BATCH = VOX_LDS_THREADS * VOX_LDS_UNR;
    if (n > 0) load_batch(0, px, py, pz);
    for (int i0 = 0; i0 < n; i0 += BATCH) {
        float nx[VOX_LDS_UNR], ny[VOX_LDS_UNR], nz[VOX_LDS_UNR];
        const bool more = i0 + BATCH < n;   // workgroup-uniform
        if (more) load_batch(i0 + BATCH, nx, ny, nz);
#pragma unroll
        for (int u = 0; u < VOX_LDS_UNR; ++u)
            vox_lds_point(i0 + u * VOX_LDS_THREADS + tid < n, (double)px[u], (double)py[u], (double)pz[u], vp, sgrid);
        if (more) {
#pragma unroll
            for (int u = 0; u < VOX_LDS_UNR; ++u) {
                px[u] = nx[u];
                py[u] = ny[u];
                pz[u] = nz[u];
            }
        }
    }
    __syncthreads();
    // expand 16 -> 32 bits on the way out: LDS words 2k, 2k+1 -> pixels 4k .. 4k+3 (one 16-B store)
    uint32_t *grid = bits + (size_t)cloud * vp.X * vp.Y;
    for (int k = tid; k < (n_words >> 1); k += VOX_LDS_THREADS) {
        const uint2 w = *reinterpret_cast<const uint2 *>(&sgrid[2 * k]);
        *reinterpret_cast<uint4 *>(&grid[4 * k]) = make_uint4(w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16);
    }
}

// Early fusion (upperbound): job j scatters source cloud src[j], moved by the rigid transform xform[j] (row-major
// 3x4, fp32), into target grid dst[j].  The transform is evaluated in fp32 with every multiply and add rounded
// separately, in the fixed order ((x*m0 + y*m1) + z*m2) + m3 (no FMA contraction), so the oracle can restate it
// bit for bit; the voxel index of the moved point then follows the a1 spec (fp64 compare / divide).
__global__ __launch_bounds__(256) void voxelize_fused_scatter_kernel(const float *__restrict__ pts,
                                                                     const int32_t *__restrict__ n_pts, int max_pts,
                                                                     int pt_stride, const float *__restrict__ xform,
                                                                     const int32_t *__restrict__ src,
                                                                     const int32_t *__restrict__ dst, VoxParams vp,
                                                                     uint32_t *__restrict__ bits, int n_clouds,
                                                                     int n_grids) {
    const int job = blockIdx.y;
    const int cloud = src[job];
    // src / dst arrive as device data the host wrapper cannot inspect: a job naming a cloud or grid that does not
    // exist is skipped instead of reading / writing out of bounds
    if ((unsigned)cloud >= (unsigned)n_clouds || (unsigned)dst[job] >= (unsigned)n_grids) return;
    const int n = min(n_pts[cloud], max_pts);
    const float *base = pts + (size_t)cloud * max_pts * pt_stride;
    const float *m = xform + (size_t)job * 12;
    uint32_t *grid = bits + (size_t)dst[job] * vp.X * vp.Y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float px = base[(size_t)i * pt_stride + 0], py = base[(size_t)i * pt_stride + 1],
                    pz = base[(size_t)i * pt_stride + 2];
        float t[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            t[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(px, m[4 * r + 0]), __fmul_rn(py, m[4 * r + 1])),
                                       __fmul_rn(pz, m[4 * r + 2])), m[4 * r + 3]);
        const double x = (double)t[0], y = (double)t[1], z = (double)t[2];
        const bool keep = (vp.lo[0] < x) && (x < vp.hi[0]) && (vp.lo[1] < y) && (y < vp.hi[1]) &&
                          (vp.lo[2] < z) && (z < vp.hi[2]);
        if (!keep) continue;
        const int ix = (int)(floor(vox_quot(x, vp, 0)) - vp.mn[0]);
        const int iy = (int)(floor(vox_quot(y, vp, 1)) - vp.mn[1]);
        const int iz = (int)(floor(vox_quot(z, vp, 2)) - vp.mn[2]);
        if ((unsigned)ix >= (unsigned)vp.X || (unsigned)iy >= (unsigned)vp.Y || (unsigned)iz >= (unsigned)vp.Z)
            continue;
        atomicOr(&grid[(size_t)ix * vp.Y + iy], 1u << iz);
    }
}

// LDS-binned form of the early fusion (round 4): ONE workgroup per TARGET grid keeps that grid in LDS as 16-bit words, like voxelize_lds_kernel,
// and streams every job whose dst is its grid through it -- it scans the whole job list for them (<= 65 535 ints, broadcast loads), so the jobs need
// no order.  The scatter kernel above does 5 x 65 536 device-scope atomicOr per grid into L2 (1.18 ms per 320 grids, 1.1 TB/s: the one kernel of
// the upperbound config far from any roof); here every atomic is a ds_or_b32 and the grid leaves the CU once, as linear 16-byte stores -- no memset
// pass.  Arithmetic per point is the scatter kernel's (same fp32 transform with separately rounded operations, same fp64 indexing): bit-identical.
template <bool VEC4>
__global__ __launch_bounds__(VOX_LDS_THREADS) void voxelize_fused_lds_kernel(const float *__restrict__ pts, const int32_t *__restrict__ n_pts,
                                                                             int max_pts, int pt_stride, const float *__restrict__ xform,
                                                                             const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                                             int n_jobs, VoxParams vp, uint32_t *__restrict__ bits, int n_clouds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sgrid[];   // [X*Y/2]: two 16-bit pixels per word
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    const int n_words = (vp.X * vp.Y) >> 1;
    for (int i = tid * 4; i < n_words; i += VOX_LDS_THREADS * 4) *reinterpret_cast<uint4 *>(&sgrid[i]) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int job = 0; job < n_jobs; ++job) {
        if (dst[job] != g) continue;                       // workgroup-uniform
        const int cloud = src[job];
        if ((unsigned)cloud >= (unsigned)n_clouds) continue;
        const int n = min(n_pts[cloud], max_pts);
        const float *base = pts + (size_t)cloud * max_pts * pt_stride;
        float m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = xform[(size_t)job * 12 + k];
        for (int i0 = 0; i0 < n; i0 += VOX_LDS_THREADS * VOX_LDS_UNR) {
            float px[VOX_LDS_UNR], py[VOX_LDS_UNR], pz[VOX_LDS_UNR];
#pragma unroll
            for (int u = 0; u < VOX_LDS_UNR; ++u) {          // unconditional loads (index clamped): all UNR of them in flight
                const int i = min(i0 + u * VOX_LDS_THREADS + tid, n - 1);
                if (VEC4) {
                    const float4 p = reinterpret_cast<const float4 *>(base)[i];
                    px[u] = p.x;
                    py[u] = p.y;
                    pz[u] = p.z;
                } else {
                    px[u] = base[(size_t)i * pt_stride + 0];
                    py[u] = base[(size_t)i * pt_stride + 1];
                    pz[u] = base[(size_t)i * pt_stride + 2];
                }
            }
#pragma unroll
            for (int u = 0; u < VOX_LDS_UNR; ++u) {
                float t[3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    t[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(px[u], m[4 * r + 0]), __fmul_rn(py[u], m[4 * r + 1])), __fmul_rn(pz[u], m[4 * r + 2])),
                                     m[4 * r + 3]);
                vox_lds_point(i0 + u * VOX_LDS_THREADS + tid < n, (double)t[0], (double)t[1], (double)t[2], vp, sgrid);
            }
        }
    }
    __syncthreads();
    uint32_t *grid = bits + (size_t)g * vp.X * vp.Y;
    for (int k = tid; k < (n_words >> 1); k += VOX_LDS_THREADS) {
        const uint2 w = *reinterpret_cast<const uint2 *>(&sgrid[2 * k]);
        *reinterpret_cast<uint4 *>(&grid[4 * k]) = make_uint4(w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16);
    }
}

// [n][X][Y] words -> [n][X][Y][Z] fp32, one thread per output element (coalesced 4-B stores)
__global__ __launch_bounds__(256) void bits_to_dense_f32_kernel(const uint32_t *__restrict__ bits, size_t n_pix,
                                                                int Z, float *__restrict__ out) {
    const size_t total = n_pix * (size_t)Z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Z;
        const int z = (int)(i - pix * Z);
        out[i] = (float)((bits[pix] >> z) & 1u);
    }
}

// [n][X][Y] words -> NHWC bf16 [n][X][Y][c_pad]; one thread per 8-channel (16-B) group
__global__ __launch_bounds__(256) void bits_to_nhwc_bf16_kernel(const uint32_t *__restrict__ bits, size_t n_pix,
                                                                int Z, int c_pad, uint16_t *__restrict__ out) {
    const int groups = c_pad >> 3;
    const size_t total = n_pix * (size_t)groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / groups;
        const int g = (int)(i - pix * groups);
        const uint32_t zmask = (Z >= 32) ? 0xffffffffu : ((1u << Z) - 1u);
        const uint32_t w = (bits[pix] & zmask) >> (g * 8);
        uint4 v;
        const uint32_t one = 0x3f80u;  // bf16(1.0)
        v.x = ((w & 1u) ? one : 0u) | ((w & 2u) ? (one << 16) : 0u);
        v.y = ((w & 4u) ? one : 0u) | ((w & 8u) ? (one << 16) : 0u);
        v.z = ((w & 16u) ? one : 0u) | ((w & 32u) ? (one << 16) : 0u);
        v.w = ((w & 64u) ? one : 0u) | ((w & 128u) ? (one << 16) : 0u);
        *reinterpret_cast<uint4 *>(out + i * 8) = v;
    }
}

// dense fp32 [n][X][Y][Z] -> NHWC bf16 [n][X][Y][c_pad]; one thread per 8-channel group
__global__ __launch_bounds__(256) void dense_f32_to_nhwc_bf16_kernel(const float *__restrict__ bev, size_t n_pix,
                                                                     int Z, int c_pad, uint16_t *__restrict__ out) {
    const int groups = c_pad >> 3;
    const size_t total = n_pix * (size_t)groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / groups;
        const int g = (int)(i - pix * groups);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int z = g * 8 + e;
            f[e] = (z < Z) ? bev[pix * Z + z] : 0.f;
        }
        uint4 v;
        v.x = pack_bf16x2(f[0], f[1]);
        v.y = pack_bf16x2(f[2], f[3]);
        v.z = pack_bf16x2(f[4], f[5]);
        v.w = pack_bf16x2(f[6], f[7]);
        *reinterpret_cast<uint4 *>(out + i * 8) = v;
    }
}

// ---- ordered compaction: bits -> sorted (x,y,z) index list -------------------------------
__device__ __forceinline__ int block_reduce_sum(int v, int *sh) {
    // blockDim.x multiple of 64, <= 1024
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    int tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) tot += sh[w];
    __syncthreads();
    return tot;
}

__global__ void row_count_kernel(const uint32_t *__restrict__ bits, int X, int Y, int Z,
                                 int32_t *__restrict__ row_counts) {
    __shared__ int sh[16];
    const int x = blockIdx.x, cloud = blockIdx.y;
    const uint32_t zmask = (Z >= 32) ? 0xffffffffu : ((1u << Z) - 1u);
    int c = 0;
    for (int y = threadIdx.x; y < Y; y += blockDim.x) c += __popc(bits[((size_t)cloud * X + x) * Y + y] & zmask);
    const int tot = block_reduce_sum(c, sh);
    if (threadIdx.x == 0) row_counts[cloud * X + x] = tot;
}

__global__ void row_emit_kernel(const uint32_t *__restrict__ bits, int X, int Y, int Z,
                                const int32_t *__restrict__ row_counts, int32_t *__restrict__ idx, int cap,
                                int32_t *__restrict__ counts) {
    __shared__ int sh[16];
    __shared__ int wave_off[17];
    const int x = blockIdx.x, cloud = blockIdx.y;
    const uint32_t zmask = (Z >= 32) ? 0xffffffffu : ((1u << Z) - 1u);
    // offset of this x-row = sum of the counts of the rows before it
    int part = 0;
    for (int r = threadIdx.x; r < x; r += blockDim.x) part += row_counts[cloud * X + r];
    const int row_off = block_reduce_sum(part, sh);
    if (x == X - 1 && threadIdx.x == 0) counts[cloud] = row_off + row_counts[cloud * X + x];
    // blockDim.x >= Y: thread y owns pixel (x, y)
    const int y = threadIdx.x;
    const uint32_t w = (y < Y) ? (bits[((size_t)cloud * X + x) * Y + y] & zmask) : 0u;
    const int c = __popc(w);
    // inclusive scan within the wave
    int inc = c;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_off[wid + 1] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        wave_off[0] = 0;
        const int nw = blockDim.x >> 6;
        for (int k = 1; k <= nw; ++k) wave_off[k] += wave_off[k - 1];
    }
    __syncthreads();
    int pos = row_off + wave_off[wid] + inc - c;
    uint32_t ww = w;
    while (ww) {
        const int z = __ffs(ww) - 1;
        ww &= ww - 1;
        if (pos < cap) {
            int32_t *o = idx + ((size_t)cloud * cap + pos) * 3;
            o[0] = x;
            o[1] = y;
            o[2] = z;
        }
        ++pos;
    }
}

// ---- densify: sparse voxel indices (the parsed dataset's storage format) -> bit grid -------------
__global__ __launch_bounds__(256) void indices_to_bits_kernel(const int32_t *__restrict__ idx,
                                                              const int32_t *__restrict__ counts, int cap, int X,
                                                              int Y, int Z, uint32_t *__restrict__ bits) {
    const int cloud = blockIdx.y;
    const int n = min(counts[cloud], cap);
    const int32_t *src = idx + (size_t)cloud * cap * 3;
    uint32_t *grid = bits + (size_t)cloud * X * Y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = src[3 * i + 0], y = src[3 * i + 1], z = src[3 * i + 2];
        // numpy fancy indexing would raise IndexError on an out-of-range index; we drop it
        if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z)
            atomicOr(&grid[(size_t)x * Y + y], 1u << z);
    }
}

static inline int grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;  // grid-stride the rest (guide: cap at ~2048 blocks)
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int v2x_voxelize_bits(const float *pts, const int32_t *n_pts, int n_clouds, int max_pts, int pt_stride,
                                 const double *extents, const double *voxel, const int32_t *dims_xyz,
                                 uint32_t *bits, v2x_stream_t stream) {
    V2X_REQUIRE(pts && n_pts && extents && voxel && dims_xyz && bits, "v2x_voxelize_bits: null pointer");
    V2X_REQUIRE(n_clouds >= 0 && max_pts >= 0 && pt_stride >= 3, "v2x_voxelize_bits: bad sizes (stride >= 3)");
    V2X_REQUIRE(dims_xyz[2] >= 1 && dims_xyz[2] <= 32, "v2x_voxelize_bits: Z=%d must be in [1,32]", dims_xyz[2]);
    V2X_REQUIRE(dims_xyz[0] >= 1 && dims_xyz[1] >= 1, "v2x_voxelize_bits: bad grid dims");
    if (n_clouds == 0) return V2X_OK;
    hipStream_t s = (hipStream_t)stream;
    VoxParams vp;
    for (int a = 0; a < 3; ++a) {
        vp.lo[a] = extents[2 * a];
        vp.hi[a] = extents[2 * a + 1];
        vp.vs[a] = voxel[a];
        V2X_REQUIRE(voxel[a] > 0.0, "v2x_voxelize_bits: voxel size must be > 0");
        vox_reciprocal(vp, a);
        vp.mn[a] = floor(extents[2 * a] / voxel[a]);
    }
    vp.X = dims_xyz[0];
    vp.Y = dims_xyz[1];
    vp.Z = dims_xyz[2];
    // LDS-binned form: the cloud's grid as 16-bit words must fit one CU's LDS (and split into 16-B pieces)
    const size_t lds_bytes = (size_t)vp.X * vp.Y * 2;
    // (tests switch VOXELIZE_LDS to compare the two forms).  The LDS form is one workgroup per cloud: with few clouds most CUs idle and
    // the global-atomic form -- many workgroups per cloud, bit-identical bits -- is faster (measured: 5 clouds 61 -> 21 us, 40 clouds 64 ->
    // 53 us, 320 clouds 152 vs 310 us); VOXELIZE_LDS = 2 forces the LDS form at any count.
    const int lds_mode = v2x_tune(V2X_TUNE_VOXELIZE_LDS);
    const bool lds_off = lds_mode == 0 || (lds_mode == 1 && n_clouds <= 48);
    if (!lds_off && vp.Z <= 16 && lds_bytes <= 128 * 1024 && ((size_t)vp.X * vp.Y) % 8 == 0 && max_pts > 0 &&
        (reinterpret_cast<uintptr_t>(bits) & 15) == 0) {
        const bool vec4 = pt_stride == 4 && (reinterpret_cast<uintptr_t>(pts) & 15) == 0;
        static v2x_once_per_device attr_once;
        if (v2x_first_use_on_device(attr_once)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(voxelize_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(voxelize_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        }
        if (vec4)
            hipLaunchKernelGGL(voxelize_lds_kernel<true>, dim3(n_clouds), dim3(VOX_LDS_THREADS), lds_bytes, s, pts, n_pts, max_pts, pt_stride, vp, bits);
        else
            hipLaunchKernelGGL(voxelize_lds_kernel<false>, dim3(n_clouds), dim3(VOX_LDS_THREADS), lds_bytes, s, pts, n_pts, max_pts, pt_stride, vp, bits);
        V2X_CHECK_LAUNCH("voxelize_lds_kernel");
        return V2X_OK;
    }
    if (hipMemsetAsync(bits, 0, (size_t)n_clouds * vp.X * vp.Y * sizeof(uint32_t), s) != hipSuccess) {
        v2x_set_error("v2x_voxelize_bits: memset failed");
        return V2X_EIO;
    }
    if (max_pts == 0) return V2X_OK;
    dim3 grid((max_pts + 255) / 256, n_clouds);
    if (grid.x > 512) grid.x = 512;
    hipLaunchKernelGGL(voxelize_scatter_kernel, grid, dim3(256), 0, s, pts, n_pts, max_pts, pt_stride, vp, bits);
    V2X_CHECK_LAUNCH("voxelize_scatter_kernel");
    return V2X_OK;
}

extern "C" int v2x_bits_to_dense_f32(const uint32_t *bits, int n, int X, int Y, int Z, float *out,
                                     v2x_stream_t stream) {
    V2X_REQUIRE(bits && out, "v2x_bits_to_dense_f32: null pointer");
    V2X_REQUIRE(n >= 0 && X > 0 && Y > 0 && Z > 0 && Z <= 32, "v2x_bits_to_dense_f32: bad dims");
    if (n == 0) return V2X_OK;
    const size_t n_pix = (size_t)n * X * Y;
    hipLaunchKernelGGL(bits_to_dense_f32_kernel, dim3(grid_for(n_pix * Z, 256)), dim3(256), 0, (hipStream_t)stream,
                       bits, n_pix, Z, out);
    V2X_CHECK_LAUNCH("bits_to_dense_f32_kernel");
    return V2X_OK;
}

extern "C" int v2x_bits_to_nhwc_bf16(const uint32_t *bits, int n, int X, int Y, int Z, int c_pad, uint16_t *out,
                                     v2x_stream_t stream) {
    V2X_REQUIRE(bits && out, "v2x_bits_to_nhwc_bf16: null pointer");
    V2X_REQUIRE(n >= 0 && X > 0 && Y > 0 && Z > 0 && Z <= 32, "v2x_bits_to_nhwc_bf16: bad dims");
    V2X_REQUIRE(c_pad >= Z && c_pad % 8 == 0 && c_pad <= 32, "v2x_bits_to_nhwc_bf16: c_pad=%d must be a multiple of 8 in [Z,32]", c_pad);
    if (n == 0) return V2X_OK;
    const size_t n_pix = (size_t)n * X * Y;
    hipLaunchKernelGGL(bits_to_nhwc_bf16_kernel, dim3(grid_for(n_pix * (c_pad / 8), 256)), dim3(256), 0,
                       (hipStream_t)stream, bits, n_pix, Z, c_pad, out);
    V2X_CHECK_LAUNCH("bits_to_nhwc_bf16_kernel");
    return V2X_OK;
}

extern "C" int v2x_dense_f32_to_nhwc_bf16(const float *bev, int n, int X, int Y, int Z, int c_pad, uint16_t *out,
                                          v2x_stream_t stream) {
    V2X_REQUIRE(bev && out, "v2x_dense_f32_to_nhwc_bf16: null pointer");
    V2X_REQUIRE(n >= 0 && X > 0 && Y > 0 && Z > 0, "v2x_dense_f32_to_nhwc_bf16: bad dims");
    V2X_REQUIRE(c_pad >= Z && c_pad % 8 == 0, "v2x_dense_f32_to_nhwc_bf16: c_pad=%d must be a multiple of 8 >= Z", c_pad);
    if (n == 0) return V2X_OK;
    const size_t n_pix = (size_t)n * X * Y;
    hipLaunchKernelGGL(dense_f32_to_nhwc_bf16_kernel, dim3(grid_for(n_pix * (c_pad / 8), 256)), dim3(256), 0,
                       (hipStream_t)stream, bev, n_pix, Z, c_pad, out);
    V2X_CHECK_LAUNCH("dense_f32_to_nhwc_bf16_kernel");
    return V2X_OK;
}

extern "C" int v2x_bits_to_indices(const uint32_t *bits, int n, int X, int Y, int Z, int32_t *idx, int cap,
                                   int32_t *counts, int32_t *scratch, v2x_stream_t stream) {
    V2X_REQUIRE(bits && idx && counts && scratch, "v2x_bits_to_indices: null pointer");
    V2X_REQUIRE(n >= 0 && X > 0 && Y > 0 && Y <= 1024 && Z > 0 && Z <= 32 && cap >= 0, "v2x_bits_to_indices: bad dims (Y <= 1024)");
    if (n == 0) return V2X_OK;
    const int block = ((Y + 63) / 64) * 64;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_count_kernel, dim3(X, n), dim3(block), 0, s, bits, X, Y, Z, scratch);
    V2X_CHECK_LAUNCH("row_count_kernel");
    hipLaunchKernelGGL(row_emit_kernel, dim3(X, n), dim3(block), 0, s, bits, X, Y, Z, scratch, idx, cap, counts);
    V2X_CHECK_LAUNCH("row_emit_kernel");
    return V2X_OK;
}

extern "C" int v2x_indices_to_bits(const int32_t *idx, const int32_t *counts, int n, int cap, int X, int Y, int Z,
                                   uint32_t *bits, v2x_stream_t stream) {
    V2X_REQUIRE(idx && counts && bits, "v2x_indices_to_bits: null pointer");
    V2X_REQUIRE(n >= 0 && cap >= 0 && X > 0 && Y > 0 && Z > 0 && Z <= 32, "v2x_indices_to_bits: bad dims");
    if (n == 0) return V2X_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(bits, 0, (size_t)n * X * Y * sizeof(uint32_t), s) != hipSuccess) {
        v2x_set_error("v2x_indices_to_bits: memset failed");
        return V2X_EIO;
    }
    if (cap == 0) return V2X_OK;
    dim3 grid((cap + 255) / 256, n);
    if (grid.x > 512) grid.x = 512;
    hipLaunchKernelGGL(indices_to_bits_kernel, grid, dim3(256), 0, s, idx, counts, cap, X, Y, Z, bits);
    V2X_CHECK_LAUNCH("indices_to_bits_kernel");
    return V2X_OK;
}

extern "C" int v2x_voxelize_fused_bits(const float *pts, const int32_t *n_pts, int n_clouds, int max_pts, int pt_stride,
                                       const float *xform, const int32_t *src_cloud, const int32_t *dst_grid, int n_jobs,
                                       int n_grids, const double *extents, const double *voxel,
                                       const int32_t *dims_xyz, uint32_t *bits, v2x_stream_t stream) {
    V2X_REQUIRE(pts && n_pts && xform && src_cloud && dst_grid && extents && voxel && dims_xyz && bits,
                "v2x_voxelize_fused_bits: null pointer");
    V2X_REQUIRE(n_clouds >= 0 && max_pts >= 0 && pt_stride >= 3 && n_jobs >= 0 && n_grids >= 0 && n_jobs <= 65535,
                "v2x_voxelize_fused_bits: bad sizes");
    V2X_REQUIRE(dims_xyz[2] >= 1 && dims_xyz[2] <= 32 && dims_xyz[0] >= 1 && dims_xyz[1] >= 1,
                "v2x_voxelize_fused_bits: bad grid dims");
    if (n_grids == 0) return V2X_OK;
    hipStream_t s = (hipStream_t)stream;
    VoxParams vp;
    for (int a = 0; a < 3; ++a) {
        vp.lo[a] = extents[2 * a];
        vp.hi[a] = extents[2 * a + 1];
        vp.vs[a] = voxel[a];
        V2X_REQUIRE(voxel[a] > 0.0, "v2x_voxelize_fused_bits: voxel size must be > 0");
        vox_reciprocal(vp, a);
        vp.mn[a] = floor(extents[2 * a] / voxel[a]);
    }
    vp.X = dims_xyz[0];
    vp.Y = dims_xyz[1];
    vp.Z = dims_xyz[2];
    // LDS-binned form: one workgroup per target grid (the grid as 16-bit words must fit one CU's LDS); VOXELIZE_LDS = 0 keeps the global-atomic form
    // (the bitwise-equality test compares the two), = 1 takes it from 48 grids on like v2x_voxelize_bits, = 2 always
    const size_t lds_bytes = (size_t)vp.X * vp.Y * 2;
    const int lds_mode = v2x_tune(V2X_TUNE_VOXELIZE_LDS);
    if (lds_mode != 0 && !(lds_mode == 1 && n_grids <= 48) && vp.Z <= 16 && lds_bytes <= 128 * 1024 && ((size_t)vp.X * vp.Y) % 8 == 0 && max_pts > 0 &&
        n_jobs > 0) {
        static v2x_once_per_device attr_once;
        if (v2x_first_use_on_device(attr_once)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(voxelize_fused_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(voxelize_fused_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        }
        if (pt_stride == 4 && (reinterpret_cast<uintptr_t>(pts) & 15) == 0)
            hipLaunchKernelGGL(voxelize_fused_lds_kernel<true>, dim3(n_grids), dim3(VOX_LDS_THREADS), lds_bytes, s, pts, n_pts, max_pts, pt_stride, xform,
                               src_cloud, dst_grid, n_jobs, vp, bits, n_clouds);
        else
            hipLaunchKernelGGL(voxelize_fused_lds_kernel<false>, dim3(n_grids), dim3(VOX_LDS_THREADS), lds_bytes, s, pts, n_pts, max_pts, pt_stride, xform,
                               src_cloud, dst_grid, n_jobs, vp, bits, n_clouds);
        V2X_CHECK_LAUNCH("voxelize_fused_lds_kernel");
        return V2X_OK;
    }
    if (hipMemsetAsync(bits, 0, (size_t)n_grids * vp.X * vp.Y * sizeof(uint32_t), s) != hipSuccess) {
        v2x_set_error("v2x_voxelize_fused_bits: memset failed");
        return V2X_EIO;
    }
    if (max_pts == 0 || n_jobs == 0) return V2X_OK;
    dim3 grid((max_pts + 255) / 256, n_jobs);
    if (grid.x > 256) grid.x = 256;
    hipLaunchKernelGGL(voxelize_fused_scatter_kernel, grid, dim3(256), 0, s, pts, n_pts, max_pts, pt_stride, xform,
                       src_cloud, dst_grid, vp, bits, n_clouds, n_grids);
    V2X_CHECK_LAUNCH("voxelize_fused_scatter_kernel");
    return V2X_OK;
}
