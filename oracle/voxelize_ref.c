/* ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the
 * point-cloud -> BEV occupancy voxel scatter.
 *
 * PARITY UNPINNED: /root/reference holds no code (README.md + .gitmodules
 * only; the implementation lives in the un-vendored, un-pinned `coperception`
 * submodule, /root/reference/.gitmodules:1-3, README.md:45, README.md:101).
 * This restates upstream coperception/utils/data_util.py::voxelize_occupy
 * from its published algorithm; see oracle/voxelize_ref.py for the numpy
 * restatement it is cross-checked against (tests/test_oracle_voxel.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Scalar, single-threaded: cpu_baseline "cores" = 1.
 *
 * Spec: keep iff lo < p < hi strictly (fp64 compare of the promoted fp32);
 * idx = (int)(floor((double)p / voxel) - floor(lo / voxel)); occupancy only.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* pts: n x stride floats (x,y,z first); extents: {xlo,xhi,ylo,yhi,zlo,zhi};
 * voxel: {vx,vy,vz} as doubles; dims: {X,Y,Z}; occ: X*Y*Z bytes (x-major,
 * z fastest -- the (256,256,13) layout of the reference's padded_voxel_points).
 * Returns number of points kept by the range filter. */
int64_t oracle_voxelize_occupy(const float *pts, int64_t n, int stride,
                               const double *extents, const double *voxel,
                               const int *dims, uint8_t *occ)
{
    const double mnx = floor(extents[0] / voxel[0]);
    const double mny = floor(extents[2] / voxel[1]);
    const double mnz = floor(extents[4] / voxel[2]);
    int64_t kept = 0;
    memset(occ, 0, (size_t)dims[0] * dims[1] * dims[2]);
    for (int64_t i = 0; i < n; ++i) {
        const double x = (double)pts[i * stride + 0];
        const double y = (double)pts[i * stride + 1];
        const double z = (double)pts[i * stride + 2];
        if (!(extents[0] < x && x < extents[1])) continue;
        if (!(extents[2] < y && y < extents[3])) continue;
        if (!(extents[4] < z && z < extents[5])) continue;
        const int ix = (int)(floor(x / voxel[0]) - mnx);
        const int iy = (int)(floor(y / voxel[1]) - mny);
        const int iz = (int)(floor(z / voxel[2]) - mnz);
        occ[((size_t)ix * dims[1] + iy) * dims[2] + iz] = 1;
        ++kept;
    }
    return kept;
}

/* Emits the occupied voxel indices in lexicographic (x,y,z) order, as the
 * reference's voxelize_occupy(return_indices=True) does.  Returns the count. */
int64_t oracle_occupancy_indices(const uint8_t *occ, const int *dims, int32_t *out_idx)
{
    int64_t m = 0;
    for (int x = 0; x < dims[0]; ++x)
        for (int y = 0; y < dims[1]; ++y)
            for (int z = 0; z < dims[2]; ++z)
                if (occ[((size_t)x * dims[1] + y) * dims[2] + z]) {
                    out_idx[3 * m + 0] = x;
                    out_idx[3 * m + 1] = y;
                    out_idx[3 * m + 2] = z;
                    ++m;
                }
    return m;
}
