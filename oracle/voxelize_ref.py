"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the
point-cloud -> BEV occupancy voxel scatter.

PARITY UNPINNED: the reference checkout (/root/reference) contains only
README.md and .gitmodules; the code that implements this stage lives in the
un-vendored, un-pinned third-party submodule `coperception`
(/root/reference/.gitmodules:1-3, /root/reference/README.md:45 "parse this
dataset yourself with create_data.py", README.md:101).  This file restates the
published algorithm of upstream `coperception/utils/data_util.py::
voxelize_occupy` and the densify step of `coperception/datasets/V2XSimDet.py::
__getitem__` from recollection; no reference golden vector exists to pin it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Frozen numerical spec:
  * a point is kept iff  lo < p < hi  strictly on every axis, compared after
    exact promotion of the fp32 coordinate to fp64 (extents are fp64);
  * voxel coordinate  = floor(fp64(p) / fp64(voxel))  -- numpy promotes a
    float32 array divided by a tuple of python floats to float64, so the
    division is an IEEE fp64 division (matters for z: 0.4 is inexact);
  * index             = coordinate - floor(lo / voxel)  (fp64), as int32;
  * grid dims         = ceil(hi / voxel) - 1 - floor(lo / voxel) + 1;
  * duplicates collapse; indices are emitted sorted lexicographically (x,y,z).
"""
import numpy as np

VOXEL_SIZE = (0.25, 0.25, 0.4)
AREA_EXTENTS = np.array([[-32.0, 32.0], [-32.0, 32.0], [-3.0, 2.0]])


def grid_dims(voxel_size=VOXEL_SIZE, extents=AREA_EXTENTS):
    vs = np.asarray(voxel_size, dtype=np.float64)
    mn = np.floor(extents.T[0] / vs)
    mx = np.ceil(extents.T[1] / vs) - 1
    return ((mx - mn) + 1).astype(np.int32)


def voxelize_occupy(pts, voxel_size=VOXEL_SIZE, extents=AREA_EXTENTS, return_indices=False):
    """Restates upstream voxelize_occupy (lexsort + unique formulation).

    pts: (N, >=3) float32.  Returns the dense float32 occupancy grid of shape
    grid_dims() and, optionally, the sorted unique (M, 3) voxel indices.
    """
    pts = np.asarray(pts)
    if extents.shape != (3, 2):
        raise ValueError("Extents are the wrong shape {}".format(extents.shape))
    keep = np.where(
        (extents[0, 0] < pts[:, 0]) & (pts[:, 0] < extents[0, 1])
        & (extents[1, 0] < pts[:, 1]) & (pts[:, 1] < extents[1, 1])
        & (extents[2, 0] < pts[:, 2]) & (pts[:, 2] < extents[2, 1])
    )[0]
    pts = pts[keep]
    # float32 array / tuple-of-python-floats -> float64 division
    discrete = np.floor(pts[:, :3] / voxel_size).astype(np.int32)
    order = np.lexsort((discrete[:, 2], discrete[:, 1], discrete[:, 0]))
    discrete = discrete[order]
    contiguous = np.ascontiguousarray(discrete).view(
        np.dtype((np.void, discrete.dtype.itemsize * discrete.shape[1])))
    _, uniq = np.unique(contiguous, return_index=True)
    uniq.sort()
    voxel_coords = discrete[uniq]
    mn = np.floor(extents.T[0] / voxel_size)
    mx = np.ceil(extents.T[1] / voxel_size) - 1
    num_divisions = ((mx - mn) + 1).astype(np.int32)
    voxel_indices = (voxel_coords - mn).astype(int)
    leaf = np.zeros(num_divisions.astype(int), dtype=np.float32)
    leaf[voxel_indices[:, 0], voxel_indices[:, 1], voxel_indices[:, 2]] = 1.0
    if return_indices:
        return leaf, voxel_indices
    return leaf


def densify(voxel_indices, dims):
    """Restates the Dataset densify: sparse (M,3) indices -> bool grid."""
    grid = np.zeros(tuple(int(d) for d in dims), dtype=bool)
    grid[voxel_indices[:, 0], voxel_indices[:, 1], voxel_indices[:, 2]] = 1
    return grid


def voxelize_direct(pts, voxel_size=VOXEL_SIZE, extents=AREA_EXTENTS):
    """Sort-free formulation of the same spec (used to cross-check the
    lexsort/unique restatement above and the C restatement)."""
    pts = np.asarray(pts)
    vs = np.asarray(voxel_size, dtype=np.float64)
    p = pts[:, :3].astype(np.float64)
    keep = np.all((extents[:, 0] < p) & (p < extents[:, 1]), axis=1)
    p = p[keep]
    mn = np.floor(extents.T[0] / vs)
    idx = (np.floor(p / vs) - mn).astype(np.int64)
    dims = grid_dims(voxel_size, extents)
    grid = np.zeros(tuple(int(d) for d in dims), dtype=np.float32)
    grid[idx[:, 0], idx[:, 1], idx[:, 2]] = 1.0
    return grid


def synthetic_points(n, seed, n_dup_frac=0.01, n_edge=64):
    """SURVEY.md section 8(d) synthetic sweep: x,y~U(-40,40), z~U(-5,4) fp32,
    plus ~1 % exact duplicates and `n_edge` points exactly on extents / voxel
    boundaries (exercise strict '<' and floor on exact multiples)."""
    rng = np.random.default_rng(seed)
    n_dup = int(n * n_dup_frac)
    n_rand = n - n_dup - n_edge
    pts = np.empty((n, 4), dtype=np.float32)
    pts[:n_rand, 0] = rng.uniform(-40, 40, n_rand)
    pts[:n_rand, 1] = rng.uniform(-40, 40, n_rand)
    pts[:n_rand, 2] = rng.uniform(-5, 4, n_rand)
    pts[:n_rand, 3] = rng.uniform(0, 1, n_rand)
    # exact duplicates of earlier points
    src = rng.integers(0, n_rand, n_dup)
    pts[n_rand:n_rand + n_dup] = pts[src]
    # boundary points: on extents, on voxel boundaries (multiples of voxel size)
    e = np.zeros((n_edge, 4), dtype=np.float32)
    xs = np.array([-32.0, 32.0, -31.75, 31.75, 0.0, 0.25, -0.25, 31.999998, -31.999998], dtype=np.float32)
    zs = np.array([-3.0, 2.0, -2.8, -2.4, -2.0, 0.0, 0.4, 0.8, 1.2, 1.6, 1.9999999, -2.9999998,
                   np.float32(0.4) * 3, np.float32(0.4) * -7], dtype=np.float32)
    for i in range(n_edge):
        e[i, 0] = xs[rng.integers(0, len(xs))]
        e[i, 1] = xs[rng.integers(0, len(xs))]
        e[i, 2] = zs[rng.integers(0, len(zs))]
    pts[n_rand + n_dup:] = e
    return pts


def transform_points_f32(pts, T):
    """Early-fusion spec: rigid transform of fp32 points by the fp32 matrix T (3x4 or 4x4),
    every product and sum rounded to fp32, in the order ((x*m0 + y*m1) + z*m2) + m3."""
    p = np.asarray(pts, dtype=np.float32)
    m = np.asarray(T, dtype=np.float32)
    out = p.copy()
    for r in range(3):
        acc = (p[:, 0] * m[r, 0]).astype(np.float32)
        acc = (acc + (p[:, 1] * m[r, 1]).astype(np.float32)).astype(np.float32)
        acc = (acc + (p[:, 2] * m[r, 2]).astype(np.float32)).astype(np.float32)
        out[:, r] = (acc + m[r, 3]).astype(np.float32)
    return out


def voxelize_early_fusion(clouds, transforms, voxel_size=VOXEL_SIZE, extents=AREA_EXTENTS):
    """Upperbound input of one ego: union of all agents' sweeps moved into the ego frame, then voxelize_occupy."""
    merged = np.concatenate([transform_points_f32(c, T) for c, T in zip(clouds, transforms)], axis=0)
    return voxelize_occupy(merged, voxel_size, extents)
