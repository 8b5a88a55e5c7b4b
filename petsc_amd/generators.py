"""Synthetic matrices of the BASELINE.json configs, as host CSR (int32 / float64), rows in natural order (x fastest).

They stand in for DMDA-assembled matrices (src/dm/impls/da/fdda.c:1024 preallocates the full stencil, truncated at the
domain boundary, and keeps structural zeros); the reference's DM layer is out of scope.
Every generator accepts a row range [rstart, rend) so that a rank can build just the rows MPIAIJ gives it.
"""
import numpy as np


def _stencil_csr(dims, offsets, value_fn, rstart=0, rend=None):
    nx, ny, nz = dims
    N = nx * ny * nz
    rend = N if rend is None else rend
    rows = np.arange(rstart, rend, dtype=np.int64)
    i = rows % nx
    j = (rows // nx) % ny
    k = rows // (nx * ny)
    offs = sorted(offsets, key=lambda o: (o[2], o[1], o[0]))  # ascending column order
    m = len(rows)
    cols = np.empty((m, len(offs)), dtype=np.int32)
    mask = np.empty((m, len(offs)), dtype=bool)
    vals = np.empty((m, len(offs)), dtype=np.float64)
    for c, (dx, dy, dz) in enumerate(offs):
        ii, jj, kk = i + dx, j + dy, k + dz
        ok = (ii >= 0) & (ii < nx) & (jj >= 0) & (jj < ny) & (kk >= 0) & (kk < nz)
        mask[:, c] = ok
        cols[:, c] = np.where(ok, ii + nx * (jj + ny * kk), 0).astype(np.int32)
        vals[:, c] = value_fn(dx, dy, dz, i, j, k)
    ai = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(mask.sum(axis=1), out=ai[1:])
    return ai.astype(np.int32), cols[mask], vals[mask]


def laplace2d_5pt(m, n, rstart=0, rend=None):
    """src/ksp/ksp/examples/tutorials/ex2.c:88-95: I = j + n*i, diag 4, -1 to I-n, I+n, I-1, I+1 (config 1)."""
    offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0)]

    def val(dx, dy, dz, i, j, k):
        return np.full(i.shape, 4.0 if (dx, dy) == (0, 0) else -1.0)

    return _stencil_csr((n, m, 1), offs, val, rstart, rend)


def laplace3d_7pt(nx, ny=None, nz=None, rstart=0, rend=None, dirichlet_rows=True):
    """Config 2 / ex45.c:126-146: interior rows -HxHy/Hz, -HxHz/Hy, -HyHz/Hx, diag 2(sum); boundary rows keep the
    (truncated) stencil pattern with zeros and the same diagonal.  dirichlet_rows=False gives the plain SPD 7-point
    Laplacian (diag 6, off -1) on the same pattern."""
    ny = ny or nx
    nz = nz or nx
    offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    Hx, Hy, Hz = 1.0 / (nx - 1), 1.0 / (ny - 1), 1.0 / (nz - 1)
    cz, cy, cx = Hx * Hy / Hz, Hx * Hz / Hy, Hy * Hz / Hx
    diag = 2.0 * (cz + cy + cx)

    def val(dx, dy, dz, i, j, k):
        if not dirichlet_rows:
            return np.full(i.shape, 6.0 if (dx, dy, dz) == (0, 0, 0) else -1.0)
        bnd = (i == 0) | (j == 0) | (k == 0) | (i == nx - 1) | (j == ny - 1) | (k == nz - 1)
        if (dx, dy, dz) == (0, 0, 0):
            return np.full(i.shape, diag)
        c = -cx if dx else (-cy if dy else -cz)
        return np.where(bnd, 0.0, c)

    return _stencil_csr((nx, ny, nz), offs, val, rstart, rend)


def box3d_27pt(nx, ny=None, nz=None, rstart=0, rend=None):
    """Config 3: 27-point box stencil, diag 26, off -1 (SPD, synthetic)."""
    ny = ny or nx
    nz = nz or nx
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]

    def val(dx, dy, dz, i, j, k):
        return np.full(i.shape, 26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0)

    return _stencil_csr((nx, ny, nz), offs, val, rstart, rend)


def convdiff3d_7pt(nx, ny=None, nz=None, c=0.5, rstart=0, rend=None):
    """Config 4: -Laplace(u) + beta.grad(u), first-order upwind: diag 6+3c, west/south/down -1-c, east/north/up -1."""
    ny = ny or nx
    nz = nz or nx
    offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]

    def val(dx, dy, dz, i, j, k):
        if (dx, dy, dz) == (0, 0, 0):
            return np.full(i.shape, 6.0 + 3.0 * c)
        return np.full(i.shape, -1.0 - c if (dx + dy + dz) < 0 else -1.0)

    return _stencil_csr((nx, ny, nz), offs, val, rstart, rend)


def rmat(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, seed=20240601, chunk=1 << 23):
    """Config 5: R-MAT (a,b,c,d) power-law matrix, 2^scale rows, edge_factor*2^scale samples, duplicates summed,
    self-loops kept, values uniform(0,1).  Edges are drawn in chunks, keyed row*2^scale+col, sorted once and merged."""
    n = 1 << scale
    ne = edge_factor * n
    rng = np.random.default_rng(seed)
    keys = np.empty(ne, dtype=np.uint64)
    vals = np.empty(ne, dtype=np.float64)
    done = 0
    while done < ne:
        cnt = min(chunk, ne - done)
        r = np.zeros(cnt, dtype=np.uint64)
        cc = np.zeros(cnt, dtype=np.uint64)
        for _ in range(scale):
            u = rng.random(cnt)
            rbit = u >= a + b
            cbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
            r = (r << np.uint64(1)) | rbit.astype(np.uint64)
            cc = (cc << np.uint64(1)) | cbit.astype(np.uint64)
        keys[done:done + cnt] = (r << np.uint64(scale)) | cc
        vals[done:done + cnt] = rng.random(cnt)
        done += cnt
    order = np.argsort(keys, kind="stable")
    keys = keys[order]
    vals = vals[order]
    del order
    first = np.empty(ne, dtype=bool)
    first[0] = True
    np.not_equal(keys[1:], keys[:-1], out=first[1:])
    starts = np.flatnonzero(first)
    aa = np.add.reduceat(vals, starts)
    ukeys = keys[starts]
    del keys, vals, first
    rows = (ukeys >> np.uint64(scale)).astype(np.int64)
    aj = (ukeys & np.uint64(n - 1)).astype(np.int32)
    ai = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ai[1:])
    return ai.astype(np.int32), aj, aa.astype(np.float64)


def random_vector(n, seed=12345):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


def stencil_rows_native(kind, nx, ny, nz, rstart=0, rend=None):
    """The constant-coefficient 5- / 7- / 27-point matrices above (laplace2d_5pt(ny, nx), laplace3d_7pt(dirichlet_rows=False), box3d_27pt)
    built by the C loop of libpetschip_host.so (PetscHipStencilRows): the same arrays, 10-20 times faster at 256^3.  bench.py uses it."""
    import ctypes as C
    from . import petsc as P
    L = P.lib()
    N = nx * ny * nz
    rend = N if rend is None else rend
    m = rend - rstart
    ai = np.zeros(m + 1, np.int32)
    P.chk(L.PetscHipStencilRows(kind, nx, ny, nz, C.c_int64(rstart), C.c_int64(rend), ai.ctypes.data_as(C.c_void_p), None, None))
    aj = np.empty(int(ai[-1]), np.int32)
    aa = np.empty(int(ai[-1]), np.float64)
    P.chk(L.PetscHipStencilRows(kind, nx, ny, nz, C.c_int64(rstart), C.c_int64(rend), ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p)))
    return ai, aj, aa
