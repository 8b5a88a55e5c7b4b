"""CPU tests of the host layer (libpetschip_host.so) on planning-only contexts: assembly products, PetscBinary IO, error
behaviour, and -- in world_size 2/3 gloo process groups -- the MPIAIJ split, garray and the VecScatter plan, all bit-exact
against the golden fixtures and the oracle.  No numerical kernel runs here (there is no CPU fallback to run)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["tiny", "small", "medium", "arco1", "poisson1", "spd-real-int32-float64", "ns-real-int32-float64", "poisson2", "fem1", "afiro", "golan", "A4", "P4"]


@pytest.fixture(scope="module")
def comm():
    from petsc_amd import petsc as P
    c = P.Comm(device=-1, backend="self")
    yield c
    c.finalize()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, "mat_%s.npz" % name), allow_pickle=False)


@pytest.mark.parametrize("name", NAMES)
def test_seq_assembly_products(comm, orc, golden_dir, name):
    """MatSetValues in shuffled order with too little preallocation + MatAssemblyEnd == the file's CSR, bit for bit."""
    from petsc_amd import petsc as P
    g = load(golden_dir, name)
    ai, aj, aa, M, N = g["i"], g["j"], g["a"], int(g["M"]), int(g["N"])
    A = P.Mat(comm, M, N, M, N, "aijhip")
    assert A.type() == "seqaijhip"  # base name -> seq type on one rank (matreg.c:47-57)
    A.seqaij_set_preallocation(2)
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(ai))
    for k in np.random.default_rng(7).permutation(len(aj)):
        A.set_values([rows[k]], [aj[k]], [aa[k]])
    A.assemble()
    blk = A.block(0)
    np.testing.assert_array_equal(blk["i"], ai)
    np.testing.assert_array_equal(blk["j"], aj)
    np.testing.assert_array_equal(blk["a"], aa)
    np.testing.assert_array_equal(blk["diag"], g["diagpos"])
    assert blk["rmax"] == int(np.diff(ai).max()) and blk["nonzerorowcnt"] == int(np.count_nonzero(np.diff(ai)))
    use, cpi, ridx = orc.check_compressed_row(ai)
    assert blk["cprow_use"] == use
    # ADD_VALUES onto existing entries, dense block insertion
    A.set_values([rows[0], rows[0]], [aj[0]], [1.5, 2.5], P.ADD_VALUES)  # rows[0]: the first row that has an entry (A4's row 0 has none)
    A.assemble()
    assert A.block(0)["a"][0] == aa[0] + 1.5 + 2.5
    A.destroy()


def test_seq_with_arrays_and_cprow(comm, orc):
    from petsc_amd import petsc as P
    ai = np.array([0, 0, 2, 2, 2, 3, 3, 3, 3, 3, 3], dtype=np.int32)
    aj = np.array([1, 4, 0], dtype=np.int32)
    A = P.Mat.seqaij_with_arrays(comm, 10, 10, ai, aj, np.array([1.0, 2.0, 3.0]))
    blk = A.block(0)
    assert blk["cprow_use"]
    np.testing.assert_array_equal(blk["cprow_i"], [0, 2, 3])
    np.testing.assert_array_equal(blk["cprow_rindex"], [1, 4])
    np.testing.assert_array_equal(blk["diag"], [0, 0, 2, 2, 2, 3, 3, 3, 3, 3][:10] if False else orc.mark_diagonal(ai, aj))
    A.destroy()


@pytest.mark.parametrize("name", ["tiny", "small", "spd-real-int32-float64"])
def test_matload_matview_binary(comm, golden_dir, tmp_path, name):
    """MatLoad of the reference's own data files == the reference reader's CSR; MatView writes the same bytes back."""
    from petsc_amd import petsc as P
    g = load(golden_dir, name)
    src = os.path.join(golden_dir, "petscbin_%s.dat" % name)
    A = P.Mat(comm)
    off = A.load(src)
    blk = A.block(0)
    np.testing.assert_array_equal(blk["i"], g["i"])
    np.testing.assert_array_equal(blk["j"], g["j"])
    np.testing.assert_array_equal(blk["a"], g["a"])
    nz = len(g["j"])
    assert off == 16 + 4 * int(g["M"]) + 12 * nz
    out = str(tmp_path / "out.dat")
    A.view(out)
    assert open(out, "rb").read() == open(src, "rb").read()[:off]
    A.destroy()


def test_matload_unsorted_rows(comm, golden_dir, tmp_path):
    """The file format does not promise sorted rows: MatLoad lays sorted slabs straight into the blocks and falls back to sorted
    insertion (MatSetValues row by row, aij.c:4163-4171) otherwise -- same matrix either way."""
    from petsc_amd import petsc as P
    g = load(golden_dir, "small")
    ai, aj, aa = g["i"], g["j"].copy(), g["a"].copy()
    rng = np.random.default_rng(2)
    for r in range(len(ai) - 1):
        p = rng.permutation(ai[r + 1] - ai[r]) + ai[r]
        aj[ai[r]:ai[r + 1]], aa[ai[r]:ai[r + 1]] = aj[p], aa[p]
    path = str(tmp_path / "shuffled.dat")
    with open(path, "wb") as f:
        f.write(np.array([1211216, int(g["M"]), int(g["N"]), len(aj)], ">i4").tobytes())
        f.write(np.diff(ai).astype(">i4").tobytes()); f.write(aj.astype(">i4").tobytes()); f.write(aa.astype(">f8").tobytes())
    A = P.Mat(comm)
    A.load(path)
    blk = A.block(0)
    np.testing.assert_array_equal(blk["i"], g["i"]); np.testing.assert_array_equal(blk["j"], g["j"]); np.testing.assert_array_equal(blk["a"], g["a"])
    A.destroy()


def test_error_behaviour(comm):
    from petsc_amd import petsc as P
    with pytest.raises(P.PetscError) as e:
        P.Mat(comm, 4, 4, 4, 4, "nosuchtype")
    assert e.value.code == 86  # PETSC_ERR_ARG_UNKNOWN_TYPE, matreg.c:63
    A = P.Mat(comm, 4, 4, 4, 4, "seqaijhip")
    with pytest.raises(P.PetscError) as e:
        A.set_values([0], [7], [1.0])
    assert e.value.code == 63  # column too large, aij.c:452
    with pytest.raises(P.PetscError) as e:
        A.set_values([9], [0], [1.0])
    assert e.value.code == 63
    A.set_values([-1], [0], [1.0])  # negative rows are ignored (aij.c:444)
    A.set_values([0, 1, 2, 3], [0], [1.0, 1.0, 1.0, 1.0])
    x = P.Vec(comm, 4, 4, "hip")
    y = x.duplicate()
    with pytest.raises(P.PetscError) as e:
        A.mult(x, y)
    assert e.value.code == 73  # not for unassembled matrix, matrix.c:2209
    A.assemble()
    with pytest.raises(P.PetscError) as e:
        A.mult(x, x)
    assert e.value.code == 61  # x and y must be different vectors
    z = P.Vec(comm, 5, 5, "hip")
    with pytest.raises(P.PetscError) as e:
        A.mult(z, y)
    assert e.value.code == 60
    with pytest.raises(P.PetscError) as e:
        A.mult(x, y)
    assert e.value.code == 56 and "no CPU fallback" in str(e.value)  # planning-only context: numerical ops are unavailable
    with pytest.raises(P.PetscError) as e:
        P.Vec(comm, 4, 4, "cusp")
    assert e.value.code == 86
    assert x.type() == "seqhip" and x.ownership_range() == (0, 4)
    for o in (x, y, z, A):
        o.destroy()


def test_split_ownership(comm, orc):
    import ctypes as C
    from petsc_amd import petsc as P
    n, N = C.c_int32(P.PETSC_DECIDE), C.c_int32(17)
    P.chk(P.lib().PetscSplitOwnership(comm.h, C.byref(n), C.byref(N)))
    assert n.value == 17
    for size in (1, 2, 3, 7, 8):
        assert sum(orc.split_ownership(1000003, size, r) for r in range(size)) == 1000003


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(world, name, mode, comm="callbacks"):
    port = _free_port()
    procs = []
    key = os.urandom(128).hex()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   PHIP_TEST_COMM=comm, PHIP_TEST_KEY=key, PHIP_HOST_RING_TIMEOUT="60")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), name, mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])


@pytest.mark.parametrize("world,name,mode", [(2, "arco1", "arrays"), (2, "medium", "local"), (2, "small", "stash"),
                                             (3, "medium", "arrays"), (3, "spd-real-int32-float64", "stash"), (4, "small", "local")])
def test_mpiaij_plan_gloo(world, name, mode):
    _run_world(world, name, mode)


@pytest.mark.parametrize("world,name,mode", [(2, "medium", "arrays"), (3, "small", "stash"), (4, "arco1", "local")])
def test_mpiaij_plan_shm(world, name, mode):
    """The same index work with the library's own multi-process communicator (backend "shm": shared-memory ring) carrying the
    handshake -- the host side of what bench.py runs with under torch.distributed.run."""
    _run_world(world, name, mode, comm="shm")


def test_native_stencil_generator_matches_numpy():
    """bench.py builds its matrices with the C loop of the host library (PetscHipStencilRows); the numpy generators the parity tests use
    must give the same arrays, for whole matrices and for the row ranges the ranks of a multi-GPU run ask for."""
    from petsc_amd import generators as G
    for kind, fn in ((7, lambda a, b: G.laplace3d_7pt(9, 7, 11, rstart=a, rend=b, dirichlet_rows=False)), (27, lambda a, b: G.box3d_27pt(9, 7, 11, rstart=a, rend=b))):
        for a, b in ((0, None), (100, 433), (0, 0), (692, 693)):
            for u, v in zip(fn(a, b), G.stencil_rows_native(kind, 9, 7, 11, a, b)):
                np.testing.assert_array_equal(u, v)
    for u, v in zip(G.laplace2d_5pt(13, 17), G.stencil_rows_native(5, 17, 13, 1)):
        np.testing.assert_array_equal(u, v)
