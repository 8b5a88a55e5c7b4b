#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

What it does
  * imports the reference's own reader  /root/reference/bin/PetscBinaryIO.py  (Python, importable
    here; it never travels) and reads the matrices + right-hand sides that the reference's tests
    use from /root/reference/share/petsc/datafiles/matrices;
  * stores each as CSR (i, j, a) + rhs in  tests/golden/mat_<name>.npz  together with
      - y = A @ x computed by scipy.sparse (csr_matvec accumulates left to right from 0, the same
        order as PetscSparseDensePlusDot, src/mat/impls/aij/seq/aij.h:409-411) for a seeded x,
      - the main diagonal and "position of the diagonal" array derived with numpy (independent of
        the C oracle),
      - MPIAIJ split products for 2/3/4 ranks derived independently with numpy: ownership ranges
        (psplit.c:88), garray (mmaij.c:69-101), compact B column ids, and the VecScatter PtoS
        plan lists (vpscat.c:2327-2447);
  * copies three small reference data files byte-for-byte (data, not source) for the
    PetscBinary reader/writer tests.
The fixtures are data only: inputs and expected outputs.
"""
import os
import shutil
import sys

import numpy as np
import scipy.sparse as sp

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "bin"))
sys.path.insert(0, HERE)
import PetscBinaryIO  # noqa: E402  (reference's reader)

NAMES = ["tiny", "small", "medium", "arco1", "poisson1", "spd-real-int32-float64", "ns-real-int32-float64", "poisson2", "fem1"]
SUBDIR = {"afiro": "indefinite/afiro", "golan": "indefinite/golan", "A4": "matmatmult/A4.BGriffith", "P4": "matmatmult/P4.BGriffith"}  # fixture name -> file
RAW_COPIES = ["tiny", "small", "spd-real-int32-float64"]


from plan_numpy import ranges, split_products  # noqa: E402,F401  (numpy derivation of the MPIAIJ split products)


def main():
    io = PetscBinaryIO.PetscBinaryIO(precision="double", indices="32bit", complexscalars=False)
    for name in NAMES + list(SUBDIR):
        path = os.path.join(REF, "share/petsc/datafiles/matrices", SUBDIR.get(name, name))
        objs = io.readBinaryFile(path)
        (M, N), (ai, aj, aa) = objs[0]
        ai = np.asarray(ai, dtype=np.int32)
        aj = np.asarray(aj, dtype=np.int32)
        aa = np.asarray(aa, dtype=np.float64)
        vecs = [np.asarray(o, dtype=np.float64) for o in objs[1:] if isinstance(o, PetscBinaryIO.Vec)]
        rng = np.random.default_rng(12345)
        x = rng.uniform(-1.0, 1.0, N)
        A = sp.csr_matrix((aa, aj, ai), shape=(M, N))
        y = A @ x
        y0 = rng.uniform(-1.0, 1.0, M)
        diag = np.zeros(M)
        dpos = ai[1:].copy()
        for r in range(M):
            cols = aj[ai[r]:ai[r + 1]]
            hit = np.nonzero(cols == r)[0]
            if len(hit):
                dpos[r] = ai[r] + hit[0]
                diag[r] = aa[dpos[r]]
        fx = dict(M=np.int32(M), N=np.int32(N), i=ai, j=aj, a=aa, x=x, y=y, y0=y0, diag=diag, diagpos=dpos.astype(np.int32))
        for k, v in enumerate(vecs):
            fx["vec%d" % k] = v
        if M == N:
            for size in (2, 3, 4):
                for key, val in split_products(ai, aj, N, size).items():
                    fx["s%d_%s" % (size, key)] = val
        np.savez_compressed(os.path.join(HERE, "mat_%s.npz" % name), **fx)
        print(name, M, N, len(aa), "vecs", len(vecs))
    for name in RAW_COPIES:
        shutil.copyfile(os.path.join(REF, "share/petsc/datafiles/matrices", name), os.path.join(HERE, "petscbin_%s.dat" % name))
        os.chmod(os.path.join(HERE, "petscbin_%s.dat" % name), 0o644)


if __name__ == "__main__":
    main()
