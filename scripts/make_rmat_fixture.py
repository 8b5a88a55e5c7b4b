#!/usr/bin/env python3
"""BASELINE config 5 as a committed fixture: the R-MAT matrix (a,b,c,d) = (0.57,0.19,0.19,0.05), scale 16, 16 * 2^16 edge samples,
duplicates summed, self-loops kept, values uniform(0,1), seed 20240601 (petsc_amd/generators.py: rmat), written in the PetscBinary
format of MatView_SeqAIJ_Binary (src/mat/impls/aij/seq/aij.c:547-582: header {1211216, M, N, nz}, row lengths, columns, values, all
big-endian) BY THE REFERENCE'S OWN WRITER /root/reference/bin/PetscBinaryIO.py (writeMatSparse).  Run only in the build container:

    python scripts/make_rmat_fixture.py          ->  tests/golden/rmat16.petscbin  +  tests/golden/rmat16_expect.npz

rmat16_expect.npz holds the expected outputs that do not depend on this repository's code: sizes, y = A x for a seeded x by
scipy.sparse (csr_matvec adds left to right from 0, the order of aij.h:409-411), and per-rank nonzero counts of the 8-rank split.
The scale-24 instance of the benchmark is generated on the GPU box by the same generator (scripts/bench_configs.py)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/bin")
import PetscBinaryIO  # noqa: E402  (the reference's writer; never travels)
from petsc_amd import generators as G  # noqa: E402

SCALE = 16


def main():
    ai, aj, aa = G.rmat(SCALE)
    n = 1 << SCALE
    io = PetscBinaryIO.PetscBinaryIO(precision="double", indices="32bit", complexscalars=False)
    out = os.path.join(ROOT, "tests", "golden", "rmat%d.petscbin" % SCALE)
    with open(out, "wb") as fh:
        io.writeMatSparse(fh, ((n, n), (ai, aj, aa)))
    x = np.random.default_rng(12345).uniform(-1.0, 1.0, n)
    y = sp.csr_matrix((aa, aj, ai), shape=(n, n)) @ x
    size = 8
    rg = np.concatenate([[0], np.cumsum([n // size + (1 if (n % size) > r else 0) for r in range(size)])])
    nnz8 = np.array([int(ai[rg[r + 1]] - ai[rg[r]]) for r in range(size)], np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rmat%d_expect.npz" % SCALE), M=np.int32(n), N=np.int32(n), nz=np.int64(len(aa)),
                        x=x, y=y, nnz_rank8=nnz8, rowlen_max=np.int32(np.diff(ai).max()), empty_rows=np.int32(np.count_nonzero(np.diff(ai) == 0)))
    print(out, os.path.getsize(out), "bytes; nz", len(aa), "nnz per rank (8):", nnz8.tolist(), "imbalance %.2f" % (nnz8.max() / nnz8.mean()))


if __name__ == "__main__":
    main()
