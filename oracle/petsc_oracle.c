/*
 * petsc_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the PETSc 3.6.1 sparse-solve hot path,
 * written from the reference's loops in the reference's loop order.  It is the checker for the
 * HIP product path and the "port" CPU baseline of bench.py.  Nothing in petsc_amd/ links, loads
 * or calls it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * Parity pinning (see oracle/README.md): the integer/index work is pinned bit-exact by fixtures
 * derived from the reference's own data files; SpMV is pinned against scipy's csr_matvec (same
 * left-to-right accumulate as aij.h:409-411) on the reference's share/petsc/datafiles matrices;
 * the GMRES+Jacobi drivers are pinned by the reference's golden stdout files
 * (src/ksp/ksp/examples/tests/output/ex3_1.out, tutorials/output/ex5_1.out).  BLAS level-1
 * (ddot/daxpy/dscal/dasum) is an un-vendored dependency of the reference (no source, no pinned
 * version); it is restated as the netlib reference-BLAS sequential loops, so dot/norm are
 * "parity unpinned" at the ulp level and bound only by the contract tolerances.
 *
 * All file:line cites are relative to /root/reference.
 * PetscInt = int32, PetscScalar = double.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>

typedef int32_t OInt;

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* Row ownership: src/sys/utils/psplit.c:88  n = N/size + ((N % size) > rank);
   ranges = prefix sums, src/vec/is/utils/pmap.c:146-158.                                       */
ORC_EXPORT OInt orc_split_ownership(OInt N, OInt size, OInt rank)
{
  return N / size + ((N % size) > rank);
}

ORC_EXPORT void orc_ownership_ranges(OInt N, OInt size, OInt *range /* size+1 */)
{
  OInt r;
  range[0] = 0;
  for (r = 0; r < size; r++) range[r + 1] = range[r] + orc_split_ownership(N, size, r);
}

/* ------------------------------------------------------------------------------------------ */
/* SeqAIJ under assembly: src/mat/impls/aij/seq/aij.h:10-38,106-124 (i, j, a, imax, ilen, diag,
   nonzerorowcnt, rmax, compressedrow).                                                         */
typedef struct {
  OInt    m, n;          /* rows, columns */
  OInt   *i, *j;         /* i[m+1], j[maxnz] */
  double *a;
  OInt   *imax, *ilen;   /* allocated / used length per row */
  OInt   *diag;
  OInt    nz, maxnz, nonzerorowcnt, rmax, reallocs;
  /* Mat_CompressedRow, include/petsc/private/matimpl.h:308-313 */
  int     cprow_use;
  OInt    cprow_nrows;
  OInt   *cprow_i, *cprow_rindex;
  int     assembled;
} OrcSeqAIJ;

/* MatSeqAIJSetPreallocation_SeqAIJ (aij.c:3547-3640): nz per row (uniform or nnz[]); rows laid out
   back to back with imax[i] slots each.                                                        */
ORC_EXPORT OrcSeqAIJ *orc_seqaij_create(OInt m, OInt n, OInt nz, const OInt *nnz)
{
  OrcSeqAIJ *A = (OrcSeqAIJ *)calloc(1, sizeof(*A));
  OInt       r, tot = 0;
  A->m = m; A->n = n;
  A->imax = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  A->ilen = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  A->i    = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  A->diag = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  for (r = 0; r < m; r++) { A->imax[r] = nnz ? nnz[r] : nz; tot += A->imax[r]; }
  A->maxnz = tot;
  A->j = (OInt *)calloc((size_t)tot + 1, sizeof(OInt));
  A->a = (double *)calloc((size_t)tot + 1, sizeof(double));
  for (r = 0; r < m; r++) A->i[r + 1] = A->i[r] + A->imax[r];
  return A;
}

ORC_EXPORT void orc_seqaij_destroy(OrcSeqAIJ *A)
{
  if (!A) return;
  free(A->i); free(A->j); free(A->a); free(A->imax); free(A->ilen); free(A->diag);
  free(A->cprow_i); free(A->cprow_rindex); free(A);
}

/* MatSeqXAIJReallocateAIJ, aij.h:148-179: CHUNKSIZE 15 extra slots opened after the row. */
static void orc_seqaij_grow_row(OrcSeqAIJ *A, OInt row, OInt nrow)
{
  const OInt CHUNK = 15;
  OInt       m = A->m, new_nz = A->i[m] + CHUNK, ii, len;
  OInt      *ni = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt));
  OInt      *nj = (OInt *)calloc((size_t)new_nz + 1, sizeof(OInt));
  double    *na = (double *)calloc((size_t)new_nz + 1, sizeof(double));
  for (ii = 0; ii < row + 1; ii++) ni[ii] = A->i[ii];
  for (ii = row + 1; ii < m + 1; ii++) ni[ii] = A->i[ii] + CHUNK;
  memcpy(nj, A->j, (size_t)(A->i[row] + nrow) * sizeof(OInt));
  len = new_nz - CHUNK - A->i[row] - nrow;
  memcpy(nj + A->i[row] + nrow + CHUNK, A->j + A->i[row] + nrow, (size_t)len * sizeof(OInt));
  memcpy(na, A->a, (size_t)(A->i[row] + nrow) * sizeof(double));
  memcpy(na + A->i[row] + nrow + CHUNK, A->a + A->i[row] + nrow, (size_t)len * sizeof(double));
  free(A->i); free(A->j); free(A->a);
  A->i = ni; A->j = nj; A->a = na;
  A->imax[row] += CHUNK; A->maxnz += CHUNK; A->reallocs++;
}

/* MatSetValues_SeqAIJ, aij.c:430-501 (row-oriented v; is_add: ADD_VALUES else INSERT_VALUES).
   Sorted insert: bisect while the window is > 5 wide, then linear scan.                        */
ORC_EXPORT int orc_seqaij_setvalues(OrcSeqAIJ *A, OInt m, const OInt *im, OInt n, const OInt *in,
                                    const double *v, int is_add)
{
  OInt k, l, low, high, t, ii, row, nrow, i, col, rmax, N, lastcol = -1;
  for (k = 0; k < m; k++) {
    OInt   *rp;
    double *ap;
    row = im[k];
    if (row < 0) continue;
    if (row >= A->m) return 63; /* PETSC_ERR_ARG_OUTOFRANGE */
    rp = A->j + A->i[row]; ap = A->a + A->i[row];
    rmax = A->imax[row]; nrow = A->ilen[row];
    low = 0; high = nrow;
    for (l = 0; l < n; l++) {
      double value;
      if (in[l] < 0) continue;
      if (in[l] >= A->n) return 63;
      col = in[l];
      value = v[l + k * n];
      if (col <= lastcol) low = 0; else high = nrow;
      lastcol = col;
      while (high - low > 5) {
        t = (low + high) / 2;
        if (rp[t] > col) high = t; else low = t;
      }
      for (i = low; i < high; i++) {
        if (rp[i] > col) break;
        if (rp[i] == col) {
          if (is_add) ap[i] += value; else ap[i] = value;
          low = i + 1;
          goto noinsert;
        }
      }
      if (nrow >= rmax) {
        orc_seqaij_grow_row(A, row, nrow);
        rp = A->j + A->i[row]; ap = A->a + A->i[row]; rmax = A->imax[row];
      }
      N = nrow++ - 1; A->nz++; high++;
      for (ii = N; ii >= i; ii--) { rp[ii + 1] = rp[ii]; ap[ii + 1] = ap[ii]; }
      rp[i] = col; ap[i] = value; low = i + 1;
noinsert:;
    }
    A->ilen[row] = nrow;
  }
  return 0;
}

/* MatMarkDiagonal_SeqAIJ, aij.c:1506-1527: diag[i] = position of (i,i) or i[i+1] when absent. */
ORC_EXPORT void orc_mark_diagonal(OInt m, const OInt *ai, const OInt *aj, OInt *diag)
{
  OInt i, j;
  for (i = 0; i < m; i++) {
    diag[i] = ai[i + 1];
    for (j = ai[i]; j < ai[i + 1]; j++) {
      if (aj[j] == i) { diag[i] = j; break; }
    }
  }
}

/* MatCheckCompressedRow, src/mat/utils/compressedrow.c:32-74.  Returns use flag; when used fills
   cpi[nrows+1], ridx[nrows] (caller allocates m+1 / m).  nzrows = rows with entries.           */
ORC_EXPORT int orc_check_compressed_row(OInt mbs, const OInt *ai, OInt nzrows, double ratio,
                                        OInt *cprow_nrows, OInt *cpi, OInt *ridx)
{
  OInt nrows = mbs - nzrows, i, row, nz; /* number of zero rows, compressedrow.c:46 */
  if (nrows < ratio * mbs) { *cprow_nrows = 0; return 0; }
  nrows = mbs - nrows;
  row = 0; cpi[0] = 0;
  for (i = 0; i < mbs; i++) {
    nz = ai[i + 1] - ai[i];
    if (nz == 0) continue;
    cpi[row + 1] = ai[i + 1];
    ridx[row++]  = i;
  }
  *cprow_nrows = nrows;
  return 1;
}

/* MatAssemblyEnd_SeqAIJ, aij.c:968-1024: squeeze out preallocation slack, rebuild i[], count
   nonzero rows, rmax, MatMarkDiagonal, MatCheckCompressedRow(ratio 0.6).                       */
ORC_EXPORT void orc_seqaij_assembly_end(OrcSeqAIJ *A)
{
  OInt  fshift = 0, i, j, m = A->m, N, rmax = 0;
  OInt *ai = A->i, *aj = A->j, *imax = A->imax, *ailen = A->ilen;
  double *aa = A->a;
  if (m) rmax = ailen[0];
  for (i = 1; i < m; i++) {
    fshift += imax[i - 1] - ailen[i - 1];
    if (ailen[i] > rmax) rmax = ailen[i];
    if (fshift) {
      OInt   *ip = aj + ai[i];
      double *ap = aa + ai[i];
      N = ailen[i];
      for (j = 0; j < N; j++) { ip[j - fshift] = ip[j]; ap[j - fshift] = ap[j]; }
    }
    ai[i] = ai[i - 1] + ailen[i - 1];
  }
  if (m) { fshift += imax[m - 1] - ailen[m - 1]; ai[m] = ai[m - 1] + ailen[m - 1]; }
  A->nonzerorowcnt = 0;
  for (i = 0; i < m; i++) {
    ailen[i] = imax[i] = ai[i + 1] - ai[i];
    A->nonzerorowcnt += ((ai[i + 1] - ai[i]) > 0);
  }
  A->nz = ai[m];
  orc_mark_diagonal(m, ai, aj, A->diag);
  A->rmax = rmax;
  free(A->cprow_i); free(A->cprow_rindex);
  A->cprow_i = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  A->cprow_rindex = (OInt *)calloc((size_t)m + 1, sizeof(OInt));
  A->cprow_use = orc_check_compressed_row(m, ai, A->nonzerorowcnt, 0.6, &A->cprow_nrows, A->cprow_i, A->cprow_rindex);
  A->assembled = 1;
}

/* accessors for ctypes */
ORC_EXPORT void orc_seqaij_info(const OrcSeqAIJ *A, OInt *out /* m,n,nz,nonzerorowcnt,rmax,cprow_use,cprow_nrows,reallocs */)
{
  out[0] = A->m; out[1] = A->n; out[2] = A->nz; out[3] = A->nonzerorowcnt; out[4] = A->rmax;
  out[5] = A->cprow_use; out[6] = A->cprow_nrows; out[7] = A->reallocs;
}
ORC_EXPORT void orc_seqaij_get(const OrcSeqAIJ *A, OInt *ai, OInt *aj, double *aa, OInt *diag, OInt *cpi, OInt *ridx)
{
  if (ai) memcpy(ai, A->i, ((size_t)A->m + 1) * sizeof(OInt));
  if (aj) memcpy(aj, A->j, (size_t)A->nz * sizeof(OInt));
  if (aa) memcpy(aa, A->a, (size_t)A->nz * sizeof(double));
  if (diag) memcpy(diag, A->diag, (size_t)A->m * sizeof(OInt));
  if (cpi && A->cprow_use) memcpy(cpi, A->cprow_i, ((size_t)A->cprow_nrows + 1) * sizeof(OInt));
  if (ridx && A->cprow_use) memcpy(ridx, A->cprow_rindex, (size_t)A->cprow_nrows * sizeof(OInt));
}

/* ------------------------------------------------------------------------------------------ */
/* SpMV.  PetscSparseDensePlusDot default branch, aij.h:409-411: for i<nnz: sum += xv[i]*r[xi[i]]
   -- left-to-right, starting from the caller's sum.                                            */
#define ORC_SPARSE_DENSE_PLUS_DOT(sum, r, xv, xi, nnz) \
  { OInt __i; for (__i = 0; __i < (nnz); __i++) (sum) += (xv)[__i] * (r)[(xi)[__i]]; }

/* MatMult_SeqAIJ, aij.c:1272-1327.  cprow branch :1295-1308 zero-fills y then writes the nonzero
   rows; plain branch :1313-1320.                                                               */
ORC_EXPORT void orc_matmult_seqaij(OInt m, const OInt *ai, const OInt *aj, const double *aa,
                                   int usecprow, OInt cprow_nrows, const OInt *cprow_i, const OInt *cprow_rindex,
                                   const double *x, double *y)
{
  OInt i, n;
  double sum;
  if (usecprow) {
    memset(y, 0, (size_t)m * sizeof(double));
    for (i = 0; i < cprow_nrows; i++) {
      n = cprow_i[i + 1] - cprow_i[i];
      sum = 0.0;
      ORC_SPARSE_DENSE_PLUS_DOT(sum, x, aa + cprow_i[i], aj + cprow_i[i], n);
      y[cprow_rindex[i]] = sum;
    }
  } else {
    for (i = 0; i < m; i++) {
      n = ai[i + 1] - ai[i];
      sum = 0.0;
      ORC_SPARSE_DENSE_PLUS_DOT(sum, x, aa + ai[i], aj + ai[i], n);
      y[i] = sum;
    }
  }
}

/* MatMultAdd_SeqAIJ, aij.c:1440-1499: z = y + A x, sum starts at y[i]; cprow branch copies y to z
   when distinct (:1460-1462) and updates only the nonzero rows.                                */
ORC_EXPORT void orc_matmultadd_seqaij(OInt m, const OInt *ai, const OInt *aj, const double *aa,
                                      int usecprow, OInt cprow_nrows, const OInt *cprow_i, const OInt *cprow_rindex,
                                      const double *x, const double *y, double *z)
{
  OInt i, n;
  double sum;
  if (usecprow) {
    if (z != y) memcpy(z, y, (size_t)m * sizeof(double));
    for (i = 0; i < cprow_nrows; i++) {
      n = cprow_i[i + 1] - cprow_i[i];
      sum = y[cprow_rindex[i]];
      ORC_SPARSE_DENSE_PLUS_DOT(sum, x, aa + cprow_i[i], aj + cprow_i[i], n);
      z[cprow_rindex[i]] = sum;
    }
  } else {
    for (i = 0; i < m; i++) {
      n = ai[i + 1] - ai[i];
      sum = y[i];
      ORC_SPARSE_DENSE_PLUS_DOT(sum, x, aa + ai[i], aj + ai[i], n);
      z[i] = sum;
    }
  }
}

/* MatMultTransposeAdd_SeqAIJ, aij.c:1207-1254: y = z when distinct (:1224), then for every (compressed) row i
   alpha = x[i]; y[idx[j]] += alpha*v[j] (:1238-1248).  n = number of columns.                    */
ORC_EXPORT void orc_matmulttransposeadd_seqaij(OInt m, OInt n, const OInt *ai, const OInt *aj, const double *aa,
                                               int usecprow, OInt cprow_nrows, const OInt *cprow_i, const OInt *cprow_rindex,
                                               const double *x, const double *zz, double *yy)
{
  OInt i, j, nn;
  const OInt *ii = usecprow ? cprow_i : ai;
  double alpha;
  if (zz != yy) memcpy(yy, zz, (size_t)n * sizeof(double));
  if (usecprow) m = cprow_nrows;
  for (i = 0; i < m; i++) {
    const OInt   *idx = aj + ii[i];
    const double *v = aa + ii[i];
    nn = ii[i + 1] - ii[i];
    alpha = usecprow ? x[cprow_rindex[i]] : x[i];
    for (j = 0; j < nn; j++) yy[idx[j]] += alpha * v[j];
  }
}

/* MatMultTranspose_SeqAIJ, aij.c:1258-1266: VecSet(yy,0.0) then the Add routine with zz == yy. */
ORC_EXPORT void orc_matmulttranspose_seqaij(OInt m, OInt n, const OInt *ai, const OInt *aj, const double *aa,
                                            int usecprow, OInt cprow_nrows, const OInt *cprow_i, const OInt *cprow_rindex,
                                            const double *x, double *y)
{
  OInt i;
  for (i = 0; i < n; i++) y[i] = 0.0;
  orc_matmulttransposeadd_seqaij(m, n, ai, aj, aa, usecprow, cprow_nrows, cprow_i, cprow_rindex, x, y, y);
}

/* MatGetDiagonal_SeqAIJ, aij.c:1169-1202 (non-factored branch): first aj[j]==i, else 0. */
ORC_EXPORT void orc_getdiagonal_seqaij(OInt n, const OInt *ai, const OInt *aj, const double *aa, double *x)
{
  OInt i, j;
  for (i = 0; i < n; i++) x[i] = 0.0;
  for (i = 0; i < n; i++) {
    for (j = ai[i]; j < ai[i + 1]; j++) {
      if (aj[j] == i) { x[i] = aa[j]; break; }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* MPIAIJ diag/off-diag routing of one rank's rows: MatSetValues_MPIAIJ, mpiaij.c:553-618
   (col in [cstart,cend) -> A with local col; else -> B with GLOBAL col before first assembly),
   then MatSetUpMultiply_MPIAIJ, mmaij.c:69-101 (garray = sorted unique off-proc cols; B->j
   rewritten to compact ids).  Input: this rank's rows as CSR with global, sorted columns.
   Outputs are caller-allocated with the sizes returned by orc_mpiaij_split_count.              */
ORC_EXPORT void orc_mpiaij_split_count(OInt m, const OInt *gi, const OInt *gj, OInt cstart, OInt cend,
                                       OInt *nzA, OInt *nzB)
{
  OInt r, k, a = 0, b = 0;
  for (r = 0; r < m; r++)
    for (k = gi[r]; k < gi[r + 1]; k++) { if (gj[k] >= cstart && gj[k] < cend) a++; else b++; }
  *nzA = a; *nzB = b;
}

ORC_EXPORT OInt orc_mpiaij_split(OInt m, OInt N, const OInt *gi, const OInt *gj, const double *ga,
                                 OInt cstart, OInt cend,
                                 OInt *Ai, OInt *Aj, double *Aa, OInt *Bi, OInt *Bj, double *Ba,
                                 OInt *garray /* capacity nzB */)
{
  OInt  r, k, na = 0, nb = 0, ec = 0, i;
  OInt *indices = (OInt *)calloc((size_t)N + 1, sizeof(OInt)); /* mmaij.c:71 */
  Ai[0] = 0; Bi[0] = 0;
  for (r = 0; r < m; r++) {
    for (k = gi[r]; k < gi[r + 1]; k++) {
      if (gj[k] >= cstart && gj[k] < cend) { Aj[na] = gj[k] - cstart; Aa[na++] = ga[k]; }
      else                                 { Bj[nb] = gj[k];          Ba[nb++] = ga[k]; }
    }
    Ai[r + 1] = na; Bi[r + 1] = nb;
  }
  for (k = 0; k < nb; k++) { if (!indices[Bj[k]]) ec++; indices[Bj[k]] = 1; } /* mmaij.c:72-77 */
  ec = 0;
  for (i = 0; i < N; i++) if (indices[i]) garray[ec++] = i;                   /* :82-84 */
  for (i = 0; i < ec; i++) indices[garray[i]] = i;                            /* :87-89 */
  for (k = 0; k < nb; k++) Bj[k] = indices[Bj[k]];                            /* :92-96 */
  free(indices);
  return ec;
}

/* ------------------------------------------------------------------------------------------ */
/* VecScatterCreate_PtoS, src/vec/vec/utils/vpscat.c:2297-2495, for the MatSetUpMultiply call
   (inidx = garray, inidy = 0..ec-1, bs = 1), restated for ALL ranks at once with the MPI
   handshake replaced by direct reads of the other ranks' lists.
   Per rank r the plan is
     from (receive side): from_n, from_procs[] ascending (:2434-2440), from_starts[], from_indices[]
                          (indices into lvec, :2442-2447)
     to   (send side):    to_n, to_procs[] ascending (PetscSortMPIIntWithArray :2348), to_starts[],
                          to_indices[] = global - owners[rank] (:2416)
     local part:          nlocal, to_local[], from_local[] (:2459-2472)
   Flattened output layout (caller allocates generously):
     hdr[r*4+0..3] = from_n, to_n, nlocal, from_contiq
     procs/starts/indices are concatenated per rank with offsets in off[r*6 + ...].             */
ORC_EXPORT void orc_scatter_plan_rank(OInt size, OInt rank, const OInt *owners /* size+1 */,
                                      const OInt *ec_all /* size */, const OInt *const *garray_all,
                                      OInt *from_n, OInt *from_procs, OInt *from_starts, OInt *from_indices,
                                      OInt *to_n, OInt *to_procs, OInt *to_starts, OInt *to_indices,
                                      OInt *nlocal, OInt *to_local, OInt *from_local, OInt *from_contiq)
{
  OInt  nx = ec_all[rank], i, j, idx, nsends = 0, count, p;
  const OInt *inidx = garray_all[rank];
  OInt *nprocs = (OInt *)calloc((size_t)size, sizeof(OInt));
  OInt *owner  = (OInt *)calloc((size_t)nx + 1, sizeof(OInt));
  OInt *lowner = (OInt *)calloc((size_t)size, sizeof(OInt));
  OInt *start  = (OInt *)calloc((size_t)size + 1, sizeof(OInt));
  OInt  nprocslocal, n;
  /* :2327-2340 owner of every needed index (the j cursor only resets when idx decreases) */
  j = 0;
  for (i = 0; i < nx; i++) {
    idx = inidx[i];
    if (idx < owners[j]) j = 0;
    for (; j < size; j++) {
      if (idx < owners[j + 1]) { if (!nprocs[j]++) nsends++; owner[i] = j; break; }
    }
  }
  nprocslocal = nprocs[rank]; nprocs[rank] = 0;
  if (nprocslocal) nsends--;
  /* receive side of the scatter = the messages this rank will get: :2423-2447 */
  *from_n = nsends;
  count = 0; from_starts[0] = start[0] = 0;
  for (i = 0; i < size; i++) {
    if (nprocs[i]) {
      lowner[i] = count;
      from_procs[count++] = i;
      from_starts[count] = start[count] = start[count - 1] + nprocs[i];
    }
  }
  for (i = 0; i < nx; i++) {
    if (owner[i] != rank) from_indices[start[lowner[owner[i]]]++] = i; /* inidy[i] = i */
  }
  /* local part :2459-2472 */
  *nlocal = nprocslocal;
  if (nprocslocal) {
    OInt nt = 0;
    for (i = 0; i < nx; i++) {
      idx = inidx[i];
      if (idx >= owners[rank] && idx < owners[rank + 1]) { to_local[nt] = idx - owners[rank]; from_local[nt++] = i; }
    }
  }
  /* contiguity check VecScatterCreateCommon_PtoS :2524-2532 */
  n = from_starts[*from_n];
  *from_contiq = 1;
  for (i = 1; i < n; i++) if (from_indices[i] != from_indices[i - 1] + 1) { *from_contiq = 0; break; }
  /* send side: what every other rank p asks of me, p ascending (:2348), in p's request order
     (svalues filled in inidx order, :2371-2373), minus my base (:2416)                        */
  *to_n = 0; to_starts[0] = 0;
  for (p = 0; p < size; p++) {
    OInt cnt = 0, base = owners[rank];
    if (p == rank) continue;
    for (i = 0; i < ec_all[p]; i++) {
      idx = garray_all[p][i];
      if (idx >= owners[rank] && idx < owners[rank + 1]) { to_indices[to_starts[*to_n] + cnt] = idx - base; cnt++; }
    }
    if (cnt) { to_procs[*to_n] = p; to_starts[*to_n + 1] = to_starts[*to_n] + cnt; (*to_n)++; }
  }
  free(nprocs); free(owner); free(lowner); free(start);
}

/* Pack_1 (vpscat.c:494-498): y[i] = x[idx[i]];  UnPack_1 INSERT (vpscat.c:502-510): y[idx[i]] = x[i]. */
ORC_EXPORT void orc_pack(OInt n, const double *x, const OInt *idx, double *y)
{
  OInt i; for (i = 0; i < n; i++) y[i] = x[idx[i]];
}
ORC_EXPORT void orc_unpack_insert(OInt n, const double *x, const OInt *idx, double *y)
{
  OInt i; for (i = 0; i < n; i++) y[idx[i]] = x[i];
}

/* ------------------------------------------------------------------------------------------ */
/* Vector kernels.  BLAS-1 restated as reference-BLAS sequential loops (see header).           */

/* VecDot_Seq bvec1.c:12-30 (BLASdot_(n,ya,1,xa,1)); VecTDot_Seq :34-51 -- identical for reals. */
ORC_EXPORT double orc_vecdot(OInt n, const double *x, const double *y)
{
  OInt i; double s = 0.0;
  for (i = 0; i < n; i++) s += y[i] * x[i];
  return s;
}

/* VecNorm_Seq bvec2.c:233-271.  type: 0 NORM_1, 1 NORM_2, 2 NORM_FROBENIUS, 3 NORM_INFINITY,
   4 NORM_1_AND_2 (include/petscvec.h NormType).  z must hold 2 doubles for 1_AND_2.            */
ORC_EXPORT void orc_vecnorm(OInt n, const double *x, int type, double *z)
{
  OInt i;
  if (type == 1 || type == 2) {
    double s = 0.0;
    for (i = 0; i < n; i++) s += x[i] * x[i];  /* sqrt(dot(x,x)), :244-245 -- not a scaled nrm2 */
    *z = sqrt(s);
  } else if (type == 3) {
    double max = 0.0, tmp;
    for (i = 0; i < n; i++) {
      if ((tmp = fabs(x[i])) > max) max = tmp;
      if (tmp != tmp) { max = tmp; break; } /* NaN propagates, :256 */
    }
    *z = max;
  } else if (type == 0) {
    double s = 0.0;
    for (i = 0; i < n; i++) s += fabs(x[i]);
    *z = s;
  } else if (type == 4) {
    orc_vecnorm(n, x, 0, z);
    orc_vecnorm(n, x, 1, z + 1);
  }
}

/* VecMDot_Seq, dvec2.c:98-294: remainder group nv&3 first, then groups of 4 vectors; inside a
   group the j&3 head is handled first in the order 2,1,0 and the body adds, per 4 entries,
   x0*y[0] + x1*y[1] + x2*y[2] + x3*y[3] as ONE expression to the running sum.                  */
static void orc_mdot_group(OInt n, const double *xbase, int g, const double *const *yy, double *z)
{
  double sum[4] = {0., 0., 0., 0.};
  const double *x = xbase, *y[4];
  OInt j = n, j_rem = n & 0x3;
  int  k;
  for (k = 0; k < g; k++) y[k] = yy[k];
  switch (j_rem) {
  case 3: for (k = 0; k < g; k++) sum[k] += x[2] * y[k][2]; /* fallthrough */
  case 2: for (k = 0; k < g; k++) sum[k] += x[1] * y[k][1]; /* fallthrough */
  case 1: for (k = 0; k < g; k++) sum[k] += x[0] * y[k][0]; /* fallthrough */
  case 0:
    x += j_rem; for (k = 0; k < g; k++) y[k] += j_rem;
    j -= j_rem;
    break;
  }
  while (j > 0) {
    double x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    x += 4;
    for (k = 0; k < g; k++) { sum[k] += x0 * y[k][0] + x1 * y[k][1] + x2 * y[k][2] + x3 * y[k][3]; y[k] += 4; }
    j -= 4;
  }
  for (k = 0; k < g; k++) z[k] = sum[k];
}

ORC_EXPORT void orc_vecmdot(OInt n, const double *x, OInt nv, const double *const *y, double *z)
{
  OInt nv_rem = nv & 0x3, i = nv;
  if (nv_rem) orc_mdot_group(n, x, (int)nv_rem, y, z);
  z += nv_rem; i -= nv_rem; y += nv_rem;
  while (i > 0) { orc_mdot_group(n, x, 4, y, z); z += 4; i -= 4; y += 4; }
}

/* VecMAXPY_Seq, dvec2.c:570-638 with the default PetscKernelAXPY{,2,3,4} (petscaxpy.h:94-97):
   remainder nv&3 first, then groups of 4: x[i] += a0*y0[i] + a1*y1[i] + a2*y2[i] + a3*y3[i].   */
ORC_EXPORT void orc_vecmaxpy(OInt n, double *x, OInt nv, const double *alpha, const double *const *y)
{
  OInt i, j, j_rem = nv & 0x3;
  switch (j_rem) {
  case 3: for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i] + alpha[1] * y[1][i] + alpha[2] * y[2][i]; break;
  case 2: for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i] + alpha[1] * y[1][i]; break;
  case 1: { double a1 = alpha[0]; for (i = 0; i < n; i++) x[i] += a1 * y[0][i]; } break;
  }
  alpha += j_rem; y += j_rem;
  for (j = j_rem; j < nv; j += 4) {
    for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i] + alpha[1] * y[1][i] + alpha[2] * y[2][i] + alpha[3] * y[3][i];
    alpha += 4; y += 4;
  }
}

/* VecAXPY_Seq bvec1.c:76-95: skipped when alpha == 0; daxpy: y[i] += alpha*x[i]. */
ORC_EXPORT void orc_vecaxpy(OInt n, double *y, double alpha, const double *x)
{
  OInt i;
  if (alpha != 0.0) for (i = 0; i < n; i++) y[i] += alpha * x[i];
}

ORC_EXPORT void orc_veccopy(OInt n, const double *x, double *y) /* bvec2.c:192-207 */
{
  if (x != y) memcpy(y, x, (size_t)n * sizeof(double));
}

ORC_EXPORT void orc_vecset(OInt n, double *x, double alpha) /* dvec2.c:551-566 */
{
  OInt i;
  if (alpha == 0.0) memset(x, 0, (size_t)n * sizeof(double));
  else for (i = 0; i < n; i++) x[i] = alpha;
}

/* VecAYPX_Seq dvec2.c:644-686: y = x + alpha*y with alpha in {0, 1, -1} special cases. */
ORC_EXPORT void orc_vecaypx(OInt n, double *y, double alpha, const double *x)
{
  OInt i;
  if (alpha == 0.0) orc_veccopy(n, x, y);
  else if (alpha == 1.0) orc_vecaxpy(n, y, alpha, x);
  else if (alpha == -1.0) for (i = 0; i < n; i++) y[i] = x[i] - y[i];
  else for (i = 0; i < n; i++) y[i] = x[i] + alpha * y[i];
}

/* VecWAXPY_Seq dvec2.c:697-730: w = y + alpha*x; alpha in {1,-1,0} special cases. */
ORC_EXPORT void orc_vecwaxpy(OInt n, double *w, double alpha, const double *x, const double *y)
{
  OInt i;
  if (alpha == 1.0) for (i = 0; i < n; i++) w[i] = y[i] + x[i];
  else if (alpha == -1.0) for (i = 0; i < n; i++) w[i] = y[i] - x[i];
  else if (alpha == 0.0) memcpy(w, y, (size_t)n * sizeof(double));
  else for (i = 0; i < n; i++) w[i] = y[i] + alpha * x[i];
}

/* VecScale_Seq bvec1.c:55-72: alpha == 0 -> VecSet, alpha == 1 -> nothing, else dscal. */
ORC_EXPORT void orc_vecscale(OInt n, double *x, double alpha)
{
  OInt i;
  if (alpha == 0.0) orc_vecset(n, x, 0.0);
  else if (alpha != 1.0) for (i = 0; i < n; i++) x[i] = alpha * x[i];
}

/* VecAXPBY_Seq bvec1.c:99-133. */
ORC_EXPORT void orc_vecaxpby(OInt n, double *y, double a, double b, const double *x)
{
  OInt i;
  if (a == 0.0) orc_vecscale(n, y, b);
  else if (b == 1.0) orc_vecaxpy(n, y, a, x);
  else if (a == 1.0) orc_vecaypx(n, y, b, x);
  else if (b == 0.0) for (i = 0; i < n; i++) y[i] = a * x[i];
  else for (i = 0; i < n; i++) y[i] = a * x[i] + b * y[i];
}

/* VecAXPBYPCZ_Seq bvec1.c:137-165. */
ORC_EXPORT void orc_vecaxpbypcz(OInt n, double *z, double alpha, double beta, double gamma, const double *x, const double *y)
{
  OInt i;
  if (alpha == 1.0) for (i = 0; i < n; i++) z[i] = x[i] + beta * y[i] + gamma * z[i];
  else if (gamma == 1.0) for (i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i] + z[i];
  else if (gamma == 0.0) for (i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i];
  else for (i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i] + gamma * z[i];
}

/* VecPointwiseMult_Seq bvec2.c:84-110 (aliasing cases), VecPointwiseDivide_Seq :114-132. */
ORC_EXPORT void orc_vecpointwisemult(OInt n, double *w, const double *x, const double *y)
{
  OInt i;
  if (w == x) for (i = 0; i < n; i++) w[i] *= y[i];
  else if (w == y) for (i = 0; i < n; i++) w[i] *= x[i];
  else for (i = 0; i < n; i++) w[i] = x[i] * y[i];
}
ORC_EXPORT void orc_vecpointwisedivide(OInt n, double *w, const double *x, const double *y)
{
  OInt i; for (i = 0; i < n; i++) w[i] = x[i] / y[i];
}

ORC_EXPORT void orc_vecswap(OInt n, double *x, double *y) /* bvec2.c:211-227, dswap */
{
  OInt i; if (x == y) return;
  for (i = 0; i < n; i++) { double t = x[i]; x[i] = y[i]; y[i] = t; }
}

/* VecReciprocal_Default, src/vec/vec/utils/vinv.c:1213-1227: nonzero -> 1/x, zero stays zero. */
ORC_EXPORT void orc_vecreciprocal(OInt n, double *x)
{
  OInt i; for (i = 0; i < n; i++) if (x[i] != 0.0) x[i] = 1.0 / x[i];
}

/* PCSetUp_Jacobi diag branch, src/ksp/pc/impls/jacobi/jacobi.c:174-195: d = diag(A); reciprocal;
   optional abs; 0 -> 1.0.  kind: 0 diagonal (given in d on entry), the rowmax/rowsum variants
   supply their own d.  PCApply_Jacobi :279-290 is orc_vecpointwisemult(y, x, d).               */
ORC_EXPORT int orc_jacobi_setup(OInt n, double *d, int useabs)
{
  OInt i; int zeroflag = 0;
  orc_vecreciprocal(n, d);
  if (useabs) for (i = 0; i < n; i++) d[i] = fabs(d[i]);
  for (i = 0; i < n; i++) if (d[i] == 0.0) { d[i] = 1.0; zeroflag = 1; }
  return zeroflag;
}
/* sqrt variant :196-214 for symmetric application */
ORC_EXPORT int orc_jacobi_setup_sqrt(OInt n, double *d)
{
  OInt i; int zeroflag = 0;
  for (i = 0; i < n; i++) { if (d[i] != 0.0) d[i] = 1.0 / sqrt(fabs(d[i])); else { d[i] = 1.0; zeroflag = 1; } }
  return zeroflag;
}
/* MatGetRowMaxAbs_SeqAIJ (aij.c:2755-2790) / MatGetRowSum (matrix.c:4445: A*ones) for -pc_jacobi_type */
ORC_EXPORT void orc_getrowmaxabs(OInt m, const OInt *ai, const double *aa, double *x)
{
  OInt i, j;
  for (i = 0; i < m; i++) {
    double v = 0.0;
    for (j = ai[i]; j < ai[i + 1]; j++) if (fabs(aa[j]) > fabs(v)) v = fabs(aa[j]);
    x[i] = v;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* A row-block distributed operator as MPIAIJ sees it, emulated in one process: rank r owns rows
   [range[r], range[r+1]); A_r (local columns), B_r (compact columns), garray_r.
   MatMult_MPIAIJ, mpiaij.c:984-998: lvec = x[garray] (the scatter), y = A x_loc, y += B lvec.  */
typedef struct {
  OInt size, N;
  OInt *range;              /* size+1 */
  OInt **Ai, **Aj; double **Aa;
  OInt **Bi, **Bj; double **Ba;
  OInt *ec; OInt **garray;
  /* compressed-row data of B_r (MatAssemblyEnd_SeqAIJ on B, aij.c:1020) */
  int *Bcp_use; OInt *Bcp_n; OInt **Bcp_i, **Bcp_r;
  double *lvec;             /* scratch, max ec */
  double **lvecs;           /* one ghost buffer per rank (lvec of mmaij.c:104) */
  int threads;              /* > 1: the ranks' work runs on that many OpenMP threads, one rank per thread at a time -- the CPU baseline
                               "P ranks of MPIAIJ on P cores" (SURVEY.md 8d); needs a build with -fopenmp, otherwise sequential.
                               Results do not depend on it: ranks are independent and reductions are summed in rank order. */
} OrcOperator;

/* loop over the emulated ranks, in parallel when the operator asks for it (the pragma is ignored without -fopenmp) */
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_FOR_RANKS(op, r) ORC_PRAGMA(omp parallel for schedule(static, 1) num_threads((op)->threads > 1 ? (op)->threads : 1)) \
  for (r = 0; r < (op)->size; r++)

static OrcOperator *orc_operator_create_t(OInt N, OInt size, const OInt *gi, const OInt *gj, const double *ga, int threads);
ORC_EXPORT OrcOperator *orc_operator_create(OInt N, OInt size, const OInt *gi, const OInt *gj, const double *ga)
{
  return orc_operator_create_t(N, size, gi, gj, ga, 1);
}
/* the same operator with its ranks spread over OpenMP threads (CPU baseline only) */
ORC_EXPORT OrcOperator *orc_operator_create_threaded(OInt N, OInt size, const OInt *gi, const OInt *gj, const double *ga, int threads)
{
  return orc_operator_create_t(N, size, gi, gj, ga, threads);
}
static OrcOperator *orc_operator_create_t(OInt N, OInt size, const OInt *gi, const OInt *gj, const double *ga, int threads)
{
  OrcOperator *op = (OrcOperator *)calloc(1, sizeof(*op));
  OInt r, maxec = 0;
  op->size = size; op->N = N;
  op->range = (OInt *)calloc((size_t)size + 1, sizeof(OInt));
  orc_ownership_ranges(N, size, op->range);
  op->Ai = (OInt **)calloc(size, sizeof(OInt *)); op->Aj = (OInt **)calloc(size, sizeof(OInt *)); op->Aa = (double **)calloc(size, sizeof(double *));
  op->Bi = (OInt **)calloc(size, sizeof(OInt *)); op->Bj = (OInt **)calloc(size, sizeof(OInt *)); op->Ba = (double **)calloc(size, sizeof(double *));
  op->ec = (OInt *)calloc(size, sizeof(OInt)); op->garray = (OInt **)calloc(size, sizeof(OInt *));
  op->Bcp_use = (int *)calloc(size, sizeof(int)); op->Bcp_n = (OInt *)calloc(size, sizeof(OInt));
  op->Bcp_i = (OInt **)calloc(size, sizeof(OInt *)); op->Bcp_r = (OInt **)calloc(size, sizeof(OInt *));
  op->lvecs = (double **)calloc(size, sizeof(double *));
  op->threads = threads;
  ORC_FOR_RANKS(op, r) { /* each rank's blocks are allocated and filled by the thread that will use them (first touch) */
    OInt rs = op->range[r], re = op->range[r + 1], m = re - rs, nzA, nzB, k, nzrows = 0;
    OInt *li = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt));
    for (k = 0; k <= m; k++) li[k] = gi[rs + k] - gi[rs];
    orc_mpiaij_split_count(m, li, gj + gi[rs], rs, re, &nzA, &nzB);
    op->Ai[r] = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt)); op->Aj[r] = (OInt *)malloc(((size_t)nzA + 1) * sizeof(OInt)); op->Aa[r] = (double *)malloc(((size_t)nzA + 1) * sizeof(double));
    op->Bi[r] = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt)); op->Bj[r] = (OInt *)malloc(((size_t)nzB + 1) * sizeof(OInt)); op->Ba[r] = (double *)malloc(((size_t)nzB + 1) * sizeof(double));
    op->garray[r] = (OInt *)malloc(((size_t)nzB + 1) * sizeof(OInt));
    op->ec[r] = orc_mpiaij_split(m, N, li, gj + gi[rs], ga + gi[rs], rs, re, op->Ai[r], op->Aj[r], op->Aa[r], op->Bi[r], op->Bj[r], op->Ba[r], op->garray[r]);
    op->lvecs[r] = (double *)calloc((size_t)op->ec[r] + 1, sizeof(double));
    for (k = 0; k < m; k++) nzrows += (op->Bi[r][k + 1] - op->Bi[r][k]) > 0;
    op->Bcp_i[r] = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt)); op->Bcp_r[r] = (OInt *)malloc(((size_t)m + 1) * sizeof(OInt));
    op->Bcp_use[r] = orc_check_compressed_row(m, op->Bi[r], nzrows, 0.6, &op->Bcp_n[r], op->Bcp_i[r], op->Bcp_r[r]);
    free(li);
  }
  for (r = 0; r < size; r++) if (op->ec[r] > maxec) maxec = op->ec[r];
  op->lvec = (double *)malloc(((size_t)maxec + 1) * sizeof(double));
  return op;
}

ORC_EXPORT void orc_operator_destroy(OrcOperator *op)
{
  OInt r;
  if (!op) return;
  for (r = 0; r < op->size; r++) {
    free(op->Ai[r]); free(op->Aj[r]); free(op->Aa[r]); free(op->Bi[r]); free(op->Bj[r]); free(op->Ba[r]);
    free(op->garray[r]); free(op->Bcp_i[r]); free(op->Bcp_r[r]); free(op->lvecs[r]);
  }
  free(op->lvecs);
  free(op->Ai); free(op->Aj); free(op->Aa); free(op->Bi); free(op->Bj); free(op->Ba); free(op->ec); free(op->garray);
  free(op->Bcp_use); free(op->Bcp_n); free(op->Bcp_i); free(op->Bcp_r); free(op->range); free(op->lvec); free(op);
}

ORC_EXPORT void orc_operator_mult(OrcOperator *op, const double *x, double *y)
{
  OInt r;
  ORC_FOR_RANKS(op, r) {
    OInt rs = op->range[r], m = op->range[r + 1] - rs, k;
    double *lvec = op->lvecs[r];
    for (k = 0; k < op->ec[r]; k++) lvec[k] = x[op->garray[r][k]];                /* VecScatterBegin/End */
    orc_matmult_seqaij(m, op->Ai[r], op->Aj[r], op->Aa[r], 0, 0, NULL, NULL, x + rs, y + rs);  /* A->mult */
    orc_matmultadd_seqaij(m, op->Bi[r], op->Bj[r], op->Ba[r], op->Bcp_use[r], op->Bcp_n[r], op->Bcp_i[r], op->Bcp_r[r],
                          lvec, y + rs, y + rs);                                  /* B->multadd(lvec,y,y) */
  }
}

/* MatMultTranspose_MPIAIJ, mpiaij.c:1029-1056 / MatMultTransposeAdd_MPIAIJ :1102-1117 (add != 0: z = y + A^T x):
   lvec_r = B_r^T x_r; z_r = (y_r +) A_r^T x_r; reverse scatter with ADD_VALUES: z[garray_r[k]] += lvec_r[k].
   The reference adds the messages in arrival order; this emulation adds them in ascending sender rank. */
ORC_EXPORT void orc_operator_mult_transpose(OrcOperator *op, const double *x, const double *y, double *z, int add)
{
  OInt r, k;
  for (r = 0; r < op->size; r++) {
    OInt rs = op->range[r], m = op->range[r + 1] - rs;
    if (add) orc_matmulttransposeadd_seqaij(m, m, op->Ai[r], op->Aj[r], op->Aa[r], 0, 0, NULL, NULL, x + rs, y + rs, z + rs);
    else     orc_matmulttranspose_seqaij(m, m, op->Ai[r], op->Aj[r], op->Aa[r], 0, 0, NULL, NULL, x + rs, z + rs);
  }
  for (r = 0; r < op->size; r++) {
    OInt rs = op->range[r], m = op->range[r + 1] - rs;
    orc_matmulttranspose_seqaij(m, op->ec[r], op->Bi[r], op->Bj[r], op->Ba[r], op->Bcp_use[r], op->Bcp_n[r], op->Bcp_i[r], op->Bcp_r[r],
                                x + rs, op->lvec);
    for (k = 0; k < op->ec[r]; k++) z[op->garray[r][k]] += op->lvec[k];
  }
}

ORC_EXPORT void orc_operator_getdiagonal(OrcOperator *op, double *d) /* mpiaij.c:1125-1135: diagonal of A block */
{
  OInt r;
  for (r = 0; r < op->size; r++) {
    OInt rs = op->range[r], m = op->range[r + 1] - rs;
    orc_getdiagonal_seqaij(m, op->Ai[r], op->Aj[r], op->Aa[r], d + rs);
  }
}

/* reductions as VecDot_MPI pbvec.c:9-19 / VecNorm_MPI pvec2.c:49-85 / VecMDot_MPI pvec2.c:10-25:
   local kernel per rank, then a SUM over ranks (rank order here).                              */
static double orc_op_dot(const OrcOperator *op, const double *x, const double *y)
{
  OInt r; double s = 0.0;
  double *part = (double *)malloc((size_t)op->size * sizeof(double));
  ORC_FOR_RANKS(op, r) part[r] = orc_vecdot(op->range[r + 1] - op->range[r], x + op->range[r], y + op->range[r]);
  for (r = 0; r < op->size; r++) s += part[r];  /* MPI_Allreduce(SUM): rank order here */
  free(part);
  return s;
}
static double orc_op_norm2(const OrcOperator *op, const double *x)
{
  OInt r; double s = 0.0;
  double *part = (double *)malloc((size_t)op->size * sizeof(double));
  ORC_FOR_RANKS(op, r) {
    OInt n = op->range[r + 1] - op->range[r], i; const double *xx = x + op->range[r]; double w = 0.0;
    for (i = 0; i < n; i++) w += xx[i] * xx[i];
    part[r] = w;
  }
  for (r = 0; r < op->size; r++) s += part[r]; /* pvec2.c:59-65: local dot, Allreduce SUM, sqrt */
  free(part);
  return sqrt(s);
}
static void orc_op_mdot(const OrcOperator *op, const double *x, OInt nv, double *const *y, double *z)
{
  OInt r, k; double work[128]; const double *yy[128];
  for (k = 0; k < nv; k++) z[k] = 0.0;
  for (r = 0; r < op->size; r++) {
    for (k = 0; k < nv; k++) yy[k] = y[k] + op->range[r];
    orc_vecmdot(op->range[r + 1] - op->range[r], x + op->range[r], nv, yy, work);
    for (k = 0; k < nv; k++) z[k] += work[k];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* KSP drivers.  Convergence test KSPConvergedDefault, src/ksp/ksp/interface/iterativ.c:744-817.
   reason codes: include/petscksp.h KSPConvergedReason.                                         */
enum { ORC_CONVERGED_RTOL = 2, ORC_CONVERGED_ATOL = 3, ORC_CONVERGED_ITS = 4,
       ORC_DIVERGED_NULL = -2, ORC_DIVERGED_ITS = -3, ORC_DIVERGED_DTOL = -4, ORC_DIVERGED_BREAKDOWN = -5,
       ORC_DIVERGED_NANORINF = -9, ORC_DIVERGED_INDEFINITE_PC = -8, ORC_DIVERGED_INDEFINITE_MAT = -10 };

typedef struct {
  double rtol, abstol, divtol; OInt max_it;
  int    guess_nonzero;
  int    normtype;        /* 1 PRECONDITIONED, 2 UNPRECONDITIONED, 3 NATURAL (KSPNormType) */
  int    pc_type;         /* 0 none, 1 jacobi */
  OInt   restart;         /* GMRES */
  int    cgs_refine;      /* 0 never, 1 ifneeded, 2 always (KSPGMRESCGSRefinementType) */
  double haptol;
  /* outputs */
  OInt   its; int reason; double rnorm, rnorm0, ttol;
  OInt   nhist; double *hist; OInt hist_cap;
} OrcKSP;

static int orc_converged_default(OrcKSP *k, OInt n, double rnorm, const OrcOperator *op, const double *b, const double *d, double *work)
{
  if (!n) {
    if (k->guess_nonzero) { /* iterativ.c:761-788, left PC: snorm = ||B b|| (preconditioned) or ||b|| */
      double snorm;
      if (k->normtype == 2) snorm = orc_op_norm2(op, b);
      else {
        if (d) orc_vecpointwisemult(op->N, work, b, d); else orc_veccopy(op->N, b, work);
        if (k->normtype == 1) snorm = orc_op_norm2(op, work);
        else snorm = sqrt(fabs(orc_op_dot(op, b, work)));
      }
      if (!snorm) snorm = rnorm;
      k->rnorm0 = snorm;
    } else k->rnorm0 = rnorm;
    k->ttol = (k->rtol * k->rnorm0 > k->abstol) ? k->rtol * k->rnorm0 : k->abstol;
  }
  if (isnan(rnorm) || isinf(rnorm)) return ORC_DIVERGED_NANORINF;
  if (rnorm <= k->ttol) return (rnorm < k->abstol) ? ORC_CONVERGED_ATOL : ORC_CONVERGED_RTOL;
  if (rnorm >= k->divtol * k->rnorm0) return ORC_DIVERGED_DTOL;
  return 0;
}

static void orc_hist(OrcKSP *k, double r) { if (k->hist && k->nhist < k->hist_cap) k->hist[k->nhist] = r; k->nhist++; }

/* elementwise kernels rank by rank (every rank owns a slice of the vectors, as the MPI Vec type does): the same values as one
   call over the whole vector, on op->threads threads when asked */
static void orc_op_pointwisemult(const OrcOperator *op, double *w, const double *x, const double *y)
{ OInt r; ORC_FOR_RANKS(op, r) orc_vecpointwisemult(op->range[r + 1] - op->range[r], w + op->range[r], x + op->range[r], y + op->range[r]); }
static void orc_op_copy(const OrcOperator *op, const double *x, double *y)
{ OInt r; ORC_FOR_RANKS(op, r) orc_veccopy(op->range[r + 1] - op->range[r], x + op->range[r], y + op->range[r]); }
static void orc_op_set(const OrcOperator *op, double *x, double a)
{ OInt r; ORC_FOR_RANKS(op, r) orc_vecset(op->range[r + 1] - op->range[r], x + op->range[r], a); }
static void orc_op_axpy(const OrcOperator *op, double *y, double a, const double *x)
{ OInt r; ORC_FOR_RANKS(op, r) orc_vecaxpy(op->range[r + 1] - op->range[r], y + op->range[r], a, x + op->range[r]); }
static void orc_op_aypx(const OrcOperator *op, double *y, double a, const double *x)
{ OInt r; ORC_FOR_RANKS(op, r) orc_vecaypx(op->range[r + 1] - op->range[r], y + op->range[r], a, x + op->range[r]); }

static void orc_pcapply(const OrcOperator *op, const double *d, const double *x, double *y)
{
  if (d) orc_op_pointwisemult(op, y, x, d);    /* PCApply_Jacobi jacobi.c:288 */
  else orc_op_copy(op, x, y);                  /* PCNONE */
}

/* KSPSolve_CG, src/ksp/ksp/impls/cg/cg.c:93-279 (no single-reduction, no eigenvalue estimate);
   KSPSolve's zero-guess VecSet(x,0) itfunc.c:580.  d = Jacobi diagonal (already inverted) or NULL. */
ORC_EXPORT int orc_ksp_cg(OrcOperator *op, OrcKSP *ksp, const double *d, const double *B, double *X)
{
  OInt   N = op->N, i;
  double dpi = 0.0, a = 1.0, beta = 0.0, betaold = 1.0, b = 0, dpiold, dp = 0.0;
  double *R = (double *)malloc((size_t)N * sizeof(double)), *Z = (double *)malloc((size_t)N * sizeof(double));
  double *P = (double *)malloc((size_t)N * sizeof(double)), *W = Z, *T = (double *)malloc((size_t)N * sizeof(double));
  ksp->its = 0; ksp->reason = 0; ksp->nhist = 0;
  if (ksp->guess_nonzero) { orc_operator_mult(op, X, R); orc_op_aypx(op, R, -1.0, B); } /* cg.c:131-132 */
  else { orc_op_set(op, X, 0.0); orc_op_copy(op, B, R); }                                /* :134 */
  switch (ksp->normtype) {
  case 1: orc_pcapply(op, d, R, Z); dp = orc_op_norm2(op, Z); break;                   /* :139-140 */
  case 2: dp = orc_op_norm2(op, R); break;                                             /* :143 */
  case 3: orc_pcapply(op, d, R, Z); beta = orc_op_dot(op, Z, R); dp = sqrt(fabs(beta)); break; /* :146-153 */
  default: dp = 0.0;
  }
  orc_hist(ksp, dp); ksp->rnorm = dp;
  ksp->reason = orc_converged_default(ksp, 0, dp, op, B, d, T);                        /* :164 */
  if (ksp->reason) goto done;
  if (ksp->normtype != 1 && ksp->normtype != 3) orc_pcapply(op, d, R, Z);              /* :167-169 */
  if (ksp->normtype != 3) beta = orc_op_dot(op, Z, R);                                 /* :175 */
  i = 0;
  do {
    ksp->its = i + 1;
    if (beta == 0.0) { ksp->reason = ORC_CONVERGED_ATOL; break; }                      /* :182-185 */
    else if ((i > 0) && (beta * betaold < 0.0)) { ksp->reason = ORC_DIVERGED_INDEFINITE_PC; break; }
    if (!i) { orc_op_copy(op, Z, P); b = 0.0; }                                         /* :194 */
    else { b = beta / betaold; orc_op_aypx(op, P, b, Z); }                              /* :197,202 */
    dpiold = dpi;
    orc_operator_mult(op, P, W);                                                       /* :206 */
    dpi = orc_op_dot(op, P, W);                                                        /* :207 */
    betaold = beta;
    if (isnan(beta) || isinf(beta)) { ksp->reason = ORC_DIVERGED_NANORINF; break; }    /* KSPCheckDot :213 */
    if ((dpi == 0.0) || ((i > 0) && (dpi * dpiold <= 0.0))) { ksp->reason = ORC_DIVERGED_INDEFINITE_MAT; break; }
    a = beta / dpi;                                                                    /* :220 */
    orc_op_axpy(op, X, a, P);                                                           /* :222 */
    orc_op_axpy(op, R, -a, W);                                                          /* :223 */
    if (ksp->normtype == 1) { orc_pcapply(op, d, R, Z); dp = orc_op_norm2(op, Z); }    /* :225,229 */
    else if (ksp->normtype == 2) dp = orc_op_norm2(op, R);                             /* :231 */
    else if (ksp->normtype == 3) { orc_pcapply(op, d, R, Z); beta = orc_op_dot(op, Z, R); dp = sqrt(fabs(beta)); }
    else dp = 0.0;
    ksp->rnorm = dp; orc_hist(ksp, dp);
    ksp->reason = orc_converged_default(ksp, i + 1, dp, op, B, d, T);                  /* :252 */
    if (ksp->reason) break;
    if (ksp->normtype != 1 && ksp->normtype != 3) orc_pcapply(op, d, R, Z);            /* :255-256 */
    if (ksp->normtype != 3) beta = orc_op_dot(op, Z, R);                               /* :269 */
    i++;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = ORC_DIVERGED_ITS;                                /* :276 */
done:
  free(R); free(Z); free(P); free(T);
  return 0;
}

/* KSPSolve_PIPECG, src/ksp/ksp/impls/cg/pipecg/pipecg.c:31-169.  The split reductions (VecNormBegin/VecDotBegin ... End, comb.c)
   deliver the same values as the blocking calls: local sums, then one SUM over the ranks.                       */
ORC_EXPORT int orc_ksp_pipecg(OrcOperator *op, OrcKSP *ksp, const double *d, const double *B, double *X)
{
  OInt   N = op->N, i, q;
  double alpha = 0.0, beta = 0.0, gamma = 0.0, gammaold = 0.0, delta = 0.0, dp = 0.0;
  double *w[9], *M, *Z, *P, *Nn, *W, *Q, *U, *R, *S;
  for (q = 0; q < 9; q++) w[q] = (double *)calloc((size_t)N + 1, sizeof(double));
  M = w[0]; Z = w[1]; P = w[2]; Nn = w[3]; W = w[4]; Q = w[5]; U = w[6]; R = w[7]; S = w[8];
  ksp->its = 0; ksp->reason = 0; ksp->nhist = 0;
  if (ksp->guess_nonzero) { orc_operator_mult(op, X, R); orc_op_aypx(op, R, -1.0, B); }   /* :62-63 */
  else { orc_op_set(op, X, 0.0); orc_op_copy(op, B, R); }                                 /* itfunc.c:580, :65 */
  orc_pcapply(op, d, R, U);                                                                /* :68 */
  switch (ksp->normtype) {
  case 1: orc_operator_mult(op, U, W); dp = orc_op_norm2(op, U); break;                    /* :72-75 */
  case 2: orc_operator_mult(op, U, W); dp = orc_op_norm2(op, R); break;                    /* :78-81 */
  case 3: orc_operator_mult(op, U, W); gamma = orc_op_dot(op, R, U);                       /* :84-87 */
          if (isnan(gamma) || isinf(gamma)) { ksp->reason = ORC_DIVERGED_NANORINF; goto done; }
          dp = sqrt(fabs(gamma)); break;
  default: orc_operator_mult(op, U, W); dp = 0.0;
  }
  orc_hist(ksp, dp); ksp->rnorm = dp;
  ksp->reason = orc_converged_default(ksp, 0, dp, op, B, d, M);                            /* :101 */
  if (ksp->reason) goto done;
  i = 0;
  do {
    orc_pcapply(op, d, W, M);                                                              /* :117 */
    orc_operator_mult(op, M, Nn);                                                          /* :118 */
    if (i > 0 && ksp->normtype == 2) dp = orc_op_norm2(op, R);                             /* :106-110 / :120-124 */
    else if (i > 0 && ksp->normtype == 1) dp = orc_op_norm2(op, U);
    if (!(i == 0 && ksp->normtype == 3)) gamma = orc_op_dot(op, R, U);                     /* :111-113 / :125-127 */
    delta = orc_op_dot(op, W, U);                                                          /* :114 / :128 */
    if (i > 0) {
      if (ksp->normtype == 3) dp = sqrt(fabs(gamma));
      else if (ksp->normtype != 1 && ksp->normtype != 2) dp = 0.0;
      ksp->rnorm = dp; orc_hist(ksp, dp);
      ksp->reason = orc_converged_default(ksp, i, dp, op, B, d, M);                      /* :137 (the work vector is only written at i == 0) */
      if (ksp->reason) break;
    }
    if (i == 0) {
      alpha = gamma / delta;
      orc_op_copy(op, Nn, Z); orc_op_copy(op, M, Q); orc_op_copy(op, U, P); orc_op_copy(op, W, S);   /* :143-146 */
    } else {
      beta = gamma / gammaold;
      alpha = gamma / (delta - beta / alpha * gamma);                                      /* :149 */
      orc_op_aypx(op, Z, beta, Nn); orc_op_aypx(op, Q, beta, M); orc_op_aypx(op, P, beta, U); orc_op_aypx(op, S, beta, W); /* :150-153 */
    }
    orc_op_axpy(op, X, alpha, P); orc_op_axpy(op, U, -alpha, Q); orc_op_axpy(op, W, -alpha, Z); orc_op_axpy(op, R, -alpha, S); /* :155-158 */
    gammaold = gamma;
    i++;
    ksp->its = i;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it && !ksp->reason) ksp->reason = ORC_DIVERGED_ITS;
done:
  for (q = 0; q < 9; q++) free(w[q]);
  return 0;
}

/* KSPSolve_GMRES / KSPGMRESCycle / KSPGMRESBuildSoln / KSPGMRESUpdateHessenberg,
   src/ksp/ksp/impls/gmres/gmres.c:118-246,316-418, classical Gram-Schmidt borthog2.c:36-111,
   KSPInitialResidual itres.c:39-70, left preconditioning KSP_PCApplyBAorAB = MatMult then PCApply
   (src/ksp/pc/interface/precon.c:722-734), VecNormalize rvector.c:328-347.                     */
ORC_EXPORT int orc_ksp_gmres(OrcOperator *op, OrcKSP *ksp, const double *d, const double *B, double *X)
{
  OInt   N = op->N, max_k = ksp->restart, it, j, itcount = 0, its;
  OInt   ldh = max_k + 2;
  double *hh = (double *)calloc((size_t)(max_k + 2) * (max_k + 1), sizeof(double));
  double *rs = (double *)calloc((size_t)max_k + 2, sizeof(double));
  double *cc = (double *)calloc((size_t)max_k + 1, sizeof(double)), *ss = (double *)calloc((size_t)max_k + 1, sizeof(double));
  double *lhh = (double *)calloc((size_t)max_k + 2, sizeof(double)), *nrs = (double *)calloc((size_t)max_k + 2, sizeof(double));
  double **VV = (double **)calloc((size_t)max_k + 2, sizeof(double *));
  double *TEMP = (double *)malloc((size_t)N * sizeof(double)), *TMATOP = (double *)malloc((size_t)N * sizeof(double));
  double *T = (double *)malloc((size_t)N * sizeof(double));
  int    guess_zero = !ksp->guess_nonzero;
#define HH(a, b) (hh + (b) * ldh + (a))
  for (j = 0; j < max_k + 2; j++) VV[j] = (double *)malloc((size_t)N * sizeof(double));
  ksp->its = 0; ksp->reason = 0; ksp->nhist = 0;
  if (guess_zero) orc_vecset(N, X, 0.0);
  while (!ksp->reason) {
    double res_norm, res, tt, hapbnd; int hapend = 0;
    /* KSPInitialResidual itres.c:51-68 */
    if (!guess_zero) {
      orc_operator_mult(op, X, TEMP); orc_veccopy(N, B, TMATOP); orc_vecaxpy(N, TMATOP, -1.0, TEMP);
      orc_pcapply(op, d, TMATOP, VV[0]);
    } else { orc_veccopy(N, B, TMATOP); orc_pcapply(op, d, B, VV[0]); }
    /* KSPGMRESCycle gmres.c:118-214 */
    it = 0;
    res_norm = orc_op_norm2(op, VV[0]);                                     /* VecNormalize :127 */
    if (res_norm != 0.0 && res_norm != 1.0) orc_vecscale(N, VV[0], 1.0 / res_norm);
    res = res_norm; rs[0] = res_norm;
    ksp->rnorm = res;
    orc_hist(ksp, res);
    if (!res) { ksp->reason = ORC_CONVERGED_ATOL; its = 0; goto cycle_done; }
    ksp->reason = orc_converged_default(ksp, ksp->its, res, op, B, d, T);
    while (!ksp->reason && it < max_k && ksp->its < ksp->max_it) {
      int refine = (ksp->cgs_refine == 2);
      if (it) orc_hist(ksp, res);
      orc_operator_mult(op, VV[it], TMATOP); orc_pcapply(op, d, TMATOP, VV[it + 1]); /* :155 */
      /* classical GS, borthog2.c:53-108 */
      for (j = 0; j <= it; j++) *HH(j, it) = 0.0;
      orc_op_mdot(op, VV[it + 1], it + 1, VV, lhh);
      for (j = 0; j <= it; j++) {
        if (isnan(lhh[j]) || isinf(lhh[j])) { ksp->reason = ORC_DIVERGED_NANORINF; }
        lhh[j] = -lhh[j];
      }
      if (ksp->reason) break;
      orc_vecmaxpy(N, VV[it + 1], it + 1, lhh, (const double *const *)VV);
      for (j = 0; j <= it; j++) *HH(j, it) -= lhh[j];
      if (ksp->cgs_refine == 1) {
        double hnrm = 0.0, wnrm;
        for (j = 0; j <= it; j++) hnrm += lhh[j] * lhh[j];
        hnrm = sqrt(hnrm);
        wnrm = orc_op_norm2(op, VV[it + 1]);
        if (wnrm < hnrm) refine = 1;
      }
      if (refine) {
        orc_op_mdot(op, VV[it + 1], it + 1, VV, lhh);
        for (j = 0; j <= it; j++) lhh[j] = -lhh[j];
        orc_vecmaxpy(N, VV[it + 1], it + 1, lhh, (const double *const *)VV);
        for (j = 0; j <= it; j++) *HH(j, it) -= lhh[j];
      }
      tt = orc_op_norm2(op, VV[it + 1]);                                    /* VecNormalize :162 */
      if (tt != 0.0 && tt != 1.0) orc_vecscale(N, VV[it + 1], 1.0 / tt);
      *HH(it + 1, it) = tt;
      hapbnd = fabs(tt / rs[it]);                                           /* :169-174 */
      if (hapbnd > ksp->haptol) hapbnd = ksp->haptol;
      if (tt < hapbnd) hapend = 1;
      { /* KSPGMRESUpdateHessenberg :369-418 */
        double *h = HH(0, it), *c = cc, *s = ss, t2;
        for (j = 1; j <= it; j++) { t2 = *h; *h = *c * t2 + *s * *(h + 1); h++; *h = *c++ * *h - (*s++ * t2); }
        if (!hapend) {
          t2 = sqrt(*h * *h + *(h + 1) * *(h + 1));
          if (t2 == 0.0) { ksp->reason = ORC_DIVERGED_NULL; }
          else {
            *c = *h / t2; *s = *(h + 1) / t2;
            rs[it + 1] = -(*s * rs[it]);
            rs[it] = *c * rs[it];
            *h = *c * *h + *s * *(h + 1);
            res = fabs(rs[it + 1]);
          }
        } else res = 0.0;
      }
      it++; ksp->its++; ksp->rnorm = res;
      if (ksp->reason) break;
      ksp->reason = orc_converged_default(ksp, ksp->its, res, op, B, d, T);
      if (hapend && !ksp->reason) { ksp->reason = ORC_DIVERGED_BREAKDOWN; break; }
    }
    if (it && (ksp->reason || ksp->its >= ksp->max_it)) orc_hist(ksp, res);  /* :198-201 */
    its = it;
    /* KSPGMRESBuildSoln(GRS(0), x, x, it-1) :316-363 */
    {
      OInt itb = it - 1, ii, k2;
      if (itb >= 0) {
        if (*HH(itb, itb) != 0.0) nrs[itb] = rs[itb] / *HH(itb, itb);
        else { ksp->reason = ORC_DIVERGED_BREAKDOWN; goto cycle_done; }
        for (ii = 1; ii <= itb; ii++) {
          double t3;
          k2 = itb - ii; t3 = rs[k2];
          for (j = k2 + 1; j <= itb; j++) t3 = t3 - *HH(k2, j) * nrs[j];
          if (*HH(k2, k2) == 0.0) { ksp->reason = ORC_DIVERGED_BREAKDOWN; goto cycle_done; }
          nrs[k2] = t3 / *HH(k2, k2);
        }
        orc_vecset(N, TEMP, 0.0);
        orc_vecmaxpy(N, TEMP, itb + 1, nrs, (const double *const *)VV);
        orc_vecaxpy(N, X, 1.0, TEMP);   /* left PC: KSPUnwindPreconditioner is a no-op, itres.c:114-116 */
      }
    }
cycle_done:
    itcount += its;
    if (itcount >= ksp->max_it) { if (!ksp->reason) ksp->reason = ORC_DIVERGED_ITS; break; } /* gmres.c:238-241 */
    guess_zero = 0;
  }
#undef HH
  for (j = 0; j < max_k + 2; j++) free(VV[j]);
  free(VV); free(hh); free(rs); free(cc); free(ss); free(lhh); free(nrs); free(TEMP); free(TMATOP); free(T);
  return 0;
}

ORC_EXPORT OrcKSP *orc_ksp_create(void)
{
  OrcKSP *k = (OrcKSP *)calloc(1, sizeof(*k));
  /* defaults: src/ksp/ksp/interface/itcreate.c:706-714; GMRES restart 30, haptol 1e-30, refine never
     (src/ksp/ksp/impls/gmres/gmres.c:925-933) */
  k->rtol = 1e-5; k->abstol = 1e-50; k->divtol = 1e4; k->max_it = 10000;
  k->normtype = 1; k->pc_type = 1; k->restart = 30; k->cgs_refine = 0; k->haptol = 1e-30;
  return k;
}
ORC_EXPORT void orc_ksp_set(OrcKSP *k, double rtol, double abstol, double divtol, OInt max_it, int guess_nonzero,
                            int normtype, OInt restart, int cgs_refine)
{
  k->rtol = rtol; k->abstol = abstol; k->divtol = divtol; k->max_it = max_it; k->guess_nonzero = guess_nonzero;
  k->normtype = normtype; k->restart = restart; k->cgs_refine = cgs_refine;
}
ORC_EXPORT void orc_ksp_set_history(OrcKSP *k, double *hist, OInt cap) { k->hist = hist; k->hist_cap = cap; k->nhist = 0; }
ORC_EXPORT void orc_ksp_get(const OrcKSP *k, OInt *its, int *reason, double *rnorm, OInt *nhist)
{
  *its = k->its; *reason = k->reason; *rnorm = k->rnorm; *nhist = k->nhist;
}
ORC_EXPORT void orc_ksp_destroy(OrcKSP *k) { free(k); }
