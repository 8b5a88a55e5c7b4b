// Device CSR matrix and the row-block SpMV of libpetschip (see include/petschip.h).
//
// Data layout in HBM (all library-owned):
//   rowptr[R+1] int32   R = m, or cprow.nrows when the host matrix uses compressed rows
//   ridx[R]     int32   only with compressed rows: local row of each stored row (Mat_CompressedRow.rindex)
//   col[nz+pad] int32   sorted within a row, exactly Mat_SeqAIJ.j
//   val[nz+pad] f64     exactly Mat_SeqAIJ.a
//   blkrow[B+1] int32   row-block plan: block b owns stored rows [blkrow[b], blkrow[b+1])
//   blktpr[B]   uint8   log2(threads per row) the block's reduce phase uses
//
// Kernel (one 256-thread workgroup per row block, "CSR-stream"):
//   phase 1  the block's nonzeros (<= SPMV_CAP, contiguous in col/val) are streamed with 16-byte (val) and 8-byte (col)
//            fully coalesced loads, multiplied with the gathered x entries and the PRODUCTS are staged in LDS;
//   phase 2  each row is summed from LDS by 1..64 lanes.  With one lane per row the sum runs k-ascending from 0.0 (or from
//            y[i] for MultAdd) over products rounded to double -- the arithmetic of PetscSparseDensePlusDot
//            (src/mat/impls/aij/seq/aij.h:409-411) compiled without FMA contraction, bit for bit.
//   A row with more than SPMV_CAP nonzeros gets a block of its own and is reduced by the whole workgroup straight from HBM.
// Roofline: HBM.  Algorithmic bytes per launch: 12*nz + 20*m + 4 (MultAdd: + 8*m).
#include "phip_internal.h"
#include <stdlib.h>
#include <vector>

#define SPMV_CAP 2046          /* nonzeros per row block; +2 alignment slack = 2048 LDS doubles = 16 KiB */
#define SPMV_PAIRS 1024        /* (SPMV_CAP + 2) / 2 */
#define SPMV_ROWS_MAX 256      /* rows per block */

struct phip_mat_s {
  phip_ctx ctx;
  int32_t  m, n;
  int64_t  nnz;
  int      use_cprow;
  int32_t  R; /* stored rows */
  int32_t *d_rowptr, *d_ridx, *d_col;
  double  *d_val;
  int32_t *d_blkrow;
  uint8_t *d_blktpr;
  int32_t  nblocks;
  int32_t  tpr_user;
  std::vector<int32_t> *h_rowptr, *h_blkrow;
  std::vector<uint8_t> *h_blktpr;
};

template <int TPR>
__device__ __forceinline__ void spmv_reduce_rows(const double *prod, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx,
                                                 const double *__restrict__ yin, double *__restrict__ yout, int r0, int r1, int abase,
                                                 bool add, bool cprow)
{
  const int t = threadIdx.x;
  constexpr int RPP = PHIP_THREADS / TPR; /* rows per pass */
  const int     nrows = r1 - r0;
  const int     passes = (nrows + RPP - 1) / RPP;
  const int     sub = t % TPR;
  for (int p = 0; p < passes; p++) {
    const int  r = r0 + p * RPP + t / TPR;
    const bool valid = r < r1;
    int        s = 0, e = 0;
    if (valid) { s = rowptr[r] - abase; e = rowptr[r + 1] - abase; }
    double sum = 0.0;
    int    rout = 0;
    if (TPR == 1) {
      if (valid) {
        rout = cprow ? ridx[r] : r;
        if (add) sum = yin[rout];
        for (int j = s; j < e; j++) sum += prod[j];
        yout[rout] = sum;
      }
    } else {
      for (int j = s + sub; j < e; j += TPR) sum += prod[j];
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (valid && sub == 0) {
        rout = cprow ? ridx[r] : r;
        if (add) sum += yin[rout];
        yout[rout] = sum;
      }
    }
  }
}

__global__ __launch_bounds__(PHIP_THREADS) void spmv_rowblock_kernel(const int32_t *__restrict__ blkrow, const uint8_t *__restrict__ blktpr,
                                                                     const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx,
                                                                     const int32_t *__restrict__ col, const double *__restrict__ val,
                                                                     const double *__restrict__ x, const double *__restrict__ yin,
                                                                     double *__restrict__ yout, int add, int cprow)
{
  __shared__ __attribute__((aligned(16))) double prod[2 * SPMV_PAIRS];
  const int t = threadIdx.x;
  const int b = blockIdx.x;
  const int r0 = blkrow[b], r1 = blkrow[b + 1];
  const int base = rowptr[r0], end = rowptr[r1];

  if (end - base > SPMV_CAP) {
    /* one long row (the plan guarantees r1 == r0 + 1): whole workgroup strides over it */
    double sum = 0.0;
    for (int j = base + t; j < end; j += PHIP_THREADS) sum += val[j] * x[col[j]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((t & 63) == 0) prod[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
      const int rout = cprow ? ridx[r0] : r0;
      double    s = (prod[0] + prod[1]) + (prod[2] + prod[3]);
      if (add) s += yin[rout];
      yout[rout] = s;
    }
    return;
  }

  /* phase 1: stream products into LDS.  abase is even so val pairs are 16-byte aligned. */
  const int abase = base & ~1;
  double2   a[4];
  int2      c[4];
  bool      v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int e = abase + 2 * (t + k * PHIP_THREADS);
    v[k] = e < end;
    if (v[k]) {
      a[k] = *reinterpret_cast<const double2 *>(val + e);
      c[k] = *reinterpret_cast<const int2 *>(col + e);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (v[k]) {
      double2 p;
      p.x = a[k].x * x[c[k].x];
      p.y = a[k].y * x[c[k].y];
      *reinterpret_cast<double2 *>(prod + 2 * (t + k * PHIP_THREADS)) = p;
    }
  }
  __syncthreads();

  /* phase 2: per-row sums out of LDS */
  switch (blktpr[b]) {
  case 0: spmv_reduce_rows<1>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  case 1: spmv_reduce_rows<2>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  case 2: spmv_reduce_rows<4>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  case 3: spmv_reduce_rows<8>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  case 4: spmv_reduce_rows<16>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  case 5: spmv_reduce_rows<32>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  default: spmv_reduce_rows<64>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow); break;
  }
}

__global__ void getdiag_kernel(int32_t m, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx, int32_t R,
                               const int32_t *__restrict__ col, const double *__restrict__ val, double *__restrict__ v, int cprow)
{
  /* MatGetDiagonal_SeqAIJ, aij.c:1190-1199: first k with j[k]==i, else 0 (v pre-zeroed by the launcher) */
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int i = cprow ? ridx[r] : r;
  double    d = 0.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; k++) {
    if (col[k] == i) { d = val[k]; break; }
  }
  v[i] = d;
}

/* ---- host: row-block plan ---------------------------------------------------------------------- */
static uint8_t choose_tpr_log2(int nrows, int maxlen)
{
  int best = 0, bestcost = 1 << 30;
  for (int l = 0; l <= 6; l++) {
    const int tpr = 1 << l, rpp = PHIP_THREADS / tpr;
    const int cost = ((nrows + rpp - 1) / rpp) * (((maxlen + tpr - 1) / tpr) + 2) + 2 * l;
    if (cost < bestcost) { bestcost = cost; best = l; }
  }
  return (uint8_t)best;
}

static void build_plan(const std::vector<int32_t> &rp, int32_t R, int32_t tpr_user, std::vector<int32_t> &blkrow, std::vector<uint8_t> &blktpr)
{
  blkrow.clear(); blktpr.clear();
  blkrow.push_back(0);
  int32_t r = 0;
  uint8_t forced = 255;
  if (tpr_user > 0) { forced = 0; while ((1 << forced) < tpr_user && forced < 6) forced++; }
  while (r < R) {
    const int32_t start = r;
    int32_t       len = rp[r + 1] - rp[r];
    int           maxlen = 0;
    if (len > SPMV_CAP) { r++; maxlen = len; }
    else {
      int64_t nnz = 0;
      while (r < R && r - start < SPMV_ROWS_MAX) {
        len = rp[r + 1] - rp[r];
        if (nnz + len > SPMV_CAP) break;
        nnz += len;
        if (len > maxlen) maxlen = len;
        r++;
      }
    }
    blkrow.push_back(r);
    blktpr.push_back(forced != 255 ? forced : choose_tpr_log2(r - start, maxlen));
  }
}

static int upload_plan(phip_mat A)
{
  build_plan(*A->h_rowptr, A->R, A->tpr_user, *A->h_blkrow, *A->h_blktpr);
  A->nblocks = (int32_t)A->h_blktpr->size();
  if (A->ctx->device < 0) return 0; /* planning-only: the plan stays on the host */
  if (A->d_blkrow) { hipFree(A->d_blkrow); A->d_blkrow = NULL; }
  if (A->d_blktpr) { hipFree(A->d_blktpr); A->d_blktpr = NULL; }
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_blkrow, sizeof(int32_t) * ((size_t)A->nblocks + 1)));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_blktpr, (size_t)A->nblocks + 16));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_blkrow, A->h_blkrow->data(), sizeof(int32_t) * ((size_t)A->nblocks + 1), hipMemcpyHostToDevice, A->ctx->stream));
  if (A->nblocks) PHIP_CHECK_HIP(hipMemcpyAsync(A->d_blktpr, A->h_blktpr->data(), (size_t)A->nblocks, hipMemcpyHostToDevice, A->ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  return 0;
}

extern "C" int phip_mat_create(phip_ctx ctx, int32_t m, int32_t n, const int32_t *h_i, const int32_t *h_j, const double *h_a,
                               int use_cprow, int32_t cprow_nrows, const int32_t *h_cprow_i, const int32_t *h_cprow_rindex, phip_mat *out)
{
  PHIP_REQUIRE(ctx && out && h_i, PHIP_ERR_ARG_NULL, "phip_mat_create: null argument");
  PHIP_REQUIRE(m >= 0 && n >= 0, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_create: negative size %d x %d", m, n);
  PHIP_REQUIRE(h_i[0] == 0, PHIP_ERR_ARG_WRONGSTATE, "phip_mat_create: i[0] = %d, matrix not assembled", h_i[0]);
  const int64_t nnz = h_i[m];
  PHIP_REQUIRE(nnz == 0 || (h_j && h_a), PHIP_ERR_ARG_NULL, "phip_mat_create: null j/a with nz=%lld", (long long)nnz);
  if (use_cprow) PHIP_REQUIRE(h_cprow_i && (h_cprow_rindex || !cprow_nrows), PHIP_ERR_ARG_NULL, "phip_mat_create: compressed row arrays missing");
  /* operand checks the kernel relies on: monotone row pointer, column range */
  for (int32_t r = 0; r < m; r++) PHIP_REQUIRE(h_i[r + 1] >= h_i[r], PHIP_ERR_ARG_WRONGSTATE, "phip_mat_create: row pointer decreases at row %d", r);
  for (int64_t k = 0; k < nnz; k++) PHIP_REQUIRE(h_j[k] >= 0 && h_j[k] < n, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_create: column %d out of range [0,%d) at entry %lld", h_j[k], n, (long long)k);
  if (ctx->device >= 0) PHIP_CHECK_HIP(hipSetDevice(ctx->device));
  phip_mat A = new phip_mat_s();
  memset(A, 0, sizeof(*A));
  A->ctx = ctx; A->m = m; A->n = n; A->nnz = nnz; A->use_cprow = use_cprow ? 1 : 0;
  A->R = use_cprow ? cprow_nrows : m;
  A->h_rowptr = new std::vector<int32_t>((size_t)A->R + 1);
  A->h_blkrow = new std::vector<int32_t>();
  A->h_blktpr = new std::vector<uint8_t>();
  const int32_t *rp = use_cprow ? h_cprow_i : h_i;
  for (int32_t r = 0; r <= A->R; r++) (*A->h_rowptr)[r] = rp[r];
  if (use_cprow) {
    PHIP_REQUIRE(rp[A->R] == nnz, PHIP_ERR_ARG_WRONGSTATE, "phip_mat_create: compressed row pointer ends at %d, nz=%lld", rp[A->R], (long long)nnz);
    for (int32_t r = 0; r < A->R; r++) PHIP_REQUIRE(h_cprow_rindex[r] >= 0 && h_cprow_rindex[r] < m, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_create: rindex[%d]=%d", r, h_cprow_rindex[r]);
  }
  const size_t pad = 8;
  if (ctx->device < 0) { /* planning-only: no device copy */
    int prc = upload_plan(A);
    if (prc) return prc;
    *out = A;
    return 0;
  }
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_rowptr, sizeof(int32_t) * ((size_t)A->R + 1)));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_col, sizeof(int32_t) * ((size_t)nnz + pad)));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_val, sizeof(double) * ((size_t)nnz + pad)));
  PHIP_CHECK_HIP(hipMemsetAsync(A->d_col + nnz, 0, sizeof(int32_t) * pad, ctx->stream));
  PHIP_CHECK_HIP(hipMemsetAsync(A->d_val + nnz, 0, sizeof(double) * pad, ctx->stream));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_rowptr, A->h_rowptr->data(), sizeof(int32_t) * ((size_t)A->R + 1), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) {
    PHIP_CHECK_HIP(hipMemcpyAsync(A->d_col, h_j, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    PHIP_CHECK_HIP(hipMemcpyAsync(A->d_val, h_a, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  if (use_cprow) {
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_ridx, sizeof(int32_t) * ((size_t)A->R + 1)));
    if (A->R) PHIP_CHECK_HIP(hipMemcpyAsync(A->d_ridx, h_cprow_rindex, sizeof(int32_t) * (size_t)A->R, hipMemcpyHostToDevice, ctx->stream));
  }
  PHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  int rc = upload_plan(A);
  if (rc) return rc;
  *out = A;
  return 0;
}

extern "C" int phip_mat_update_values(phip_mat A, const double *h_a)
{
  PHIP_REQUIRE(A && (h_a || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_update_values: null argument");
  if (A->ctx->device < 0) return 0;
  PHIP_CHECK_HIP(hipSetDevice(A->ctx->device));
  if (A->nnz) PHIP_CHECK_HIP(hipMemcpyAsync(A->d_val, h_a, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice, A->ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  return 0;
}

extern "C" int phip_mat_destroy(phip_mat A)
{
  if (!A) return 0;
  if (A->ctx->device < 0) { delete A->h_rowptr; delete A->h_blkrow; delete A->h_blktpr; delete A; return 0; }
  hipSetDevice(A->ctx->device);
  hipStreamSynchronize(A->ctx->stream);
  hipFree(A->d_rowptr); hipFree(A->d_ridx); hipFree(A->d_col); hipFree(A->d_val); hipFree(A->d_blkrow); hipFree(A->d_blktpr);
  delete A->h_rowptr; delete A->h_blkrow; delete A->h_blktpr;
  delete A;
  return 0;
}

static int launch_spmv(phip_mat A, const double *d_x, const double *d_yin, double *d_yout, int add)
{
  if (!A->nblocks) return 0;
  hipEvent_t e0 = NULL, e1 = NULL;
  if (A->ctx->prof_on) {
    int rc = phip_prof_open(A->ctx, add ? 1 : 0, &e0, &e1);
    if (rc) return rc;
    PHIP_CHECK_HIP(hipEventRecord(e0, A->ctx->stream));
  }
  hipLaunchKernelGGL(spmv_rowblock_kernel, dim3((unsigned)A->nblocks), dim3(PHIP_THREADS), 0, A->ctx->stream, A->d_blkrow, A->d_blktpr, A->d_rowptr,
                     A->d_ridx, A->d_col, A->d_val, d_x, d_yin, d_yout, add, A->use_cprow);
  PHIP_CHECK_HIP(hipGetLastError());
  if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, A->ctx->stream));
  return 0;
}

extern "C" int phip_mat_mult(phip_mat A, const double *d_x, double *d_y)
{
  PHIP_REQUIRE(A && (d_y || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_mult: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_mult");
  PHIP_REQUIRE(d_x != d_y || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_mult: x and y must be different vectors (matrix.c:2213)");
  if (A->use_cprow && A->m) PHIP_CHECK_HIP(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)A->m, A->ctx->stream)); /* aij.c:1296 */
  return launch_spmv(A, d_x, NULL, d_y, 0);
}

extern "C" int phip_mat_multadd(phip_mat A, const double *d_x, const double *d_y, double *d_z)
{
  PHIP_REQUIRE(A && ((d_y && d_z) || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_multadd: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_multadd");
  PHIP_REQUIRE(d_x != d_z || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_multadd: x and z must be different vectors (matrix.c:2361)");
  if (A->use_cprow && d_z != d_y && A->m)
    PHIP_CHECK_HIP(hipMemcpyAsync(d_z, d_y, sizeof(double) * (size_t)A->m, hipMemcpyDeviceToDevice, A->ctx->stream)); /* aij.c:1460-1462 */
  return launch_spmv(A, d_x, d_y, d_z, 1);
}

extern "C" int phip_mat_getdiagonal(phip_mat A, double *d_v)
{
  PHIP_REQUIRE(A && (d_v || !A->m), PHIP_ERR_ARG_NULL, "phip_mat_getdiagonal: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_getdiagonal");
  if (!A->m) return 0;
  PHIP_CHECK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)A->m, A->ctx->stream)); /* VecSet(v,0) aij.c:1188 */
  if (A->R) {
    hipLaunchKernelGGL(getdiag_kernel, dim3((unsigned)((A->R + 255) / 256)), dim3(256), 0, A->ctx->stream, A->m, A->d_rowptr, A->d_ridx, A->R, A->d_col,
                       A->d_val, d_v, A->use_cprow);
    PHIP_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int phip_mat_plan_info(phip_mat A, int32_t *nblocks, int32_t *block_nnz_cap, int32_t *threads_per_row, int64_t *nnz)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_plan_info: null matrix");
  if (nblocks) *nblocks = A->nblocks;
  if (block_nnz_cap) *block_nnz_cap = SPMV_CAP;
  if (threads_per_row) *threads_per_row = A->nblocks ? (1 << (*A->h_blktpr)[0]) : 0;
  if (nnz) *nnz = A->nnz;
  return 0;
}

extern "C" int phip_mat_plan_get(phip_mat A, int32_t *h_block_row)
{
  PHIP_REQUIRE(A && h_block_row, PHIP_ERR_ARG_NULL, "phip_mat_plan_get: null argument");
  memcpy(h_block_row, A->h_blkrow->data(), sizeof(int32_t) * A->h_blkrow->size());
  return 0;
}

extern "C" int phip_mat_set_threads_per_row(phip_mat A, int32_t tpr)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_threads_per_row: null matrix");
  PHIP_REQUIRE(tpr >= 0 && tpr <= 64 && (tpr & (tpr - 1)) == 0, PHIP_ERR_ARG_OUTOFRANGE, "threads per row must be 0 or a power of two <= 64, got %d", tpr);
  A->tpr_user = tpr;
  return upload_plan(A);
}
