/*
 * petschip.h -- C ABI of libpetschip.so: the MI355X (gfx950) kernels behind PETSc's AIJ/Vec/Jacobi
 * ops tables.  Plain C, plain pointers and sizes; no PETSc, torch or C++ types cross this boundary.
 *
 * This is the layer a PETSc-side glue file binds: each MATSEQAIJHIP/MATMPIAIJHIP/VECSEQHIP/VECMPIHIP
 * ops-table entry unpacks the PETSc structs and calls exactly one function below (INTEGRATION.md shows
 * the glue).  Each declaration cites the reference routine (file:line under the PETSc 3.6.1 tree) whose
 * ops-table slot it fills and whose result it reproduces.
 *
 * Conventions
 *   - PetscInt = int32_t, PetscScalar = PetscReal = double.
 *   - Return value: 0 on success, otherwise a PetscErrorCode value (include/petscerror.h:24-68):
 *       PHIP_ERR_MEM 55, PHIP_ERR_SUP 56, PHIP_ERR_ARG_SIZ 60, PHIP_ERR_ARG_WRONG 62, PHIP_ERR_ARG_OUTOFRANGE 63,
 *       PHIP_ERR_ARG_WRONGSTATE 73, PHIP_ERR_LIB 76 (any HIP / RCCL failure), PHIP_ERR_ARG_NULL 85.
 *     phip_last_error() returns the text of the last failure on the calling thread.
 *   - "d_" pointers are device pointers obtained from phip_malloc (or any hipMalloc'd memory);
 *     "h_" pointers are host pointers.  The library owns what it allocates; callers own the rest.
 *   - One phip_ctx per process and GPU.  All work of a context is issued on its compute stream, in
 *     call order; calls that return a host scalar (dot/norm/mdot) synchronise that stream, everything
 *     else is asynchronous.  Contexts are not thread-safe (PETSc 3.6 objects are not either).
 *   - There is no CPU fallback: without a usable GPU phip_init fails with PHIP_ERR_LIB.
 *   - phip_init(-1, ...) makes a PLANNING-ONLY context for hosts without a GPU: index work (row-block plan, ghost plan, set-up
 *     collectives) runs; uploads are dropped; every numerical entry point fails with PHIP_ERR_SUP.  It exists for CPU-side tests
 *     of the host logic and never computes anything.
 */
#ifndef PETSCHIP_H
#define PETSCHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHIP_ERR_MEM            55
#define PHIP_ERR_SUP            56
#define PHIP_ERR_ARG_SIZ        60
#define PHIP_ERR_ARG_WRONG      62
#define PHIP_ERR_ARG_OUTOFRANGE 63
#define PHIP_ERR_ARG_WRONGSTATE 73
#define PHIP_ERR_LIB            76
#define PHIP_ERR_ARG_NULL       85

/* NormType, include/petscvec.h:155 */
#define PHIP_NORM_1         0
#define PHIP_NORM_2         1
#define PHIP_NORM_FROBENIUS 2
#define PHIP_NORM_INFINITY  3
#define PHIP_NORM_1_AND_2   4

#define PHIP_MAX_MVEC 128 /* VecMDot_MPI's stack bound, src/vec/vec/impls/mpi/pvec2.c:13 */

typedef struct phip_ctx_s  *phip_ctx;
typedef struct phip_mat_s  *phip_mat;
typedef struct phip_comm_s *phip_comm;
typedef struct phip_halo_s *phip_halo;

/* ---- context / memory -------------------------------------------------------------------------
 * Replaces the CUSP precedent's implicit global state (VecCUSPAllocateCheck / CopyToGPU / CopyFromGPU,
 * src/vec/vec/impls/seq/seqcusp/veccusp.cu:55-180) with an explicit per-process context. */
int         phip_init(int device, phip_ctx *ctx);
int         phip_finalize(phip_ctx ctx);
const char *phip_last_error(void);
int         phip_device_count(int *count); /* GPUs this process can see (hipGetDeviceCount; 0 when there is no usable driver); needs no context */
int         phip_device_info(phip_ctx ctx, int *device, int *cu_count, size_t *hbm_bytes, char *name, int name_len);
int         phip_malloc(phip_ctx ctx, size_t bytes, void **d_ptr);
int         phip_free(phip_ctx ctx, void *d_ptr);
int         phip_memcpy_h2d(phip_ctx ctx, void *d_dst, const void *h_src, size_t bytes); /* VecCUSPCopyToGPU   veccusp.cu:90  */
int         phip_memcpy_d2h(phip_ctx ctx, void *h_dst, const void *d_src, size_t bytes); /* VecCUSPCopyFromGPU veccusp.cu:152 (synchronises) */
int         phip_memcpy_d2d(phip_ctx ctx, void *d_dst, const void *d_src, size_t bytes);
int         phip_sync(phip_ctx ctx);                                                     /* WaitForGPU, cuspvecimpl.h:66 */
int         phip_device_sync(phip_ctx ctx);                                              /* hipDeviceSynchronize: every stream of the device */
void       *phip_stream(phip_ctx ctx);          /* the hipStream_t all work is issued on */
/* hipEvent timing on the context's stream (PetscLogEventBegin/End's role for device work, include/petsclog.h:284-295):
 * record two events around a region, then phip_event_elapsed_ms synchronises on the second and returns the device time. */
int         phip_event_create(phip_ctx ctx, void **event);
int         phip_event_record(phip_ctx ctx, void *event);
int         phip_event_elapsed_ms(phip_ctx ctx, void *start, void *stop, float *ms);
int         phip_event_destroy(phip_ctx ctx, void *event);
/* Per-launch device timing of the SpMV kernel (the role of the MAT_Mult event of -log_view, src/mat/interface/matrix.c:2224-2226):
 * while enabled, every phip_mat_mult (kind 0) / phip_mat_multadd (kind 1) launch is bracketed by a hipEvent pair on the compute
 * stream; so are the launch of a batch of elementwise operations that leaves with a reducing pointwise product (kind 2: KSPCG's
 * VecAXPY(R,-a,W) + PCApply with its sums) and every other launch of pending elementwise operations (kind 3: KSPCG's VecAXPY(X,a,P) +
 * VecAYPX(P,b,Z)).  phip_profile_read synchronises and returns the number of launches recorded since enable and their summed / extreme
 * durations in ms.  Enabling (or re-enabling) clears the records. */
int         phip_profile_enable(phip_ctx ctx, int on);
int         phip_profile_read(phip_ctx ctx, int kind, int32_t *count, double *sum_ms, double *min_ms, double *max_ms);
/* the individual durations (ms, launch order) behind phip_profile_read: at most cap are written to h_ms, *count = launches recorded */
int         phip_profile_durations(phip_ctx ctx, int kind, int32_t cap, float *h_ms, int32_t *count);

/* ---- device CSR matrix (Mat_SeqAIJ on the GPU) --------------------------------------------------
 * phip_mat_create uploads a host CSR exactly as MatAssemblyEnd_SeqAIJ leaves it (src/mat/impls/aij/seq/aij.c:
 * 968-1024: i[m+1], j[nz] sorted per row, a[nz]) and builds the row-block plan the SpMV kernel runs on.
 * Precedent slot: MatCUSPCopyToGPU, src/mat/impls/aij/seq/seqcusp/aijcusp.cu:110-208.
 * use_cprow/cprow_*: the Mat_CompressedRow of MatCheckCompressedRow (src/mat/utils/compressedrow.c:32-74);
 * pass use_cprow = 0 and NULLs when the host matrix does not use it. */
int phip_mat_create(phip_ctx ctx, int32_t m, int32_t n, const int32_t *h_i, const int32_t *h_j, const double *h_a,
                    int use_cprow, int32_t cprow_nrows, const int32_t *h_cprow_i, const int32_t *h_cprow_rindex,
                    phip_mat *A);
/* same pattern, new values (MatAssemblyEnd with unchanged nonzerostate; aijcusp.cu:451-462 re-uploads everything) */
int phip_mat_update_values(phip_mat A, const double *h_a);
/* Value updates that keep the nonzero pattern, on the device (no value crosses PCIe): the CSR values, the SELL-dict copy, the pair
 * dictionary / diagonal masks (scaled or shifted exactly as the entries they stand for; dropped by a diagonal scaling, whose values
 * depend on the position) and the transposed copy.  phip_mat_get_values brings the current values back to a host CSR of the same
 * pattern when the host needs them again (MatGetRow, MatView, further MatSetValues ...).
 *   phip_mat_scale            a *= alpha                       MatScale_SeqAIJ, aij.c:2475-2488
 *   phip_mat_shift            a(i,i) += alpha                  MatShift (axpy.c:154) -> MatShift_Basic (gcreate.c:6-24); rows without a stored
 *                                                              diagonal entry are left alone: the caller checks (a missing entry changes the pattern)
 *   phip_mat_diagonal_scale   a(i,j) = (a(i,j) l_i) r_j        MatDiagonalScale_SeqAIJ, aij.c:2200-2235; d_l (m entries) or d_r (n entries) may be NULL
 *   phip_mat_zero_entries     a = 0                            MatZeroEntries_SeqAIJ, aij.c:1058-1067 */
int phip_mat_scale(phip_mat A, double alpha);
int phip_mat_shift(phip_mat A, double alpha);
int phip_mat_diagonal_scale(phip_mat A, const double *d_l, const double *d_r);
int phip_mat_zero_entries(phip_mat A);
int phip_mat_get_values(phip_mat A, double *h_a);
int phip_mat_destroy(phip_mat A);
/* y = A x.  MatMult_SeqAIJ, aij.c:1272-1327 (cprow branch :1295-1308 zero-fills y). d_x has n entries, d_y m. */
int phip_mat_mult(phip_mat A, const double *d_x, double *d_y);
/* z = y + A x.  MatMultAdd_SeqAIJ, aij.c:1440-1499; d_z may alias d_y (mpiaij.c:996 calls it that way). */
int phip_mat_multadd(phip_mat A, const double *d_x, const double *d_y, double *d_z);
/* v[i] = a[k] where j[k]==i, else 0.  MatGetDiagonal_SeqAIJ, aij.c:1169-1202. */
/* y = A^T x (MatMultTranspose_SeqAIJ aij.c:1258-1266) and z = y + A^T x (MatMultTransposeAdd_SeqAIJ aij.c:1207-1254; z may be y).
 * phip_mat_transpose_setup builds the operator from the host CSR the matrix was created from (full row pointer, also for a
 * compressed-row matrix): a second device copy holding A^T row by row, whose k-ascending row sums add the products of a column of A
 * in ascending row order -- the order of the reference's scatter-add loop (aij.c:1238-1248) -- with no atomics: the reference's bits
 * in the layouts that sum a row in one lane (SELL-dict; CSR with one lane per row), within 1e-12 otherwise, as for the forward product.
 * phip_mat_update_values refreshes that copy too.  The products fail with PHIP_ERR_ARG_WRONGSTATE before the set-up call. */
int phip_mat_transpose_setup(phip_mat A, const int32_t *h_i, const int32_t *h_j, const double *h_a);
int phip_mat_mult_transpose(phip_mat A, const double *d_x /* m */, double *d_y /* n */);
int phip_mat_mult_transpose_add(phip_mat A, const double *d_x /* m */, const double *d_y /* n */, double *d_z /* n */);
int phip_mat_getdiagonal(phip_mat A, double *d_v);
/* Row-block plan introspection (tests): number of blocks, nnz capacity per block, threads per row used. */
int phip_mat_plan_info(phip_mat A, int32_t *nblocks, int32_t *block_nnz_cap, int32_t *threads_per_row, int64_t *nnz);
int phip_mat_plan_get(phip_mat A, int32_t *h_block_row /* nblocks+1 */);
/* the row binning of the plan (BASELINE config 5 "load-imbalance / row binning"): h_hist[l], l = 0..6, = number of row blocks whose rows
 * are summed by 2^l lanes each; h_hist[7] = number of rows longer than a block's capacity (a whole workgroup strides over such a row) */
int phip_mat_plan_lanes(phip_mat A, int64_t *h_hist /* [8] */);
/* Device layout.  1 = CSR row blocks (any matrix).  2 = "SELL-dict": 512-row slices stored slot-major with a one-byte index into the
 * table of distinct diagonals (col - row) in place of the 4-byte column: 9 instead of 12 bytes per nonzero; built at upload when the
 * matrix has <= 256 distinct diagonals, rows of <= 255 entries and < 15 % slice padding (the stencil matrices of DMDA; the
 * reference's precedent for a second layout is MatCUSPStorageFormat csr/ell/dia, aijcusp.cu:92-100,171-183).  0 = auto.
 * Both layouts give the same bits.  phip_set_default_mat_format applies to matrices created afterwards. */
int phip_set_default_mat_format(int fmt);
int phip_mat_set_format(phip_mat A, int fmt);
int phip_mat_format_info(phip_mat A, int *format, int32_t *ndiagonals, int64_t *padded_entries);
/* Pair dictionary of the SELL-dict layout.  Matrices assembled from constant-coefficient stencils hold few distinct values; when the
 * stored entries form <= 256 distinct (diagonal, value bit pattern) pairs, one byte per entry names both and the kernel reads the
 * value from a 256-entry table in LDS: 1 byte of HBM traffic per nonzero instead of 9.  Lossless: the table holds the doubles the
 * caller stored (keyed on their bits), the sums run in the same order -- same bits as with streamed values.  Built at upload and
 * rebuilt by phip_mat_update_values (a matrix that stops fitting goes back to streamed values).  phip_set_default_value_dict(0) /
 * PHIP_VALUE_DICT=0: never build; phip_mat_set_value_dict switches a built dictionary off and on (A/B measurements, tests). */
int phip_set_default_value_dict(int mode);
int phip_mat_set_value_dict(phip_mat A, int on);
int phip_mat_value_dict_info(phip_mat A, int *built, int *in_use, int32_t *npairs);
/* Diagonal-mask layout, on top of the pair dictionary when it has <= 32 pairs and every row meets its pairs in ascending diagonal
 * order (sorted rows without repeated columns): the pairs are sorted by diagonal and a row is a 32-bit set of the pairs it holds --
 * 4 bytes per ROW.  The kernel loops over the pairs; two adjacent rows of a lane read x with one 16-byte load.  Same products, same
 * k-ascending sums: same bits.  phip_set_default_diag_mask(0) / PHIP_DIAG_MASK=0: never build; 1 (default): build, and use it for
 * the plain products and, from 40 M nonzeros on, for the products with a fused dot (which then take a second, one-workgroup launch
 * that sums one partial per slice; smaller matrices keep the single-launch pair-code kernel there); 2: for every product.
 * phip_mat_set_diag_mask switches a built layout off (back to the pair codes) and on. */
int phip_set_default_diag_mask(int mode);
/* host threads of the set-up passes of phip_mat_create / phip_mat_update_values (SELL-dict arrays, pair codes, masks; matrices of 2^20
 * stored entries and more): 0 (default) = PHIP_SETUP_THREADS or the machine's hardware threads, at most 16.  The layout built does not
 * depend on the number of threads.  27-point 256^3 (449 M nonzeros): phip_mat_create 3.8 s on one thread, 1.1 s on 16. */
int phip_set_setup_threads(int nthreads);
/* which of the two diagonal-mask kernels runs (same bits): 0 (default) by size -- spmv_dmaskg_kernel (group descriptors and transposed
 * masks read by the scalar unit, the masks as execution masks) from 4 Mi rows on or with 12 pairs and more, spmv_dmask4_kernel (a bit
 * set per row) otherwise --
 * 4 / 5: that kernel for every size.  PHIP_DMASK_KERNEL sets the initial value. */
int phip_set_dmask_kernel(int kind);
/* tuning, process-wide: how the CSR row-block and SELL-dict kernels deal their work items (row blocks / slices) to the eight XCDs, each of
 * which has an L2 of its own: runs of `chunk` consecutive items per XCD (a power of two; 0 = the kernels' defaults, 128 row blocks / 8 slices;
 * PHIP_XCD_CHUNK sets the initial value), 1 = item i on XCD i mod 8 (round robin, the hardware's order).  Neighbouring items of a banded matrix share lines of x; in runs
 * they share them inside one L2 instead of fetching them over the fabric once per XCD.  Same bits either way (no reference counterpart:
 * MatMult_SeqAIJ, aij.c:1313-1320, is one loop on one core). */
int phip_set_xcd_chunk(int chunk);
/* the transposed masks of spmv_dmaskg_kernel: waves (of 256 rows) and distinct sets of mask words among them (equal sets are stored
 * once, 1056 bytes each, plus a 4-byte set number per wave); 0 / 0 without the diagonal-mask layout */
int phip_mat_mask_sets_info(phip_mat A, int64_t *nsets, int64_t *nwaves);
int phip_mat_set_diag_mask(phip_mat A, int on);
int phip_mat_diag_mask_info(phip_mat A, int *built, int *in_use, int32_t *npairs);
/* tuning: loads of the matrix stream: -1 (default) non-temporal when the stream exceeds 192 MB (it would only displace x from the caches),
 * 0 always cached, 1 always non-temporal.  Same bits either way. */
int phip_mat_set_nontemporal(phip_mat A, int mode);
/* tuning: threads_per_row (0 = auto), 1,2,4,...,64 */
int phip_mat_set_threads_per_row(phip_mat A, int32_t tpr);
/* Hot-column order of the CSR row-block layout, for irregular matrices whose x is far larger than L2 and whose columns are used very
 * unevenly (power-law graphs; BASELINE config 5): the columns are numbered by descending use, x is copied into that order by every
 * product (20 B per column) and the gathers then find the often-used entries sharing cache lines.  The nonzeros stay where they are,
 * so every row adds the same products in the same order: same bits as without (the reference has no counterpart; its closest relative
 * is a MatPermute by the user, which changes the operator seen by the solver, this does not).  Built at phip_mat_create when
 * 8 n >= 48 MiB and the share of gathers that fall into the 32768 hottest lines of x rises by >= 0.04 (mode 1, the default;
 * PHIP_HOT_COLUMNS or phip_set_default_hot_columns: 0 never, 2 whenever there are nonzeros); phip_mat_set_hot_columns switches an
 * order that was built off and on. */
int phip_set_default_hot_columns(int mode);
int phip_mat_set_hot_columns(phip_mat A, int on);
int phip_mat_hot_columns_info(phip_mat A, int *built, int *in_use, double *share_before, double *share_after);


/* ---- vector kernels (the _VecOps slots, include/petsc/private/vecimpl.h:20-96) --------------------
 * n = local length.  Results of reductions are written to host memory and are valid on return.
 * The elementwise operations (axpy ... reciprocal, set) are asynchronous like every launch, and in addition DEFERRED: they wait in the
 * context until another entry point of this library needs the stream (every one of them launches what is pending first, including
 * phip_sync, the copies, phip_free and phip_stream), and up to four of them leave in one launch that applies them in call order to each
 * entry -- same bits as one launch per operation, fewer launches, operands re-read from cache (PHIP_EW_BATCH=0 in the environment
 * switches the deferral off).  A caller that touches the vectors outside this library asks for phip_stream() or phip_sync() first, as it
 * must for any asynchronous launch. */
/* tuning, process-wide: non-temporal loads (and stores of X and R in the KSPCG batch) in the elementwise kernels: -1 (default) for vectors
 * of 48 MiB and more, 0 never, 1 always.  Same bits either way.  PHIP_EW_NT in the environment sets the initial value. */
int phip_set_ew_nontemporal(int mode);
/* KSPCG's VecAXPY(X,a,P) (cg.c:222) is independent of the rest of its batch (VecAXPY(R,-a,W); PCApply with its sums) and P is read again
 * by the VecAYPX(P,b,Z) of the next iteration (cg.c:202): by default it stays pending when the batch leaves and is launched together with
 * that operation in one kernel that reads P once (xp_update_kernel).  Same arithmetic, same bits; every other entry point launches what
 * is pending first, so nobody can see the old X.  0 switches it off (PHIP_CG_DEFER_X=0 sets the initial value). */
int phip_set_cg_defer_x(int on);
int phip_vec_dot(phip_ctx ctx, int32_t n, const double *d_x, const double *d_y, double *h_z);        /* VecDot_Seq/VecTDot_Seq bvec1.c:12-51 */
int phip_vec_norm(phip_ctx ctx, int32_t n, const double *d_x, int type, double *h_z);                /* VecNorm_Seq bvec2.c:233-271 (h_z[2] for 1_AND_2) */
int phip_vec_mdot(phip_ctx ctx, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *h_z); /* VecMDot_Seq dvec2.c:98-294 */
int phip_vec_maxpy(phip_ctx ctx, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy); /* VecMAXPY_Seq dvec2.c:570-638 */
int phip_vec_axpy(phip_ctx ctx, int32_t n, double *d_y, double alpha, const double *d_x);            /* VecAXPY_Seq bvec1.c:76-95 */
int phip_vec_aypx(phip_ctx ctx, int32_t n, double *d_y, double alpha, const double *d_x);            /* VecAYPX_Seq dvec2.c:644-686 */
int phip_vec_waxpy(phip_ctx ctx, int32_t n, double *d_w, double alpha, const double *d_x, const double *d_y); /* VecWAXPY_Seq dvec2.c:697-730 */
int phip_vec_axpby(phip_ctx ctx, int32_t n, double *d_y, double alpha, double beta, const double *d_x);       /* VecAXPBY_Seq bvec1.c:99-133 */
int phip_vec_axpbypcz(phip_ctx ctx, int32_t n, double *d_z, double alpha, double beta, double gamma, const double *d_x, const double *d_y); /* bvec1.c:137-165 */
int phip_vec_pointwise_mult(phip_ctx ctx, int32_t n, double *d_w, const double *d_x, const double *d_y);      /* VecPointwiseMult_Seq bvec2.c:84-110 */
int phip_vec_pointwise_divide(phip_ctx ctx, int32_t n, double *d_w, const double *d_x, const double *d_y);    /* VecPointwiseDivide_Seq bvec2.c:114-132 */
int phip_vec_scale(phip_ctx ctx, int32_t n, double *d_x, double alpha);                              /* VecScale_Seq bvec1.c:55-72 */
int phip_vec_copy(phip_ctx ctx, int32_t n, const double *d_x, double *d_y);                          /* VecCopy_Seq bvec2.c:192-207 */
int phip_vec_set(phip_ctx ctx, int32_t n, double *d_x, double alpha);                                /* VecSet_Seq dvec2.c:551-566 */
int phip_vec_swap(phip_ctx ctx, int32_t n, double *d_x, double *d_y);                                /* VecSwap_Seq bvec2.c:211-227 */
int phip_vec_reciprocal(phip_ctx ctx, int32_t n, double *d_x);                                       /* VecReciprocal_Default vinv.c:1213-1227 */
/* PCSetUp_Jacobi's host loop (src/ksp/pc/impls/jacobi/jacobi.c:182-195) on the device: d = 1/d (0 stays 0), optional |d|, 0 -> 1.0.
 * h_zeroflag (may be NULL) receives 1 when a zero was replaced.  sqrt_variant selects :196-214 (1/sqrt|d|). */
int phip_jacobi_setup(phip_ctx ctx, int32_t n, double *d_diag, int useabs, int sqrt_variant, int *h_zeroflag);

/* Local halves of the distributed reductions (VecDot_MPI pbvec.c:9-19, VecNorm_MPI pvec2.c:49-85, VecMDot_MPI pvec2.c:10-25):
 * the local kernel leaves its result in device memory (d_out, nv doubles) so that phip_comm_allreduce can combine it without
 * a host round trip; phip_read_scalars then brings the final values to the host.
 * kind: 0 dot(x,y[0]) / mdot(x,y[0..nv-1]);  1 sum x_i^2 (nv=1);  2 sum |x_i| ;  3 max |x_i| (NaN propagates);  4 {sum|x|, sum x^2} */
int phip_vec_reduce_local(phip_ctx ctx, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_out);
/* the same local results delivered to HOST memory (one launch, pinned slot, no stream synchronize): the input of
 * phip_comm_allreduce_host.  Raw sums: kind 1 returns sum x^2, not its square root. */
int phip_vec_reduce_local_host(phip_ctx ctx, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *h_out);
int phip_read_scalars(phip_ctx ctx, const double *d_src, int32_t n, double *h_dst);
/* d_z[k] = sum_i x_i y_k,i, k < nv: phip_vec_reduce_local's kind 0 as an entry point that ALLOCATES NOTHING and needs no context.
 * It has no counterpart in the reference; it is for callers that cannot call hipMalloc: code under HIP graph capture, tight serving
 * loops, harnesses that forbid allocation.  The partial sums of the workgroups live in a file-scope __device__ array of the library
 * (1 MiB + 256 B per GPU, held until the process ends) instead of a context's hipMalloc'ed rows; the kernel is launched on `stream`
 * (a hipStream_t, NULL = the default stream) of the CURRENT device; inputs and the nv results stay in device memory, nothing is
 * synchronised.  Limits: 1 <= nv <= 16 per call (PHIP_ERR_ARG_OUTOFRANGE beyond; split longer lists), n >= 0.  Same kernel, grid and
 * summation order as phip_vec_reduce_local: the same bits.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race; calls on one stream, or ordered by events, do not.  The array is left clean by every launch (the
 * arrival counter returns to 0), so a captured launch may be replayed.  h_dy is a host array of nv device pointers, read before return.
 * It takes no context, so it cannot launch a context's DEFERRED elementwise operations (see "vector kernels" above: phip_vec_axpy and
 * its siblings wait in their context; KSPCG's VecAXPY(X,a,P) stays pending past the reducing product by default): when an operand was
 * last written through a context, call phip_stream(ctx) or phip_sync(ctx) for that context immediately before, or the sums are taken
 * over the data as it was before the pending operations. */
int phip_vec_mdot_static(void *stream, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_z);

/* ---- by-product reductions --------------------------------------------------------------------------
 * On this hardware a reduction fused into a kernel that streams the operands anyway costs a few microseconds, while a reduction of
 * its own costs a pass over HBM plus a launch and a host round trip.  The calls below produce, next to their normal result, the
 * sums the Krylov callers ask for right afterwards (cg.c:206-207 MatMult then VecTDot(P,W); cg.c:225-229,269 PCApply then VecNorm(Z)
 * and VecTDot(Z,R)), and leave them in a slot of a small ring named by a ticket.  The glue keeps the ticket with the output vector,
 * keyed on the PetscObjectState of the operands (the mechanism of VecNorm's own cache, rvector.c:237-247), and serves a matching
 * VecDot / VecNorm from it: phip_ticket_wait (one rank; publish != 0 made the kernel write the pinned slot the host spins on), or
 * phip_ticket_values + phip_comm_allreduce + phip_read_scalars (several ranks).  A ticket expires after 63 newer ones.
 * Values are local sums, bitwise deterministic for a given length and grid. */
typedef struct { int32_t slot, nvals; uint64_t seq; } phip_ticket;
/* w = x .* y (VecPointwiseMult_Seq bvec2.c:84-110, w aliasing neither x nor y); values {sum w^2, sum w x, sum w y} */
int phip_vec_pointwise_mult_reduce(phip_ctx ctx, int32_t n, double *d_w, const double *d_x, const double *d_y, int publish, phip_ticket *t);
/* x += sum_k alpha_k y_k (phip_vec_maxpy, nv >= 1); value {sum x_new^2}: the VecNorm of the VecNormalize that follows the
 * orthogonalisation in KSPGMRESCycle (gmres.c:170-176 after borthog2.c:60-67) without another pass over x */
int phip_vec_maxpy_reduce(phip_ctx ctx, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy, int publish, phip_ticket *t);
/* y = A x (phip_mat_mult); value {sum_i v_i y_i}, d_v has m entries (the caller passes x for a square matrix) */
int phip_mat_mult_dot(phip_mat A, const double *d_x, double *d_y, const double *d_v, int publish, phip_ticket *t);
/* y += A x in place (phip_mat_multadd with z == y, mpiaij.c:996); adds sum_i v_i (y_new_i - y_old_i) to the value of the live ticket t;
 * publish != 0 also delivers the accumulated value to the ticket's host slot (phip_ticket_wait), for a ticket opened without publish */
int phip_mat_multadd_dot(phip_mat A, const double *d_x, double *d_y, const double *d_v, int publish, const phip_ticket *t);
/* Split reductions (VecDotBegin/VecNormBegin ... VecDotEnd, src/vec/vec/utils/comb.c:  begin the reduction, queue other work, collect the
 * result later): the local kernel is launched without waiting and delivers its raw sums (kind and layout as phip_vec_reduce_local, at most
 * 7 values) to the ticket; phip_ticket_wait collects them, phip_comm_allreduce_host combines the ranks. */
int phip_vec_reduce_begin(phip_ctx ctx, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, phip_ticket *t);
int phip_ticket_values(phip_ctx ctx, const phip_ticket *t, double **d_vals);
int phip_ticket_wait(phip_ctx ctx, const phip_ticket *t, double *h_vals);

/* ---- communicator: one rank per GPU, RCCL over xGMI ---------------------------------------------
 * Replaces MPI_Allreduce in the MPI Vec types and the MPI persistent send/recv of VecScatterBegin/End
 * (src/vec/vec/utils/vpscat.h:14-196).  backend "rccl": id128 is the 128-byte ncclUniqueId made by rank 0 with
 * phip_comm_unique_id and handed to every rank by the embedding application (bench.py uses torch.distributed).
 * backend "self": size-1 communicator, no library needed.  backend "threads": ranks are host threads of ONE process
 * sharing one GPU (test harness for the distributed code path on a single-GPU box; id128 then holds a group key). */
typedef int (*phip_allgather_fn)(void *user, const void *h_send, void *h_recv, size_t bytes_per_rank);
typedef int (*phip_alltoallv_i32_fn)(void *user, const int32_t *h_send, const int32_t *send_counts, int32_t *h_recv, const int32_t *recv_counts);
/* A communicator whose SET-UP collectives are supplied by the embedding application (host MPI inside PETSc, torch.distributed/gloo in
 * the CPU tests).  It carries no device traffic: phip_comm_allreduce / phip_halo_begin fail with PHIP_ERR_SUP on it. */
int phip_comm_create_callbacks(phip_ctx ctx, int nranks, int rank, phip_allgather_fn allgather, phip_alltoallv_i32_fn alltoallv, void *user, phip_comm *comm);
int phip_comm_unique_id(void *id128);
int phip_comm_create(phip_ctx ctx, const char *backend, const void *id128, int nranks, int rank, phip_comm *comm);
int phip_comm_destroy(phip_comm comm);
int phip_comm_rank_size(phip_comm comm, int *rank, int *size);
/* in-place all-reduce of n doubles in device memory, op 0 = SUM, 1 = MAX; stream-ordered on the context's stream */
int phip_comm_allreduce(phip_comm comm, double *d_buf, int32_t n, int op);
/* The MPI_Allreduce of VecDot_MPI / VecNorm_MPI / VecMDot_MPI (pbvec.c:9-19, pvec2.c:10-85) on HOST values: op 0 SUM, 1 MAX (NaN
 * propagates), n <= PHIP_MAX_MVEC, combined in rank order on every rank (identical bits everywhere).  Ranks of one node ("rccl",
 * "shm", "threads" backends) meet in a shared-memory ring -- no kernel, no RCCL call, about a microsecond; ranks on several nodes
 * fall back to ncclAllReduce through a device staging buffer; the callbacks backend uses the application's all-gather.
 * backend "shm" (phip_comm_create): processes of one node joined by that ring alone (id128 = any 128-byte key the ranks share).
 * Its host collectives run through the ring; its ghost exchanges are staged through the host (functional, slow): several processes may
 * share ONE GPU with it, which RCCL refuses, so the multi-process path can be rehearsed on a one-GPU box.  No phip_comm_allreduce. */
int phip_comm_allreduce_host(phip_comm comm, double *h_vals, int32_t n, int op);
/* host-side collectives used at set-up (the PetscGatherNumberOfMessages/MessageLengths + Isend/Irecv handshake of
 * VecScatterCreate_PtoS, vpscat.c:2346-2392): all-gather of fixed-size records and variable all-to-all of int32 lists */
int phip_comm_allgather_host(phip_comm comm, const void *h_send, void *h_recv, size_t bytes_per_rank);
int phip_comm_alltoallv_i32(phip_comm comm, const int32_t *h_send, const int32_t *send_counts, int32_t *h_recv, const int32_t *recv_counts);
int phip_comm_barrier(phip_comm comm);

/* ---- ghost exchange (the VecScatter of MatMult_MPIAIJ) -------------------------------------------
 * Plan = VecScatter_MPI_General "to" and "from" (include/petsc/private/vecimpl.h:245-273) as built by
 * VecScatterCreate_PtoS (vpscat.c:2297-2495): send side to_procs ascending / to_starts / to_indices (local offsets),
 * receive side from_procs ascending / from_starts, landing contiguously in lvec (from->contiq, vpscat.c:2524-2532).
 * begin: pack kernel (Pack_1, vpscat.c:494-498) + grouped send/recv on the communication stream (VecScatterBegin_1 vpscat.h:14-126);
 * end:   the compute stream waits for the exchange (VecScatterEnd_1 vpscat.h:132-196).  The diagonal-block SpMV issued between
 * begin and end overlaps the exchange exactly as mpiaij.c:990-996 intends. */
int phip_halo_create(phip_comm comm, int32_t to_n, const int32_t *to_procs, const int32_t *to_starts, const int32_t *to_indices,
                     int32_t from_n, const int32_t *from_procs, const int32_t *from_starts, phip_halo *halo);
int phip_halo_destroy(phip_halo halo);
/* The forward exchange as PEER WRITES: the ranks map each other's receive buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle), the pack
 * kernel of phip_halo_begin stores x[to_indices] straight into them and raises a flag word there; phip_halo_end launches one kernel that
 * waits for the flags, copies to lvec and acknowledges.  No send/receive kernels, no second stream.  PHIP_HALO_IPC (alike on every rank,
 * read by phip_halo_create): unset -- tried with the "rccl" backend when all ranks share the node, and kept only if three trial exchanges
 * deliver on every rank what was sent (otherwise the grouped ncclSend/ncclRecv path, silently; PHIP_VERBOSE=1 says why); 1 -- required
 * ("rccl" and "shm" backends), an error if unavailable; 0 -- never.  It has run between processes sharing one GPU (tests), not yet
 * between GPUs.  phip_halo_ipc_count: exchanges this process made that way (trial ones included). */
int phip_halo_ipc_count(unsigned long long *n);
int phip_halo_begin(phip_halo halo, const double *d_x_local, double *d_lvec);
int phip_halo_end(phip_halo halo);
/* The same plan run backwards with ADD_VALUES (VecScatterBegin/End(Mvctx, lvec, y, ADD_VALUES, SCATTER_REVERSE) of
 * MatMultTranspose_MPIAIJ, mpiaij.c:1041-1046, and MatMultTransposeAdd_MPIAIJ :1111-1115): begin sends every segment of lvec back to
 * the rank that owns those entries; end adds what arrived into y[to.indices] (UnPack_1 with ADD_VALUES, vpscat.c:512-515), the
 * contributions to one entry in ascending neighbour rank (the reference: arrival order).  Nothing is added to y before end, as
 * mpiaij.c:1044-1045 requires, so the diagonal block's transposed product may run in between. */
int phip_halo_reverse_begin(phip_halo halo, const double *d_lvec);
int phip_halo_reverse_end(phip_halo halo, double *d_y);

#ifdef __cplusplus
}
#endif
#endif /* PETSCHIP_H */
