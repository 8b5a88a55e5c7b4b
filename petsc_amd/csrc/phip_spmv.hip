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
//
// Second layout, chosen at upload when the matrix is stencil-like ("SELL-dict"): the set of distinct diagonals col-row has at most
// 256 members, no row has more than 255 entries and slice padding costs < 15 %.  Rows are cut into slices of 512; inside a slice the
// entries are stored slot-major (entry s of every row, then entry s+1, ...), values as f64 and the column as a ONE-BYTE index into the
// diagonal table, so a nonzero costs 9 bytes of HBM traffic instead of 12:
//   sval[P] f64, scode[P] uint8 (P = padded entries), slen[m] uint8 row lengths, sptr[S+1] int64 slice offsets, soffs[<=256] int32
// Kernel: one 256-thread workgroup per slice, each lane owns two adjacent rows (16-byte value loads, 2-byte code loads, all fully
// coalesced), looks the diagonals up in LDS, gathers x[row + diag] and accumulates ITS rows k-ascending from 0.0 (or y[i]) with
// separately rounded multiply and add: again the arithmetic of PetscSparseDensePlusDot bit for bit.  The CSR arrays stay on the
// device for MatGetDiagonal and as the source of truth.
//
// On top of it, when the stored entries form at most 256 distinct (diagonal, value) pairs -- matrices assembled from constant-
// coefficient stencils -- a "pair dictionary": one byte per entry names both, the values come from a table in LDS and a nonzero costs
// 1 byte of HBM traffic (setup_pairs, spmv_pair_kernel).  Lossless, same summation order, same bits.
// For irregular matrices whose x outgrows L2 and whose columns are used very unevenly: the hot-column order (setup_hot).
#include "phip_internal.h"
#include "phip_reduce.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

#define SPMV_ALIGN 32          /* a block's stream starts at a multiple of 32 entries: whole 128-byte lines of col (and of val) per wave-instruction */
#define SPMV_CAP 2016          /* nonzeros per row block; + (SPMV_ALIGN - 1) entries of alignment slack in front <= 2048 LDS doubles = 16 KiB */
#define SPMV_PAIRS 1024        /* (SPMV_CAP + SPMV_ALIGN) / 2 */
#ifndef SPMV_ROWS_MAX
#define SPMV_ROWS_MAX 256      /* rows per block */
#endif
#ifndef SPMV_WAVES_ATTR
#define SPMV_WAVES_ATTR
#endif
#define SELL_H 512             /* rows per SELL slice = 2 rows per lane */
#define SELL_MAX_OFFS 256
#define SELL_MAX_PAD 1.15

static int g_default_fmt = 0; /* phip_set_default_mat_format */
/* The host passes of phip_mat_create (SELL-dict arrays, pair codes, masks) are independent per slice of rows: they run on
   PHIP_SETUP_THREADS threads (default: the machine's, at most 16) over contiguous chunks in ascending order, so that tables numbered by
   first appearance come out as in one pass -- the layout does not depend on the number of threads. */
static int g_setup_threads = 0; /* phip_set_setup_threads; 0: PHIP_SETUP_THREADS or the machine's, at most 16 */
static int setup_threads(int64_t work)
{
  static const int T = [] {
    int t = getenv("PHIP_SETUP_THREADS") ? atoi(getenv("PHIP_SETUP_THREADS")) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return t < 1 ? 1 : (t > 64 ? 64 : t);
  }();
  return work < (1 << 20) ? 1 : (g_setup_threads > 0 ? g_setup_threads : T);
}
template <class F>
static void par_chunks(int64_t n, int T, F fn) /* fn(chunk, begin, end) */
{
  if (T <= 1 || n <= 0) { fn(0, (int64_t)0, n); return; }
  const int64_t            per = (n + T - 1) / T;
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) {
    const int64_t b = std::min<int64_t>(n, per * t), e = std::min<int64_t>(n, per * (t + 1));
    th.emplace_back([&fn, t, b, e] { fn(t, b, e); });
  }
  fn(0, (int64_t)0, std::min<int64_t>(n, per));
  for (auto &x : th) x.join();
}
/* a vector of doubles that is not zero-filled when it is sized (the 3.6 GB of padded values of a 27-point 256^3 matrix would be
   touched once by one thread before the threads fill them) */
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  template <class U, class... Args> void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
  template <class U> void construct(U *p) { ::new ((void *)p) U; }
};
typedef std::vector<double, NoInitAlloc<double>> raw_dvec;

/* PHIP_SETUP_TIMING=1: phip_mat_create reports where its time goes (stderr) */
struct SetupClock {
  bool on; std::chrono::steady_clock::time_point t;
  SetupClock() : on(getenv("PHIP_SETUP_TIMING") && atoi(getenv("PHIP_SETUP_TIMING"))), t(std::chrono::steady_clock::now()) {}
  void lap(const char *what) { if (!on) return; const auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[phip setup] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = n; }
};
static int g_hot_mode = -1;   /* phip_set_default_hot_columns; -1: take PHIP_HOT_COLUMNS or 1 at the first matrix */

struct phip_mat_s {
  phip_ctx ctx;
  int32_t  m, n;
  int64_t  nnz;
  int      use_cprow;
  int32_t  R; /* stored rows */
  int32_t *d_rowptr, *d_ridx, *d_col;
  double  *d_val;
  int4    *d_blkrec; /* [nblocks + 1] {first row, rows | log2(threads per row) << 24, first nonzero, end}: all a workgroup needs of its block in one scalar load */
  int32_t  nblocks;
  int32_t  tpr_user;
  std::vector<int32_t> *h_rowptr, *h_blkrow;
  std::vector<uint8_t> *h_blktpr;
  /* SELL-dict layout (stencil-like matrices) */
  int      fmt_user;   /* 0 auto, 1 CSR row blocks, 2 SELL-dict (error if not eligible) */
  int      sell_on, sell_built;
  int32_t  sell_nslices, sell_noffs, sell_maxlen;
  int64_t  sell_padded;
  double  *d_sval;
  uint8_t *d_scode, *d_slen;
  int64_t *d_sptr;
  int32_t *d_soffs;
  std::vector<int64_t> *h_sptr;
  /* pair dictionary of the SELL-dict layout (setup_pairs): one byte per stored entry names (diagonal, value) */
  int      vd_built, vd_on;
  int32_t  vd_npairs;
  uint32_t *d_pgrp;   /* pair codes: [slice][group of 4 slots][row 0..511], one dword = the codes of 4 consecutive slots of a row */
  int64_t  *d_pptr;   /* [S+1] first dword of each slice in d_pgrp */
  int64_t   pgrp_dwords;
  int32_t *d_spoffs;  /* [256] diagonal of each pair */
  double  *d_svtab;   /* [256] value of each pair */
  /* diagonal-mask layout (setup_dmask): <= 32 pairs sorted by diagonal, one presence bit per pair and row */
  int       dm_built, dm_on;
  int32_t   dm_npairs;
  uint32_t *d_rmask;  /* [S * 512] bit q of rmask[r]: row r holds pair q */
  unsigned long long *d_tmask; /* the same bits transposed (spmv_dmaskg_kernel): [set][pair q, 33 entries][i] bit l: row 256 wave + 4 l + i holds pair q, for the waves of that set */
  int32_t  *d_widx;   /* [wave of 256 rows] the wave's set in d_tmask (equal sets are stored once) */
  int64_t   dm_nsets;
  int       dm_dmin, dm_dmax; /* smallest / largest diagonal of the pair table */
  void     *d_gdesc;  /* [dm_ndesc + 3] DmGroup: the groups of spmv_dmaskg_kernel (at most one pair per place of the window) */
  int       dm_ndesc;
  double   *d_dotpart; /* [4 S] the fused dot's partial sum of each wave (spmv_dmask_kernel<DOT>, summed by dot_wave_partials_kernel) */
  int32_t  *d_dmoffs; /* [32] */
  double   *d_dmvals; /* [32] */
  int32_t  *d_dmgrp;  /* [dm_ngroups][4]: groups of pairs whose diagonals lie within DM4_SPAN of each other: {base diagonal, set of its pairs, span, 0} */
  int32_t   dm_ngroups;
  std::vector<uint8_t> *h_scode; /* host copies of scode and soffs: phip_mat_update_values rebuilds the pairs from them */
  std::vector<int32_t> *h_soffs;
  /* operator of y = A^T x: a second matrix holding A^T row by row, built on request (phip_mat_transpose_setup) */
  phip_mat              T;
  std::vector<int32_t> *t_perm; /* entry k of A^T is entry t_perm[k] of A */
  /* hot-column order (row-block layout, x larger than the caches, skewed column use): see setup_hot */
  int      hot_built, hot_on;
  int32_t *d_hcol;   /* [nnz + pad] column indices in the hot order */
  int32_t *d_horder; /* [n] x'[k] = x[horder[k]] */
  double  *d_hx;     /* [n] x in the hot order, refilled by every product */
  double   hot_share[2]; /* share of the gathers that fall into the hottest HOT_LINES lines of x, before / after */
  int      nt_user;  /* phip_mat_set_nontemporal: -1 by size (default), 0 cached loads, 1 non-temporal loads of the matrix stream */
};

/* Matrix entries are read once per launch.  When the matrix is much larger than the 256 MiB Infinity Cache, non-temporal loads keep
   the stream from displacing x in L2 / Infinity Cache (27-point 160^3: 221 -> 205 us); a matrix that fits the cache is better left
   cacheable, because the next launch finds it there (7-point 128^3: 26.8 us cacheable, 32.0 us non-temporal). */
#define PHIP_NT_MATRIX_BYTES (192ll << 20)
template <bool NT>
__device__ __forceinline__ double2 ld_stream(const double2 *p)
{
  if (NT) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
    return make_double2(v.x, v.y);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ int2 ld_stream(const int2 *p)
{
  if (NT) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i v = __builtin_nontemporal_load(reinterpret_cast<const v2i *>(p));
    return make_int2(v.x, v.y);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ uchar2 ld_stream(const uchar2 *p)
{
  if (NT) {
    const unsigned short v = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p));
    return make_uchar2((unsigned char)(v & 0xff), (unsigned char)(v >> 8));
  }
  return *p;
}

/* two adjacent results; nt: past the caches (results of PHIP_NT_RESULT_ROWS rows and more, see launch_spmv_t) */
__device__ __forceinline__ void st_result2(double *p, double a, double b, bool nt)
{
  if (nt) { typedef double v2d __attribute__((ext_vector_type(2))); v2d q; q.x = a; q.y = b; __builtin_nontemporal_store(q, reinterpret_cast<v2d *>(p)); }
  else *reinterpret_cast<double2 *>(p) = make_double2(a, b);
}

/* Workgroups are dealt to the eight XCDs round robin (workgroup i runs on XCD i mod 8) and every XCD has an L2 of its own, so with
   work item i = workgroup i the neighbours i - 1, i, i + 1 -- which share lines of x in any banded matrix (7-point 256^3: the row block
   one grid line up or down IS the next block) -- sit on three XCDs and each of those lines crosses the fabric three times (PMC, round 2:
   2.07 GB for 1.74 GB of algorithmic bytes).  xcd_chunk_map deals the work items out in runs of 2^lc consecutive items per XCD instead:
   item = ((chunk * 8 + xcd) << lc) + k for the k-th workgroup of that XCD's chunk.  All XCDs still sweep the same 8 * 2^lc items at a
   time (one contiguous eighth of the matrix per XCD was slower: DESIGN.md section 3).  Items behind the last full span keep i. */
__device__ __forceinline__ int xcd_chunk_map(int i, int n, int lc)
{
  if (lc <= 0) return i;
  const int span = 8 << lc;
  if (i >= (n & ~(span - 1))) return i;
  const int j = i >> 3;
  return (((((j >> lc) << 3) + (i & 7)) << lc) + (j & ((1 << lc) - 1)));
}
static int g_xcd_chunk_log2 = -1; /* phip_set_xcd_chunk; -1: PHIP_XCD_CHUNK (a power of two, 1 = off) or the kernel's own default */
/* measured (7-point 256^3, every value perturbed; chunk 1 / 16 / 64 / 128): CSR row blocks 311 / 317 / 303 / 302 us, SELL-dict slices
   219 / 216 / 221 / 222 us (best 4..16: 216 us) */
#define XCD_CHUNK_LOG2_CSR 7
#define XCD_CHUNK_LOG2_SELL 3
static int xcd_chunk_log2(int kernel_default)
{
  static const int env = [] {
    if (!getenv("PHIP_XCD_CHUNK")) return -1;
    int c = atoi(getenv("PHIP_XCD_CHUNK")), l = 0;
    while ((2 << l) <= c && l < 12) l++;
    return l;
  }();
  return g_xcd_chunk_log2 >= 0 ? g_xcd_chunk_log2 : (env >= 0 ? env : kernel_default);
}

extern "C" int phip_set_xcd_chunk(int chunk)
{
  if (chunk == 0) { g_xcd_chunk_log2 = -1; return 0; } /* back to the kernels' defaults */
  PHIP_REQUIRE(chunk >= 1 && chunk <= 4096 && (chunk & (chunk - 1)) == 0, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_xcd_chunk: a power of two in 1..4096, or 0 for the defaults");
  int l = 0;
  while ((1 << l) < chunk) l++;
  g_xcd_chunk_log2 = l;
  return 0;
}

/* sum += prod[j] for j = j0, j0 + STEP, ... below e, in that order; eight LDS reads are requested before the first is added (the plain
   loop waits out one LDS latency per entry, and with rows of very different lengths in a block that wait is what the block's time is) */
template <int STEP>
__device__ __forceinline__ double lds_ordered_sum(const double *prod, int j0, int e, double sum)
{
  int j = j0;
  for (; j + 7 * STEP < e; j += 8 * STEP) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = prod[j + u * STEP];
#pragma unroll
    for (int u = 0; u < 8; u++) sum += v[u];
  }
  for (; j < e; j += STEP) sum += prod[j];
  return sum;
}

template <int TPR, bool DOT>
__device__ __forceinline__ void spmv_reduce_rows(const double *prod, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx,
                                                 const double *__restrict__ yin, double *__restrict__ yout, int r0, int r1, int abase,
                                                 bool add, bool cprow, const double *__restrict__ dotv, double &dacc, int ps, int pe)
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
    if (p == 0) { s = ps - abase; e = pe - abase; } /* requested by the kernel together with the block's stream */
    else if (valid) { s = rowptr[r] - abase; e = rowptr[r + 1] - abase; }
    if (!valid) s = e = 0;
    double sum = 0.0;
    int    rout = 0;
    if (TPR == 1) {
      if (valid) {
        rout = cprow ? ridx[r] : r;
        const double y0 = add ? yin[rout] : 0.0;
        sum = lds_ordered_sum<1>(prod, s, e, y0);
        yout[rout] = sum;
        if (DOT) dacc += dotv[rout] * (add ? sum - y0 : sum);
      }
    } else {
      sum = lds_ordered_sum<TPR>(prod, s + sub, e, 0.0);
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (valid && sub == 0) {
        rout = cprow ? ridx[r] : r;
        if (DOT) dacc += dotv[rout] * sum;
        if (add) sum += yin[rout];
        yout[rout] = sum;
      }
    }
  }
}

/* DOT: also accumulate sum_i dotv[i] * (A x)_i (the increment, for MultAdd) and leave it where ctl says; the grid then strides over
   the row blocks with a capped number of workgroups (launch_spmv_t).  Without DOT the grid is one workgroup per row block. */
template <bool DOT, bool NT>
__global__ __launch_bounds__(PHIP_THREADS) SPMV_WAVES_ATTR void spmv_rowblock_kernel(int32_t nblocks, const int4 *__restrict__ blkrec,
                                                                     const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx,
                                                                     const int32_t *__restrict__ col, const double *__restrict__ val,
                                                                     const double *__restrict__ x, const double *__restrict__ yin,
                                                                     double *__restrict__ yout, int add, int cprow, const double *__restrict__ dotv, RedCtl ctl, int xlc)
{
  __shared__ __attribute__((aligned(16))) double prod[2 * SPMV_PAIRS];
  const int t = threadIdx.x;
  double    dacc = 0.0;
  for (int bi = blockIdx.x; bi < nblocks; bi += gridDim.x) {
    const int  b = xcd_chunk_map(bi, nblocks, xlc);
    const int4 rec = blkrec[b]; /* one scalar load instead of the chain blkrow -> rowptr -> (after the barrier) blktpr */
    const int  r0 = rec.x, r1 = rec.x + (rec.y & 0xffffff), tprl = (int)((unsigned)rec.y >> 24);
    const int  base = rec.z, end = rec.w;

    if (end - base > SPMV_CAP) {
      /* one long row (the plan guarantees r1 == r0 + 1): whole workgroup strides over it */
      /* eight entries a thread requested together (columns, values, then the eight gathers), added in the order of the plain loop */
      double        sum = 0.0;
      constexpr int LU = 8;
      /* the walk starts at a line boundary in front of the row, so that a wave-instruction covers whole 128-byte lines */
      for (int j0 = (base & ~(SPMV_ALIGN - 1)) + t; j0 < end; j0 += LU * PHIP_THREADS) {
        int    cj[LU];
        double vj[LU], xj[LU];
#pragma unroll
        for (int u = 0; u < LU; u++) {
          const int  j = j0 + u * PHIP_THREADS;
          const bool in = j >= base && j < end;
          cj[u] = in ? col[j] : col[base];
          vj[u] = in ? val[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < LU; u++) xj[u] = x[cj[u]];
#pragma unroll
        for (int u = 0; u < LU; u++) { const int j = j0 + u * PHIP_THREADS; if (j >= base && j < end) sum += vj[u] * xj[u]; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if ((t & 63) == 0) prod[t >> 6] = sum;
      __syncthreads();
      if (t == 0) {
        const int rout = cprow ? ridx[r0] : r0;
        double    s = (prod[0] + prod[1]) + (prod[2] + prod[3]);
        if (DOT) dacc += dotv[rout] * s;
        if (add) s += yin[rout];
        yout[rout] = s;
      }
    } else {
      /* phase 1: stream products into LDS.  abase is a multiple of 32 entries: every wave-instruction of the stream (64 lanes x 16 bytes
         of val, x 8 bytes of col) then covers whole 128-byte lines.  With abase = base & ~1 nearly every block started inside a line
         and each of its 28 wave-instructions shared its first and last line with a neighbour; read with non-temporal loads those lines
         crossed the fabric twice (PMC, 7-point 256^3 with perturbed values: 15.1 M 128-byte read requests for 12.5 M lines of
         algorithmic bytes).  The up to 31 entries in front of the block belong to the block before: loaded (same line), multiplied,
         never summed. */
      const int abase = base & ~(SPMV_ALIGN - 1);
      /* the bounds of the row this thread sums first (row r0 + t / threads per row; clamped: a thread without a row ignores them) */
      const int rfirst = min(r0 + (t >> tprl), r1 - 1);
      const int ps = rowptr[rfirst], pe = rowptr[rfirst + 1];
      if (end > base) { /* (a block of empty rows has no products, and x may be null for a matrix without nonzeros) */
      double2   a[4], g[4];
      int2      c[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { /* a pair behind the block's end repeats the block's first pair: no test, its product is never read */
        const int e = abase + 2 * (t + k * PHIP_THREADS), ec = e < end ? e : abase;
        a[k] = ld_stream<NT>(reinterpret_cast<const double2 *>(val + ec));
        c[k] = ld_stream<NT>(reinterpret_cast<const int2 *>(col + ec));
      }
      /* the eight stream loads and then the eight gathers of a thread are requested together: inside a test per pair, each pair
         waited for its own loads (four memory round trips a block for the stream and four for the gathers instead of one each) */
#pragma unroll
      for (int k = 0; k < 4; k++) { g[k].x = x[c[k].x]; g[k].y = x[c[k].y]; }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double2 p;
        p.x = a[k].x * g[k].x;
        p.y = a[k].y * g[k].y;
        *reinterpret_cast<double2 *>(prod + 2 * (t + k * PHIP_THREADS)) = p;
      }
      }
      __syncthreads();

      /* phase 2: per-row sums out of LDS */
      switch (tprl) {
      case 0: spmv_reduce_rows<1, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      case 1: spmv_reduce_rows<2, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      case 2: spmv_reduce_rows<4, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      case 3: spmv_reduce_rows<8, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      case 4: spmv_reduce_rows<16, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      case 5: spmv_reduce_rows<32, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      default: spmv_reduce_rows<64, DOT>(prod, rowptr, ridx, yin, yout, r0, r1, abase, add, cprow, dotv, dacc, ps, pe); break;
      }
    }
    if (DOT) __syncthreads(); /* the next row block of this workgroup reuses prod[] */
  }
  if (DOT) {
    double a1[1] = {dacc};
    block_sum<1>(a1, prod);
    __syncthreads();
    reduce_tail<1, 0>(a1, prod, ctl);
  }
}

/* SELL-dict SpMV: see the header comment.  Lane t of slice b owns rows 512 b + 2 t and + 1.  DOT as in spmv_rowblock_kernel.
   (Rounds 1-2 carried a variant that staged the stretches of x a slice needs in LDS; measured slower every time -- 7-point 128^3
   26.9 -> 31.8 us, 27-point 160^3 206 -> 220 us: the gathers are not what bounds this kernel -- and retired in round 3.) */
template <bool NT, bool DOT, int U>
__global__ __launch_bounds__(PHIP_THREADS) void spmv_sell_kernel(int32_t m, int32_t n, int32_t nslices, const int64_t *__restrict__ sptr, const uint8_t *__restrict__ slen,
                                                                 const double *__restrict__ sval, const uint8_t *__restrict__ scode,
                                                                 const int32_t *__restrict__ soffs, int noffs,
                                                                 const double *__restrict__ x, const double *__restrict__ yin, double *__restrict__ yout, int add,
                                                                 const double *__restrict__ dotv, RedCtl ctl, int nty, int xlc)
{
#pragma clang fp contract(off) /* multiply and add round separately, as the reference's compiled loop does (no FMA) */
  __shared__ int32_t offs[SELL_MAX_OFFS]; /* diagonal per code */
  __shared__ double  lds[4];
  const int t = threadIdx.x;
  offs[t] = (t < noffs) ? soffs[t] : 0;
  __syncthreads();
  double dacc = 0.0;
  for (int bi = blockIdx.x; bi < nslices; bi += gridDim.x) {
    const int b = xcd_chunk_map(bi, nslices, xlc);
    const int  r0 = b * SELL_H + 2 * t;
    const bool act = r0 < m;
    const bool has1 = r0 + 1 < m;
    const int  l0 = act ? slen[r0] : 0, l1 = has1 ? slen[r0 + 1] : 0;
    const int  lm = l0 > l1 ? l0 : l1;
    const int64_t  base = sptr[b];
    const double2 *vp = reinterpret_cast<const double2 *>(sval + base) + t;
    const uchar2  *cp = reinterpret_cast<const uchar2 *>(scode + base) + t;
    const int      g0 = r0;
    const double  *xv = x;
    double y0 = 0.0, y1 = 0.0;
    if (add && act) { y0 = yin[r0]; if (has1) y1 = yin[r0 + 1]; }
    double dv0 = 0.0, dv1 = 0.0; /* the fused dot's weights: requested here so that their latency hides behind the row sums */
    if (DOT && act) { dv0 = dotv[r0]; if (has1) dv1 = dotv[r0 + 1]; }
    double s0 = y0, s1 = y1;
    int    s = 0;
    for (; s + U <= lm; s += U) { /* full batches: U slots requested, then their gathers, then the sums in slot order */
      double2 v[U];
      uchar2  c[U];
#pragma unroll
      for (int u = 0; u < U; u++) { v[u] = ld_stream<NT>(vp + (size_t)(s + u) * (SELL_H / 2)); c[u] = ld_stream<NT>(cp + (size_t)(s + u) * (SELL_H / 2)); }
      double xa[U], xb[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool a0 = s + u < l0, a1 = s + u < l1;
        xa[u] = xv[a0 ? g0 + offs[c[u].x] : 0];
        xb[u] = xv[a1 ? g0 + 1 + offs[c[u].y] : 0];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (s + u < l0) s0 = s0 + v[u].x * xa[u];
        if (s + u < l1) s1 = s1 + v[u].y * xb[u];
      }
    }
    for (; s < lm; s += 4) { /* remainder in partly masked batches of 4, not slot by slot (7-point: fused-dot SpMV 31.2 -> 28.7 us) */
      double2 v[4];
      uchar2  c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (s + u < lm) { v[u] = ld_stream<NT>(vp + (size_t)(s + u) * (SELL_H / 2)); c[u] = ld_stream<NT>(cp + (size_t)(s + u) * (SELL_H / 2)); }
        else { v[u] = make_double2(0.0, 0.0); c[u] = make_uchar2(0, 0); }
      }
      double xa[4], xb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool a0 = s + u < l0, a1 = s + u < l1;
        xa[u] = xv[a0 ? g0 + offs[c[u].x] : 0];
        xb[u] = xv[a1 ? g0 + 1 + offs[c[u].y] : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (s + u < l0) s0 = s0 + v[u].x * xa[u];
        if (s + u < l1) s1 = s1 + v[u].y * xb[u];
      }
    }
    if (has1) st_result2(yout + r0, s0, s1, nty);
    else if (act) yout[r0] = s0;
    if (DOT && act) {
      dacc += dv0 * (s0 - y0);
      if (has1) dacc += dv1 * (s1 - y1);
    }
  }
  if (DOT) {
    double a1[1] = {dacc};
    block_sum<1>(a1, lds);
    __syncthreads();
    reduce_tail<1, 0>(a1, lds, ctl);
  }
}

/* SpMV over the pair dictionary (setup_pairs).  With the values in a table nothing forces a lane to own two ADJACENT rows (that was
   for 16-byte value loads): lane t owns rows t and t + 256 of the slice, so the gathers of a wave read consecutive entries of x (one
   512-byte run per instruction when the rows share the diagonal, as the interior of a stencil does) instead of every other one, and
   a dword brings the codes of 4 slots of a row.  G groups of 4 slots are requested before their gathers start.  Per-row sums run
   k-ascending with separately rounded multiply and add, as in spmv_sell_kernel: the reference's bits. */
template <bool NT, bool DOT, int G>
__global__ __launch_bounds__(PHIP_THREADS) void spmv_pair_kernel(int32_t m, int32_t nslices, const int64_t *__restrict__ pptr, const uint8_t *__restrict__ slen,
                                                                 const uint32_t *__restrict__ pgrp, const int32_t *__restrict__ poffs, const double *__restrict__ pvals,
                                                                 int npairs, const double *__restrict__ x, const double *__restrict__ yin, double *__restrict__ yout,
                                                                 int add, const double *__restrict__ dotv, RedCtl ctl, int nty)
{
#pragma clang fp contract(off)
  __shared__ int4   tab[SELL_MAX_OFFS]; /* x: diagonal, z/w: the value's low and high words */
  __shared__ double lds[4];
  const int t = threadIdx.x;
  {
    const double v = (t < npairs) ? pvals[t] : 0.0;
    tab[t] = make_int4((t < npairs) ? poffs[t] : 0, 0, __double2loint(v), __double2hiint(v));
  }
  __syncthreads();
  double dacc = 0.0;
  for (int b = blockIdx.x; b < nslices; b += gridDim.x) {
    const int  rA = b * SELL_H + t, rB = rA + SELL_H / 2;
    const bool actA = rA < m, actB = rB < m;
    const int  lA = actA ? slen[rA] : 0, lB = actB ? slen[rB] : 0;
    const int  lm = lA > lB ? lA : lB;
    const uint32_t *cp = pgrp + pptr[b] + t;
    double yA = 0.0, yB = 0.0;
    if (add) { if (actA) yA = yin[rA]; if (actB) yB = yin[rB]; }
    double dvA = 0.0, dvB = 0.0;
    if (DOT) { if (actA) dvA = dotv[rA]; if (actB) dvB = dotv[rB]; }
    double sA = yA, sB = yB;
    for (int g = 0; 4 * g < lm; g += G) {
      uint32_t cA[G], cB[G];
#pragma unroll
      for (int u = 0; u < G; u++) {
        const bool in = 4 * (g + u) < lm;
        const uint32_t *q = cp + (size_t)(g + u) * SELL_H;
        cA[u] = in ? (NT ? __builtin_nontemporal_load(q) : *q) : 0u;
        cB[u] = in ? (NT ? __builtin_nontemporal_load(q + SELL_H / 2) : q[SELL_H / 2]) : 0u;
      }
      double xa[4 * G], xb[4 * G], va[4 * G], vb[4 * G];
#pragma unroll
      for (int u = 0; u < G; u++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int  sl = 4 * (g + u) + k;
          const int4 eA = tab[(cA[u] >> (8 * k)) & 255u], eB = tab[(cB[u] >> (8 * k)) & 255u];
          xa[4 * u + k] = x[sl < lA ? rA + eA.x : 0];
          xb[4 * u + k] = x[sl < lB ? rB + eB.x : 0];
          va[4 * u + k] = __hiloint2double(eA.w, eA.z);
          vb[4 * u + k] = __hiloint2double(eB.w, eB.z);
        }
      }
#pragma unroll
      for (int j = 0; j < 4 * G; j++) {
        if (4 * g + j < lA) sA = sA + va[j] * xa[j];
        if (4 * g + j < lB) sB = sB + vb[j] * xb[j];
      }
    }
    if (nty) { if (actA) __builtin_nontemporal_store(sA, yout + rA); if (actB) __builtin_nontemporal_store(sB, yout + rB); }
    else { if (actA) yout[rA] = sA; if (actB) yout[rB] = sB; }
    if (DOT) {
      if (actA) dacc += dvA * (sA - yA);
      if (actB) dacc += dvB * (sB - yB);
    }
  }
  if (DOT) {
    double a1[1] = {dacc};
    block_sum<1>(a1, lds);
    __syncthreads();
    reduce_tail<1, 0>(a1, lds, ctl);
  }
}

/* SpMV in the diagonal-mask layout (setup_dmask).  The pair kernel above runs at the rate of the texture addresser: every gather is a
   separate 8-byte load per lane, and a CU accepts 4 lane addresses per clock whatever their width (measured: TA busy for the whole
   kernel, 1.9 T lane loads/s of a 2.2 T/s ceiling, VALU 48 % busy; profiles/r01_pmc_pair_kernel_27pt_192.json).  When the pairs are
   few (<= 32) the matrix can be held in DIAGONAL space instead of slot space: the pairs are sorted by diagonal, a row is a 32-bit set of
   the pairs it holds, and the loop runs over the pairs, not over the row's slots.  Then the diagonal is the same for every row of a
   wave, so a lane that owns two ADJACENT rows reads x[r + d] and x[r + 1 + d] with ONE 16-byte load: half the lane addresses per
   nonzero, no code stream, no LDS table (diagonal and value of the current pair live in scalar registers).  Ascending diagonal is
   ascending column, so a row still adds its products k-ascending with separately rounded multiply and add: the reference's bits.
   Pairs no row of the wave holds are skipped; a wave whose 16-byte loads could leave x (first and last rows of the matrix, far
   diagonals) takes the entries one by one.  (Round 1's kernel of this layout, two rows a lane, is retired: the two kernels below --
   four rows a lane -- took its place in round 2.) */
typedef double phip_v2d_u __attribute__((ext_vector_type(2), aligned(8)));

/* The diagonal masks with FOUR adjacent rows per lane and the pairs taken in GROUPS of neighbouring diagonals.  spmv_dmask_kernel above
   still pushes every x entry through the vector L1 once per diagonal that touches it (27 times for a 27-point stencil): 216 B per row,
   and the 64 B/clock/CU of that path, not HBM, is what its 205 us at 27-point 256^3 are made of (HBM traffic: 335 MB).  Neighbouring
   rows use neighbouring entries, and neighbouring diagonals use them again: the rows r .. r+3 of a lane need, for the diagonals
   D, D+1, D+2 of one group, just x[r+D .. r+D+5] -- three 16-byte loads instead of six (two rows a lane, one diagonal a load).  The
   window stays in registers while the group's pairs are worked off in ascending diagonal order, so a row still adds its products
   k-ascending with separately rounded multiply and add: the same bits.  A group whose pairs share one diagonal loads four entries (two
   loads).  Waves whose windows could leave x (first and last rows, far diagonals) load entry by entry with clamped indices; an entry
   outside the row's own pairs is never used.  UG groups are requested before the sums of the first begin.  One workgroup = 1024 rows. */
#define DMASK_MAX_PAIRS 32
#define DM4_ROWS 4
#define DM4_SPAN 3 /* diagonals D .. D + DM4_SPAN - 1 share a window of DM4_ROWS + DM4_SPAN - 1 entries */
#define DM4_WIN (DM4_ROWS + DM4_SPAN - 1)

template <int OFF>
__device__ __forceinline__ void dm4_step(double (&s)[DM4_ROWS], const double (&w)[DM4_WIN], double v, const uint32_t (&mk)[DM4_ROWS], int q)
{
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < DM4_ROWS; i++)
    if ((mk[i] >> q) & 1u) s[i] = s[i] + v * w[i + OFF];
}

/* the same for a pair that EVERY row of the wave holds: no tests, no selects (a third of the instructions) */
template <int OFF>
__device__ __forceinline__ void dm4_step_all(double (&s)[DM4_ROWS], const double (&w)[DM4_WIN], double v)
{
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < DM4_ROWS; i++) s[i] = s[i] + v * w[i + OFF];
}

template <bool DOT, bool ADD, int UG>
__global__ __launch_bounds__(PHIP_THREADS) void spmv_dmask4_kernel(int32_t m, int32_t n, const uint32_t *__restrict__ rmask, const int32_t *__restrict__ poffs,
                                                                   const double *__restrict__ pvals, const int4 *__restrict__ gtab, int ngroups,
                                                                   const double *__restrict__ x, const double *__restrict__ yin, double *__restrict__ yout,
                                                                   const double *__restrict__ dotv, double *__restrict__ wavepart, int nty)
{
#pragma clang fp contract(off)
  const int      t = threadIdx.x;
  const int      r0 = (int)blockIdx.x * (PHIP_THREADS * DM4_ROWS) + DM4_ROWS * t;
  const uint4    mq = reinterpret_cast<const uint4 *>(rmask)[(size_t)blockIdx.x * PHIP_THREADS + t]; /* the mask array is padded to whole workgroups */
  const uint32_t mk[DM4_ROWS] = {mq.x, mq.y, mq.z, mq.w};
  const int      nact = m - r0; /* rows r0 .. r0 + 3 exist while i < nact */
  double         y0[DM4_ROWS] = {0.0, 0.0, 0.0, 0.0}, s[DM4_ROWS], dv[DM4_ROWS] = {0.0, 0.0, 0.0, 0.0};
  if (ADD) {
    if (nact >= DM4_ROWS) {
      const phip_v2d_u a = *reinterpret_cast<const phip_v2d_u *>(yin + r0), b = *reinterpret_cast<const phip_v2d_u *>(yin + r0 + 2);
      y0[0] = a.x; y0[1] = a.y; y0[2] = b.x; y0[3] = b.y;
    } else {
#pragma unroll
      for (int i = 0; i < DM4_ROWS; i++) if (i < nact) y0[i] = yin[r0 + i];
    }
  }
#pragma unroll
  for (int i = 0; i < DM4_ROWS; i++) s[i] = y0[i];
  uint32_t um = mk[0] | mk[1] | mk[2] | mk[3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) um |= (uint32_t)__shfl_xor((int)um, o, 64);
  const uint32_t rem = (uint32_t)__builtin_amdgcn_readfirstlane((int)um); /* the pairs some row of this wave holds */
  uint32_t am = mk[0] & mk[1] & mk[2] & mk[3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am &= (uint32_t)__shfl_xor((int)am, o, 64);
  const uint32_t uni = (uint32_t)__builtin_amdgcn_readfirstlane((int)am); /* the pairs EVERY row of this wave holds */
  const int      w0 = (int)blockIdx.x * (PHIP_THREADS * DM4_ROWS) + (t & ~63) * DM4_ROWS; /* first of the 256 rows of this wave */
  /* the group table and the pair table live in the lanes of the wave (entry l in lane l) and are read with v_readlane: no memory
     latency inside the loops below */
  const int  lane = t & 63;
  const int4 gl = lane < ngroups ? gtab[lane] : make_int4(0, 0, 1, 0);
  const int    pl_off = lane < DMASK_MAX_PAIRS ? poffs[lane] : 0;
  const double pl_val = lane < DMASK_MAX_PAIRS ? pvals[lane] : 0.0;
  const int    pl_lo = __double2loint(pl_val), pl_hi = __double2hiint(pl_val);
  int          gi = 0;
  while (gi < ngroups) {
    int      gD[UG], gS[UG];
    uint32_t gM[UG];
#pragma unroll
    for (int u = 0; u < UG; u++) { /* the next UG groups that some row of the wave takes part in */
      gM[u] = 0u; gD[u] = 0; gS[u] = 1;
      while (gi < ngroups) {
        const uint32_t set = (uint32_t)__builtin_amdgcn_readlane(gl.y, gi);
        if (set & rem) { gD[u] = __builtin_amdgcn_readlane(gl.x, gi); gS[u] = __builtin_amdgcn_readlane(gl.z, gi); gM[u] = set & rem; gi++; break; }
        gi++;
      }
    }
    double w[UG][DM4_WIN];
#pragma unroll
    for (int u = 0; u < UG; u++) {
#pragma unroll
      for (int j = 0; j < DM4_WIN; j++) w[u][j] = 0.0;
      if (gM[u]) {
        const int D = gD[u];
        if ((int64_t)w0 + D >= 0 && (int64_t)w0 + 64 * DM4_ROWS - 1 + D + (DM4_SPAN - 1) < (int64_t)n) { /* the wave's windows stay inside x */
          const double     *p = x + ((int64_t)r0 + D);
          const phip_v2d_u  a = *reinterpret_cast<const phip_v2d_u *>(p), b = *reinterpret_cast<const phip_v2d_u *>(p + 2);
          w[u][0] = a.x; w[u][1] = a.y; w[u][2] = b.x; w[u][3] = b.y;
          if (gS[u] > 1) { const phip_v2d_u c = *reinterpret_cast<const phip_v2d_u *>(p + 4); w[u][4] = c.x; w[u][5] = c.y; }
        } else {
#pragma unroll
          for (int j = 0; j < DM4_WIN; j++) {
            int64_t k = (int64_t)r0 + D + j;
            k = k < 0 ? 0 : (k >= (int64_t)n ? (int64_t)n - 1 : k);
            w[u][j] = x[k];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UG; u++) {
      uint32_t gm = gM[u];
      while (gm) {
        const int q = __builtin_ctz(gm);
        gm &= gm - 1;
        const int    off = __builtin_amdgcn_readlane(pl_off, q) - gD[u];
        const double v = __hiloint2double(__builtin_amdgcn_readlane(pl_hi, q), __builtin_amdgcn_readlane(pl_lo, q));
        if ((uni >> q) & 1u) {
          if (off == 0) dm4_step_all<0>(s, w[u], v);
          else if (off == 1) dm4_step_all<1>(s, w[u], v);
          else dm4_step_all<2>(s, w[u], v);
        } else {
          if (off == 0) dm4_step<0>(s, w[u], v, mk, q);
          else if (off == 1) dm4_step<1>(s, w[u], v, mk, q);
          else dm4_step<2>(s, w[u], v, mk, q);
        }
      }
    }
  }
  if (nact >= DM4_ROWS) {
    st_result2(yout + r0, s[0], s[1], nty);
    st_result2(yout + r0 + 2, s[2], s[3], nty);
  } else {
#pragma unroll
    for (int i = 0; i < DM4_ROWS; i++) if (i < nact) yout[r0 + i] = s[i];
  }
  if (DOT) { /* one partial per wave, stored and forgotten (dot_wave_partials_kernel sums them).  The weights are requested only now:
                held across the loop they cost 24 VGPRs and a sixth of the occupancy; the entries were gathered a moment ago (CG: dotv = x) */
    if (nact >= DM4_ROWS) {
      const phip_v2d_u a = *reinterpret_cast<const phip_v2d_u *>(dotv + r0), b = *reinterpret_cast<const phip_v2d_u *>(dotv + r0 + 2);
      dv[0] = a.x; dv[1] = a.y; dv[2] = b.x; dv[3] = b.y;
    } else {
#pragma unroll
      for (int i = 0; i < DM4_ROWS; i++) if (i < nact) dv[i] = dotv[r0 + i];
    }
    double dacc = 0.0;
#pragma unroll
    for (int i = 0; i < DM4_ROWS; i++) if (i < nact) dacc += dv[i] * (ADD ? s[i] - y0[i] : s[i]); /* MultAdd: the increment A x, not z */
    dacc = wave_sum(dacc);
    if ((t & 63) == 0) wavepart[(size_t)blockIdx.x * (PHIP_THREADS / 64) + (t >> 6)] = dacc;
  }
}

struct alignas(32) DmWords { unsigned long long w[4]; };
/* a group: base diagonal D, span (1 .. DM4_SPAN diagonals), and for each place j of the window the pair on diagonal D + j: its index
   (DMASK_MAX_PAIRS: none) and its value */
struct alignas(64) DmGroup { int32_t D, span; int32_t q[DM4_SPAN]; int32_t pad1[3]; double v[DM4_SPAN]; int32_t pad2[2]; };
static_assert(sizeof(DmGroup) == 64, "one group descriptor = one s_load_dwordx16");
typedef uint32_t phip_u32x16 __attribute__((ext_vector_type(16)));
#define DMG_WAVE_ROWS 256               /* rows of a wave: lane l holds the rows 4 l .. 4 l + 3 of them */
#define DMG_WG_ROWS (DMG_WAVE_ROWS * (PHIP_THREADS / 64))
#define DMG_WORDS (DMASK_MAX_PAIRS + 1) /* mask entries of a wave: one per pair and an empty one */

/* s[i] += v * w[i + OFF] for the rows i whose lanes the words select.  The four products are formed in all lanes; each sum is then taken
   with the word as the execution mask, set by the scalar unit -- no test, no select, no branch: two vector instructions a row.
   EXEC is written behind the compiler's back, so the block keeps it whole: the mask it finds is saved in a scalar pair first and put
   back last (not "all ones": today every caller is wave-uniform code of a fully populated 256-thread workgroup, but an edit that puts
   a call under divergence must not switch lanes on that the compiler believes off), every write of EXEC sits inside this one asm
   statement (the compiler schedules nothing into it), and the mask words are ANDed with nothing -- a lane outside the saved mask may
   add into its own dead copy of s[], which no live lane reads.  tests/test_kernels_gpu.py::test_mask_kernels_partial_last_wave runs the
   kernels on row counts that leave the last wave and the last workgroup partly without rows. */
template <int OFF>
__device__ __forceinline__ void dmg_step(double (&s)[4], const double (&w)[DM4_WIN], double v, const DmWords &mw)
{
  double             t0, t1, t2, t3;
  unsigned long long saved;
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               "v_mul_f64 %[t0], %[v], %[w0]\n\t"
               "v_mul_f64 %[t1], %[v], %[w1]\n\t"
               "v_mul_f64 %[t2], %[v], %[w2]\n\t"
               "v_mul_f64 %[t3], %[v], %[w3]\n\t"
               "s_mov_b64 exec, %[m0]\n\t"
               "v_add_f64 %[s0], %[s0], %[t0]\n\t"
               "s_mov_b64 exec, %[m1]\n\t"
               "v_add_f64 %[s1], %[s1], %[t1]\n\t"
               "s_mov_b64 exec, %[m2]\n\t"
               "v_add_f64 %[s2], %[s2], %[t2]\n\t"
               "s_mov_b64 exec, %[m3]\n\t"
               "v_add_f64 %[s3], %[s3], %[t3]\n\t"
               "s_mov_b64 exec, %[sv]"
               : [s0] "+v"(s[0]), [s1] "+v"(s[1]), [s2] "+v"(s[2]), [s3] "+v"(s[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
                 [sv] "=&s"(saved)
               : [v] "s"(v), [w0] "v"(w[OFF]), [w1] "v"(w[OFF + 1]), [w2] "v"(w[OFF + 2]), [w3] "v"(w[OFF + 3]), [m0] "s"(mw.w[0]), [m1] "s"(mw.w[1]),
                 [m2] "s"(mw.w[2]), [m3] "s"(mw.w[3]));
}

/* spmv_dmask4_kernel without its instruction overhead.  Its counters (profiles/r02_pmc_spmv_dmask4.json, 27-point 256^3) show 858 vector
   and 670 scalar instructions per wave for 108 multiply-add pairs a lane: finding the pairs by walking bit sets, table reads with
   v_readlane, a three-way dispatch on the pair's place in the window, a test and a select per row and pair.  Here a group is one
   64-byte descriptor fetched with one scalar load, one group ahead; the places of the window are compile-time; the masks come
   TRANSPOSED, per wave and pair one 64-bit word for each of the lane's four rows (tmask), fetched by scalar loads and used as
   execution masks; a pair every row holds runs through the same code with full words: 297 vector, 476 scalar instructions and 80
   scalar loads per wave (profiles/r02_pmc_spmv_dmaskg.json), 27-point 256^3 116 -> 90 us.  What is left is the vector L1: one access
   a clock and CU for the whole launch (TCP_TOTAL_CACHE_ACCESSES 53.7 M = 210 k per CU in 222 k clocks), the waves waiting on memory
   half of their time.  Tried on top of this and no faster, all bit-exact: two windows in flight (130 us: more lines in flight than
   the L1 holds, its hits on the windows' overlaps are what the kernel lives on), contiguous lanes with pairs of rows (91 us), entries
   4 and 5 of the window from the next lane through the lane-shift network (v_mov_b32 wave_shl:1; 93 us with a third fewer bytes
   requested), the same with 16-byte aligned loads and 252 or 248 rows a wave (106 / 112 us: 65 % more L2 misses), eight rows a
   lane (198 us), the windows staged in LDS (spmv_dmasks_kernel, deleted in round 3: 108 us).  The same sums in the same order as spmv_dmask4_kernel: the same bits.
   dmin / dmax: smallest and largest diagonal of the table: a wave all of whose windows stay inside x (all but the first and last few)
   takes the loop without range tests. */
template <bool DOT, bool ADD>
__global__ __launch_bounds__(PHIP_THREADS) void spmv_dmaskg_kernel(int32_t m, int32_t n, const DmWords *__restrict__ tmask, const int32_t *__restrict__ widx, const phip_u32x16 *__restrict__ gdesc,
                                                                   int ngroups, int dmin, int dmax, const double *__restrict__ x, const double *__restrict__ yin,
                                                                   double *__restrict__ yout, const double *__restrict__ dotv, double *__restrict__ wavepart, int nty, int xcdmap)
{
#pragma clang fp contract(off)
  const int      t = threadIdx.x;
  /* workgroups go to the XCDs round robin (b mod 8), and each XCD has an L2 of its own.  PHIP_SPMV_XCD_MAP=1 gives every XCD one
     contiguous eighth of the rows, so that the stretches of x neighbouring row blocks share are fetched into one L2 instead of two:
     27-point 256^3 back to back (x and masks in the Infinity Cache) 89.3 -> 84.4 us, but from cold caches -- the product inside KSPCG
     -- 131 -> 159 us and KSPCG 2862 -> 2796 it/s: eight XCDs streaming eight distant regions of x and y suit the memory worse than all
     of them sweeping one region together.  Off by default. */
  int bid = (int)blockIdx.x;
  if (xcdmap) {
    const int per = (int)gridDim.x >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const int      wave = __builtin_amdgcn_readfirstlane(bid * (PHIP_THREADS / 64) + (t >> 6));
  const int64_t  w0 = (int64_t)wave * DMG_WAVE_ROWS; /* first row of this wave */
  const int      r0 = bid * DMG_WG_ROWS + DM4_ROWS * t;
  const int      nact = m - r0; /* rows r0 .. r0 + 3 exist while i < nact */
  double         y0[DM4_ROWS] = {0.0, 0.0, 0.0, 0.0}, s[DM4_ROWS];
  if (ADD) {
    if (nact >= DM4_ROWS) {
      const phip_v2d_u a = *reinterpret_cast<const phip_v2d_u *>(yin + r0), b = *reinterpret_cast<const phip_v2d_u *>(yin + r0 + 2);
      y0[0] = a.x; y0[1] = a.y; y0[2] = b.x; y0[3] = b.y;
    } else {
#pragma unroll
      for (int i = 0; i < DM4_ROWS; i++) if (i < nact) y0[i] = yin[r0 + i];
    }
  }
#pragma unroll
  for (int i = 0; i < DM4_ROWS; i++) s[i] = y0[i];
  const DmWords *tmw = tmask + (size_t)__builtin_amdgcn_readfirstlane(widx[wave]) * DMG_WORDS; /* the wave's set of mask words */
  auto run = [&](auto fastc) {
    constexpr bool FAST = decltype(fastc)::value;
    DmGroup nxt = __builtin_bit_cast(DmGroup, gdesc[0]);
    for (int g = 0; g < ngroups; g++) {
      const DmGroup d = nxt;
      nxt = __builtin_bit_cast(DmGroup, gdesc[g + 1]); /* the table has a spare descriptor behind the last group */
      double w[DM4_WIN] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (FAST) { /* no range tests */
        const double    *p = x + ((int64_t)r0 + d.D);
        const phip_v2d_u a = *reinterpret_cast<const phip_v2d_u *>(p), b = *reinterpret_cast<const phip_v2d_u *>(p + 2);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
        if (d.span > 1) { const phip_v2d_u c = *reinterpret_cast<const phip_v2d_u *>(p + 4); w[4] = c.x; w[5] = c.y; }
      } else { /* entry by entry with clamped indices (an entry outside the row's own pairs is never used) */
#pragma unroll
        for (int j = 0; j < DM4_WIN; j++) {
          int64_t k = (int64_t)r0 + d.D + j;
          k = k < 0 ? 0 : (k >= (int64_t)n ? (int64_t)n - 1 : k);
          w[j] = x[k];
        }
      }
      /* the mask words, requested right behind the window and held here (the empty statement "uses" them): left to itself the compiler
         moves each request to its use, where the wave would sit out the latency alone */
      const DmWords m0 = tmw[d.q[0]], m1 = tmw[d.q[1]], m2 = tmw[d.q[2]];
      asm volatile("" ::"s"(m0.w[0]), "s"(m0.w[1]), "s"(m0.w[2]), "s"(m0.w[3]), "s"(m1.w[0]), "s"(m1.w[1]), "s"(m1.w[2]), "s"(m1.w[3]));
      asm volatile("" ::"s"(m2.w[0]), "s"(m2.w[1]), "s"(m2.w[2]), "s"(m2.w[3]));
      dmg_step<0>(s, w, d.v[0], m0); /* the first place of a group always carries a pair */
      if (d.span > 1) {
        dmg_step<1>(s, w, d.v[1], m1); /* a place without a pair: the empty words, nothing is added */
        dmg_step<2>(s, w, d.v[2], m2);
      }
    }
  };
  if (w0 + dmin >= 0 && w0 + DMG_WAVE_ROWS - 1 + dmax + (DM4_SPAN - 1) < (int64_t)n) run(std::true_type());
  else run(std::false_type());
  if (nact >= DM4_ROWS) {
    st_result2(yout + r0, s[0], s[1], nty);
    st_result2(yout + r0 + 2, s[2], s[3], nty);
  } else {
#pragma unroll
    for (int i = 0; i < DM4_ROWS; i++) if (i < nact) yout[r0 + i] = s[i];
  }
  if (DOT) { /* one partial per wave, stored and forgotten (dot_wave_partials_kernel sums them); the weights are requested only now */
    double dv[DM4_ROWS] = {0.0, 0.0, 0.0, 0.0};
    if (nact >= DM4_ROWS) {
      const phip_v2d_u a = *reinterpret_cast<const phip_v2d_u *>(dotv + r0), b = *reinterpret_cast<const phip_v2d_u *>(dotv + r0 + 2);
      dv[0] = a.x; dv[1] = a.y; dv[2] = b.x; dv[3] = b.y;
    } else {
#pragma unroll
      for (int i = 0; i < DM4_ROWS; i++) if (i < nact) dv[i] = dotv[r0 + i];
    }
    double dacc = 0.0;
#pragma unroll
    for (int i = 0; i < DM4_ROWS; i++) if (i < nact) dacc += dv[i] * (ADD ? s[i] - y0[i] : s[i]); /* MultAdd: the increment A x, not z */
    dacc = wave_sum(dacc);
    if ((t & 63) == 0) wavepart[(size_t)bid * (PHIP_THREADS / 64) + (t >> 6)] = dacc;
  }
}

/* second half of the fused dot of spmv_dmask_kernel: np wave partials, summed in a fixed order (thread t of workgroup b adds the entries
   (b * 256 + t) + k * gridDim.x * 256, k ascending, sixteen loads in flight; then block_sum and reduce_tail) and delivered as
   reduce_tail delivers */
__global__ __launch_bounds__(PHIP_THREADS) void dot_wave_partials_kernel(const double *__restrict__ part, int np, RedCtl c)
{
  __shared__ double lds[4];
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  double        a1[1] = {0.0};
  int64_t       i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x;
  for (; i + 15 * stride < np; i += 16 * stride) {
    double p[16];
#pragma unroll
    for (int u = 0; u < 16; u++) p[u] = part[i + u * stride];
#pragma unroll
    for (int u = 0; u < 16; u++) a1[0] += p[u];
  }
  for (; i < np; i += stride) a1[0] += part[i];
  block_sum<1>(a1, lds);
  __syncthreads();
  reduce_tail<1, 0>(a1, lds, c);
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

/* ---- hot-column order --------------------------------------------------------------------------------------------------------------
   An irregular matrix whose x does not fit L2 pays for every gather that misses with a 128-byte line from the Infinity Cache or HBM,
   and the product runs at the rate of those line fetches, not of the 12 B/nonzero matrix stream (DESIGN.md section 3, R-MAT).  Power-law
   matrices use few columns very often.  Numbering the columns by descending use puts the hot entries of x side by side, so the lines
   L2 keeps hold 16 hot entries each instead of one hot entry and its cold neighbours: x' = x[order] (one pass of 20 B per column per
   product), column indices renumbered, nonzeros left where they are -- every row adds the same products in the same order, so the
   result keeps its bits.  Measured, R-MAT scale 24 (x = 128 MiB): 3407 -> 2110 us + the pass over x; no gain while x is within reach of
   L2 (scale 22, 32 MiB: 503 -> 482 us, less than the pass costs), hence the size threshold.  Built when the share of gathers that land
   in the hottest HOT_LINES lines (one XCD's L2) rises by HOT_MIN_GAIN; PHIP_HOT_COLUMNS / phip_set_default_hot_columns: 0 never, 1 by that rule, 2 whenever the matrix has nonzeros. */
#define HOT_LINES 32768
#define HOT_MIN_BYTES (48ll << 20)
#define HOT_MIN_GAIN 0.04

__global__ __launch_bounds__(256) void hot_gather_kernel(int32_t n, const int32_t *__restrict__ order, const double *__restrict__ x, double *__restrict__ xp)
{
  for (int32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) xp[k] = x[order[k]];
}

static double hot_line_share(const int32_t *cols, int64_t nnz, int32_t n, const int32_t *newcol)
{
  std::vector<int64_t> lc(((size_t)n + 15) / 16, 0);
  for (int64_t k = 0; k < nnz; k++) lc[(size_t)((newcol ? newcol[cols[k]] : cols[k]) >> 4)]++;
  if (lc.size() > HOT_LINES) std::nth_element(lc.begin(), lc.begin() + HOT_LINES, lc.end(), [](int64_t a, int64_t b) { return a > b; });
  int64_t top = 0;
  for (size_t q = 0; q < lc.size() && q < HOT_LINES; q++) top += lc[q];
  return nnz ? (double)top / (double)nnz : 0.0;
}

static void free_hot(phip_mat A)
{
  if (A->ctx->device >= 0) { hipFree(A->d_hcol); hipFree(A->d_horder); hipFree(A->d_hx); }
  A->d_hcol = NULL; A->d_horder = NULL; A->d_hx = NULL; A->hot_built = 0; A->hot_on = 0;
}

/* columns by descending use, ties by ascending index (a counting sort over the use counts): order[k] = column in place k, newcol = its inverse */
static void column_use_order(const int32_t *h_j, int64_t nnz, int32_t n, std::vector<int32_t> &order, std::vector<int32_t> &newcol)
{
  std::vector<int32_t> cnt((size_t)n, 0);
  order.assign((size_t)n, 0); newcol.assign((size_t)n, 0);
  for (int64_t k = 0; k < nnz; k++) cnt[(size_t)h_j[k]]++;
  int32_t maxc = 0;
  for (int32_t c = 0; c < n; c++) maxc = std::max(maxc, cnt[(size_t)c]);
  std::vector<int64_t> start((size_t)maxc + 2, 0);
  for (int32_t c = 0; c < n; c++) start[(size_t)(maxc - cnt[(size_t)c]) + 1]++;
  for (size_t q = 1; q < start.size(); q++) start[q] += start[q - 1];
  for (int32_t c = 0; c < n; c++) { const int32_t k = (int32_t)start[(size_t)(maxc - cnt[(size_t)c])]++; order[(size_t)k] = c; newcol[(size_t)c] = k; }
}

static int setup_hot(phip_mat A, const int32_t *h_j, std::vector<int32_t> &order, std::vector<int32_t> &newcol)
{
  free_hot(A);
  if (g_hot_mode < 0) g_hot_mode = getenv("PHIP_HOT_COLUMNS") ? atoi(getenv("PHIP_HOT_COLUMNS")) : 1;
  const int mode = g_hot_mode;
  if (A->ctx->device < 0 || mode == 0 || A->nnz == 0 || A->n == 0 || A->sell_on) return 0;
  if (mode != 2 && (int64_t)sizeof(double) * A->n < HOT_MIN_BYTES) return 0;
  const int32_t n = A->n;
  column_use_order(h_j, A->nnz, n, order, newcol);
  A->hot_share[0] = hot_line_share(h_j, A->nnz, n, NULL);
  A->hot_share[1] = hot_line_share(h_j, A->nnz, n, newcol.data());
  if (mode != 2 && A->hot_share[1] - A->hot_share[0] < HOT_MIN_GAIN) return 0;
  std::vector<int32_t> hj((size_t)A->nnz + 8, 0);
  for (int64_t k = 0; k < A->nnz; k++) hj[(size_t)k] = newcol[(size_t)h_j[k]];
  hipStream_t st = A->ctx->stream;
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_hcol, sizeof(int32_t) * hj.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_horder, sizeof(int32_t) * (size_t)n));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_hx, sizeof(double) * (size_t)n));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_hcol, hj.data(), sizeof(int32_t) * hj.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_horder, order.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipStreamSynchronize(st));
  A->hot_built = 1; A->hot_on = 1;
  return 0;
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


/* ---- host: SELL-dict layout --------------------------------------------------------------------- */
/* Decide eligibility and build sval/scode/slen/sptr/soffs from the host CSR.  Returns 1 when built, 0 when not eligible. */
static int build_sell(const std::vector<int32_t> &rp, int32_t m, int32_t n, const int32_t *cj, const double *ca, std::vector<int64_t> &sptr,
                      std::vector<uint8_t> &slen, std::vector<int32_t> &soffs, raw_dvec *sval, std::vector<uint8_t> &scode)
{
  if (m <= 0 || n <= 0) return 0;
  const int64_t nnz = rp[(size_t)m];
  if (nnz == 0) return 0;
  const int32_t S = (m + SELL_H - 1) / SELL_H;
  const int     T = setup_threads(nnz);
  sptr.assign((size_t)S + 1, 0);
  slen.assign((size_t)S * SELL_H + 2, 0);
  std::atomic<int> bad(0);
  par_chunks(S, T, [&](int, int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      int L = 0;
      for (int64_t r = b * SELL_H; r < m && r < (b + 1) * SELL_H; r++) {
        const int len = rp[(size_t)r + 1] - rp[(size_t)r];
        if (len > 255) { bad = 1; return; }
        slen[(size_t)r] = (uint8_t)len;
        if (len > L) L = len;
      }
      sptr[(size_t)b + 1] = (int64_t)L * SELL_H;
    }
  });
  if (bad) return 0;
  for (int32_t b = 0; b < S; b++) sptr[(size_t)b + 1] += sptr[(size_t)b];
  const int64_t P = sptr[(size_t)S];
  if ((double)P > SELL_MAX_PAD * (double)nnz + SELL_H * 8) return 0;
  /* the diagonals, numbered by first appearance in row order: every chunk lists its own in that order, the lists are merged in
     chunk order */
  std::vector<std::vector<int32_t>> local((size_t)T);
  par_chunks(S, T, [&](int c, int64_t b0, int64_t b1) {
    std::vector<int32_t> &mine = local[(size_t)c];
    uint8_t               guess[256];
    memset(guess, 0, sizeof(guess));
    for (int64_t r = b0 * SELL_H; r < m && r < b1 * SELL_H; r++) {
      const int32_t k0 = rp[(size_t)r];
      const int     len = rp[(size_t)r + 1] - k0;
      for (int sl = 0; sl < len; sl++) {
        const int32_t d = cj[k0 + sl] - (int32_t)r;
        const int     gcode = guess[sl];
        if (gcode < (int)mine.size() && mine[(size_t)gcode] == d) continue;
        int code = -1;
        for (size_t q = 0; q < mine.size(); q++) if (mine[q] == d) { code = (int)q; break; }
        if (code < 0) {
          if (mine.size() >= SELL_MAX_OFFS) { bad = 1; return; }
          code = (int)mine.size();
          mine.push_back(d);
        }
        guess[sl] = (uint8_t)code;
      }
    }
  });
  if (bad) return 0;
  soffs.clear();
  for (const auto &mine : local)
    for (int32_t d : mine) {
      bool have = false;
      for (int32_t e : soffs) if (e == d) { have = true; break; }
      if (!have) { if (soffs.size() >= SELL_MAX_OFFS) return 0; soffs.push_back(d); }
    }
  scode.resize((size_t)P + 16);
  if (sval) sval->resize((size_t)P + 16);
  par_chunks(S, T, [&](int, int64_t b0, int64_t b1) {
    uint8_t guess[256];
    memset(guess, 0, sizeof(guess));
    for (int64_t b = b0; b < b1; b++) {
      const int64_t base = sptr[(size_t)b];
      const int     L = (int)((sptr[(size_t)b + 1] - base) / SELL_H);
      for (int64_t r = b * SELL_H; r < (b + 1) * SELL_H; r++) {
        const int64_t pos0 = base + (r - b * SELL_H);
        int           len = 0;
        if (r < m) {
          const int32_t k0 = rp[(size_t)r];
          len = rp[(size_t)r + 1] - k0;
          for (int sl = 0; sl < len; sl++) {
            const int32_t d = cj[k0 + sl] - (int32_t)r;
            int           code = guess[sl];
            if (code >= (int)soffs.size() || soffs[(size_t)code] != d) {
              for (size_t q = 0; q < soffs.size(); q++) if (soffs[q] == d) { code = (int)q; break; }
              guess[sl] = (uint8_t)code;
            }
            scode[(size_t)(pos0 + (int64_t)sl * SELL_H)] = (uint8_t)code;
            if (sval) (*sval)[(size_t)(pos0 + (int64_t)sl * SELL_H)] = ca[k0 + sl];
          }
        }
        for (int sl = len; sl < L; sl++) { /* padding */
          scode[(size_t)(pos0 + (int64_t)sl * SELL_H)] = 0;
          if (sval) (*sval)[(size_t)(pos0 + (int64_t)sl * SELL_H)] = 0.0;
        }
      }
    }
  });
  for (int k = 0; k < 16; k++) { scode[(size_t)P + k] = 0; if (sval) (*sval)[(size_t)P + k] = 0.0; }
  return 1;
}

static void fill_sell_values(const std::vector<int32_t> &rp, int32_t m, const double *ca, const std::vector<int64_t> &sptr, raw_dvec &sval)
{
  const int32_t S = (m + SELL_H - 1) / SELL_H;
  const int64_t P = sptr[(size_t)S];
  sval.resize((size_t)P + 16);
  par_chunks(S, setup_threads(rp[(size_t)m]), [&](int, int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      const int64_t base = sptr[(size_t)b];
      const int     L = (int)((sptr[(size_t)b + 1] - base) / SELL_H);
      for (int64_t r = b * SELL_H; r < (b + 1) * SELL_H; r++) {
        const int64_t pos0 = base + (r - b * SELL_H);
        int           len = 0;
        if (r < m) {
          const int32_t k0 = rp[(size_t)r];
          len = rp[(size_t)r + 1] - k0;
          for (int sl = 0; sl < len; sl++) sval[(size_t)(pos0 + (int64_t)sl * SELL_H)] = ca[k0 + sl];
        }
        for (int sl = len; sl < L; sl++) sval[(size_t)(pos0 + (int64_t)sl * SELL_H)] = 0.0;
      }
    }
  });
  for (int k = 0; k < 16; k++) sval[(size_t)P + k] = 0.0;
}

/* ---- pair dictionary -------------------------------------------------------------------------------------------------------------
   A matrix assembled from a constant-coefficient stencil holds few distinct values: the 7-point Laplacian two, each on its own
   diagonals.  When the stored entries of a SELL-dict matrix form at most 256 distinct (diagonal, value bit pattern) pairs, ONE byte
   per entry names both, the two tables sit in LDS, and a nonzero costs 1 byte of HBM traffic instead of 9 (12 in CSR).  The value a
   lane multiplies with is the table's copy of the very double the caller stored (keyed on its bits: -0.0, NaN payloads and
   denormals survive), and the sums run as before, so the result keeps the reference's bits.  phip_mat_update_values rebuilds the
   pairs; a matrix that stops fitting goes back to streamed values.  PHIP_VALUE_DICT=0 / phip_set_default_value_dict(0) never builds.
   Measured on MI355X (streamed values -> pair dictionary): 7-point 128^3 25.2 -> 14.1 us (KSPCG 13.6 k -> 16.3 k it/s); 27-point 256^3
   841 -> 300 us (KSPCG 851 -> 1585 it/s); convection-diffusion 192^3 94 -> 39.5 us.  What bounds it now is no longer HBM (the 27-point
   case moves 0.72 GB in 300 us) but the rate at which a CU turns codes into gathers: ~1.4 T nonzeros/s, the rate of the bare coalesced
   gather loop of scripts/micro/gather_rate.hip (1.17 T/s). */
static int g_vd_mode = -1;

static int build_pairs(const std::vector<uint8_t> &scode, const raw_dvec &sval, int64_t P, const std::vector<int32_t> &soffs,
                       std::vector<uint8_t> &pcode, std::vector<int32_t> &poffs, std::vector<double> &pvals)
{
  struct Slot { uint64_t bits; int32_t code; int32_t pair; };
  struct Found { uint64_t bits; int32_t code; };
  const int HB = 2048; /* open addressing, load <= 1/8 */
  auto      hash = [](uint64_t bits, int code) { return (size_t)((((bits * 0x9E3779B97F4A7C15ull) >> 40) ^ ((uint64_t)code * 0x9E37u)) & (uint64_t)(HB - 1)); };
  /* looks (code, bits) up in tab; adds it as pair `next` when absent and next >= 0; returns the pair or -1 */
  auto lookup = [&](std::vector<Slot> &tab, uint64_t bits, int code, int next) {
    for (size_t h = hash(bits, code);; h = (h + 1) & (HB - 1)) {
      Slot &sl = tab[h];
      if (sl.pair < 0) {
        if (next < 0) return -1;
        sl.bits = bits; sl.code = code; sl.pair = next;
        return next;
      }
      if (sl.bits == bits && sl.code == code) return (int)sl.pair;
    }
  };
  const int T = setup_threads(P);
  /* the pairs, numbered by first appearance: every chunk lists its own in that order, the lists are merged in chunk order */
  std::vector<std::vector<Found>> local((size_t)T);
  std::atomic<int>                bad(0);
  par_chunks(P, T, [&](int c, int64_t p0, int64_t p1) {
    std::vector<Slot>   tab((size_t)HB, Slot{0, -1, -1});
    std::vector<Found> &mine = local[(size_t)c];
    uint64_t            lastbits = 0;
    int                 lastcode = -1;
    for (int64_t p = p0; p < p1; p++) {
      uint64_t bits;
      memcpy(&bits, &sval[(size_t)p], sizeof(bits));
      const int code = scode[(size_t)p];
      if (code == lastcode && bits == lastbits) continue;
      lastbits = bits; lastcode = code;
      if (lookup(tab, bits, code, (int)mine.size()) == (int)mine.size()) {
        if (mine.size() >= SELL_MAX_OFFS) { bad = 1; return; }
        mine.push_back(Found{bits, code});
      }
    }
  });
  if (bad) return 0;
  std::vector<Slot> gtab((size_t)HB, Slot{0, -1, -1});
  poffs.clear(); pvals.clear();
  for (const auto &mine : local)
    for (const Found &f : mine)
      if (lookup(gtab, f.bits, f.code, (int)poffs.size()) == (int)poffs.size()) {
        if (poffs.size() >= SELL_MAX_OFFS) return 0;
        double v;
        memcpy(&v, &f.bits, sizeof(v));
        poffs.push_back(soffs[(size_t)f.code]);
        pvals.push_back(v);
      }
  pcode.resize((size_t)P + 16);
  par_chunks(P, T, [&](int, int64_t p0, int64_t p1) {
    std::vector<Slot> tab(gtab);
    uint64_t          lastbits = 0;
    int               lastcode = -1, lastpair = 0;
    for (int64_t p = p0; p < p1; p++) {
      uint64_t bits;
      memcpy(&bits, &sval[(size_t)p], sizeof(bits));
      const int code = scode[(size_t)p];
      if (code != lastcode || bits != lastbits) { lastpair = lookup(tab, bits, code, -1); lastbits = bits; lastcode = code; }
      pcode[(size_t)p] = (uint8_t)lastpair;
    }
  });
  for (int k = 0; k < 16; k++) pcode[(size_t)P + k] = 0;
  return 1;
}

static void free_dmask(phip_mat A);
static void free_pairs(phip_mat A)
{
  free_dmask(A);
  if (A->ctx->device >= 0) { hipFree(A->d_spoffs); hipFree(A->d_svtab); hipFree(A->d_pgrp); hipFree(A->d_pptr); }
  A->d_spoffs = NULL; A->d_svtab = NULL; A->d_pgrp = NULL; A->d_pptr = NULL; A->vd_built = 0; A->vd_on = 0; A->vd_npairs = 0;
}

static void free_dmask(phip_mat A)
{
  if (A->ctx->device >= 0) { hipFree(A->d_rmask); hipFree(A->d_dmoffs); hipFree(A->d_dmvals); hipFree(A->d_dotpart); hipFree(A->d_dmgrp); hipFree(A->d_tmask); hipFree(A->d_widx); hipFree(A->d_gdesc); }
  A->d_widx = NULL; A->dm_nsets = 0;
  A->d_tmask = NULL; A->d_gdesc = NULL; A->dm_ndesc = 0;
  A->d_rmask = NULL; A->d_dmoffs = NULL; A->d_dmvals = NULL; A->d_dotpart = NULL; A->d_dmgrp = NULL; A->dm_ngroups = 0; A->dm_built = 0; A->dm_on = 0; A->dm_npairs = 0;
}

/* diagonal-mask layout out of the pair codes (slot-major bytes, laid out like scode).  Not built when there are more than 32 pairs or
   when some row does not meet its pairs in strictly ascending diagonal order (unsorted or repeated columns): such a row's k-ascending
   sum is not the ascending-diagonal sum. */
static int g_dm_mode = -1;
static int g_dm_kernel = -1; /* 0 by size, 4 spmv_dmask4_kernel, 5 spmv_dmaskg_kernel */
#define DMG_MIN_ROWS (4 << 20)
static int setup_dmask(phip_mat A, const std::vector<uint8_t> &pcode, const std::vector<int32_t> &poffs, const std::vector<double> &pvals, int npairs)
{
  const int was_on = A->dm_built ? A->dm_on : 1;
  if (g_dm_mode < 0) g_dm_mode = getenv("PHIP_DIAG_MASK") ? atoi(getenv("PHIP_DIAG_MASK")) : 1;
  if (A->ctx->device < 0 || g_dm_mode == 0 || npairs > DMASK_MAX_PAIRS || npairs < 1) { free_dmask(A); return 0; }
  int order[DMASK_MAX_PAIRS], rank[DMASK_MAX_PAIRS];
  for (int q = 0; q < npairs; q++) order[q] = q;
  std::stable_sort(order, order + npairs, [&](int a, int b) { return poffs[(size_t)a] < poffs[(size_t)b]; });
  for (int q = 0; q < npairs; q++) rank[order[q]] = q;
  const int32_t         S = A->sell_nslices;
  std::vector<uint32_t> rmask((size_t)S * SELL_H + PHIP_THREADS * 8 + 2, 0u); /* padded to whole workgroups of spmv_dmask4_kernel / spmv_dmaskg_kernel */
  const std::vector<int32_t> &rp = *A->h_rowptr;
  const int        T = setup_threads(A->sell_padded);
  std::atomic<int> unordered(0);
  par_chunks(A->m, T, [&](int, int64_t r0, int64_t r1) {
    for (int64_t r = r0; r < r1; r++) {
      const int64_t b = r / SELL_H;
      const int64_t pos0 = (*A->h_sptr)[(size_t)b] + (r - b * SELL_H);
      const int     len = rp[(size_t)r + 1] - rp[(size_t)r];
      int           last = -1, lastoff = 0;
      uint32_t      mk = 0;
      for (int sl = 0; sl < len; sl++) {
        const int q = rank[pcode[(size_t)(pos0 + (int64_t)sl * SELL_H)]];
        const int off = poffs[(size_t)order[q]];
        if (q <= last || (last >= 0 && off <= lastoff)) { unordered = 1; return; }
        last = q; lastoff = off;
        mk |= 1u << q;
      }
      rmask[(size_t)r] = mk;
    }
  });
  if (unordered) { free_dmask(A); return 0; }
  int32_t offs[DMASK_MAX_PAIRS] = {0};
  double  vals[DMASK_MAX_PAIRS] = {0.0};
  for (int q = 0; q < npairs; q++) { offs[q] = poffs[(size_t)order[q]]; vals[q] = pvals[(size_t)order[q]]; }
  /* groups of pairs whose diagonals lie within DM4_SPAN of the group's first (spmv_dmask4_kernel): {base diagonal, pair set, span, 0} */
  int32_t grp[DMASK_MAX_PAIRS][4];
  int     ngroups = 0;
  for (int q = 0; q < npairs;) {
    const int32_t D = offs[q];
    uint32_t      set = 0;
    int32_t       span = 1;
    while (q < npairs && (int64_t)offs[q] - D < DM4_SPAN) { set |= 1u << q; span = offs[q] - D + 1; q++; }
    grp[ngroups][0] = D; grp[ngroups][1] = (int32_t)set; grp[ngroups][2] = span; grp[ngroups][3] = 0;
    ngroups++;
  }
  A->dm_ngroups = ngroups;
  /* the descriptors of spmv_dmaskg_kernel: as above, but a place of the window holds at most one pair (a diagonal that carries several
     values opens a new group for each further one) */
  DmGroup gd[DMASK_MAX_PAIRS + 4];
  memset(gd, 0, sizeof(gd));
  for (auto &G : gd) { G.span = 1; for (int j = 0; j < DM4_SPAN; j++) G.q[j] = DMASK_MAX_PAIRS; }
  int ndesc = 0;
  for (int q = 0; q < npairs;) {
    DmGroup &G = gd[ndesc++];
    G.D = offs[q];
    while (q < npairs && (int64_t)offs[q] - G.D < DM4_SPAN && G.q[offs[q] - G.D] == DMASK_MAX_PAIRS) {
      const int j = offs[q] - G.D;
      G.q[j] = q; G.v[j] = vals[q]; G.span = j + 1;
      q++;
    }
  }
  A->dm_dmin = offs[0]; A->dm_dmax = offs[npairs - 1];
  A->dm_ndesc = ndesc;
  /* the transposed masks: one wave = 256 rows, lane l holds the rows 4 l .. 4 l + 3 of them (words 0 .. 3); entry DMASK_MAX_PAIRS of
     a wave stays empty (the places of a window that carry no pair) */
  const size_t nwaves = (((size_t)A->m + DMG_WG_ROWS - 1) / DMG_WG_ROWS) * (PHIP_THREADS / 64);
  std::vector<unsigned long long> tmask(std::max<size_t>(nwaves, 1) * DMG_WORDS * 4, 0ull);
  par_chunks((int64_t)nwaves, T, [&](int, int64_t wv0, int64_t wv1) {
  for (int64_t wv = wv0; wv < wv1; wv++) {
    const int64_t first = (int64_t)wv * DMG_WAVE_ROWS;
    const int     nrows = (int)std::min<int64_t>(DMG_WAVE_ROWS, std::max<int64_t>((int64_t)A->m - first, 0));
    uint32_t      any = 0u, all = ~0u;
    for (int k = 0; k < nrows; k++) { any |= rmask[(size_t)(first + k)]; all &= rmask[(size_t)(first + k)]; }
    if (!nrows) continue;
    unsigned long long *tw = tmask.data() + wv * DMG_WORDS * 4;
    unsigned long long  act[4]; /* the lanes whose row i exists */
    for (int i = 0; i < 4; i++) { const int nl = (nrows - i + 3) / 4; act[i] = nl >= 64 ? ~0ull : ((1ull << (nl > 0 ? nl : 0)) - 1ull); }
    for (uint32_t pm = any; pm; pm &= pm - 1) { const int q = __builtin_ctz(pm); for (int i = 0; i < 4; i++) tw[q * 4 + i] = act[i]; } /* start from "all rows" */
    const uint32_t part = any & ~all;
    if (!part) continue;
    for (int k = 0; k < nrows; k++) /* ... and take out the few rows that miss the pair */
      for (uint32_t miss = part & ~rmask[(size_t)(first + k)]; miss; miss &= miss - 1) tw[__builtin_ctz(miss) * 4 + (k & 3)] &= ~(1ull << (k >> 2));
  }
  });
  /* most waves of a stencil matrix hold the same words (every interior line of the grid looks alike): keep each distinct set once and
     give every wave the number of its set -- 27-point 256^3: 65 536 waves, a few dozen sets, 70 MB of masks become 0.3 MB, which is
     what the product has to fetch again from HBM when it runs behind kernels that swept the caches (inside KSPCG) */
  const size_t          setw = (size_t)DMG_WORDS * 4;
  std::vector<int32_t>  widx(std::max<size_t>(nwaves, 1), 0);
  size_t                nsets = 0;
  {
    std::unordered_map<uint64_t, std::vector<int32_t>> seen;
    for (size_t wv = 0; wv < nwaves; wv++) {
      const unsigned long long *tw = tmask.data() + wv * setw;
      uint64_t                  hsh = 1469598103934665603ull;
      for (size_t k = 0; k < setw; k++) { hsh ^= tw[k]; hsh *= 1099511628211ull; hsh ^= hsh >> 29; }
      std::vector<int32_t> &cand = seen[hsh];
      int32_t               found = -1;
      for (int32_t c : cand) if (!memcmp(tmask.data() + (size_t)c * setw, tw, sizeof(unsigned long long) * setw)) { found = c; break; }
      if (found < 0) { /* sets are compacted to the front of tmask in place: set number nsets <= wv */
        found = (int32_t)nsets;
        if (nsets != wv) memmove(tmask.data() + nsets * setw, tw, sizeof(unsigned long long) * setw);
        cand.push_back(found);
        nsets++;
      }
      widx[wv] = found;
    }
  }
  A->dm_nsets = (int64_t)nsets;
  hipStream_t st = A->ctx->stream;
  if (!A->d_rmask) {
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_widx, sizeof(int32_t) * widx.size()));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_tmask, sizeof(unsigned long long) * tmask.size()));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_gdesc, sizeof(gd)));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_dmgrp, sizeof(grp)));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_rmask, sizeof(uint32_t) * rmask.size()));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_dmoffs, sizeof(offs)));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_dmvals, sizeof(vals)));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_dotpart, sizeof(double) * ((size_t)S * (PHIP_THREADS / 64) + 2)));
  }
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_rmask, rmask.data(), sizeof(uint32_t) * rmask.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_tmask, tmask.data(), sizeof(unsigned long long) * std::max<size_t>(nsets, 1) * setw, hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_widx, widx.data(), sizeof(int32_t) * widx.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_gdesc, gd, sizeof(gd), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_dmoffs, offs, sizeof(offs), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_dmvals, vals, sizeof(vals), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_dmgrp, grp, sizeof(grp), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipStreamSynchronize(st));
  A->dm_built = 1; A->dm_on = was_on; A->dm_npairs = npairs;
  return 0;
}

/* scode / sval: the SELL-dict arrays on the host; soffs: the diagonal table (its first sell_noffs entries) */
static int setup_pairs(phip_mat A, const std::vector<uint8_t> &scode, const raw_dvec &sval, const std::vector<int32_t> &soffs)
{
  const int was_on = A->vd_built ? A->vd_on : 1;
  if (g_vd_mode < 0) g_vd_mode = getenv("PHIP_VALUE_DICT") ? atoi(getenv("PHIP_VALUE_DICT")) : 1;
  if (A->ctx->device < 0 || g_vd_mode == 0) { free_pairs(A); return 0; }
  std::vector<uint8_t> pcode;
  std::vector<int32_t> poffs;
  std::vector<double>  pvals;
  SetupClock clk;
  if (!build_pairs(scode, sval, A->sell_padded, soffs, pcode, poffs, pvals)) { free_pairs(A); return 0; }
  clk.lap("    build_pairs");
  /* the grouped copy for spmv_pair_kernel */
  const int32_t S = A->sell_nslices;
  std::vector<int64_t> pptr((size_t)S + 1, 0);
  for (int32_t b = 0; b < S; b++) {
    const int64_t L = ((*A->h_sptr)[(size_t)b + 1] - (*A->h_sptr)[(size_t)b]) / SELL_H;
    pptr[(size_t)b + 1] = pptr[(size_t)b] + ((L + 3) / 4) * SELL_H;
  }
  std::vector<uint32_t> pgrp((size_t)pptr[(size_t)S] + 4, 0u);
  par_chunks(S, setup_threads(A->sell_padded), [&](int, int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; b++) {
      const int64_t  base = (*A->h_sptr)[(size_t)b], L = ((*A->h_sptr)[(size_t)b + 1] - base) / SELL_H;
      const uint8_t *src = pcode.data() + base;
      uint32_t      *dst = pgrp.data() + pptr[(size_t)b];
      for (int64_t sl = 0; sl < L; sl++) {
        const int      sh = 8 * (int)(sl & 3);
        uint32_t      *row = dst + (sl >> 2) * SELL_H;
        const uint8_t *from = src + sl * SELL_H;
        for (int r = 0; r < SELL_H; r++) row[r] |= (uint32_t)from[r] << sh;
      }
    }
  });
  clk.lap("    pair codes grouped");
  hipStream_t st = A->ctx->stream;
  if (!A->d_pgrp) {
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_pgrp, sizeof(uint32_t) * pgrp.size()));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_pptr, sizeof(int64_t) * pptr.size()));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_spoffs, sizeof(int32_t) * SELL_MAX_OFFS));
    PHIP_CHECK_HIP(hipMalloc((void **)&A->d_svtab, sizeof(double) * SELL_MAX_OFFS));
  }
  A->vd_npairs = (int32_t)poffs.size();
  {
    int rc = setup_dmask(A, pcode, poffs, pvals, A->vd_npairs);
    if (rc) return rc;
    clk.lap("    diagonal masks");
  }
  poffs.resize(SELL_MAX_OFFS, 0); pvals.resize(SELL_MAX_OFFS, 0.0);
  A->pgrp_dwords = pptr[(size_t)S];
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_pgrp, pgrp.data(), sizeof(uint32_t) * pgrp.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_pptr, pptr.data(), sizeof(int64_t) * pptr.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_spoffs, poffs.data(), sizeof(int32_t) * SELL_MAX_OFFS, hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_svtab, pvals.data(), sizeof(double) * SELL_MAX_OFFS, hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipStreamSynchronize(st));
  A->vd_built = 1; A->vd_on = was_on;
  return 0;
}

static void free_sell(phip_mat A)
{
  free_pairs(A);
  delete A->h_scode; A->h_scode = NULL;
  delete A->h_soffs; A->h_soffs = NULL;
  if (A->ctx->device >= 0) { hipFree(A->d_sval); hipFree(A->d_scode); hipFree(A->d_slen); hipFree(A->d_sptr); hipFree(A->d_soffs); }
  A->d_sval = NULL; A->d_scode = NULL; A->d_slen = NULL; A->d_sptr = NULL; A->d_soffs = NULL;
  A->sell_on = 0; A->sell_built = 0;
}

/* (re)build the SELL-dict copy according to fmt_user; h_j/h_a = host CSR */
static int setup_sell(phip_mat A, const int32_t *h_j, const double *h_a)
{
  free_sell(A);
  if (A->fmt_user == 1 || A->use_cprow) {
    PHIP_REQUIRE(A->fmt_user != 2, PHIP_ERR_SUP, "phip_mat: the SELL-dict layout does not apply to compressed-row matrices");
    return 0;
  }
  std::vector<uint8_t> slen, scode;
  std::vector<int32_t> soffs;
  raw_dvec             sval;
  SetupClock           clk;
  const int ok = build_sell(*A->h_rowptr, A->m, A->n, h_j, h_a, *A->h_sptr, slen, soffs, &sval, scode);
  clk.lap("  build_sell");
  if (!ok) {
    PHIP_REQUIRE(A->fmt_user != 2, PHIP_ERR_SUP, "phip_mat: matrix is not eligible for the SELL-dict layout (more than %d distinct diagonals, a row longer than 255, or padding above %.0f %%)", SELL_MAX_OFFS, (SELL_MAX_PAD - 1) * 100);
    return 0;
  }
  A->sell_nslices = (A->m + SELL_H - 1) / SELL_H;
  A->sell_noffs = (int32_t)soffs.size();
  A->sell_padded = (*A->h_sptr)[(size_t)A->sell_nslices];
  A->sell_maxlen = 0;
  for (int32_t r = 0; r < A->R; r++) { const int32_t L = (*A->h_rowptr)[(size_t)r + 1] - (*A->h_rowptr)[(size_t)r]; if (L > A->sell_maxlen) A->sell_maxlen = L; }
  A->sell_on = 1; A->sell_built = 1;
  if (A->ctx->device < 0) return 0;
  hipStream_t st = A->ctx->stream;
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_sval, sizeof(double) * sval.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_scode, scode.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_slen, slen.size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_sptr, sizeof(int64_t) * A->h_sptr->size()));
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_soffs, sizeof(int32_t) * SELL_MAX_OFFS));
  soffs.resize(SELL_MAX_OFFS, 0);
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_sval, sval.data(), sizeof(double) * sval.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_scode, scode.data(), scode.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_slen, slen.data(), slen.size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_sptr, A->h_sptr->data(), sizeof(int64_t) * A->h_sptr->size(), hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_soffs, soffs.data(), sizeof(int32_t) * SELL_MAX_OFFS, hipMemcpyHostToDevice, st));
  PHIP_CHECK_HIP(hipStreamSynchronize(st));
  clk.lap("  SELL arrays to the device");
  int rc = setup_pairs(A, scode, sval, soffs);
  if (rc) return rc;
  clk.lap("  setup_pairs (+ masks)");
  A->h_scode = new std::vector<uint8_t>();
  A->h_scode->swap(scode);
  A->h_soffs = new std::vector<int32_t>(soffs);
  return 0;
}

static int upload_plan(phip_mat A)
{
  build_plan(*A->h_rowptr, A->R, A->tpr_user, *A->h_blkrow, *A->h_blktpr);
  A->nblocks = (int32_t)A->h_blktpr->size();
  if (A->ctx->device < 0) return 0; /* planning-only: the plan stays on the host */
  if (A->d_blkrec) { hipFree(A->d_blkrec); A->d_blkrec = NULL; }
  std::vector<int4> rec((size_t)A->nblocks + 1, make_int4(0, 0, 0, 0));
  for (int32_t b = 0; b < A->nblocks; b++) {
    const int32_t r0 = (*A->h_blkrow)[(size_t)b], r1 = (*A->h_blkrow)[(size_t)b + 1];
    rec[(size_t)b] = make_int4(r0, (r1 - r0) | ((int32_t)(*A->h_blktpr)[(size_t)b] << 24), (*A->h_rowptr)[(size_t)r0], (*A->h_rowptr)[(size_t)r1]);
  }
  PHIP_CHECK_HIP(hipMalloc((void **)&A->d_blkrec, sizeof(int4) * rec.size()));
  PHIP_CHECK_HIP(hipMemcpyAsync(A->d_blkrec, rec.data(), sizeof(int4) * rec.size(), hipMemcpyHostToDevice, A->ctx->stream));
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
  SetupClock    clk;
  PHIP_REQUIRE(nnz == 0 || (h_j && h_a), PHIP_ERR_ARG_NULL, "phip_mat_create: null j/a with nz=%lld", (long long)nnz);
  if (use_cprow) PHIP_REQUIRE(h_cprow_i && (h_cprow_rindex || !cprow_nrows), PHIP_ERR_ARG_NULL, "phip_mat_create: compressed row arrays missing");
  /* operand checks the kernel relies on: monotone row pointer, column range */
  for (int32_t r = 0; r < m; r++) PHIP_REQUIRE(h_i[r + 1] >= h_i[r], PHIP_ERR_ARG_WRONGSTATE, "phip_mat_create: row pointer decreases at row %d", r);
  for (int64_t k = 0; k < nnz; k++) PHIP_REQUIRE(h_j[k] >= 0 && h_j[k] < n, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_create: column %d out of range [0,%d) at entry %lld", h_j[k], n, (long long)k);
  clk.lap("operand checks");
  if (ctx->device >= 0) PHIP_CHECK_HIP(hipSetDevice(ctx->device));
  phip_mat A = new phip_mat_s();
  memset(A, 0, sizeof(*A));
  A->ctx = ctx; A->m = m; A->n = n; A->nnz = nnz; A->use_cprow = use_cprow ? 1 : 0;
  A->R = use_cprow ? cprow_nrows : m;
  A->h_rowptr = new std::vector<int32_t>((size_t)A->R + 1);
  A->h_blkrow = new std::vector<int32_t>();
  A->h_blktpr = new std::vector<uint8_t>();
  A->h_sptr = new std::vector<int64_t>();
  A->fmt_user = g_default_fmt;
  A->nt_user = -1;
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
    if ((prc = setup_sell(A, h_j, h_a))) return prc;
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
  clk.lap("CSR to the device");
  int rc = upload_plan(A);
  if (rc) return rc;
  clk.lap("row-block plan");
  if ((rc = setup_sell(A, h_j, h_a))) return rc;
  clk.lap("SELL / pairs / masks");
  {
    std::vector<int32_t> order, newcol;
    if ((rc = setup_hot(A, h_j, order, newcol))) return rc;
  }
  clk.lap("hot columns");
  *out = A;
  return 0;
}

extern "C" int phip_mat_update_values(phip_mat A, const double *h_a)
{
  PHIP_REQUIRE(A && (h_a || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_update_values: null argument");
  if (A->ctx->device < 0) return 0;
  if (A->T) { /* same nonzero pattern: the transposed copy takes the new values through its permutation */
    std::vector<double> ta((size_t)A->nnz + 1);
    for (int64_t k = 0; k < A->nnz; k++) ta[(size_t)k] = h_a[(*A->t_perm)[(size_t)k]];
    int rc = phip_mat_update_values(A->T, ta.data());
    if (rc) return rc;
  }
  PHIP_CHECK_HIP(hipSetDevice(A->ctx->device));
  if (A->nnz) PHIP_CHECK_HIP(hipMemcpyAsync(A->d_val, h_a, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice, A->ctx->stream));
  if (A->sell_built) {
    raw_dvec sval;
    fill_sell_values(*A->h_rowptr, A->m, h_a, *A->h_sptr, sval);
    PHIP_CHECK_HIP(hipMemcpyAsync(A->d_sval, sval.data(), sizeof(double) * sval.size(), hipMemcpyHostToDevice, A->ctx->stream));
    PHIP_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
    return setup_pairs(A, *A->h_scode, sval, *A->h_soffs); /* the new values may form other pairs, or too many */
  }
  PHIP_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  return 0;
}

/* ---- value updates on the device --------------------------------------------------------------------------------------------------
   MatScale / MatShift / MatDiagonalScale / MatZeroEntries keep the nonzero pattern, so nothing but values has to change, and the
   values are on the device already: the kernels below update the CSR values in place, refresh the SELL-dict copy from them (an entry's
   place in it follows from its row and slot, no map is stored), and carry the change into the pair dictionary and the diagonal masks
   where that is exact -- scaling multiplies the table's values exactly as it multiplies the entries they stand for, a shift adds to
   the pairs on diagonal 0 -- or drop the dictionary (diagonal scaling makes the values depend on the position: the matrix then streams
   its values, 9 B/nz).  The transposed copy follows.  No value crosses PCIe; the host CSR of the caller is brought up to date with
   phip_mat_get_values when (if) the host asks for values again.  Round 1 did all of this on the host and re-sent every value. */
__global__ __launch_bounds__(PHIP_THREADS) void mat_scale_values_kernel(int64_t n, double *__restrict__ v, double alpha)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n; i += stride) v[i] = v[i] * alpha;
}

/* a[k] = (a[k] * l[row]) * r[col[k]]: the two loops of MatDiagonalScale_SeqAIJ (aij.c:2214-2228), two roundings per entry; l or r may be NULL */
__global__ __launch_bounds__(PHIP_THREADS) void mat_diagonal_scale_kernel(int32_t R, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx, int cprow,
                                                                         const int32_t *__restrict__ col, double *__restrict__ val, const double *__restrict__ l,
                                                                         const double *__restrict__ r, int colfirst)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t q = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; q < R; q += stride) {
    const double lv = l ? l[cprow ? ridx[q] : (int32_t)q] : 1.0;
    for (int32_t k = rowptr[q]; k < rowptr[q + 1]; k++) {
      double a = val[k];
      if (colfirst) { /* the transposed copy: its column factor is the original's row factor, which the reference applies first */
        if (r) a = a * r[col[k]];
        if (l) a = a * lv;
      } else {
        if (l) a = a * lv;
        if (r) a = a * r[col[k]];
      }
      val[k] = a;
    }
  }
}

/* a[k] += alpha where col[k] == row (MatShift_Basic, gcreate.c:6-24, on a matrix whose rows hold their diagonal entries) */
__global__ __launch_bounds__(PHIP_THREADS) void mat_shift_kernel(int32_t R, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ridx, int cprow,
                                                                const int32_t *__restrict__ col, double *__restrict__ val, double alpha)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t q = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; q < R; q += stride) {
    const int32_t i = cprow ? ridx[q] : (int32_t)q;
    for (int32_t k = rowptr[q]; k < rowptr[q + 1]; k++)
      if (col[k] == i) { val[k] = val[k] + alpha; break; }
  }
}

/* the SELL-dict values out of the CSR values: entry sl of row r lives at sptr[r / 512] + r % 512 + 512 sl */
__global__ __launch_bounds__(PHIP_THREADS) void sell_refresh_values_kernel(int32_t m, const int32_t *__restrict__ rowptr, const double *__restrict__ val,
                                                                          const int64_t *__restrict__ sptr, double *__restrict__ sval)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t r = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; r < m; r += stride) {
    const int64_t pos0 = sptr[r / SELL_H] + (r % SELL_H);
    const int32_t k0 = rowptr[r], len = rowptr[r + 1] - k0;
    for (int32_t sl = 0; sl < len; sl++) sval[pos0 + (int64_t)sl * SELL_H] = val[k0 + sl];
  }
}

/* the pair tables after a shift: the pairs on diagonal 0 stand for the diagonal entries */
/* the group descriptors carry copies of the pair values: bring them in line with the pair table after it changed */
__global__ void dmask_desc_refresh_kernel(int ndesc, DmGroup *__restrict__ gd, const double *__restrict__ vals)
{
  const int g = threadIdx.x / DM4_SPAN, j = threadIdx.x % DM4_SPAN;
  if (g < ndesc && gd[g].q[j] < DMASK_MAX_PAIRS) gd[g].v[j] = vals[gd[g].q[j]];
}
#define DMASK_DESC_REFRESH(A, st) do { if ((A)->dm_built) hipLaunchKernelGGL(dmask_desc_refresh_kernel, dim3(1), dim3(DMASK_MAX_PAIRS * DM4_SPAN), 0, st, (A)->dm_ndesc, reinterpret_cast<DmGroup *>((A)->d_gdesc), (A)->d_dmvals); } while (0)

__global__ void pair_table_shift_kernel(int n, const int32_t *__restrict__ offs, double *__restrict__ vals, double alpha)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n && offs[q] == 0) vals[q] = vals[q] + alpha;
}

static int dev_grid(int64_t n) { return phip_grid_for(n, 1, 8192); }

static int refresh_sell_values(phip_mat A)
{
  if (!A->sell_built) return 0;
  hipLaunchKernelGGL(sell_refresh_values_kernel, dim3(dev_grid(A->m)), dim3(PHIP_THREADS), 0, A->ctx->stream, A->m, A->d_rowptr, A->d_val, A->d_sptr, A->d_sval);
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

#define MAT_DEV_ENTER(A, name)                                                                        \
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, name ": null matrix");                                           \
  PHIP_NEED_DEVICE((A)->ctx, name);                                                                   \
  PHIP_FLUSH((A)->ctx);                                                                               \
  PHIP_CHECK_HIP(hipSetDevice((A)->ctx->device))

extern "C" int phip_mat_scale(phip_mat A, double alpha)
{
  MAT_DEV_ENTER(A, "phip_mat_scale");
  hipStream_t st = A->ctx->stream;
  if (A->nnz) hipLaunchKernelGGL(mat_scale_values_kernel, dim3(dev_grid(A->nnz)), dim3(PHIP_THREADS), 0, st, A->nnz, A->d_val, alpha);
  if (A->sell_built) hipLaunchKernelGGL(mat_scale_values_kernel, dim3(dev_grid(A->sell_padded)), dim3(PHIP_THREADS), 0, st, A->sell_padded, A->d_sval, alpha);
  if (A->vd_built) hipLaunchKernelGGL(mat_scale_values_kernel, dim3(1), dim3(PHIP_THREADS), 0, st, (int64_t)SELL_MAX_OFFS, A->d_svtab, alpha);
  if (A->dm_built) hipLaunchKernelGGL(mat_scale_values_kernel, dim3(1), dim3(PHIP_THREADS), 0, st, (int64_t)DMASK_MAX_PAIRS, A->d_dmvals, alpha);
  DMASK_DESC_REFRESH(A, st);
  PHIP_CHECK_HIP(hipGetLastError());
  return A->T ? phip_mat_scale(A->T, alpha) : 0;
}

extern "C" int phip_mat_shift(phip_mat A, double alpha)
{
  MAT_DEV_ENTER(A, "phip_mat_shift");
  hipStream_t st = A->ctx->stream;
  if (A->R) hipLaunchKernelGGL(mat_shift_kernel, dim3(dev_grid(A->R)), dim3(PHIP_THREADS), 0, st, A->R, A->d_rowptr, A->d_ridx, A->use_cprow, A->d_col, A->d_val, alpha);
  PHIP_CHECK_HIP(hipGetLastError());
  int rc = refresh_sell_values(A);
  if (rc) return rc;
  if (A->vd_built) hipLaunchKernelGGL(pair_table_shift_kernel, dim3(1), dim3(SELL_MAX_OFFS), 0, st, A->vd_npairs, A->d_spoffs, A->d_svtab, alpha);
  if (A->dm_built) hipLaunchKernelGGL(pair_table_shift_kernel, dim3(1), dim3(DMASK_MAX_PAIRS), 0, st, A->dm_npairs, A->d_dmoffs, A->d_dmvals, alpha);
  DMASK_DESC_REFRESH(A, st);
  PHIP_CHECK_HIP(hipGetLastError());
  return A->T ? phip_mat_shift(A->T, alpha) : 0;
}

static int mat_diagonal_scale_impl(phip_mat A, const double *d_l, const double *d_r, int colfirst)
{
  MAT_DEV_ENTER(A, "phip_mat_diagonal_scale");
  if (!d_l && !d_r) return 0;
  hipStream_t st = A->ctx->stream;
  if (A->R) hipLaunchKernelGGL(mat_diagonal_scale_kernel, dim3(dev_grid(A->R)), dim3(PHIP_THREADS), 0, st, A->R, A->d_rowptr, A->d_ridx, A->use_cprow, A->d_col, A->d_val, d_l, d_r, colfirst);
  PHIP_CHECK_HIP(hipGetLastError());
  int rc = refresh_sell_values(A);
  if (rc) return rc;
  if (A->vd_built) { PHIP_CHECK_HIP(hipStreamSynchronize(st)); free_pairs(A); } /* the values now depend on the position: streamed from here on */
  return A->T ? mat_diagonal_scale_impl(A->T, d_r, d_l, !colfirst) : 0;
}

extern "C" int phip_mat_diagonal_scale(phip_mat A, const double *d_l, const double *d_r) { return mat_diagonal_scale_impl(A, d_l, d_r, 0); }

extern "C" int phip_mat_zero_entries(phip_mat A)
{
  MAT_DEV_ENTER(A, "phip_mat_zero_entries");
  hipStream_t st = A->ctx->stream;
  if (A->nnz) PHIP_CHECK_HIP(hipMemsetAsync(A->d_val, 0, sizeof(double) * (size_t)A->nnz, st));
  if (A->sell_built) PHIP_CHECK_HIP(hipMemsetAsync(A->d_sval, 0, sizeof(double) * (size_t)A->sell_padded, st));
  if (A->vd_built) PHIP_CHECK_HIP(hipMemsetAsync(A->d_svtab, 0, sizeof(double) * SELL_MAX_OFFS, st));
  if (A->dm_built) PHIP_CHECK_HIP(hipMemsetAsync(A->d_dmvals, 0, sizeof(double) * DMASK_MAX_PAIRS, st));
  DMASK_DESC_REFRESH(A, st);
  return A->T ? phip_mat_zero_entries(A->T) : 0;
}

extern "C" int phip_mat_get_values(phip_mat A, double *h_a)
{
  MAT_DEV_ENTER(A, "phip_mat_get_values");
  PHIP_REQUIRE(h_a || !A->nnz, PHIP_ERR_ARG_NULL, "phip_mat_get_values: null array");
  if (A->nnz) PHIP_CHECK_HIP(hipMemcpyAsync(h_a, A->d_val, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToHost, A->ctx->stream));
  PHIP_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  return 0;
}

extern "C" int phip_mat_destroy(phip_mat A)
{
  if (!A) return 0;
  if (A->T) { phip_mat_destroy(A->T); A->T = NULL; }
  delete A->t_perm; A->t_perm = NULL;
  if (A->ctx->device < 0) { delete A->h_rowptr; delete A->h_blkrow; delete A->h_blktpr; delete A->h_sptr; delete A; return 0; }
  hipSetDevice(A->ctx->device);
  hipStreamSynchronize(A->ctx->stream);
  free_sell(A);
  free_hot(A);
  delete A->h_sptr;
  hipFree(A->d_rowptr); hipFree(A->d_ridx); hipFree(A->d_col); hipFree(A->d_val); hipFree(A->d_blkrec);
  delete A->h_rowptr; delete A->h_blkrow; delete A->h_blktpr;
  delete A;
  return 0;
}

template <bool DOT>
static int launch_spmv_t(phip_mat A, const double *d_x, const double *d_yin, double *d_yout, int add, const double *d_v, const RedCtl &ctl)
{
  hipStream_t st = A->ctx->stream;
  if (A->sell_on) {
    /* with the fused dot the grid is capped (the tail sums one partial per workgroup).  Measured on 7-point 128^3, CG it/s: cap 1024
       11.32 k, 2048 11.08 k, 4096 11.10 k -- the cost of the fused dot is its tail (partials -> last workgroup -> pinned host slot),
       not occupancy.  PHIP_SPMV_DOT_GRID overrides (<= PHIP_RED_RESIDENT: the collector waits for workgroups that must be resident) */
    static const int dot_grid_env = getenv("PHIP_SPMV_DOT_GRID") ? atoi(getenv("PHIP_SPMV_DOT_GRID")) : 0;
    const int      dot_cap = (dot_grid_env > 0 && dot_grid_env <= PHIP_RED_RESIDENT) ? dot_grid_env : 1024;
    const unsigned grid = DOT ? (unsigned)(A->sell_nslices < dot_cap ? A->sell_nslices : dot_cap) : (unsigned)A->sell_nslices;
    const bool     nt = A->nt_user >= 0 ? A->nt_user != 0 : 9 * A->sell_padded > PHIP_NT_MATRIX_BYTES;
    const int      sxlc = (grid % 8 == 0 || !DOT) ? xcd_chunk_log2(XCD_CHUNK_LOG2_SELL) : 0;
    /* results of 64 MiB and more leave with non-temporal stores: they do not push x out of the caches while the product still
       gathers from it (diagonal masks, 7-point 256^3: 97.9 -> 80.2 us, KSPCG 2759 -> 2805 it/s; 27-point 256^3: 227 -> 205 us;
       192^3, 57 MB: no change).  PHIP_SPMV_NT_Y=0|1 overrides. */
    static const int nty_env = getenv("PHIP_SPMV_NT_Y") ? atoi(getenv("PHIP_SPMV_NT_Y")) : -1;
    const int        nty = nty_env >= 0 ? nty_env : (A->m >= (8 << 20));
    const bool     vd = A->vd_built && A->vd_on; /* the pair dictionary has a kernel of its own */
#define SELL_LAUNCH2(NTv, Uv)                                                                                                                             \
  hipLaunchKernelGGL((spmv_sell_kernel<NTv, DOT, Uv>), dim3(grid), dim3(PHIP_THREADS), 0, st, A->m, A->n, A->sell_nslices, A->d_sptr, A->d_slen, A->d_sval,    \
                     A->d_scode, A->d_soffs, A->sell_noffs, d_x, d_yin, d_yout, add, d_v, ctl, nty, sxlc)
    /* slots requested per lane before the gathers start: 8 for long rows (27-point 192^3: 361.9 -> 349.7 us), 4 otherwise (rows of 7
       never fill a batch of 8: 27.1 -> 36.1 us); PHIP_SELL_UNROLL=4|8 overrides */
    static const int unroll_env = getenv("PHIP_SELL_UNROLL") ? atoi(getenv("PHIP_SELL_UNROLL")) : 0;
    /* the fused-dot variant needs ~100 registers with batches of 8 and loses more occupancy than the longer batches give back
       (27-point 256^3 inside KSPCG: 926 us against 903 us with batches of 4).  Also tried and dropped: taking the fused dot's weight
       x[r] from the row's own diagonal gather instead of a second load (256^3: 903 -> 967 us, the per-slot test costs more) */
    const int        unroll = unroll_env ? unroll_env : ((A->sell_maxlen >= 16 && !DOT) ? 8 : 4);
#define SELL_LAUNCH(NTv) do { if (unroll == 8) SELL_LAUNCH2(NTv, 8); else SELL_LAUNCH2(NTv, 4); } while (0)
    /* groups of 4 slots requested per lane before their gathers start: 2 for short rows, 1 for long ones (7-point 128^3 inside KSPCG
       16.19 k -> 16.34 k it/s with 2; 27-point 256^3 SpMV 327 -> 300 us with 1); PHIP_PAIR_GROUPS=1|2 overrides */
    static const int pair_g_env = getenv("PHIP_PAIR_GROUPS") ? atoi(getenv("PHIP_PAIR_GROUPS")) : 0;
    /* Diagonal masks: 7-point 128^3 14.1 -> 13.1 us, 27-point 256^3 300 -> 226 us against the pair codes.  With the fused dot on a
       capped grid a workgroup walks several slices one after the other, each starting with its serial prologue (masks, wave-wide
       OR), and loses what the wider loads give (7-point 128^3 inside KSPCG 20.8 -> 23.2 us at 1024 workgroups; 27-point 192^3: 136 us
       at 4096 workgroups against 90 us for the plain product).  So the fused dot runs like the plain product, one workgroup per
       slice; every WAVE leaves one partial with a plain store and ends -- no barrier, nothing waits -- and a small kernel right
       behind it sums the partials in a fixed order and publishes (dot_wave_partials_kernel): deterministic.  What a fused dot must
       not do in a kernel of many short workgroups is hold them: one partial per workgroup (LDS, barrier) with the reduce_tail code
       in the kernel cost 75 instead of 46 VGPRs and 27-point 256^3 ran 271 us against 205 us plain; arrival counters in the kernel
       (write-through store, wait, atomic with return: ~4 us per workgroup) the same, and 7-point 128^3 27 us against 13.6 us.
       PHIP_DMASK_DOT=0 sends the fused-dot products to the pair-code kernel instead (round 1's rule for matrices under 40 M nonzeros;
       with four rows a lane the masks win there too: 7-point 128^3 fused-dot product 13.9 us against 16.1 us). */
    static const int dm_dot_env = getenv("PHIP_DMASK_DOT") ? atoi(getenv("PHIP_DMASK_DOT")) : -1;
    if (vd && A->dm_built && A->dm_on && (!DOT || g_dm_mode == 2 || dm_dot_env != 0)) {
      double        *part = DOT ? A->d_dotpart : NULL;
      static const int dm_ug_env = getenv("PHIP_DMASK_GROUPS") ? atoi(getenv("PHIP_DMASK_GROUPS")) : 0;
      int              nwg;
      /* spmv_dmaskg_kernel from 4 Mi rows on and for matrices with 12 pairs and more, spmv_dmask4_kernel otherwise (27-point 256^3:
         116 -> 87 us with the former, 27-point 128^3 21.2 -> 15.9 us and KSPCG 13.95 k -> 14.96 k it/s; 7-point 128^3 inside KSPCG 1 %
         better with the latter: its launches are 10 us long and start from cold scalar caches);
         PHIP_DMASK_KERNEL / phip_set_dmask_kernel: 4 or 5 (the group-descriptor kernel) for all sizes.  Retired in round 3 after two
         rounds behind switches, measured slower: round 1's two-rows-a-lane kernel (27-point 256^3 205 us against 116 / 87 us) and the
         group-descriptor kernel with its windows staged in LDS (108 us against 90 us); both are in the history. */
      if (g_dm_kernel < 0) g_dm_kernel = getenv("PHIP_DMASK_KERNEL") ? atoi(getenv("PHIP_DMASK_KERNEL")) : 0;
      const bool use_g = g_dm_kernel == 5 || (g_dm_kernel != 4 && (A->m >= DMG_MIN_ROWS || A->dm_npairs >= 12));
      if (use_g) {
        nwg = (A->m + DMG_WG_ROWS - 1) / DMG_WG_ROWS;
#define DMG_LAUNCH(ADDv)                                                                                                                                  \
  hipLaunchKernelGGL((spmv_dmaskg_kernel<DOT, ADDv>), dim3((unsigned)nwg), dim3(PHIP_THREADS), 0, st, A->m, A->n, reinterpret_cast<const DmWords *>(A->d_tmask), A->d_widx, \
                     reinterpret_cast<const phip_u32x16 *>(A->d_gdesc), A->dm_ndesc, A->dm_dmin, A->dm_dmax, d_x, d_yin, d_yout, d_v, part, nty_g, xcd_map)
        /* this kernel stores its result with plain stores at every size: the next kernel of a Krylov iteration reads it back (KSPCG: the
           batch behind the product), and with the kernel's small footprint the non-temporal stores no longer protect anything -- KSPCG
           27-point 256^3 2889 -> 2917 it/s, 7-point 256^3 3053 -> 3173 it/s (product 64.9 -> 54.9 us); PHIP_SPMV_NT_Y=1 forces them */
        const int nty_g = nty_env >= 0 ? nty_env : 0;
        static const int xcd_map = getenv("PHIP_SPMV_XCD_MAP") ? atoi(getenv("PHIP_SPMV_XCD_MAP")) : 0;
        if (add) DMG_LAUNCH(true); else DMG_LAUNCH(false);
#undef DMG_LAUNCH
      } else {
        nwg = (A->m + PHIP_THREADS * DM4_ROWS - 1) / (PHIP_THREADS * DM4_ROWS);
        const int ug = dm_ug_env ? dm_ug_env : 1; /* 27-point 256^3: 1 / 2 / 3 / 4 groups in flight 121.6 / 126.6 / 133.0 / 147.3 us: occupancy hides the latency better */
#define DM4_LAUNCH(ADDv, UGv)                                                                                                                             \
  hipLaunchKernelGGL((spmv_dmask4_kernel<DOT, ADDv, UGv>), dim3((unsigned)nwg), dim3(PHIP_THREADS), 0, st, A->m, A->n, A->d_rmask, A->d_dmoffs, A->d_dmvals, \
                     reinterpret_cast<const int4 *>(A->d_dmgrp), A->dm_ngroups, d_x, d_yin, d_yout, d_v, part, nty)
        if (add) { if (ug == 2) DM4_LAUNCH(true, 2); else DM4_LAUNCH(true, 1); }
        else { if (ug == 2) DM4_LAUNCH(false, 2); else DM4_LAUNCH(false, 1); }
#undef DM4_LAUNCH
      }
      if (DOT) { /* 16 entries per thread and round: 27-point 256^3 has 131 072 wave partials = 32 workgroups */
        const int np = nwg * (PHIP_THREADS / 64);
        hipLaunchKernelGGL(dot_wave_partials_kernel, dim3(phip_grid_for(np, 16, PHIP_RED_GRID)), dim3(PHIP_THREADS), 0, st, A->d_dotpart, np, ctl);
      }
    }
    else if (vd) {
      const bool pnt = A->nt_user >= 0 ? A->nt_user != 0 : 4 * A->pgrp_dwords > PHIP_NT_MATRIX_BYTES;
      const int  G = pair_g_env ? pair_g_env : (A->sell_maxlen >= 16 ? 1 : 2);
#define PAIR_LAUNCH(NTv, Gv)                                                                                                                              \
  hipLaunchKernelGGL((spmv_pair_kernel<NTv, DOT, Gv>), dim3(grid), dim3(PHIP_THREADS), 0, st, A->m, A->sell_nslices, A->d_pptr, A->d_slen, A->d_pgrp, A->d_spoffs, \
                     A->d_svtab, A->vd_npairs, d_x, d_yin, d_yout, add, d_v, ctl, nty)
      if (pnt) { if (G == 2) PAIR_LAUNCH(true, 2); else PAIR_LAUNCH(true, 1); }
      else { if (G == 2) PAIR_LAUNCH(false, 2); else PAIR_LAUNCH(false, 1); }
#undef PAIR_LAUNCH
    }
    else if (nt) SELL_LAUNCH(true);
    else SELL_LAUNCH(false);
#undef SELL_LAUNCH2
#undef SELL_LAUNCH
  } else {
    static const int dot_grid_env = getenv("PHIP_SPMV_DOT_GRID") ? atoi(getenv("PHIP_SPMV_DOT_GRID")) : 0;
    const int      dot_cap = (dot_grid_env > 0 && dot_grid_env <= PHIP_RED_RESIDENT) ? dot_grid_env : 1024;
    const unsigned grid = DOT ? (unsigned)(A->nblocks < dot_cap ? (A->nblocks > 0 ? A->nblocks : 1) : dot_cap) : (unsigned)A->nblocks;
    /* out-of-cache irregular matrices: non-temporal matrix stream (R-MAT scale 22: 546 -> 505 us); PHIP_CSR_NT=0/1 overrides */
    static const int csr_nt_env = getenv("PHIP_CSR_NT") ? atoi(getenv("PHIP_CSR_NT")) : -1;
    const bool       csr_nt = A->nt_user >= 0 ? A->nt_user != 0 : (csr_nt_env >= 0 ? csr_nt_env != 0 : (12 * A->nnz > PHIP_NT_MATRIX_BYTES && !A->use_cprow));
    const int32_t   *cols = A->d_col;
    const double    *xs = d_x;
    if (A->hot_on && A->nblocks) { /* x in the hot-column order (setup_hot) */
      const unsigned g = (unsigned)std::min<int64_t>(((int64_t)A->n + 1023) / 1024, 16384);
      hipLaunchKernelGGL(hot_gather_kernel, dim3(g), dim3(256), 0, st, A->n, A->d_horder, d_x, A->d_hx);
      cols = A->d_hcol; xs = A->d_hx;
    }
    /* row blocks per XCD run: 512 for a large matrix known to be banded (it also has the SELL-dict layout): 7-point 256^3, every value
       perturbed, 16 / 128 / 512 / 1024 blocks: 318 / 303 / 308 / 314 us with 1.02 / 1.16 / 1.09 / ~1.05 x the algorithmic bytes of fabric
       traffic (runs shorter than the far diagonals' distance, 256 blocks here, put a block and its far neighbours on different XCDs and x
       crosses the fabric three times; much longer runs make eight XCDs stream eight distant regions); 128 for other large matrices;
       16 below 32 Ki blocks: R-MAT scale 20 (10 400 blocks, the heavy rows in the first of them) 90.6 / 91.9 / 98.1 us with runs of
       1 / 16 / 128 -- long runs put the expensive start of such a matrix on one XCD */
    const int xlc = (grid % 8 == 0 || !DOT) ? xcd_chunk_log2(A->nblocks >= 32768 ? (A->sell_built ? 9 : XCD_CHUNK_LOG2_CSR) : 4) : 0; /* a strided grid keeps a workgroup on its XCD only if 8 divides it */
    if (csr_nt)
      hipLaunchKernelGGL((spmv_rowblock_kernel<DOT, true>), dim3(grid), dim3(PHIP_THREADS), 0, st, A->nblocks, A->d_blkrec, A->d_rowptr, A->d_ridx, cols,
                         A->d_val, xs, d_yin, d_yout, add, A->use_cprow, d_v, ctl, xlc);
    else
      hipLaunchKernelGGL((spmv_rowblock_kernel<DOT, false>), dim3(grid), dim3(PHIP_THREADS), 0, st, A->nblocks, A->d_blkrec, A->d_rowptr, A->d_ridx, cols,
                         A->d_val, xs, d_yin, d_yout, add, A->use_cprow, d_v, ctl, xlc);
  }
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

static int launch_spmv(phip_mat A, const double *d_x, const double *d_yin, double *d_yout, int add)
{
  PHIP_FLUSH(A->ctx);
  if (!A->nblocks) return 0;
  hipEvent_t e0 = NULL, e1 = NULL;
  if (A->ctx->prof_on) {
    int rc = phip_prof_open(A->ctx, add ? 1 : 0, &e0, &e1);
    if (rc) return rc;
    PHIP_CHECK_HIP(hipEventRecord(e0, A->ctx->stream));
  }
  RedCtl none;
  memset(&none, 0, sizeof(none));
  int rc = launch_spmv_t<false>(A, d_x, d_yin, d_yout, add, NULL, none);
  if (rc) return rc;
  if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, A->ctx->stream));
  return 0;
}

static int launch_spmv_dot(phip_mat A, const double *d_x, const double *d_yin, double *d_yout, int add, const double *d_v, const RedCtl &ctl)
{
  PHIP_FLUSH(A->ctx);
  hipEvent_t e0 = NULL, e1 = NULL;
  if (A->ctx->prof_on) {
    int rc = phip_prof_open(A->ctx, add ? 1 : 0, &e0, &e1);
    if (rc) return rc;
    PHIP_CHECK_HIP(hipEventRecord(e0, A->ctx->stream));
  }
  int rc = launch_spmv_t<true>(A, d_x, d_yin, d_yout, add, d_v, ctl); /* runs even for an empty matrix: the tail must publish 0 */
  if (rc) return rc;
  if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, A->ctx->stream));
  return 0;
}

extern "C" int phip_mat_mult(phip_mat A, const double *d_x, double *d_y)
{
  PHIP_REQUIRE(A && (d_y || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_mult: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_mult");
  PHIP_FLUSH(A->ctx);
  PHIP_REQUIRE(d_x != d_y || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_mult: x and y must be different vectors (matrix.c:2213)");
  if (A->use_cprow && A->m) PHIP_CHECK_HIP(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)A->m, A->ctx->stream)); /* aij.c:1296 */
  return launch_spmv(A, d_x, NULL, d_y, 0);
}

extern "C" int phip_mat_multadd(phip_mat A, const double *d_x, const double *d_y, double *d_z)
{
  PHIP_REQUIRE(A && ((d_y && d_z) || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_multadd: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_multadd");
  PHIP_FLUSH(A->ctx);
  PHIP_REQUIRE(d_x != d_z || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_multadd: x and z must be different vectors (matrix.c:2361)");
  if (A->use_cprow && d_z != d_y && A->m)
    PHIP_CHECK_HIP(hipMemcpyAsync(d_z, d_y, sizeof(double) * (size_t)A->m, hipMemcpyDeviceToDevice, A->ctx->stream)); /* aij.c:1460-1462 */
  return launch_spmv(A, d_x, d_y, d_z, 1);
}

extern "C" int phip_mat_mult_dot(phip_mat A, const double *d_x, double *d_y, const double *d_v, int publish, phip_ticket *t)
{
  PHIP_REQUIRE(A && t && (d_y || !A->m) && (d_v || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_mult_dot: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_mult_dot");
  PHIP_FLUSH(A->ctx);
  PHIP_REQUIRE(d_x != d_y || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_mult_dot: x and y must be different vectors (matrix.c:2213)");
  if (A->use_cprow && A->m) PHIP_CHECK_HIP(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)A->m, A->ctx->stream)); /* aij.c:1296 */
  RedCtl ctl;
  int    rc = phip_ticket_open(A->ctx, 1, publish, t, &ctl);
  if (rc) return rc;
  return launch_spmv_dot(A, d_x, NULL, d_y, 0, d_v, ctl);
}

extern "C" int phip_mat_multadd_dot(phip_mat A, const double *d_x, double *d_y, const double *d_v, int publish, const phip_ticket *t)
{
  PHIP_REQUIRE(A && t && ((d_y && d_v) || !A->m) && (d_x || !A->nnz), PHIP_ERR_ARG_NULL, "phip_mat_multadd_dot: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_multadd_dot");
  PHIP_FLUSH(A->ctx);
  PHIP_REQUIRE(d_x != d_y || !A->m, PHIP_ERR_ARG_SIZ, "phip_mat_multadd_dot: x and y must be different vectors");
  RedCtl ctl;
  int    rc = phip_ticket_accumulate(A->ctx, t, publish, &ctl);
  if (rc) return rc;
  return launch_spmv_dot(A, d_x, d_y, d_y, 1, d_v, ctl);
}

/* ---- transposed products ------------------------------------------------------------------------------------------------------
   MatMultTransposeAdd_SeqAIJ (aij.c:1207-1254) walks the rows of A and adds alpha*v[j] into y[idx[j]]: entry c of the result takes the
   products of column c in ascending row order, starting from the old y[c] (or from the 0.0 of VecSet, aij.c:1263).  Row c of the
   explicit transpose, summed k-ascending by the ordinary product kernels, adds the same rounded products in the same order, so the
   result has the reference's bits, needs no atomics, and streams at the speed of the forward product (A^T of a stencil matrix is a
   stencil matrix: SELL-dict applies).  The price is a second copy of the matrix in HBM, paid only by callers of the transpose. */
extern "C" int phip_mat_transpose_setup(phip_mat A, const int32_t *h_i, const int32_t *h_j, const double *h_a)
{
  PHIP_REQUIRE(A && h_i, PHIP_ERR_ARG_NULL, "phip_mat_transpose_setup: null argument");
  PHIP_REQUIRE(h_i[0] == 0 && (int64_t)h_i[A->m] == A->nnz, PHIP_ERR_ARG_SIZ, "phip_mat_transpose_setup: the CSR passed has %d nonzeros, the matrix %lld", h_i[A->m], (long long)A->nnz);
  PHIP_REQUIRE(!A->nnz || (h_j && h_a), PHIP_ERR_ARG_NULL, "phip_mat_transpose_setup: null j/a");
  if (A->T) { phip_mat_destroy(A->T); A->T = NULL; }
  delete A->t_perm;
  A->t_perm = new std::vector<int32_t>((size_t)A->nnz + 1);
  std::vector<int32_t> ti((size_t)A->n + 1, 0), tj((size_t)A->nnz + 1);
  std::vector<double>  ta((size_t)A->nnz + 1);
  for (int64_t k = 0; k < A->nnz; k++) {
    PHIP_REQUIRE(h_j[k] >= 0 && h_j[k] < A->n, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_transpose_setup: column %d out of range at entry %lld", h_j[k], (long long)k);
    ti[(size_t)h_j[k] + 1]++;
  }
  for (int32_t c = 0; c < A->n; c++) ti[(size_t)c + 1] += ti[(size_t)c];
  std::vector<int32_t> fill(ti.begin(), ti.end() - 1);
  for (int32_t r = 0; r < A->m; r++) /* rows ascending: a stable counting sort by column */
    for (int32_t k = h_i[r]; k < h_i[r + 1]; k++) {
      const int32_t q = fill[(size_t)h_j[k]]++;
      tj[(size_t)q] = r; ta[(size_t)q] = h_a[k]; (*A->t_perm)[(size_t)q] = k;
    }
  int rc = phip_mat_create(A->ctx, A->n, A->m, ti.data(), tj.data(), ta.data(), 0, 0, NULL, NULL, &A->T);
  if (rc) return rc;
  /* the tuning choices made for A carry over */
  if (A->tpr_user && (rc = phip_mat_set_threads_per_row(A->T, A->tpr_user))) return rc;
  if (A->sell_built && !A->sell_on) A->T->sell_on = 0;
  return 0;
}

extern "C" int phip_mat_mult_transpose(phip_mat A, const double *d_x, double *d_y)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_mult_transpose: null matrix");
  PHIP_REQUIRE(A->T, PHIP_ERR_ARG_WRONGSTATE, "phip_mat_mult_transpose: call phip_mat_transpose_setup first");
  return phip_mat_mult(A->T, d_x, d_y);
}

extern "C" int phip_mat_mult_transpose_add(phip_mat A, const double *d_x, const double *d_y, double *d_z)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_mult_transpose_add: null matrix");
  PHIP_REQUIRE(A->T, PHIP_ERR_ARG_WRONGSTATE, "phip_mat_mult_transpose_add: call phip_mat_transpose_setup first");
  return phip_mat_multadd(A->T, d_x, d_y, d_z);
}

extern "C" int phip_mat_getdiagonal(phip_mat A, double *d_v)
{
  PHIP_REQUIRE(A && (d_v || !A->m), PHIP_ERR_ARG_NULL, "phip_mat_getdiagonal: null argument");
  PHIP_NEED_DEVICE(A->ctx, "phip_mat_getdiagonal");
  PHIP_FLUSH(A->ctx);
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

extern "C" int phip_mat_plan_lanes(phip_mat A, int64_t *h_hist)
{
  PHIP_REQUIRE(A && h_hist, PHIP_ERR_ARG_NULL, "phip_mat_plan_lanes: null argument");
  for (int l = 0; l < 8; l++) h_hist[l] = 0;
  const std::vector<int32_t> &rp = *A->h_rowptr, &br = *A->h_blkrow;
  for (int32_t b = 0; b < A->nblocks; b++) {
    if (rp[(size_t)br[(size_t)b + 1]] - rp[(size_t)br[(size_t)b]] > SPMV_CAP) h_hist[7]++;
    else h_hist[(*A->h_blktpr)[(size_t)b] > 6 ? 6 : (*A->h_blktpr)[(size_t)b]]++;
  }
  return 0;
}

extern "C" int phip_mat_set_threads_per_row(phip_mat A, int32_t tpr)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_threads_per_row: null matrix");
  PHIP_REQUIRE(tpr >= 0 && tpr <= 64 && (tpr & (tpr - 1)) == 0, PHIP_ERR_ARG_OUTOFRANGE, "threads per row must be 0 or a power of two <= 64, got %d", tpr);
  A->tpr_user = tpr;
  if (A->T) { int rc = phip_mat_set_threads_per_row(A->T, tpr); if (rc) return rc; }
  return upload_plan(A);
}

/* layout choice: 0 auto (SELL-dict when eligible), 1 CSR row blocks, 2 SELL-dict or fail.  The default applies to matrices created
   afterwards; phip_mat_set_format switches an existing matrix between the layouts that were built for it. */
extern "C" int phip_set_default_mat_format(int fmt)
{
  PHIP_REQUIRE(fmt >= 0 && fmt <= 2, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_default_mat_format: %d", fmt);
  g_default_fmt = fmt;
  return 0;
}

extern "C" int phip_mat_set_format(phip_mat A, int fmt)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_format: null matrix");
  PHIP_REQUIRE(fmt >= 0 && fmt <= 2, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_set_format: %d", fmt);
  PHIP_REQUIRE(fmt != 2 || A->sell_built, PHIP_ERR_SUP, "phip_mat_set_format: no SELL-dict copy was built for this matrix");
  A->sell_on = (fmt != 1 && A->sell_built) ? 1 : 0;
  if (A->T) A->T->sell_on = (fmt != 1 && A->T->sell_built) ? 1 : 0;
  return 0;
}

/* hot-column order of the row-block layout (setup_hot): on = 0 gathers from x as it is, 1 uses the order built at creation (error if
   none was built: x small, columns used evenly, or SELL-dict layout).  info: built / in use / share of gathers in the hottest lines
   before and after (0 if not evaluated). */
extern "C" int phip_set_default_hot_columns(int mode)
{
  PHIP_REQUIRE(mode >= 0 && mode <= 2, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_default_hot_columns: %d (0 never, 1 when it pays, 2 always)", mode);
  g_hot_mode = mode;
  return 0;
}

extern "C" int phip_mat_set_hot_columns(phip_mat A, int on)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_hot_columns: null matrix");
  PHIP_REQUIRE(!on || A->hot_built, PHIP_ERR_SUP, "phip_mat_set_hot_columns: no hot-column order was built for this matrix");
  A->hot_on = on ? 1 : 0;
  return 0;
}

extern "C" int phip_mat_hot_columns_info(phip_mat A, int *built, int *in_use, double *share_before, double *share_after)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_hot_columns_info: null matrix");
  if (built) *built = A->hot_built;
  if (in_use) *in_use = A->hot_on && !A->sell_on;
  if (share_before) *share_before = A->hot_share[0];
  if (share_after) *share_after = A->hot_share[1];
  return 0;
}

/* hot columns resident in LDS (setup_lds): on = 0 runs the row-block kernel, 1 the wave kernel built at creation (error if none was
   built).  info: built / in use / K / share of the nonzeros whose column is one of the K. */
extern "C" int phip_mat_mask_sets_info(phip_mat A, int64_t *nsets, int64_t *nwaves)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_mask_sets_info: null matrix");
  if (nsets) *nsets = A->dm_built ? A->dm_nsets : 0;
  if (nwaves) *nwaves = A->dm_built ? (((int64_t)A->m + DMG_WG_ROWS - 1) / DMG_WG_ROWS) * (PHIP_THREADS / 64) : 0;
  return 0;
}

extern "C" int phip_set_setup_threads(int nthreads)
{
  PHIP_REQUIRE(nthreads >= 0 && nthreads <= 64, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_setup_threads: %d (0: default, at most 64)", nthreads);
  g_setup_threads = nthreads;
  return 0;
}

extern "C" int phip_set_dmask_kernel(int kind)
{
  PHIP_REQUIRE(kind == 0 || kind == 4 || kind == 5, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_dmask_kernel: %d (0 by size, 4 bit sets per row, 5 group descriptors)", kind);
  g_dm_kernel = kind;
  return 0;
}

extern "C" int phip_mat_format_info(phip_mat A, int *format, int32_t *ndiagonals, int64_t *padded_entries)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_format_info: null matrix");
  if (format) *format = A->sell_on ? 2 : 1;
  if (ndiagonals) *ndiagonals = A->sell_built ? A->sell_noffs : 0;
  if (padded_entries) *padded_entries = A->sell_built ? A->sell_padded : 0;
  return 0;
}

/* pair dictionary of the SELL-dict layout (setup_pairs): on = 0 streams the values, 1 uses the pairs built at upload (error if the
   matrix has more than 256 distinct (diagonal, value) pairs or is not in the SELL-dict layout) */
extern "C" int phip_set_default_value_dict(int mode)
{
  PHIP_REQUIRE(mode == 0 || mode == 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_default_value_dict: %d (0 never, 1 when the matrix allows)", mode);
  g_vd_mode = mode;
  return 0;
}

extern "C" int phip_mat_set_value_dict(phip_mat A, int on)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_value_dict: null matrix");
  PHIP_REQUIRE(!on || A->vd_built, PHIP_ERR_SUP, "phip_mat_set_value_dict: no pair dictionary was built for this matrix");
  A->vd_on = on ? 1 : 0;
  if (A->T && A->T->vd_built) A->T->vd_on = A->vd_on;
  return 0;
}

extern "C" int phip_mat_value_dict_info(phip_mat A, int *built, int *in_use, int32_t *npairs)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_value_dict_info: null matrix");
  if (built) *built = A->vd_built;
  if (in_use) *in_use = A->vd_built && A->vd_on && A->sell_on;
  if (npairs) *npairs = A->vd_built ? A->vd_npairs : 0;
  return 0;
}

/* diagonal-mask layout (setup_dmask): on = 0 falls back to the pair codes, 1 uses the masks built at upload (error if none) */
extern "C" int phip_set_default_diag_mask(int mode)
{
  PHIP_REQUIRE(mode >= 0 && mode <= 2, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_default_diag_mask: %d (0 never, 1 when the matrix allows, 2 also for the fused dot of small matrices)", mode);
  g_dm_mode = mode;
  return 0;
}

extern "C" int phip_mat_set_diag_mask(phip_mat A, int on)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_diag_mask: null matrix");
  PHIP_REQUIRE(!on || A->dm_built, PHIP_ERR_SUP, "phip_mat_set_diag_mask: no diagonal-mask layout was built for this matrix");
  A->dm_on = on ? 1 : 0;
  if (A->T && A->T->dm_built) A->T->dm_on = A->dm_on;
  return 0;
}

extern "C" int phip_mat_diag_mask_info(phip_mat A, int *built, int *in_use, int32_t *npairs)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_diag_mask_info: null matrix");
  if (built) *built = A->dm_built;
  if (in_use) *in_use = A->dm_built && A->dm_on && A->vd_built && A->vd_on && A->sell_on;
  if (npairs) *npairs = A->dm_built ? A->dm_npairs : 0;
  return 0;
}

/* tuning: how the kernels load the matrix stream: -1 by size (non-temporal from 192 MB on), 0 cached, 1 non-temporal; same bits */
extern "C" int phip_mat_set_nontemporal(phip_mat A, int mode)
{
  PHIP_REQUIRE(A, PHIP_ERR_ARG_NULL, "phip_mat_set_nontemporal: null matrix");
  PHIP_REQUIRE(mode >= -1 && mode <= 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_mat_set_nontemporal: %d (-1 by size, 0 never, 1 always)", mode);
  A->nt_user = mode;
  if (A->T) A->T->nt_user = mode;
  return 0;
}

