// Krylov vector kernels of libpetschip (see include/petschip.h): the _VecOps slots of VECSEQHIP / VECMPIHIP.
//
// Every kernel is HBM-bound: grid-stride over 16-byte (double2) accesses, tail handled by scalar code.
// Algorithmic bytes: dot 16n, norm 8n, axpy/aypx/pointwise 24n, waxpy 24n, scale 16n, copy 16n, set 8n,
// mdot(nv) 8n(nv+1), maxpy(nv) 8n(nv+2).
// Reductions: fixed grid of at most PHIP_RED_GRID workgroups; wavefront (64-lane) shuffle reduction -> LDS across the 4 waves ->
// one partial per workgroup, stored write-through (sc1) as flagged words; the workgroup with the highest index collects them in a
// fixed order (deterministic for a given n, whichever workgroup that is), writes the result to device memory and, for the calls
// that return a host scalar, into a pinned slot followed by a sequence word the host spins on: one launch per reduction,
// no second kernel, no copy, no stream synchronisation.  Hand-off form: MI355X_MICROARCH.md "Hand-offs measured with sc1 loads"
// row 1 (one lane per workgroup stores sc1, drains vmcnt, adds to an agent-scope counter; the last arriver loads sc1).
#include "phip_internal.h"

#define MV_MAX 16 /* vectors fused per mdot/maxpy launch */
#define MV_NT_MIN 8 /* vector lists from this length on are read with non-temporal loads */

struct MVPack {
  const double *y[MV_MAX];
  double        a[MV_MAX];
};

/* where a reduction launch leaves its results */
#include "phip_reduce.h"

/* z[k] = sum_i x_i * y_k,i  for k < NV : x is read once per launch.  partial layout: [k][block] */
/* COUNTED: the tail of phip_vec_mdot_static (reduce_tail_counted): ctl.partial = the static rows, ctl.h_ll = their arrival counter */
template <int NV, bool COUNTED = false>
__global__ __launch_bounds__(PHIP_THREADS) void mdot_kernel(int32_t n, const double *__restrict__ x, MVPack P, RedCtl ctl)
{
  /* many vectors, each read once: non-temporal loads keep them from pushing x out of the caches (192^3: mdot16 165 -> 155 us, mdot8
     93 -> 82 us; with 2 or 4 vectors the plain loads are faster) */
  constexpr bool nt = NV >= MV_NT_MIN;
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ double lds[4 * NV];
  double            acc[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) acc[k] = 0.0;
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double2 yv;
      if (nt) { const v2d q = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(P.y[k]) + i); yv = make_double2(q.x, q.y); }
      else yv = reinterpret_cast<const double2 *>(P.y[k])[i];
      acc[k] += xv.x * yv.x + xv.y * yv.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] += x[n - 1] * P.y[k][n - 1];
  }
  block_sum<NV>(acc, lds);
  if (COUNTED) reduce_tail_counted<NV>(acc, ctl.partial, reinterpret_cast<unsigned int *>(ctl.h_ll), ctl.d_out);
  else reduce_tail<NV, 0>(acc, lds, ctl);
}

/* kind 1: sum x^2 ; 2: sum |x| ; 3: max |x| with NaN propagation (bvec2.c:253-258) ; 4: {sum |x|, sum x^2} */
template <int KIND>
__global__ __launch_bounds__(PHIP_THREADS) void norm_kernel(int32_t n, const double *__restrict__ x, RedCtl ctl)
{
  __shared__ double lds[8];
  double            acc[2] = {0.0, 0.0};
  bool              isnan_any = false;
  const int64_t     n2 = n >> 1;
  const int64_t     stride = (int64_t)gridDim.x * PHIP_THREADS;
  auto              body = [&](double v) {
    if (KIND == 1) acc[0] += v * v;
    else if (KIND == 2) acc[0] += fabs(v);
    else if (KIND == 3) { acc[0] = fmax(acc[0], fabs(v)); isnan_any |= (v != v); }
    else { acc[0] += fabs(v); acc[1] += v * v; }
  };
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
    body(xv.x); body(xv.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) body(x[n - 1]);
  if (KIND == 3) {
    double m = wave_max(acc[0]);
    double f = wave_max(isnan_any ? 1.0 : 0.0);
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = m; lds[4 + (threadIdx.x >> 6)] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
      acc[0] = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
      acc[1] = fmax(fmax(lds[4], lds[5]), fmax(lds[6], lds[7]));
    }
    __syncthreads();
    reduce_tail<2, 1>(acc, lds, ctl);
  } else if (KIND == 4) {
    block_sum<2>(acc, lds);
    __syncthreads();
    reduce_tail<2, 0>(acc, lds, ctl);
  } else {
    double a1[1] = {acc[0]};
    block_sum<1>(a1, lds);
    __syncthreads();
    reduce_tail<1, 0>(a1, lds, ctl);
  }
}

/* x += sum_k a_k y_k : x read and written once per launch.  RED: the by-product sum x_new^2 (the VecNorm of the VecNormalize that
   follows the orthogonalisation, gmres.c:170-176 after borthog2.c:60-67) leaves through ctl; grid <= PHIP_RED_BLOCKS then */
template <int NV, bool RED = false>
__global__ __launch_bounds__(PHIP_THREADS) void maxpy_kernel(int32_t n, double *__restrict__ x, MVPack P, RedCtl ctl)
{
  constexpr bool nt = NV >= MV_NT_MIN; /* as in mdot_kernel (192^3: maxpy16 198 -> 165 us together with the larger grid; maxpy2 32.7 -> 35.7 us, hence the limit) */
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  typedef double v2d __attribute__((ext_vector_type(2)));
  double sq[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    double2 xv = reinterpret_cast<double2 *>(x)[i];
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double2 yv;
      if (nt) { const v2d q = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(P.y[k]) + i); yv = make_double2(q.x, q.y); }
      else yv = reinterpret_cast<const double2 *>(P.y[k])[i];
      xv.x += P.a[k] * yv.x;
      xv.y += P.a[k] * yv.y;
    }
    reinterpret_cast<double2 *>(x)[i] = xv;
    if (RED) sq[0] += xv.x * xv.x + xv.y * xv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v = x[n - 1];
#pragma unroll
    for (int k = 0; k < NV; k++) v += P.a[k] * P.y[k][n - 1];
    x[n - 1] = v;
    if (RED) sq[0] += v * v;
  }
  if (RED) {
    __shared__ double lds[4];
    block_sum<1>(sq, lds);
    __syncthreads();
    reduce_tail<1, 0>(sq, lds, ctl);
  }
}

/* generic elementwise kernel: out[i] = f(i) over double2 with scalar tail */
enum { EW_AXPY, EW_AYPX, EW_XMY, EW_WAXPY, EW_WXPY, EW_WYMX, EW_AXPBY, EW_SCALEX, EW_AXPBYPCZ, EW_PMULT, EW_PDIV, EW_SCALE, EW_SET, EW_RECIP, EW_SWAP };

template <int OP>
__device__ __forceinline__ double ew_apply(double a, double b, double g, double x, double y, double z)
{
  switch (OP) {
  case EW_AXPY: return y + a * x;        /* y += a x       (daxpy) */
  case EW_AYPX: return x + a * y;        /* y = x + a y    dvec2.c:678 */
  case EW_XMY: return x - y;             /* y = x - y      dvec2.c:661 */
  case EW_WAXPY: return y + a * x;       /* w = y + a x    dvec2.c:722 */
  case EW_WXPY: return y + x;            /* dvec2.c:711 */
  case EW_WYMX: return y - x;            /* dvec2.c:714 */
  case EW_AXPBY: return a * x + b * y;   /* bvec1.c:126 */
  case EW_SCALEX: return a * x;          /* bvec1.c:117 */
  case EW_AXPBYPCZ: return a * x + b * y + g * z; /* bvec1.c:158 */
  case EW_PMULT: return x * y;
  case EW_PDIV: return x / y;
  case EW_SCALE: return a * x;
  case EW_SET: return a;
  case EW_RECIP: return (x != 0.0) ? 1.0 / x : x;
  }
  return 0.0;
}

__device__ __forceinline__ double2 ld2(const double *p, int64_t i, bool nt)
{
  typedef double v2d __attribute__((ext_vector_type(2)));
  if (nt) { const v2d q = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p) + i); return make_double2(q.x, q.y); }
  return reinterpret_cast<const double2 *>(p)[i];
}

__device__ __forceinline__ void st2(double *p, int64_t i, const double2 v, bool nt)
{
  typedef double v2d __attribute__((ext_vector_type(2)));
  if (nt) { v2d q; q.x = v.x; q.y = v.y; __builtin_nontemporal_store(q, reinterpret_cast<v2d *>(p) + i); }
  else reinterpret_cast<double2 *>(p)[i] = v;
}

/* out = f(x, y, z); any of x,y,z may alias out (pure elementwise) */
template <int OP, bool HX, bool HY, bool HZ>
__global__ __launch_bounds__(PHIP_THREADS) void ew_kernel(int32_t n, double *out, const double *x, const double *y, const double *z, double a, double b, double g, int nt)
{
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    double2 xv = {0, 0}, yv = {0, 0}, zv = {0, 0}, o;
    if (HX) xv = ld2(x, i, nt & 1);
    if (HY) yv = ld2(y, i, nt & 1);
    if (HZ) zv = ld2(z, i, nt & 1);
    o.x = ew_apply<OP>(a, b, g, xv.x, yv.x, zv.x);
    o.y = ew_apply<OP>(a, b, g, xv.y, yv.y, zv.y);
    reinterpret_cast<double2 *>(out)[i] = o;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int32_t i = n - 1;
    out[i] = ew_apply<OP>(a, b, g, HX ? x[i] : 0.0, HY ? y[i] : 0.0, HZ ? z[i] : 0.0);
  }
}

__global__ __launch_bounds__(PHIP_THREADS) void swap_kernel(int32_t n, double *x, double *y)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n; i += stride) {
    const double t = x[i];
    x[i] = y[i];
    y[i] = t;
  }
}

/* PCSetUp_Jacobi host loop, jacobi.c:182-195 / :204-212 */
__global__ __launch_bounds__(PHIP_THREADS) void jacobi_setup_kernel(int32_t n, double *d, int useabs, int sqrtv, int *flag)
{
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  bool          zero = false;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n; i += stride) {
    double v = d[i];
    if (sqrtv) {
      if (v != 0.0) v = 1.0 / sqrt(fabs(v));
      else { v = 1.0; zero = true; }
    } else {
      if (v != 0.0) v = 1.0 / v;     /* VecReciprocal_Default vinv.c:1221 */
      if (useabs) v = fabs(v);
      if (v == 0.0) { v = 1.0; zero = true; }
    }
    d[i] = v;
  }
  if (zero) atomicOr(flag, 1);
}

/* w = x .* y with the by-products {sum w^2, sum w x, sum w y} (PCApply_Jacobi followed by VecNorm(Z) and VecTDot(Z,R), cg.c:225-269) */
__global__ __launch_bounds__(PHIP_THREADS) void pmult_reduce_kernel(int32_t n, double *__restrict__ w, const double *__restrict__ x,
                                                                    const double *__restrict__ y, RedCtl ctl)
{
  __shared__ double lds[12];
  double            acc[3] = {0.0, 0.0, 0.0};
  const int64_t     n2 = n >> 1;
  const int64_t     stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
    const double2 yv = reinterpret_cast<const double2 *>(y)[i];
    double2       o;
    o.x = xv.x * yv.x; o.y = xv.y * yv.y;
    reinterpret_cast<double2 *>(w)[i] = o;
    acc[0] += o.x * o.x + o.y * o.y;
    acc[1] += o.x * xv.x + o.y * xv.y;
    acc[2] += o.x * yv.x + o.y * yv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double o = x[n - 1] * y[n - 1];
    w[n - 1] = o;
    acc[0] += o * o; acc[1] += o * x[n - 1]; acc[2] += o * y[n - 1];
  }
  block_sum<3>(acc, lds);
  __syncthreads();
  reduce_tail<3, 0>(acc, lds, ctl);
}

/* ---- batched elementwise operations -----------------------------------------------------------------
   KSPSolve_CG issues VecAXPY(X,a,P); VecAXPY(R,-a,W); PCApply(R,Z) back to back (cg.c:222-225): three launches, and R is written by the
   second and read again by the third.  Elementwise operations therefore wait in the context (phip_internal.h) until something else needs
   the stream, and up to PHIP_EW_BATCH of them leave in one launch.  Every thread applies the operations IN CALL ORDER to the entries it
   owns, so a later operation sees what an earlier one stored to the same entry (the operands of one entry never depend on another
   entry), and what it stored is still in cache when the next operation asks for it.  Arithmetic per operation is that of the single
   kernels (ew_apply), so results are bit-identical to unbatched execution.  RED: the last operation is w = x .* y with the by-product
   sums of pmult_reduce_kernel. */
struct EwBatch { int nops; phip_ctx_s::phip_ew_op op[PHIP_EW_BATCH]; };

__device__ __forceinline__ double ew_apply_rt(int op, double a, double b, double g, double x, double y, double z)
{
  switch (op) {
  case EW_AXPY: return ew_apply<EW_AXPY>(a, b, g, x, y, z);
  case EW_AYPX: return ew_apply<EW_AYPX>(a, b, g, x, y, z);
  case EW_XMY: return ew_apply<EW_XMY>(a, b, g, x, y, z);
  case EW_WAXPY: return ew_apply<EW_WAXPY>(a, b, g, x, y, z);
  case EW_WXPY: return ew_apply<EW_WXPY>(a, b, g, x, y, z);
  case EW_WYMX: return ew_apply<EW_WYMX>(a, b, g, x, y, z);
  case EW_AXPBY: return ew_apply<EW_AXPBY>(a, b, g, x, y, z);
  case EW_SCALEX: return ew_apply<EW_SCALEX>(a, b, g, x, y, z);
  case EW_AXPBYPCZ: return ew_apply<EW_AXPBYPCZ>(a, b, g, x, y, z);
  case EW_PMULT: return ew_apply<EW_PMULT>(a, b, g, x, y, z);
  case EW_PDIV: return ew_apply<EW_PDIV>(a, b, g, x, y, z);
  case EW_SCALE: return ew_apply<EW_SCALE>(a, b, g, x, y, z);
  case EW_SET: return ew_apply<EW_SET>(a, b, g, x, y, z);
  default: return ew_apply<EW_RECIP>(a, b, g, x, y, z);
  }
}

template <bool RED>
__global__ __launch_bounds__(PHIP_THREADS) void ew_batch_kernel(int32_t n, EwBatch B, RedCtl ctl)
{
  __shared__ double lds[12];
  double            acc[3] = {0.0, 0.0, 0.0};
  const int64_t     n2 = n >> 1;
  const int64_t     stride = (int64_t)gridDim.x * PHIP_THREADS;
  /* EW_UNROLL entries per thread and round: the loads of one operation for all of them are in flight together (an operation cannot start
     before the previous one's stores: the compiler must assume that they alias) */
  constexpr int EW_UNROLL = 4;
  for (int64_t i0 = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i0 < n2; i0 += EW_UNROLL * stride) {
    for (int k = 0; k < B.nops; k++) {
      const phip_ctx_s::phip_ew_op &o = B.op[k];
      double2 xv[EW_UNROLL], yv[EW_UNROLL], zv[EW_UNROLL], r[EW_UNROLL];
#pragma unroll
      for (int u = 0; u < EW_UNROLL; u++) {
        const int64_t i = i0 + u * stride;
        xv[u] = yv[u] = zv[u] = make_double2(0.0, 0.0);
        if (i < n2) {
          if (o.hx) xv[u] = reinterpret_cast<const double2 *>(o.x)[i];
          if (o.hy) yv[u] = reinterpret_cast<const double2 *>(o.y)[i];
          if (o.hz) zv[u] = reinterpret_cast<const double2 *>(o.z)[i];
        }
      }
#pragma unroll
      for (int u = 0; u < EW_UNROLL; u++) {
        const int64_t i = i0 + u * stride;
        r[u].x = ew_apply_rt(o.op, o.a, o.b, o.g, xv[u].x, yv[u].x, zv[u].x);
        r[u].y = ew_apply_rt(o.op, o.a, o.b, o.g, xv[u].y, yv[u].y, zv[u].y);
        if (i < n2) reinterpret_cast<double2 *>(o.out)[i] = r[u];
      }
      if (RED && k == B.nops - 1) {
#pragma unroll
        for (int u = 0; u < EW_UNROLL; u++) {
          if (i0 + u * stride < n2) { /* entry order i0, i0 + stride, ...: the order of the one-entry-per-round loop */
            acc[0] += r[u].x * r[u].x + r[u].y * r[u].y;
            acc[1] += r[u].x * xv[u].x + r[u].y * xv[u].y;
            acc[2] += r[u].x * yv[u].x + r[u].y * yv[u].y;
          }
        }
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int32_t i = n - 1;
    for (int k = 0; k < B.nops; k++) {
      const phip_ctx_s::phip_ew_op &o = B.op[k];
      const double xv = o.hx ? o.x[i] : 0.0, yv = o.hy ? o.y[i] : 0.0, zv = o.hz ? o.z[i] : 0.0;
      const double r = ew_apply_rt(o.op, o.a, o.b, o.g, xv, yv, zv);
      o.out[i] = r;
      if (RED && k == B.nops - 1) { acc[0] += r * r; acc[1] += r * xv; acc[2] += r * yv; }
    }
  }
  if (RED) {
    block_sum<3>(acc, lds);
    __syncthreads();
    reduce_tail<3, 0>(acc, lds, ctl);
  }
}

/* The batch KSPCG produces in every iteration (cg.c:222-229), written out: VecAXPY(X,a,P); VecAXPY(R,b,W); Z = D .* R with the three
   sums.  The generic kernel above walks the operations one after the other, each waiting for its own loads; here the five operands are
   requested together, the new R goes to the product through a register, and nothing is read twice.  Same expressions (ew_apply), same
   entry order per thread, same grid: the bits of the generic kernel.  RFIRST: the product was called as Z = R .* D, else Z = D .* R.
   Z may be W (KSPCG keeps both in one work vector, cg.c:106: W = Z): a thread has read its entries of W before it stores them as Z. */
template <bool RFIRST, bool WITHX = true, int NTC = -1>
__global__ __launch_bounds__(PHIP_THREADS) void cg_update_kernel(int32_t n, double *__restrict__ x, const double *__restrict__ p, double a, double *__restrict__ r,
                                                                 const double *w, double b, double *z, const double *__restrict__ d, RedCtl ctl, int nt_arg)
{
  /* NTC >= 0: the cache hints are compile-time (the two settings in use, 0 and 7): with the hint tested per load at run time the loads
     of a thread's second entry pair waited for those of the first (s_waitcnt vmcnt(0) between them in the plain-load path) */
  const int nt = NTC >= 0 ? NTC : nt_arg;
  __shared__ double lds[12];
  double            acc[3] = {0.0, 0.0, 0.0};
  const int64_t     n2 = n >> 1;
  const int64_t     stride = (int64_t)gridDim.x * PHIP_THREADS;
  constexpr int     UN = 2;
  for (int64_t i0 = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i0 < n2; i0 += UN * stride) {
    double2 xv[UN], pv[UN], rv[UN], wv[UN], dv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t i = i0 + u * stride;
      xv[u] = pv[u] = rv[u] = wv[u] = dv[u] = make_double2(0.0, 0.0);
      if (i < n2) {
        if (WITHX) { xv[u] = ld2(x, i, nt & 1); pv[u] = ld2(p, i, nt & 1); }
        rv[u] = ld2(r, i, nt & 1); wv[u] = ld2(w, i, nt & 1); dv[u] = ld2(d, i, nt & 1);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t i = i0 + u * stride;
      if (i < n2) {
        double2 xn, rn, o;
        xn.x = ew_apply<EW_AXPY>(a, 0, 0, pv[u].x, xv[u].x, 0.0); xn.y = ew_apply<EW_AXPY>(a, 0, 0, pv[u].y, xv[u].y, 0.0);
        rn.x = ew_apply<EW_AXPY>(b, 0, 0, wv[u].x, rv[u].x, 0.0); rn.y = ew_apply<EW_AXPY>(b, 0, 0, wv[u].y, rv[u].y, 0.0);
        const double2 f = RFIRST ? rn : dv[u], g = RFIRST ? dv[u] : rn; /* Z = f .* g */
        o.x = ew_apply<EW_PMULT>(0, 0, 0, f.x, g.x, 0.0); o.y = ew_apply<EW_PMULT>(0, 0, 0, f.y, g.y, 0.0);
        if (WITHX) st2(x, i, xn, nt & 2);
        st2(r, i, rn, nt & 4);
        st2(z, i, o, nt & 8);
        acc[0] += o.x * o.x + o.y * o.y;
        acc[1] += o.x * f.x + o.y * f.y;
        acc[2] += o.x * g.x + o.y * g.y;
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int32_t i = n - 1;
    if (WITHX) x[i] = ew_apply<EW_AXPY>(a, 0, 0, p[i], x[i], 0.0);
    const double rn = ew_apply<EW_AXPY>(b, 0, 0, w[i], r[i], 0.0);
    r[i] = rn;
    const double f = RFIRST ? rn : d[i], g = RFIRST ? d[i] : rn;
    const double o = ew_apply<EW_PMULT>(0, 0, 0, f, g, 0.0);
    z[i] = o;
    acc[0] += o * o; acc[1] += o * f; acc[2] += o * g;
  }
  block_sum<3>(acc, lds);
  __syncthreads();
  reduce_tail<3, 0>(acc, lds, ctl);
}

/* The pair that KSPCG leaves when its VecAXPY(X,a,P) stays behind (see phip_vec_pointwise_mult_reduce): VecAXPY(X,a,P); VecAYPX(P,b,Z)
   (cg.c:222 and, in the next iteration, cg.c:202).  P is read once for both; same expressions and entry order as the single kernels. */
template <int NTC>
__global__ __launch_bounds__(PHIP_THREADS) void xp_update_kernel(int32_t n, double *__restrict__ x, double *__restrict__ p, double a, const double *__restrict__ z, double b, int nt_arg)
{
  const int     nt = NTC >= 0 ? NTC : nt_arg; /* as in cg_update_kernel */
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  constexpr int UN = 2;
  for (int64_t i0 = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i0 < n2; i0 += UN * stride) {
    double2 xv[UN], pv[UN], zv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t i = i0 + u * stride;
      xv[u] = pv[u] = zv[u] = make_double2(0.0, 0.0);
      if (i < n2) { xv[u] = ld2(x, i, nt & 1); pv[u] = ld2(p, i, nt & 1); zv[u] = ld2(z, i, nt & 1); }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t i = i0 + u * stride;
      if (i < n2) {
        double2 xn, pn;
        xn.x = ew_apply<EW_AXPY>(a, 0, 0, pv[u].x, xv[u].x, 0.0); xn.y = ew_apply<EW_AXPY>(a, 0, 0, pv[u].y, xv[u].y, 0.0);
        pn.x = ew_apply<EW_AYPX>(b, 0, 0, zv[u].x, pv[u].x, 0.0); pn.y = ew_apply<EW_AYPX>(b, 0, 0, zv[u].y, pv[u].y, 0.0);
        st2(x, i, xn, nt & 2); /* nobody reads X before the next iteration; P is the next product's operand */
        st2(p, i, pn, 0);
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int32_t i = n - 1;
    const double  pv = p[i];
    x[i] = ew_apply<EW_AXPY>(a, 0, 0, pv, x[i], 0.0);
    p[i] = ew_apply<EW_AYPX>(b, 0, 0, z[i], pv, 0.0);
  }
}
static bool xp_update_shape(const phip_ctx_s::phip_ew_op &o0, const phip_ctx_s::phip_ew_op &o1)
{
  if (o0.op != EW_AXPY || o1.op != EW_AYPX || o0.y != o0.out || o1.y != o1.out) return false; /* launch_ew: y = out for both */
  if (o1.out != o0.x) return false;                                                          /* the AYPX overwrites the AXPY's P */
  return o0.out != o0.x && o1.x != o0.out && o1.x != o0.x;                                    /* X, P, Z distinct */
}

/* does the batch (two pending operations + the reducing product) have the shape cg_update_kernel is written for?  The second AXPY's
   result is one factor of the product; otherwise the arrays are distinct, except that the product may overwrite W. */
static int cg_update_shape(const EwBatch &B)
{
  if (B.nops != 3) return 0;
  const phip_ctx_s::phip_ew_op &o0 = B.op[0], &o1 = B.op[1], &o2 = B.op[2];
  if (o0.op != EW_AXPY || o1.op != EW_AXPY || o2.op != EW_PMULT) return 0;
  if (o0.y != o0.out || o1.y != o1.out) return 0;                       /* launch_ew<EW_AXPY>: x = the other vector, y = out */
  const int rfirst = o2.x == o1.out ? 1 : (o2.y == o1.out ? 0 : -1);
  if (rfirst < 0) return 0;
  const double *dd = rfirst ? o2.y : o2.x;
  const double *all[6] = {o0.out, o0.x, o1.out, dd, o1.x, o2.out}; /* X P R D W Z: all distinct, except that Z may be W */
  for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) if (all[i] == all[j] && !(i == 4 && j == 5)) return 0;
  return rfirst ? 2 : 1;
}

/* ---- host launchers ---------------------------------------------------------------------------- */
#define EW_GRID(n) phip_grid_for((n) / 2 + 1, 4, 2048)
#define CHECK_CTX_Q(c, name) do { PHIP_REQUIRE((c), PHIP_ERR_ARG_NULL, name ": null context"); PHIP_NEED_DEVICE((c), name); } while (0)
/* every entry point except the queueing elementwise ones launches what is pending first */
#define CHECK_CTX(c, name) do { CHECK_CTX_Q(c, name); PHIP_FLUSH(c); } while (0)
#define CHECK_N(n, name) PHIP_REQUIRE((n) >= 0, PHIP_ERR_ARG_OUTOFRANGE, name ": negative length %d", (n))

/* Vectors of 48 MiB and more are streamed with non-temporal loads: a Krylov iteration over them has a working set of several times the
   256 MiB Infinity Cache, nothing an elementwise kernel reads is found there again, and what it would displace (the SpMV's x, the
   matrix's masks) is.  The kernels themselves take the same time; the iteration gains: KSPCG 7-point 256^3 2335 -> 2543 it/s, 27-point
   256^3 1755 -> 1845; 7-point 192^3 (57 MB vectors) 5946 -> 6010, 128^3 (cache-resident) 16.2 k -> 15.3 k, hence the limit.
   PHIP_EW_NT=0|1 overrides. */
static int g_defer_x = -1; /* phip_set_cg_defer_x; -1: take PHIP_CG_DEFER_X (default 1) at the first use */
extern "C" int phip_set_cg_defer_x(int on)
{
  g_defer_x = on ? 1 : 0;
  return 0;
}
static int g_ew_nt_mode = -2; /* phip_set_ew_nontemporal; -2: take PHIP_EW_NT (or the size rule) at the first launch */
extern "C" int phip_set_ew_nontemporal(int mode)
{
  PHIP_REQUIRE(mode >= -1 && mode <= 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_set_ew_nontemporal: %d (-1 by size, 0 never, 1 always)", mode);
  g_ew_nt_mode = mode;
  return 0;
}
static inline int ew_nt(int32_t n)
{
  if (g_ew_nt_mode == -2) g_ew_nt_mode = getenv("PHIP_EW_NT") ? atoi(getenv("PHIP_EW_NT")) : -1;
  const int env = g_ew_nt_mode;
  /* cg_update_kernel also STORES X and R past the caches (bits 1 and 2 of PHIP_EW_NT_ST; nobody reads them before the next iteration,
     several cache sizes later), but not Z, which VecAYPX reads next: KSPCG it/s with stores of none / X / X,R / X,R,Z non-temporal:
     7-point 256^3 2542 / 2654 / 2759 / 2655; 27-point 256^3 1851 / 1916 / 1973 / 1924 */
  static const int st = getenv("PHIP_EW_NT_ST") ? atoi(getenv("PHIP_EW_NT_ST")) : 3;
  const int on = env >= 0 ? env : (n >= (6 << 20)); /* 48 MiB: 7-point 192^3 (57 MB) 5933 -> 6041 it/s, 160^3 (33 MB) 10.1 k -> 9.5 k */
  return on ? (1 | (st << 1)) : 0;
}

template <int OP, bool HX, bool HY, bool HZ>
static int launch_ew_now(phip_ctx c, int32_t n, double *out, const double *x, const double *y, const double *z, double a, double b, double g)
{
  hipLaunchKernelGGL((ew_kernel<OP, HX, HY, HZ>), dim3(EW_GRID(n)), dim3(PHIP_THREADS), 0, c->stream, n, out, x, y, z, a, b, g, ew_nt(n));
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

/* one pending operation, launched by the kernel specialised for it */
static int ew_launch_single(phip_ctx c, int32_t n, const phip_ctx_s::phip_ew_op &o)
{
  switch (o.op) {
#define EW_CASE(OP, HX, HY, HZ) case OP: return launch_ew_now<OP, HX, HY, HZ>(c, n, o.out, o.x, o.y, o.z, o.a, o.b, o.g);
    EW_CASE(EW_AXPY, true, true, false) EW_CASE(EW_AYPX, true, true, false) EW_CASE(EW_XMY, true, true, false) EW_CASE(EW_WAXPY, true, true, false)
    EW_CASE(EW_WXPY, true, true, false) EW_CASE(EW_WYMX, true, true, false) EW_CASE(EW_AXPBY, true, true, false) EW_CASE(EW_SCALEX, true, false, false)
    EW_CASE(EW_AXPBYPCZ, true, true, true) EW_CASE(EW_PMULT, true, true, false) EW_CASE(EW_PDIV, true, true, false) EW_CASE(EW_SCALE, true, false, false)
    EW_CASE(EW_SET, false, false, false) EW_CASE(EW_RECIP, true, false, false)
#undef EW_CASE
  }
  PHIP_REQUIRE(0, PHIP_ERR_LIB, "ew_launch_single: unknown operation %d", o.op);
  return PHIP_ERR_LIB;
}

static int ew_flush_launch(phip_ctx c, int nops);
int phip_ew_flush(phip_ctx c)
{
  const int nops = c->ew_n;
  c->ew_n = 0;
  if (!nops) return 0;
  hipEvent_t e0 = NULL, e1 = NULL;
  if (c->prof_on) { /* bench.py: the duration of this launch inside the timed solve (kind 3) */
    int rc = phip_prof_open(c, 3, &e0, &e1);
    if (rc) return rc;
    PHIP_CHECK_HIP(hipEventRecord(e0, c->stream));
  }
  int rc = ew_flush_launch(c, nops);
  if (rc) return rc;
  if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, c->stream));
  return 0;
}
static int ew_flush_launch(phip_ctx c, int nops)
{
  if (nops == 1) return ew_launch_single(c, c->ew_len, c->ew[0]);
  if (nops == 2 && xp_update_shape(c->ew[0], c->ew[1])) {
    const int ntv = ew_nt(c->ew_len);
#define XP_LAUNCH(NTC) hipLaunchKernelGGL((xp_update_kernel<NTC>), dim3(EW_GRID(c->ew_len)), dim3(PHIP_THREADS), 0, c->stream, c->ew_len, c->ew[0].out, c->ew[1].out, c->ew[0].a, c->ew[1].x, c->ew[1].a, ntv)
    if (ntv == 0) XP_LAUNCH(0); else if (ntv == 7) XP_LAUNCH(7); else XP_LAUNCH(-1);
#undef XP_LAUNCH
    PHIP_CHECK_HIP(hipGetLastError());
    return 0;
  }
  EwBatch B;
  B.nops = nops;
  for (int k = 0; k < nops; k++) B.op[k] = c->ew[k];
  RedCtl none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL((ew_batch_kernel<false>), dim3(EW_GRID(c->ew_len)), dim3(PHIP_THREADS), 0, c->stream, c->ew_len, B, none);
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

/* two operand arrays of length n may share a batch if they are the same array or do not overlap at all */
static inline bool ew_compatible(const double *p, const double *q, int32_t n)
{
  if (!p || !q || p == q) return true;
  return p + n <= q || q + n <= p;
}
static bool ew_can_join(phip_ctx c, int32_t n, const double *out, const double *x, const double *y, const double *z)
{
  if (!c->ew_batching || !c->ew_n) return c->ew_batching != 0;
  if (c->ew_n >= PHIP_EW_BATCH || c->ew_len != n) return false;
  const double *mine[4] = {out, x, y, z};
  for (int k = 0; k < c->ew_n; k++) {
    const double *theirs[4] = {c->ew[k].out, c->ew[k].hx ? c->ew[k].x : NULL, c->ew[k].hy ? c->ew[k].y : NULL, c->ew[k].hz ? c->ew[k].z : NULL};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (!ew_compatible(mine[i], theirs[j], n)) return false;
  }
  return true;
}

template <int OP, bool HX, bool HY, bool HZ>
static int launch_ew(phip_ctx c, int32_t n, double *out, const double *x, const double *y, const double *z, double a, double b, double g)
{
  if (n == 0) return 0;
  /* vectors whose double2 accesses would be misaligned, or batching switched off: at once, after whatever is pending */
  if (!c->ew_batching) { PHIP_FLUSH(c); return launch_ew_now<OP, HX, HY, HZ>(c, n, out, x, y, z, a, b, g); }
  if (!ew_can_join(c, n, out, HX ? x : NULL, HY ? y : NULL, HZ ? z : NULL)) PHIP_FLUSH(c);
  phip_ctx_s::phip_ew_op &o = c->ew[c->ew_n++];
  o.op = OP; o.hx = HX; o.hy = HY; o.hz = HZ; o.out = out; o.x = x; o.y = y; o.z = z; o.a = a; o.b = b; o.g = g;
  c->ew_len = n;
  return 0;
}

extern "C" int phip_vec_copy(phip_ctx c, int32_t n, const double *d_x, double *d_y)
{
  CHECK_CTX(c, "phip_vec_copy"); CHECK_N(n, "phip_vec_copy");
  if (d_x != d_y && n) PHIP_CHECK_HIP(hipMemcpyAsync(d_y, d_x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

extern "C" int phip_vec_set(phip_ctx c, int32_t n, double *d_x, double alpha)
{
  CHECK_CTX_Q(c, "phip_vec_set"); CHECK_N(n, "phip_vec_set");
  if (!n) return 0;
  if (alpha == 0.0) { PHIP_FLUSH(c); PHIP_CHECK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)n, c->stream)); return 0; } /* dvec2.c:559-560 */
  return launch_ew<EW_SET, false, false, false>(c, n, d_x, NULL, NULL, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_scale(phip_ctx c, int32_t n, double *d_x, double alpha)
{
  CHECK_CTX_Q(c, "phip_vec_scale"); CHECK_N(n, "phip_vec_scale");
  if (alpha == 0.0) return phip_vec_set(c, n, d_x, 0.0); /* bvec1.c:62-63 */
  if (alpha == 1.0) return 0;                             /* bvec1.c:64 */
  return launch_ew<EW_SCALE, true, false, false>(c, n, d_x, d_x, NULL, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_axpy(phip_ctx c, int32_t n, double *d_y, double alpha, const double *d_x)
{
  CHECK_CTX_Q(c, "phip_vec_axpy"); CHECK_N(n, "phip_vec_axpy");
  if (alpha == 0.0) return 0; /* bvec1.c:86 */
  return launch_ew<EW_AXPY, true, true, false>(c, n, d_y, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_aypx(phip_ctx c, int32_t n, double *d_y, double alpha, const double *d_x)
{
  CHECK_CTX_Q(c, "phip_vec_aypx"); CHECK_N(n, "phip_vec_aypx");
  if (alpha == 0.0) return phip_vec_copy(c, n, d_x, d_y);           /* dvec2.c:652-653 */
  if (alpha == 1.0) return phip_vec_axpy(c, n, d_y, alpha, d_x);    /* :654-655 */
  if (alpha == -1.0) return launch_ew<EW_XMY, true, true, false>(c, n, d_y, d_x, d_y, NULL, 0, 0, 0); /* :656-661 */
  return launch_ew<EW_AYPX, true, true, false>(c, n, d_y, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_waxpy(phip_ctx c, int32_t n, double *d_w, double alpha, const double *d_x, const double *d_y)
{
  CHECK_CTX_Q(c, "phip_vec_waxpy"); CHECK_N(n, "phip_vec_waxpy");
  if (alpha == 1.0) return launch_ew<EW_WXPY, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
  if (alpha == -1.0) return launch_ew<EW_WYMX, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
  if (alpha == 0.0) return phip_vec_copy(c, n, d_y, d_w);
  return launch_ew<EW_WAXPY, true, true, false>(c, n, d_w, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_axpby(phip_ctx c, int32_t n, double *d_y, double a, double b, const double *d_x)
{
  CHECK_CTX_Q(c, "phip_vec_axpby"); CHECK_N(n, "phip_vec_axpby");
  if (a == 0.0) return phip_vec_scale(c, n, d_y, b);       /* bvec1.c:107-108 */
  if (b == 1.0) return phip_vec_axpy(c, n, d_y, a, d_x);   /* :109-110 */
  if (a == 1.0) return phip_vec_aypx(c, n, d_y, b, d_x);   /* :111-112 */
  if (b == 0.0) return launch_ew<EW_SCALEX, true, false, false>(c, n, d_y, d_x, NULL, NULL, a, 0, 0); /* :113-117 */
  return launch_ew<EW_AXPBY, true, true, false>(c, n, d_y, d_x, d_y, NULL, a, b, 0);
}

extern "C" int phip_vec_axpbypcz(phip_ctx c, int32_t n, double *d_z, double a, double b, double g, const double *d_x, const double *d_y)
{
  CHECK_CTX_Q(c, "phip_vec_axpbypcz"); CHECK_N(n, "phip_vec_axpbypcz");
  /* the four branches of bvec1.c:148-159 differ only in skipped multiplications by exactly 1 or 0 */
  if (g == 0.0) return launch_ew<EW_AXPBY, true, true, false>(c, n, d_z, d_x, d_y, NULL, a, b, 0);
  return launch_ew<EW_AXPBYPCZ, true, true, true>(c, n, d_z, d_x, d_y, d_z, a, b, g);
}

extern "C" int phip_vec_pointwise_mult(phip_ctx c, int32_t n, double *d_w, const double *d_x, const double *d_y)
{
  CHECK_CTX_Q(c, "phip_vec_pointwise_mult"); CHECK_N(n, "phip_vec_pointwise_mult");
  return launch_ew<EW_PMULT, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0); /* aliasing w==x / w==y is safe: pure elementwise */
}

extern "C" int phip_vec_pointwise_mult_reduce(phip_ctx c, int32_t n, double *d_w, const double *d_x, const double *d_y, int publish, phip_ticket *t)
{
  CHECK_CTX_Q(c, "phip_vec_pointwise_mult_reduce"); CHECK_N(n, "phip_vec_pointwise_mult_reduce");
  PHIP_REQUIRE(t, PHIP_ERR_ARG_NULL, "phip_vec_pointwise_mult_reduce: null ticket");
  PHIP_REQUIRE((d_w != d_x && d_w != d_y) || !n, PHIP_ERR_ARG_WRONG, "phip_vec_pointwise_mult_reduce: w must not alias x or y (use phip_vec_pointwise_mult)");
  RedCtl ctl;
  int    rc = phip_ticket_open(c, 3, publish, t, &ctl);
  if (rc) return rc;
  if (n && c->ew_n && ew_can_join(c, n, d_w, d_x, d_y, NULL)) { /* the pending operations and this one leave together */
    EwBatch B;
    B.nops = c->ew_n + 1;
    for (int k = 0; k < c->ew_n; k++) B.op[k] = c->ew[k];
    phip_ctx_s::phip_ew_op &o = B.op[c->ew_n];
    o.op = EW_PMULT; o.hx = 1; o.hy = 1; o.hz = 0; o.out = d_w; o.x = d_x; o.y = d_y; o.z = NULL; o.a = o.b = o.g = 0.0;
    c->ew_n = 0;
    static const int red_grid_env = getenv("PHIP_EW_RED_GRID") ? atoi(getenv("PHIP_EW_RED_GRID")) : 0;
    const int        red_grid = (red_grid_env > 0 && red_grid_env <= PHIP_RED_RESIDENT) ? red_grid_env : PHIP_RED_GRID;
    const dim3       grid(phip_grid_for((int64_t)n / 2 + 1, 4, red_grid));
    static const int cgk_env = getenv("PHIP_CG_UPDATE_KERNEL") ? atoi(getenv("PHIP_CG_UPDATE_KERNEL")) : 1;
    const int        shape = cgk_env ? cg_update_shape(B) : 0;
    hipEvent_t       e0 = NULL, e1 = NULL;
    if (c->prof_on) { /* bench.py: the duration of this launch inside the timed solve (kind 2) */
      int prc = phip_prof_open(c, 2, &e0, &e1);
      if (prc) return prc;
      PHIP_CHECK_HIP(hipEventRecord(e0, c->stream));
    }
    /* VecAXPY(X,a,P) has nothing to do with the product or its sums, and P is read again by the VecAYPX(P,b,Z) that follows (cg.c:202 of the
       next iteration): it stays behind and leaves with that operation (xp_update_kernel), which reads P anyway -- 8 of the
       88 bytes per row the vector kernels of an iteration move: KSPCG 27-point 256^3 2930 -> 3155 it/s, 7-point 256^3 3171 -> 3448,
       7-point 128^3 16.2 k -> 16.7 k.  Nothing reads X in between: every other entry point (and the
       device-pointer getters of the glue) launches what is pending first.  PHIP_CG_DEFER_X=0 / phip_set_cg_defer_x(0) switches it off. */
    if (g_defer_x < 0) g_defer_x = getenv("PHIP_CG_DEFER_X") ? (atoi(getenv("PHIP_CG_DEFER_X")) != 0) : 1;
    if (shape && g_defer_x) {
      const int ntv = ew_nt(n);
#define CGU_LAUNCH(RF, NTC, DD) hipLaunchKernelGGL((cg_update_kernel<RF, false, NTC>), grid, dim3(PHIP_THREADS), 0, c->stream, n, (double *)NULL, (const double *)NULL, 0.0, B.op[1].out, B.op[1].x, B.op[1].a, d_w, DD, ctl, ntv)
      if (shape == 2) { if (ntv == 0) CGU_LAUNCH(true, 0, d_y); else if (ntv == 7) CGU_LAUNCH(true, 7, d_y); else CGU_LAUNCH(true, -1, d_y); }
      else { if (ntv == 0) CGU_LAUNCH(false, 0, d_x); else if (ntv == 7) CGU_LAUNCH(false, 7, d_x); else CGU_LAUNCH(false, -1, d_x); }
#undef CGU_LAUNCH
      PHIP_CHECK_HIP(hipGetLastError());
      if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, c->stream));
      c->ew[0] = B.op[0];
      c->ew_n = 1;
      return 0;
    }
    const int ntv = ew_nt(n);
#define CGU_LAUNCH(RF, NTC, DD) hipLaunchKernelGGL((cg_update_kernel<RF, true, NTC>), grid, dim3(PHIP_THREADS), 0, c->stream, n, B.op[0].out, B.op[0].x, B.op[0].a, B.op[1].out, B.op[1].x, B.op[1].a, d_w, DD, ctl, ntv)
    if (shape == 2) { if (ntv == 0) CGU_LAUNCH(true, 0, d_y); else if (ntv == 7) CGU_LAUNCH(true, 7, d_y); else CGU_LAUNCH(true, -1, d_y); }
    else if (shape == 1) { if (ntv == 0) CGU_LAUNCH(false, 0, d_x); else if (ntv == 7) CGU_LAUNCH(false, 7, d_x); else CGU_LAUNCH(false, -1, d_x); }
#undef CGU_LAUNCH
    else
      hipLaunchKernelGGL((ew_batch_kernel<true>), grid, dim3(PHIP_THREADS), 0, c->stream, n, B, ctl);
    PHIP_CHECK_HIP(hipGetLastError());
    if (e1) PHIP_CHECK_HIP(hipEventRecord(e1, c->stream));
    return 0;
  }
  PHIP_FLUSH(c);
  hipLaunchKernelGGL(pmult_reduce_kernel, dim3(phip_grid_for((int64_t)n / 2 + 1, 4, PHIP_RED_GRID)), dim3(PHIP_THREADS), 0, c->stream, n, d_w, d_x, d_y, ctl);
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int phip_vec_pointwise_divide(phip_ctx c, int32_t n, double *d_w, const double *d_x, const double *d_y)
{
  CHECK_CTX_Q(c, "phip_vec_pointwise_divide"); CHECK_N(n, "phip_vec_pointwise_divide");
  return launch_ew<EW_PDIV, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
}

extern "C" int phip_vec_reciprocal(phip_ctx c, int32_t n, double *d_x)
{
  CHECK_CTX_Q(c, "phip_vec_reciprocal"); CHECK_N(n, "phip_vec_reciprocal");
  return launch_ew<EW_RECIP, true, false, false>(c, n, d_x, d_x, NULL, NULL, 0, 0, 0);
}

extern "C" int phip_vec_swap(phip_ctx c, int32_t n, double *d_x, double *d_y)
{
  CHECK_CTX(c, "phip_vec_swap"); CHECK_N(n, "phip_vec_swap");
  if (d_x == d_y || !n) return 0;
  hipLaunchKernelGGL(swap_kernel, dim3(phip_grid_for(n, 4, 2048)), dim3(PHIP_THREADS), 0, c->stream, n, d_x, d_y);
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int phip_jacobi_setup(phip_ctx c, int32_t n, double *d_diag, int useabs, int sqrt_variant, int *h_zeroflag)
{
  CHECK_CTX(c, "phip_jacobi_setup"); CHECK_N(n, "phip_jacobi_setup");
  PHIP_CHECK_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  if (n) {
    hipLaunchKernelGGL(jacobi_setup_kernel, dim3(phip_grid_for(n, 4, 2048)), dim3(PHIP_THREADS), 0, c->stream, n, d_diag, useabs, sqrt_variant, c->d_flag);
    PHIP_CHECK_HIP(hipGetLastError());
  }
  if (h_zeroflag) {
    PHIP_CHECK_HIP(hipMemcpyAsync(c->h_flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PHIP_CHECK_HIP(hipStreamSynchronize(c->stream));
    *h_zeroflag = c->h_flag[0];
  }
  return 0;
}

template <int NV>
static void launch_mdot(phip_ctx c, int32_t n, const double *d_x, const MVPack &P, const RedCtl &ctl, int grid)
{
  hipLaunchKernelGGL((mdot_kernel<NV>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, P, ctl);
}

static int reduce_grid(int32_t n) { return phip_grid_for((int64_t)n / 2 + 1, 4, PHIP_RED_GRID); }

/* One reduction: results into d_out[0..nres); when publish != 0 also into the pinned slot, with sequence number *seq_out */
static int reduce_launch(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_out, int publish,
                         unsigned long long *seq_out, const RedCtl *pub = NULL)
{
  /* pub: publish into that slot / sequence word (a by-product ticket) instead of the context's blocking-call slot */
  const int grid = reduce_grid(n);
  RedCtl    ctl;
  ctl.partial = c->d_partials; ctl.pflag = 0; ctl.d_out = d_out; ctl.h_ll = NULL; ctl.seq = 0; ctl.npub = 0; ctl.kofs = 0; ctl.accumulate = 0; ctl.err = c->h_err_dev;
  if (kind == 0) {
    PHIP_REQUIRE(nv >= 0 && nv <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local: nv=%d (max %d)", nv, PHIP_MAX_MVEC);
    PHIP_REQUIRE(h_dy || !nv, PHIP_ERR_ARG_NULL, "phip_vec_reduce_local: null vector list");
    /* balanced groups of at most MV_MAX vectors: x is read once per group */
    const int ngroups = (nv + MV_MAX - 1) / MV_MAX;
    for (int gi = 0, k0 = 0; gi < ngroups; gi++) {
      const int g = (nv - k0 + (ngroups - gi) - 1) / (ngroups - gi);
      MVPack    P;
      for (int k = 0; k < MV_MAX; k++) { P.y[k] = (k < g) ? h_dy[k0 + k] : d_x; P.a[k] = 0.0; }
      ctl.kofs = k0;
      ctl.pflag = phip_next_pflag(c); /* every launch marks its partials with a flag of its own */
      if (publish && k0 + g == nv) {
        if (pub) { ctl.h_ll = pub->h_ll; ctl.seq = pub->seq; ctl.npub = nv; }
        else { ctl.h_ll = c->h_ll_dev; ctl.seq = phip_next_seq(&c->seq); ctl.npub = nv; *seq_out = ctl.seq; }
      }
      switch (g) {
#define MDOT_CASE(N) case N: launch_mdot<N>(c, n, d_x, P, ctl, grid); break;
        MDOT_CASE(1) MDOT_CASE(2) MDOT_CASE(3) MDOT_CASE(4) MDOT_CASE(5) MDOT_CASE(6) MDOT_CASE(7) MDOT_CASE(8)
        MDOT_CASE(9) MDOT_CASE(10) MDOT_CASE(11) MDOT_CASE(12) MDOT_CASE(13) MDOT_CASE(14) MDOT_CASE(15)
#undef MDOT_CASE
      default: launch_mdot<16>(c, n, d_x, P, ctl, grid); break;
      }
      k0 += g;
      PHIP_CHECK_HIP(hipGetLastError());
    }
    return 0;
  }
  if (publish) {
    if (pub) { ctl.h_ll = pub->h_ll; ctl.seq = pub->seq; ctl.npub = (kind == 4) ? 2 : 1; }
    else { ctl.h_ll = c->h_ll_dev; ctl.seq = phip_next_seq(&c->seq); ctl.npub = (kind == 4) ? 2 : 1; *seq_out = ctl.seq; }
  }
  ctl.pflag = phip_next_pflag(c);
  switch (kind) {
  case 1: hipLaunchKernelGGL((norm_kernel<1>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, ctl); break;
  case 2: hipLaunchKernelGGL((norm_kernel<2>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, ctl); break;
  case 3: hipLaunchKernelGGL((norm_kernel<3>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, ctl); break;
  case 4: hipLaunchKernelGGL((norm_kernel<4>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, ctl); break;
  default: PHIP_REQUIRE(0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local: kind %d", kind);
  }
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int phip_vec_reduce_local(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_out)
{
  CHECK_CTX(c, "phip_vec_reduce_local"); CHECK_N(n, "phip_vec_reduce_local");
  PHIP_REQUIRE(d_out, PHIP_ERR_ARG_NULL, "phip_vec_reduce_local: null output");
  return reduce_launch(c, kind, n, d_x, nv, h_dy, d_out, 0, NULL);
}

/* launch + wait for the published result; r receives nres values */
static int reduce_to_host(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, int nres, double *r)
{
  unsigned long long seq = 0;
  int                rc = reduce_launch(c, kind, n, d_x, nv, h_dy, c->d_result, 1, &seq);
  if (rc) return rc;
  return phip_wait_ll(c, c->h_ll, nres, seq, r, "reduction");
}

/* the local part of a distributed reduction, delivered to the HOST (one launch, pinned slot, spin): raw sums, no square root, so that
   phip_comm_allreduce_host can combine the ranks' values.  kind as in phip_vec_reduce_local. */
extern "C" int phip_vec_reduce_local_host(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *h_out)
{
  CHECK_CTX(c, "phip_vec_reduce_local_host"); CHECK_N(n, "phip_vec_reduce_local_host");
  PHIP_REQUIRE(h_out, PHIP_ERR_ARG_NULL, "phip_vec_reduce_local_host: null output");
  PHIP_REQUIRE(kind >= 0 && kind <= 4, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local_host: kind %d", kind);
  if (kind == 0) {
    PHIP_REQUIRE(nv >= 0 && nv <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local_host: nv=%d (max %d)", nv, PHIP_MAX_MVEC);
    if (nv == 0) return 0;
  }
  return reduce_to_host(c, kind, n, d_x, nv, h_dy, kind == 0 ? nv : (kind == 4 ? 2 : 1), h_out);
}

/* The local part of a reduction launched WITHOUT waiting (VecDotBegin / VecNormBegin / VecMDotBegin, src/vec/vec/utils/comb.c): the
   raw sums land in a by-product ticket; phip_ticket_wait fetches them later, after other work has been queued behind the launch. */
extern "C" int phip_vec_reduce_begin(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, phip_ticket *t)
{
  CHECK_CTX(c, "phip_vec_reduce_begin"); CHECK_N(n, "phip_vec_reduce_begin");
  PHIP_REQUIRE(t, PHIP_ERR_ARG_NULL, "phip_vec_reduce_begin: null ticket");
  PHIP_REQUIRE(kind >= 0 && kind <= 4, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_begin: kind %d", kind);
  const int nres = kind == 0 ? nv : (kind == 4 ? 2 : 1);
  PHIP_REQUIRE(nres >= 1 && nres < PHIP_RING_VALS, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_begin: %d results (1..%d per call)", nres, PHIP_RING_VALS - 1);
  RedCtl pub;
  int    rc = phip_ticket_open(c, nres, 1, t, &pub);
  if (rc) return rc;
  return reduce_launch(c, kind, n, d_x, nv, h_dy, pub.d_out, 1, NULL, &pub);
}

extern "C" int phip_vec_dot(phip_ctx c, int32_t n, const double *d_x, const double *d_y, double *h_z)
{
  CHECK_CTX(c, "phip_vec_dot"); CHECK_N(n, "phip_vec_dot");
  PHIP_REQUIRE(h_z, PHIP_ERR_ARG_NULL, "phip_vec_dot: null result");
  const double *ys[1] = {d_y};
  return reduce_to_host(c, 0, n, d_x, 1, ys, 1, h_z);
}

extern "C" int phip_vec_mdot(phip_ctx c, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *h_z)
{
  CHECK_CTX(c, "phip_vec_mdot"); CHECK_N(n, "phip_vec_mdot");
  PHIP_REQUIRE(h_z || !nv, PHIP_ERR_ARG_NULL, "phip_vec_mdot: null result");
  PHIP_REQUIRE(nv >= 0 && nv <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_mdot: nv=%d (max %d)", nv, PHIP_MAX_MVEC);
  if (nv == 0) return 0;
  return reduce_to_host(c, 0, n, d_x, nv, h_dy, nv, h_z);
}

extern "C" int phip_vec_norm(phip_ctx c, int32_t n, const double *d_x, int type, double *h_z)
{
  CHECK_CTX(c, "phip_vec_norm"); CHECK_N(n, "phip_vec_norm");
  PHIP_REQUIRE(h_z, PHIP_ERR_ARG_NULL, "phip_vec_norm: null result");
  int    rc;
  double r[2];
  switch (type) {
  case PHIP_NORM_2:
  case PHIP_NORM_FROBENIUS:
    if ((rc = reduce_to_host(c, 1, n, d_x, 1, NULL, 1, r))) return rc;
    h_z[0] = sqrt(r[0]); /* bvec2.c:244-245 */
    return 0;
  case PHIP_NORM_1: return reduce_to_host(c, 2, n, d_x, 1, NULL, 1, h_z);
  case PHIP_NORM_INFINITY: return reduce_to_host(c, 3, n, d_x, 1, NULL, 1, h_z);
  case PHIP_NORM_1_AND_2:
    if ((rc = reduce_to_host(c, 4, n, d_x, 1, NULL, 2, r))) return rc;
    h_z[0] = r[0]; h_z[1] = sqrt(r[1]);
    return 0;
  }
  PHIP_REQUIRE(0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_norm: unknown NormType %d", type);
  return PHIP_ERR_ARG_OUTOFRANGE;
}

template <int NV>
static void launch_maxpy(phip_ctx c, int32_t n, double *d_x, const MVPack &P, const RedCtl *ctl)
{
  /* one entry pair per thread and a large grid for the long lists (maxpy16 at 192^3: 2048 workgroups 198 us, 16384 187 us) */
  const int grid = NV >= MV_NT_MIN ? phip_grid_for((int64_t)n / 2 + 1, 1, 16384) : EW_GRID(n);
  if (ctl) hipLaunchKernelGGL((maxpy_kernel<NV, true>), dim3(grid < PHIP_RED_BLOCKS ? grid : PHIP_RED_BLOCKS), dim3(PHIP_THREADS), 0, c->stream, n, d_x, P, *ctl);
  else hipLaunchKernelGGL((maxpy_kernel<NV, false>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, P, RedCtl());
}

/* the launches of one VecMAXPY; with ctl the last of them carries the by-product */
static int maxpy_groups(phip_ctx c, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy, const RedCtl *ctl)
{
  const int ngroups = (nv + MV_MAX - 1) / MV_MAX; /* balanced groups: x is read and written once per group */
  for (int gi = 0, k0 = 0; gi < ngroups; gi++) {
    const int g = (nv - k0 + (ngroups - gi) - 1) / (ngroups - gi);
    MVPack    P;
    for (int k = 0; k < MV_MAX; k++) { P.y[k] = (k < g) ? h_dy[k0 + k] : d_x; P.a[k] = (k < g) ? h_alpha[k0 + k] : 0.0; }
    const RedCtl *cl = (gi == ngroups - 1) ? ctl : NULL;
    switch (g) {
#define MAXPY_CASE(N) case N: launch_maxpy<N>(c, n, d_x, P, cl); break;
      MAXPY_CASE(1) MAXPY_CASE(2) MAXPY_CASE(3) MAXPY_CASE(4) MAXPY_CASE(5) MAXPY_CASE(6) MAXPY_CASE(7) MAXPY_CASE(8)
      MAXPY_CASE(9) MAXPY_CASE(10) MAXPY_CASE(11) MAXPY_CASE(12) MAXPY_CASE(13) MAXPY_CASE(14) MAXPY_CASE(15)
#undef MAXPY_CASE
    default: launch_maxpy<16>(c, n, d_x, P, cl); break;
    }
    k0 += g;
    PHIP_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int phip_vec_maxpy(phip_ctx c, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy)
{
  CHECK_CTX(c, "phip_vec_maxpy"); CHECK_N(n, "phip_vec_maxpy");
  PHIP_REQUIRE(nv >= 0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_maxpy: nv=%d", nv);
  PHIP_REQUIRE((h_alpha && h_dy) || !nv, PHIP_ERR_ARG_NULL, "phip_vec_maxpy: null argument");
  if (!n || !nv) return 0;
  return maxpy_groups(c, n, d_x, nv, h_alpha, h_dy, NULL);
}

/* VecMAXPY with the by-product sum x_new^2 in a ticket (the local sum: ranks combine it as they do for phip_vec_reduce_begin) */
extern "C" int phip_vec_maxpy_reduce(phip_ctx c, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy, int publish, phip_ticket *t)
{
  CHECK_CTX(c, "phip_vec_maxpy_reduce"); CHECK_N(n, "phip_vec_maxpy_reduce");
  PHIP_REQUIRE(nv >= 1, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_maxpy_reduce: nv=%d (at least one vector)", nv);
  PHIP_REQUIRE(h_alpha && h_dy, PHIP_ERR_ARG_NULL, "phip_vec_maxpy_reduce: null argument");
  PHIP_REQUIRE(t, PHIP_ERR_ARG_NULL, "phip_vec_maxpy_reduce: null ticket");
  RedCtl ctl;
  int    rc = phip_ticket_open(c, 1, publish, t, &ctl);
  if (rc) return rc;
  return maxpy_groups(c, n, d_x, nv, h_alpha, h_dy, &ctl); /* n = 0 included: one workgroup publishes the zero sum */
}

/* ---- the entry point that allocates nothing (include/petschip.h: phip_vec_mdot_static) -------------------------------------------
   phip_vec_reduce_local's kind 0 with the partial rows of the reduction in a file-scope __device__ array instead of the context's
   hipMalloc'ed rows, for callers that cannot allocate and have no context: code under HIP graph capture, serving loops, harnesses that
   forbid hipMalloc.  Same kernel, same grid, same order of the sums: the bits of phip_vec_reduce_local.
   `scratch` is a host-side stand-in in host code, not a device address: hipGetSymbolAddress gives each GPU's own copy, once per device. */
#define PHIP_STATIC_NV MV_MAX
#define PHIP_STATIC_WORDS (2 * (size_t)PHIP_STATIC_NV * PHIP_RED_BLOCKS) /* 1 MiB of partial rows ... */
static __device__ __align__(256) unsigned long long scratch[PHIP_STATIC_WORDS + 32]; /* ... + one 256-byte line for the arrival counter */

#include <mutex>
#define PHIP_STATIC_MAXDEV 64
static unsigned long long *g_scratch_ptr[PHIP_STATIC_MAXDEV];
static std::mutex          g_scratch_mutex;

/* once per device, with that device current (phip_init calls it; phip_vec_mdot_static calls it on a device that never saw phip_init) */
int phip_static_scratch_setup(unsigned long long **ptr)
{
  int dev = -1;
  PHIP_CHECK_HIP(hipGetDevice(&dev));
  PHIP_REQUIRE(dev >= 0 && dev < PHIP_STATIC_MAXDEV, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_mdot_static: device %d (the table holds %d)", dev, PHIP_STATIC_MAXDEV);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void *p = NULL;
    PHIP_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    PHIP_REQUIRE(p, PHIP_ERR_LIB, "phip_vec_mdot_static: hipGetSymbolAddress returned no address on device %d", dev);
    g_scratch_ptr[dev] = (unsigned long long *)p; /* the array is zero-initialised by the loader: the arrival counter starts at 0 */
  }
  if (ptr) *ptr = g_scratch_ptr[dev];
  return 0;
}

template <int NV>
static void launch_mdot_static(hipStream_t st, int32_t n, const double *d_x, const MVPack &P, const RedCtl &ctl, int grid)
{
  hipLaunchKernelGGL((mdot_kernel<NV, true>), dim3(grid), dim3(PHIP_THREADS), 0, st, n, d_x, P, ctl);
}

extern "C" int phip_vec_mdot_static(void *stream, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_z)
{
  CHECK_N(n, "phip_vec_mdot_static");
  PHIP_REQUIRE(nv >= 1 && nv <= PHIP_STATIC_NV, PHIP_ERR_ARG_OUTOFRANGE,
               "phip_vec_mdot_static: nv=%d, this entry point takes 1..%d vectors per call (its partial rows are a fixed %zu-byte device array); split the list or use phip_vec_reduce_local",
               nv, PHIP_STATIC_NV, sizeof(unsigned long long) * PHIP_STATIC_WORDS);
  PHIP_REQUIRE(h_dy && d_z && (d_x || !n), PHIP_ERR_ARG_NULL, "phip_vec_mdot_static: null argument");
  unsigned long long *ptr = NULL;
  int                 rc = phip_static_scratch_setup(&ptr);
  if (rc) return rc;
  const int grid = reduce_grid(n); /* <= PHIP_RED_GRID <= PHIP_RED_BLOCKS: the rows hold one partial per workgroup */
  MVPack    P;
  for (int k = 0; k < MV_MAX; k++) { P.y[k] = (k < nv) ? h_dy[k] : d_x; P.a[k] = 0.0; }
  RedCtl ctl;
  memset(&ctl, 0, sizeof(ctl));
  ctl.partial = ptr;                      /* the device address, never the array's name */
  ctl.h_ll = ptr + PHIP_STATIC_WORDS;     /* the arrival counter: 0 between launches (the last workgroup puts it back) */
  ctl.d_out = d_z;
  const hipStream_t st = (hipStream_t)stream;
  switch (nv) {
#define MDOT_CASE(N) case N: launch_mdot_static<N>(st, n, d_x, P, ctl, grid); break;
    MDOT_CASE(1) MDOT_CASE(2) MDOT_CASE(3) MDOT_CASE(4) MDOT_CASE(5) MDOT_CASE(6) MDOT_CASE(7) MDOT_CASE(8)
    MDOT_CASE(9) MDOT_CASE(10) MDOT_CASE(11) MDOT_CASE(12) MDOT_CASE(13) MDOT_CASE(14) MDOT_CASE(15)
#undef MDOT_CASE
  default: launch_mdot_static<16>(st, n, d_x, P, ctl, grid); break;
  }
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
