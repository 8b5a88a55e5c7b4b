// Krylov vector kernels of libpetschip (see include/petschip.h): the _VecOps slots of VECSEQHIP / VECMPIHIP.
//
// Every kernel is HBM-bound: grid-stride over 16-byte (double2) accesses, tail handled by scalar code.
// Algorithmic bytes: dot 16n, norm 8n, axpy/aypx/pointwise 24n, waxpy 24n, scale 16n, copy 16n, set 8n,
// mdot(nv) 8n(nv+1), maxpy(nv) 8n(nv+2).
// Reductions: fixed grid of PHIP_RED_BLOCKS workgroups; wavefront (64-lane) shuffle reduction -> LDS across the 4 waves ->
// one partial per workgroup; a second single-workgroup kernel sums the partials in a fixed order (deterministic for a given n).
#include "phip_internal.h"

#define MV_MAX 8 /* vectors fused per mdot/maxpy launch */

struct MVPack {
  const double *y[MV_MAX];
  double        a[MV_MAX];
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

/* block-level sum of NV values; result valid in thread 0 */
template <int NV>
__device__ __forceinline__ void block_sum(double (&acc)[NV], double *lds /* [4*NV] */)
{
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; k++) acc[k] = wave_sum(acc[k]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) lds[(t >> 6) * NV + k] = acc[k];
  }
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = (lds[k] + lds[NV + k]) + (lds[2 * NV + k] + lds[3 * NV + k]);
  }
}

/* z[k] = sum_i x_i * y_k,i  for k < NV : x is read once per launch.  partial layout: [k][block] */
template <int NV>
__global__ __launch_bounds__(PHIP_THREADS) void mdot_kernel(int32_t n, const double *__restrict__ x, MVPack P, double *__restrict__ partial, int kofs)
{
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
      const double2 yv = reinterpret_cast<const double2 *>(P.y[k])[i];
      acc[k] += xv.x * yv.x + xv.y * yv.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] += x[n - 1] * P.y[k][n - 1];
  }
  block_sum<NV>(acc, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) partial[(size_t)(kofs + k) * PHIP_RED_BLOCKS + blockIdx.x] = acc[k];
  }
}

/* kind 1: sum x^2 ; 2: sum |x| ; 3: max |x| with NaN propagation (bvec2.c:253-258) ; 4: {sum |x|, sum x^2} */
template <int KIND>
__global__ __launch_bounds__(PHIP_THREADS) void norm_kernel(int32_t n, const double *__restrict__ x, double *__restrict__ partial)
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
      m = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
      f = fmax(fmax(lds[4], lds[5]), fmax(lds[6], lds[7]));
      partial[blockIdx.x] = m;
      partial[PHIP_RED_BLOCKS + blockIdx.x] = f;
    }
  } else {
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
      partial[blockIdx.x] = acc[0];
      partial[PHIP_RED_BLOCKS + blockIdx.x] = acc[1];
    }
  }
}

/* final pass: out[k] = combine(partial[k][0..nb)) ; mode 0 sum, 1 max-with-nan-flag (k=0 value, k=1 flag) */
__global__ __launch_bounds__(PHIP_THREADS) void finalize_kernel(int nb, int nv, const double *__restrict__ partial, double *__restrict__ out, int mode)
{
  __shared__ double lds[8];
  if (mode == 1) {
    double m = 0.0, f = 0.0;
    for (int i = threadIdx.x; i < nb; i += PHIP_THREADS) { m = fmax(m, partial[i]); f = fmax(f, partial[PHIP_RED_BLOCKS + i]); }
    m = wave_max(m); f = wave_max(f);
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = m; lds[4 + (threadIdx.x >> 6)] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
      f = fmax(fmax(lds[4], lds[5]), fmax(lds[6], lds[7]));
      out[0] = (f > 0.0) ? __builtin_nan("") : m;
    }
    return;
  }
  for (int k = 0; k < nv; k++) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += PHIP_THREADS) s += partial[(size_t)k * PHIP_RED_BLOCKS + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    __syncthreads();
  }
}

/* x += sum_k a_k y_k : x read and written once per launch */
template <int NV>
__global__ __launch_bounds__(PHIP_THREADS) void maxpy_kernel(int32_t n, double *__restrict__ x, MVPack P)
{
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    double2 xv = reinterpret_cast<double2 *>(x)[i];
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const double2 yv = reinterpret_cast<const double2 *>(P.y[k])[i];
      xv.x += P.a[k] * yv.x;
      xv.y += P.a[k] * yv.y;
    }
    reinterpret_cast<double2 *>(x)[i] = xv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v = x[n - 1];
#pragma unroll
    for (int k = 0; k < NV; k++) v += P.a[k] * P.y[k][n - 1];
    x[n - 1] = v;
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

/* out = f(x, y, z); any of x,y,z may alias out (pure elementwise) */
template <int OP, bool HX, bool HY, bool HZ>
__global__ __launch_bounds__(PHIP_THREADS) void ew_kernel(int32_t n, double *out, const double *x, const double *y, const double *z, double a, double b, double g)
{
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * PHIP_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * PHIP_THREADS + threadIdx.x; i < n2; i += stride) {
    double2 xv = {0, 0}, yv = {0, 0}, zv = {0, 0}, o;
    if (HX) xv = reinterpret_cast<const double2 *>(x)[i];
    if (HY) yv = reinterpret_cast<const double2 *>(y)[i];
    if (HZ) zv = reinterpret_cast<const double2 *>(z)[i];
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

/* ---- host launchers ---------------------------------------------------------------------------- */
#define EW_GRID(n) phip_grid_for((n) / 2 + 1, 4, 2048)
#define CHECK_CTX(c, name) do { PHIP_REQUIRE((c), PHIP_ERR_ARG_NULL, name ": null context"); PHIP_NEED_DEVICE((c), name); } while (0)
#define CHECK_N(n, name) PHIP_REQUIRE((n) >= 0, PHIP_ERR_ARG_OUTOFRANGE, name ": negative length %d", (n))

template <int OP, bool HX, bool HY, bool HZ>
static int launch_ew(phip_ctx c, int32_t n, double *out, const double *x, const double *y, const double *z, double a, double b, double g)
{
  if (n == 0) return 0;
  hipLaunchKernelGGL((ew_kernel<OP, HX, HY, HZ>), dim3(EW_GRID(n)), dim3(PHIP_THREADS), 0, c->stream, n, out, x, y, z, a, b, g);
  PHIP_CHECK_HIP(hipGetLastError());
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
  CHECK_CTX(c, "phip_vec_set"); CHECK_N(n, "phip_vec_set");
  if (!n) return 0;
  if (alpha == 0.0) { PHIP_CHECK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)n, c->stream)); return 0; } /* dvec2.c:559-560 */
  return launch_ew<EW_SET, false, false, false>(c, n, d_x, NULL, NULL, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_scale(phip_ctx c, int32_t n, double *d_x, double alpha)
{
  CHECK_CTX(c, "phip_vec_scale"); CHECK_N(n, "phip_vec_scale");
  if (alpha == 0.0) return phip_vec_set(c, n, d_x, 0.0); /* bvec1.c:62-63 */
  if (alpha == 1.0) return 0;                             /* bvec1.c:64 */
  return launch_ew<EW_SCALE, true, false, false>(c, n, d_x, d_x, NULL, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_axpy(phip_ctx c, int32_t n, double *d_y, double alpha, const double *d_x)
{
  CHECK_CTX(c, "phip_vec_axpy"); CHECK_N(n, "phip_vec_axpy");
  if (alpha == 0.0) return 0; /* bvec1.c:86 */
  return launch_ew<EW_AXPY, true, true, false>(c, n, d_y, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_aypx(phip_ctx c, int32_t n, double *d_y, double alpha, const double *d_x)
{
  CHECK_CTX(c, "phip_vec_aypx"); CHECK_N(n, "phip_vec_aypx");
  if (alpha == 0.0) return phip_vec_copy(c, n, d_x, d_y);           /* dvec2.c:652-653 */
  if (alpha == 1.0) return phip_vec_axpy(c, n, d_y, alpha, d_x);    /* :654-655 */
  if (alpha == -1.0) return launch_ew<EW_XMY, true, true, false>(c, n, d_y, d_x, d_y, NULL, 0, 0, 0); /* :656-661 */
  return launch_ew<EW_AYPX, true, true, false>(c, n, d_y, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_waxpy(phip_ctx c, int32_t n, double *d_w, double alpha, const double *d_x, const double *d_y)
{
  CHECK_CTX(c, "phip_vec_waxpy"); CHECK_N(n, "phip_vec_waxpy");
  if (alpha == 1.0) return launch_ew<EW_WXPY, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
  if (alpha == -1.0) return launch_ew<EW_WYMX, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
  if (alpha == 0.0) return phip_vec_copy(c, n, d_y, d_w);
  return launch_ew<EW_WAXPY, true, true, false>(c, n, d_w, d_x, d_y, NULL, alpha, 0, 0);
}

extern "C" int phip_vec_axpby(phip_ctx c, int32_t n, double *d_y, double a, double b, const double *d_x)
{
  CHECK_CTX(c, "phip_vec_axpby"); CHECK_N(n, "phip_vec_axpby");
  if (a == 0.0) return phip_vec_scale(c, n, d_y, b);       /* bvec1.c:107-108 */
  if (b == 1.0) return phip_vec_axpy(c, n, d_y, a, d_x);   /* :109-110 */
  if (a == 1.0) return phip_vec_aypx(c, n, d_y, b, d_x);   /* :111-112 */
  if (b == 0.0) return launch_ew<EW_SCALEX, true, false, false>(c, n, d_y, d_x, NULL, NULL, a, 0, 0); /* :113-117 */
  return launch_ew<EW_AXPBY, true, true, false>(c, n, d_y, d_x, d_y, NULL, a, b, 0);
}

extern "C" int phip_vec_axpbypcz(phip_ctx c, int32_t n, double *d_z, double a, double b, double g, const double *d_x, const double *d_y)
{
  CHECK_CTX(c, "phip_vec_axpbypcz"); CHECK_N(n, "phip_vec_axpbypcz");
  /* the four branches of bvec1.c:148-159 differ only in skipped multiplications by exactly 1 or 0 */
  if (g == 0.0) return launch_ew<EW_AXPBY, true, true, false>(c, n, d_z, d_x, d_y, NULL, a, b, 0);
  return launch_ew<EW_AXPBYPCZ, true, true, true>(c, n, d_z, d_x, d_y, d_z, a, b, g);
}

extern "C" int phip_vec_pointwise_mult(phip_ctx c, int32_t n, double *d_w, const double *d_x, const double *d_y)
{
  CHECK_CTX(c, "phip_vec_pointwise_mult"); CHECK_N(n, "phip_vec_pointwise_mult");
  return launch_ew<EW_PMULT, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0); /* aliasing w==x / w==y is safe: pure elementwise */
}

extern "C" int phip_vec_pointwise_divide(phip_ctx c, int32_t n, double *d_w, const double *d_x, const double *d_y)
{
  CHECK_CTX(c, "phip_vec_pointwise_divide"); CHECK_N(n, "phip_vec_pointwise_divide");
  return launch_ew<EW_PDIV, true, true, false>(c, n, d_w, d_x, d_y, NULL, 0, 0, 0);
}

extern "C" int phip_vec_reciprocal(phip_ctx c, int32_t n, double *d_x)
{
  CHECK_CTX(c, "phip_vec_reciprocal"); CHECK_N(n, "phip_vec_reciprocal");
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
static void launch_mdot(phip_ctx c, int32_t n, const double *d_x, const MVPack &P, int kofs, int grid)
{
  hipLaunchKernelGGL((mdot_kernel<NV>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, P, c->d_partials, kofs);
}

static int reduce_grid(int32_t n) { return phip_grid_for((int64_t)n / 2 + 1, 4, PHIP_RED_BLOCKS); }

extern "C" int phip_vec_reduce_local(phip_ctx c, int kind, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *d_out)
{
  CHECK_CTX(c, "phip_vec_reduce_local"); CHECK_N(n, "phip_vec_reduce_local");
  PHIP_REQUIRE(d_out, PHIP_ERR_ARG_NULL, "phip_vec_reduce_local: null output");
  const int grid = reduce_grid(n);
  if (kind == 0) {
    PHIP_REQUIRE(nv >= 0 && nv <= PHIP_MAX_MVEC, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local: nv=%d (max %d)", nv, PHIP_MAX_MVEC);
    PHIP_REQUIRE(h_dy || !nv, PHIP_ERR_ARG_NULL, "phip_vec_reduce_local: null vector list");
    for (int k0 = 0; k0 < nv; k0 += MV_MAX) {
      const int g = (nv - k0 < MV_MAX) ? nv - k0 : MV_MAX;
      MVPack    P;
      for (int k = 0; k < MV_MAX; k++) { P.y[k] = (k < g) ? h_dy[k0 + k] : d_x; P.a[k] = 0.0; }
      switch (g) {
      case 1: launch_mdot<1>(c, n, d_x, P, k0, grid); break;
      case 2: launch_mdot<2>(c, n, d_x, P, k0, grid); break;
      case 3: launch_mdot<3>(c, n, d_x, P, k0, grid); break;
      case 4: launch_mdot<4>(c, n, d_x, P, k0, grid); break;
      case 5: launch_mdot<5>(c, n, d_x, P, k0, grid); break;
      case 6: launch_mdot<6>(c, n, d_x, P, k0, grid); break;
      case 7: launch_mdot<7>(c, n, d_x, P, k0, grid); break;
      default: launch_mdot<8>(c, n, d_x, P, k0, grid); break;
      }
      PHIP_CHECK_HIP(hipGetLastError());
    }
    if (nv) {
      hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(PHIP_THREADS), 0, c->stream, grid, nv, c->d_partials, d_out, 0);
      PHIP_CHECK_HIP(hipGetLastError());
    }
    return 0;
  }
  switch (kind) {
  case 1: hipLaunchKernelGGL((norm_kernel<1>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, c->d_partials); break;
  case 2: hipLaunchKernelGGL((norm_kernel<2>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, c->d_partials); break;
  case 3: hipLaunchKernelGGL((norm_kernel<3>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, c->d_partials); break;
  case 4: hipLaunchKernelGGL((norm_kernel<4>), dim3(grid), dim3(PHIP_THREADS), 0, c->stream, n, d_x, c->d_partials); break;
  default: PHIP_REQUIRE(0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_reduce_local: kind %d", kind);
  }
  PHIP_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(PHIP_THREADS), 0, c->stream, grid, (kind == 4) ? 2 : 1, c->d_partials, d_out, (kind == 3) ? 1 : 0);
  PHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int phip_vec_dot(phip_ctx c, int32_t n, const double *d_x, const double *d_y, double *h_z)
{
  CHECK_CTX(c, "phip_vec_dot");
  PHIP_REQUIRE(h_z, PHIP_ERR_ARG_NULL, "phip_vec_dot: null result");
  const double *ys[1] = {d_y};
  int           rc = phip_vec_reduce_local(c, 0, n, d_x, 1, ys, c->d_result);
  if (rc) return rc;
  return phip_read_scalars(c, c->d_result, 1, h_z);
}

extern "C" int phip_vec_mdot(phip_ctx c, int32_t n, const double *d_x, int32_t nv, const double *const *h_dy, double *h_z)
{
  CHECK_CTX(c, "phip_vec_mdot");
  PHIP_REQUIRE(h_z || !nv, PHIP_ERR_ARG_NULL, "phip_vec_mdot: null result");
  if (nv == 0) return 0;
  int rc = phip_vec_reduce_local(c, 0, n, d_x, nv, h_dy, c->d_result);
  if (rc) return rc;
  return phip_read_scalars(c, c->d_result, nv, h_z);
}

extern "C" int phip_vec_norm(phip_ctx c, int32_t n, const double *d_x, int type, double *h_z)
{
  CHECK_CTX(c, "phip_vec_norm");
  PHIP_REQUIRE(h_z, PHIP_ERR_ARG_NULL, "phip_vec_norm: null result");
  int    rc;
  double r[2];
  switch (type) {
  case PHIP_NORM_2:
  case PHIP_NORM_FROBENIUS:
    if ((rc = phip_vec_reduce_local(c, 1, n, d_x, 1, NULL, c->d_result))) return rc;
    if ((rc = phip_read_scalars(c, c->d_result, 1, r))) return rc;
    h_z[0] = sqrt(r[0]); /* bvec2.c:244-245 */
    return 0;
  case PHIP_NORM_1:
    if ((rc = phip_vec_reduce_local(c, 2, n, d_x, 1, NULL, c->d_result))) return rc;
    return phip_read_scalars(c, c->d_result, 1, h_z);
  case PHIP_NORM_INFINITY:
    if ((rc = phip_vec_reduce_local(c, 3, n, d_x, 1, NULL, c->d_result))) return rc;
    return phip_read_scalars(c, c->d_result, 1, h_z);
  case PHIP_NORM_1_AND_2:
    if ((rc = phip_vec_reduce_local(c, 4, n, d_x, 1, NULL, c->d_result))) return rc;
    if ((rc = phip_read_scalars(c, c->d_result, 2, r))) return rc;
    h_z[0] = r[0]; h_z[1] = sqrt(r[1]);
    return 0;
  }
  PHIP_REQUIRE(0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_norm: unknown NormType %d", type);
  return PHIP_ERR_ARG_OUTOFRANGE;
}

template <int NV>
static void launch_maxpy(phip_ctx c, int32_t n, double *d_x, const MVPack &P)
{
  hipLaunchKernelGGL((maxpy_kernel<NV>), dim3(EW_GRID(n)), dim3(PHIP_THREADS), 0, c->stream, n, d_x, P);
}

extern "C" int phip_vec_maxpy(phip_ctx c, int32_t n, double *d_x, int32_t nv, const double *h_alpha, const double *const *h_dy)
{
  CHECK_CTX(c, "phip_vec_maxpy"); CHECK_N(n, "phip_vec_maxpy");
  PHIP_REQUIRE(nv >= 0, PHIP_ERR_ARG_OUTOFRANGE, "phip_vec_maxpy: nv=%d", nv);
  PHIP_REQUIRE((h_alpha && h_dy) || !nv, PHIP_ERR_ARG_NULL, "phip_vec_maxpy: null argument");
  if (!n) return 0;
  for (int k0 = 0; k0 < nv; k0 += MV_MAX) {
    const int g = (nv - k0 < MV_MAX) ? nv - k0 : MV_MAX;
    MVPack    P;
    for (int k = 0; k < MV_MAX; k++) { P.y[k] = (k < g) ? h_dy[k0 + k] : d_x; P.a[k] = (k < g) ? h_alpha[k0 + k] : 0.0; }
    switch (g) {
    case 1: launch_maxpy<1>(c, n, d_x, P); break;
    case 2: launch_maxpy<2>(c, n, d_x, P); break;
    case 3: launch_maxpy<3>(c, n, d_x, P); break;
    case 4: launch_maxpy<4>(c, n, d_x, P); break;
    case 5: launch_maxpy<5>(c, n, d_x, P); break;
    case 6: launch_maxpy<6>(c, n, d_x, P); break;
    case 7: launch_maxpy<7>(c, n, d_x, P); break;
    default: launch_maxpy<8>(c, n, d_x, P); break;
    }
    PHIP_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
