// Internal declarations shared by the libpetschip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/petschip.h"

#define PHIP_RED_BLOCKS 4096 /* capacity of the partial-result rows: the largest grid a kernel with a fused reduction may have */
#define PHIP_RED_GRID 1024   /* grid cap of the vector reduction kernels (2048 measured slower); the summation tree depends on the grid only */
#define PHIP_THREADS 256
#define PHIP_RED_RESIDENT 2048 /* workgroups of 256 threads an MI355X holds at once (256 CUs x 8): the largest grid an environment override may give a
                                  kernel whose last workgroup waits for the others' partial results (reduce_tail) */

#define PHIP_RING 64
#define PHIP_RING_VALS 8
/* Results cross to the host as FLAGGED WORDS: a double travels as two 8-byte words, each carrying 32 bits of the value and the 32-bit
   sequence flag of the call.  An aligned 8-byte store is atomic, so a word whose flag matches holds valid data: no fence, no separate
   sequence word, one PCIe write round instead of two (the low-latency protocol collective libraries use for small messages). */
struct phip_ring_slot { unsigned long long ll[2 * PHIP_RING_VALS]; }; /* 128 bytes */

struct phip_ctx_s {
  int          device;
  int          cu_count;
  hipStream_t  stream;      // compute stream: all kernels, in call order
  hipStream_t  comm_stream; // ghost exchange, overlapped with the diagonal-block SpMV
  unsigned long long *d_partials; // [PHIP_MAX_MVEC][PHIP_RED_BLOCKS][2] block partials of reductions, as flagged words
  unsigned int pseq;        // flag of the last launch's partials
  double      *d_result;    // [PHIP_MAX_MVEC] final reduction results (device)
  double      *h_result;    // pinned host mirror of d_result
  unsigned long long *h_ll;     // pinned, device-visible: flagged words [2*PHIP_MAX_MVEC] published by the last workgroup of a reduction
  unsigned long long *h_ll_dev; // device address of h_ll
  unsigned int       *h_err, *h_err_dev; // pinned error word of the reduction kernels (phip_reduce.h) and its device address
  unsigned long long  seq;  // last sequence number issued (its low 32 bits, never 0, are the flag)
  /* by-product reductions (phip_ticket): ring of result slots, device side and pinned host side */
  double      *d_ring;      // [PHIP_RING][PHIP_RING_VALS]
  struct phip_ring_slot *h_ring, *h_ring_dev; // [PHIP_RING]
  unsigned long long ring_seq; // tickets issued so far
  int         *d_flag;      // small int scratch (jacobi zero flag)
  int         *h_flag;      // pinned
  /* elementwise operations wait here until something else needs the stream: up to PHIP_EW_BATCH of them leave in ONE launch
     (phip_vec.hip, "batched elementwise operations") */
  struct phip_ew_op { int op, hx, hy, hz; double *out; const double *x, *y, *z; double a, b, g; } ew[4];
  int          ew_n;        // pending operations
  int32_t      ew_len;      // their common vector length
  int          ew_batching; // 0: every operation is launched at once (PHIP_EW_BATCH=0)
  int          prof_on;     // phip_profile_enable: hipEvent pair around every SpMV launch
  struct phip_prof_s *prof;
};

/* profiling records (phip_runtime.hip): kind 0 = MatMult launches, 1 = MatMultAdd launches */
/* wait until n flagged values with sequence number seq have arrived in ll (spin, with a periodic stream health check); copies them out */
int phip_wait_ll(phip_ctx c, const unsigned long long *ll, int n, unsigned long long seq, double *out, const char *what);
/* next sequence number whose low 32 bits are not 0 (0 is what untouched slot memory holds) */
static inline unsigned long long phip_next_seq(unsigned long long *s) { if (!(unsigned int)(++*s)) ++*s; return *s; }
static inline unsigned int phip_next_pflag(phip_ctx c) { if (!++c->pseq) ++c->pseq; return c->pseq; }

/* by-product tickets (phip_runtime.hip): fill a RedCtl for a kernel that leaves nvals results in a fresh ring slot */
struct RedCtl;
int phip_ticket_open(phip_ctx c, int nvals, int publish, phip_ticket *t, struct RedCtl *ctl);
/* ... or adds one more contribution to the live ticket t */
int phip_ticket_accumulate(phip_ctx c, const phip_ticket *t, int publish, struct RedCtl *ctl);

int phip_prof_open(phip_ctx c, int kind, hipEvent_t *start, hipEvent_t *stop);

/* launch the pending elementwise operations (phip_vec.hip).  Every entry point that queues other work on the stream, reads or frees
   device memory, or waits for the stream calls this first. */
#define PHIP_EW_BATCH 4
int phip_ew_flush(phip_ctx c);
#define PHIP_FLUSH(c) do { if ((c)->ew_n) { int _frc = phip_ew_flush(c); if (_frc) return _frc; } } while (0)

void phip_set_error(const char *fmt, ...);
/* reads and clears the context's pinned error word; PHIP_ERR_LIB (with a message) when a kernel left one there (phip_runtime.hip) */
int phip_check_device_error(phip_ctx c, const char *what);

/* phip_vec.hip: device address of the static partial rows of phip_vec_mdot_static on the current device (looked up once per device) */
int phip_static_scratch_setup(unsigned long long **ptr);

#define PHIP_CHECK_HIP(call)                                                                       \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      phip_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(_e));    \
      return PHIP_ERR_LIB;                                                                         \
    }                                                                                              \
  } while (0)

#define PHIP_REQUIRE(cond, code, ...)                                                              \
  do {                                                                                             \
    if (!(cond)) { phip_set_error(__VA_ARGS__); return (code); }                                   \
  } while (0)

/* device < 0: planning-only context (no GPU): index/plan work runs, every numerical entry point fails loudly */
#define PHIP_NEED_DEVICE(c, name) PHIP_REQUIRE((c)->device >= 0, PHIP_ERR_SUP, name ": planning-only context (no GPU): numerical operations are unavailable, there is no CPU fallback")
#define PHIP_PLANNING_PTR ((void *)(uintptr_t)0x40)

static inline int phip_grid_for(int64_t n, int per_thread, int maxblocks)
{
  int64_t b = (n + (int64_t)PHIP_THREADS * per_thread - 1) / ((int64_t)PHIP_THREADS * per_thread);
  if (b < 1) b = 1;
  if (b > maxblocks) b = maxblocks;
  return (int)b;
}
