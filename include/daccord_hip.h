/*
 * daccord_hip.h — C ABI of libdaccord_hip.so, the MI355X-native replacement for
 * daccord's per-window local de Bruijn consensus path.
 *
 * The reference (gt1/daccord v0.0.14) exposes no FFI for this path.  The entry
 * points below replace, batch-granular, exactly these reference interfaces:
 *
 *   dacc_create / dacc_destroy   <- HandleContext::HandleContext(...)          src/HandleContext.hpp:332-380
 *                                   (one context per worker thread)             src/daccord.cpp:1989-2023
 *   dacc_set_error_profile       <- computeOffsetLikely(w,p_i,p_d) + KmerLimit  src/daccord.cpp:1867-1913, 1981-1988
 *   dacc_load_db                 <- DatabaseFile -> RAM + DecodedReadContainer  src/daccord.cpp:1328-1369,
 *                                                                               src/DecodedReadContainer.hpp:160-199
 *   dacc_submit_piles            <- HandleContext::operator()(out,err,ita,ite)  src/HandleContext.hpp:1699-2901
 *                                   called per A-read from the OpenMP loop      src/daccord.cpp:2107-2112, 2402-2414
 *   dacc_collect / dacc_release  <- the `out` stream of operator() (FASTA       src/HandleContext.hpp:2710-2724
 *                                   records) before wellcounter numbering       src/daccord.cpp:2481-2534
 *   dacc_last_error              <- LibMausException::what() logged by caller   src/daccord.cpp:2466-2478
 *
 * Conventions: every call returns 0 or a negative DACC_E* code and never throws
 * across the ABI.  A pile that fails internally yields zero fragments (mirrors
 * the reference's per-read try/catch that logs and skips, daccord.cpp:2466-2478).
 * Inputs are borrowed until the call returns.  One host thread per context.
 * There is NO CPU fallback: without a usable HIP device dacc_create fails.
 */
#ifndef DACCORD_HIP_H
#define DACCORD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DACC_OK          0
#define DACC_EINVAL     -1   /* bad argument / unsupported parameter combination */
#define DACC_ENODEV     -2   /* no usable HIP device (the library never falls back to the CPU) */
#define DACC_ENOMEM     -3
#define DACC_ESTATE     -4   /* call order violated (e.g. submit before load_db) */
#define DACC_EHIP       -5   /* HIP runtime error, see dacc_last_error */
#define DACC_ENOTSUP    -6   /* input outside the kernel's capacity (depth / tspace / k) */
#define DACC_EINTERNAL  -7   /* an exception of the host side that is not an allocation failure (a planner thread, a container): see dacc_last_error */

/* Run parameters: the daccord command line options that shape the path
 * (src/daccord.cpp:101-169 defaults, :1282-1305 parsing). */
typedef struct dacc_params {
	uint32_t w;              /* -w window size            (default 40); 1 <= w <= 128 (w <= 63: LDS tiers; 64..127: the wide LDS tiers; 128: generic engine) */
	uint32_t a;              /* -a advance size           (default 10) */
	uint32_t klow, khigh;    /* -k single value or lo,hi  (default 8,8); 3 <= k <= 16 */
	int32_t  minfilterfreq;  /* --minfilterfreq           (default 0) */
	int32_t  maxfilterfreq;  /* --maxfilterfreq           (default 2) */
	uint32_t minwindowcov;   /* -m                        (default 3) */
	uint64_t maxalign;       /* -d max depth              (default UINT64_MAX) */
	uint64_t eminrate;       /* -e max window error       (default UINT64_MAX) */
	uint64_t minlen;         /* -l min output length      (default 0) */
	int32_t  producefull;    /* -f                        (default 0) */
	int32_t  tspace;         /* trace point spacing of the .las (AlignmentFile::getTSpace, daccord.cpp:1375), 1..512 */
	int32_t  device;         /* HIP device ordinal */
	int32_t  verbose;
} dacc_params;

/* One DALIGNER overlap record (the 40-byte on-disk Overlap),
 * in the field meaning of libmaus2::dazzler::align::OverlapDataInterface as used at
 * src/HandleContext.hpp:1750-1962.  bbpos/bepos are coordinates in the
 * reverse-complemented B read when (flags & 1). */
typedef struct dacc_overlap {
	int32_t  aread, bread;
	uint32_t flags;          /* bit 0: B is reverse complemented (isInverse) */
	int32_t  abpos, aepos, bbpos, bepos;
	int32_t  diffs;
	int32_t  tlen;           /* number of trace values = 2 * number of tspace blocks */
	uint32_t reserved;
	uint64_t trace_off;      /* index of this overlap's first trace value in the trace array */
} dacc_overlap;

/* One pile = all overlaps of one A read, already selected (top-D) and sorted by
 * abpos exactly as the reference's caller does (src/daccord.cpp:2166-2288). */
typedef struct dacc_pile {
	int32_t  aread;
	uint32_t novl;
	uint64_t first_ovl;      /* index into the overlap array */
} dacc_pile;

/* One corrected fragment = one FASTA record of the reference
 * (">{aread+1}/{well}/{first}_{first+len} A=[{first},{last}]", HandleContext.hpp:2712).
 * The `well` field is numbered by the caller after ordered collection. */
typedef struct dacc_fragment {
	int32_t  aread;
	uint32_t first, last;    /* A=[first,last] */
	uint32_t len;            /* number of bases */
	uint64_t seq_off;        /* offset of the bases in the buffer returned by dacc_collect */
} dacc_fragment;

typedef struct dacc_ctx dacc_ctx;

/* Number of HIP devices visible to this process (0: none).  The reference has no device notion; a front end that runs one
 * context per device (`--gpus N`) uses it to tell a missing device from any other dacc_create failure. */
int  dacc_device_count(void);
int  dacc_create(dacc_ctx **ctx, const dacc_params *params);
void dacc_destroy(dacc_ctx *ctx);

/* Error profile -> OffsetLikely tables + KmerLimit tables, built on the host once
 * and uploaded (never recomputed on the device). */
int  dacc_set_error_profile(dacc_ctx *ctx, double p_i, double p_d, double est_cor);

/* 2-bit read store in the Dazzler .bps layout: 4 bases per byte, first base in the
 * two most significant bits, A,C,G,T = 0,1,2,3; read i occupies ceil(rlen[i]/4)
 * bytes starting at boff[i].  Copied to HBM and kept resident. */
int  dacc_load_db(dacc_ctx *ctx, const uint8_t *bps, uint64_t bps_bytes,
                  const uint64_t *boff, const uint32_t *rlen, uint64_t nreads);

/* Process a batch of piles on the device.  `trace` holds trace_bytes (1 or 2)
 * bytes per value, pairs (diffs_i, blen_i) per tspace block. */
int  dacc_submit_piles(dacc_ctx *ctx,
                       const dacc_pile *piles, uint64_t npiles,
                       const dacc_overlap *ovl, uint64_t novl,
                       const void *trace, uint64_t ntrace, int trace_bytes);

/* Fragments of the last submitted batch, ordered by (pile index, first).
 * Library-owned until dacc_release / next submit. */
int  dacc_collect(dacc_ctx *ctx, const dacc_fragment **frags, uint64_t *nfrags,
                  const char **bases, uint64_t *nbases);
void dacc_release(dacc_ctx *ctx);

const char *dacc_last_error(dacc_ctx *ctx);

/* Pile loader selection step (host): keep <= maxinput overlaps of one pile (records in .las
 * order) and sort them by abpos exactly as src/daccord.cpp:2120-2288 does; `out` must hold n records. */
int  dacc_pile_select(const dacc_overlap *in, uint64_t n, int trace_bytes, uint64_t maxinput,
                      dacc_overlap *out, uint64_t *nout);

/* ---- error profile estimation (host): replaces the sampling pass of src/daccord.cpp:1653-1878 ---- */

/* The estimator's own pile selection (daccord.cpp:1705-1737): the maxinput overlaps with the LOWEST error score,
 * sorted by abpos (the main path's selection, dacc_pile_select, has the reference's keep-the-worst quirk instead). */
int  dacc_pile_select_lowest(const dacc_overlap *in, uint64_t n, int trace_bytes, uint64_t maxinput,
                             dacc_overlap *out, uint64_t *nout);

/* handleIndelEstimate<8> (daccord.cpp:271-631) over batches of piles: windows of 40 bases every 5, k = 8, k-mers seen
 * at least twice, trivial traversal (DebruijnGraph.hpp:3794-3824), every window string aligned to the window consensus,
 * alignment operations counted.  The read store (same layout as dacc_load_db) is borrowed until dacc_eprof_destroy;
 * two_databases: kept for ABI stability, without effect since round 4 -- the A window ALWAYS joins a window's strings: the reference's
 * test `&RC != &RC2` (:522) compares two distinct local containers (:1775-1776) and is always true, with one database as with two.
 * dacc_eprof_finish: counts = {matches, mismatches, insertions, deletions}, prof = {p_i, p_d, est_cor}
 * (daccord.cpp:1867-1878); DACC_ENOTSUP if no window was usable. */
typedef struct dacc_eprof dacc_eprof;
int  dacc_eprof_create(dacc_eprof **e, int32_t tspace, const uint8_t *bps, const uint64_t *boff, const uint32_t *rlen,
                       uint64_t nreads, int two_databases);
int  dacc_eprof_add(dacc_eprof *e, const dacc_pile *piles, uint64_t npiles, const dacc_overlap *ovl, uint64_t novl,
                    const void *trace, uint64_t ntrace, int trace_bytes, uint64_t maxalign, int nthreads);
int  dacc_eprof_finish(dacc_eprof *e, uint64_t counts[4], uint64_t *usable, uint64_t *unusable,
                       double *eavg, double *edif, double prof[3]);
void dacc_eprof_destroy(dacc_eprof *e);
/* piles passed to dacc_eprof_add so far (*seen) and how many were left out because of malformed overlap / trace records (*skipped):
 * the reference logs such a read (src/daccord.cpp:2464-2478); a caller should report the count and refuse a profile made from a
 * minority of the piles */
int  dacc_eprof_skipped(dacc_eprof *e, uint64_t *skipped, uint64_t *seen);
/* --deepprofileonly (daccord.cpp:1442-1650, handleIndelEstimateDeep :634-995): switch the collection on before the first
 * dacc_eprof_add; dacc_eprof_deep returns, ascending, round(error rate * (2^32-1)) of every window that got a consensus
 * (:963-968).  The caller prints the cumulative distribution (:1626-1648). */
int  dacc_eprof_set_deep(dacc_eprof *e, int on);
int  dacc_eprof_deep(dacc_eprof *e, const uint32_t **values, uint64_t *n);

/* ---- measurement hooks (bench.py / profiling; not part of the data path) ---- */

/* Timings of the last dacc_submit_piles in milliseconds, measured with HIP events
 * on the stream the kernels are launched on. */
typedef struct dacc_timing {
	float h2d_ms;            /* upload of piles/overlaps/trace */
	float trace_ms;          /* trace-point block alignment kernel */
	float window_ms;         /* per-window de Bruijn consensus kernel */
	float vote_ms;           /* pile vote kernel(s) */
	float d2h_ms;            /* download of fragments */
	float total_ms;          /* first kernel start -> last kernel end */
	uint64_t nwindows;       /* windows processed */
	uint64_t nblocks;        /* trace blocks aligned */
	uint64_t algo_bytes;     /* algorithmic bytes of the batch */
	float tier_ms[3];        /* LDS capacity tiers of the window kernel (3, 2, 1 wavefronts per CU); window_ms = all + generic */
	uint32_t tier_out[3];    /* windows each tier handed on (tier_out[2] = windows run by the generic engine) */
	uint32_t first_tier;     /* kernel of the first slot: 1 = k_window_fast<1>, 4 = k_window_fast<4> (batch of deep piles) */
	uint32_t long_windows;   /* windows the second stream ran (a string of more than 64 bases, or a shape no LDS tier takes) */
	float tier0_ms;          /* size classes (shallow batches): pre-pass + k_window_fast<0>, the part of tier_ms[0] in front of k_window_fast<1>; 0 if tier 0 did not run */
	uint32_t tier0_in;       /* windows the pre-pass sent to tier 0 */
	uint32_t tier0_out;      /* windows tier 0 handed on: to tier 7 (round 6), to tier 1 when the middle class is off */
	float tier7_ms;          /* round 6, the middle size class: k_window_fast<7> (7 wavefronts per CU) between tier 0 and tier 1; 0 if it did not run */
	uint32_t tier7_in;       /* windows tier 7 ran: the pre-pass's middle class + tier 0's hand-overs (tier0_out) */
	uint32_t tier7_out;      /* windows tier 7 handed on to tier 1 */
	uint32_t pad_;
	uint32_t long_first_tier; /* of long_windows: the windows the FIRST tier found no LDS tier can run (they ran in k_window_long behind it; the rest are the pre-scan's) */
	float tier10_ms;         /* round 6, the dense-graph tier of shallow batches: k_window_fast<10> (2 wavefronts per CU) between tier 6 and tier 3, the part of tier_ms[2] in front of k_window_fast<3>; 0 if it did not run */
	uint32_t tier10_out;     /* windows tier 10 handed on to tier 3 (it ran tier_out[1] windows) */
	uint32_t tier10_ran;     /* 1: tier 10 ran in this pass (tier_out[1] went to it, not to tier 3) */
	uint32_t pad2_;
} dacc_timing;
int  dacc_last_timing(dacc_ctx *ctx, dacc_timing *t);

/* Per-pile outcome of the last dacc_submit_piles, in submission order: DACC_OK, or the reason the pile was dropped
 * (DACC_EINVAL: malformed overlap / trace records; DACC_ENOTSUP: a window beyond every engine's capacity).  A dropped
 * pile yields zero fragments and the batch goes on -- the reference logs the exception of one read and continues with
 * the next (src/daccord.cpp:2464-2478).  dacc_pile_errors returns the messages a caller would log (one per line, at
 * most 64), valid until the next call on this context. */
int  dacc_pile_status(dacc_ctx *ctx, int32_t *status, uint64_t cap, uint64_t *n);
const char *dacc_pile_errors(dacc_ctx *ctx);

/* Host-only measurement hook (no device, no context): the planner that dacc_submit_piles runs before its uploads, on the same
 * inputs; nwindows / nblocks (optional) receive the batch's window and trace block counts.  `daccord_hip --loaderonly` times the
 * host side of a run with it on a box without a GPU (reference: the input side of the OpenMP loop, src/daccord.cpp:2107-2160). */
int  dacc_plan_only(const dacc_params *par, const uint32_t *rlen, uint64_t nreads, const dacc_pile *piles, uint64_t npiles,
                    const dacc_overlap *ovl, uint64_t novl, const void *trace, uint64_t ntrace, int trace_bytes,
                    uint64_t *nwindows, uint64_t *nblocks);

/* Re-run only the device part of the last submitted batch (inputs already resident
 * in HBM); used by bench.py so the timed region excludes H2D. */
int  dacc_rerun_resident(dacc_ctx *ctx);

/* Debug/parity hook: per-window results of the last batch.
 * For window i: status (0 insufficient depth, 1 consensus found, 2 path failed),
 * consensus length and bases (the first 79), number of strings, elength. */
typedef struct dacc_window_result {
	int32_t  pile;           /* pile index in the batch */
	int32_t  y;              /* window index */
	int32_t  status;
	int32_t  mao;            /* number of strings in the window (A included) */
	int32_t  elength;
	int32_t  k;              /* k of the accepted graph */
	int32_t  filterfreq;     /* filter frequency at which the consensus was found */
	int32_t  conslen;
	uint64_t minrate;        /* summed edit distance of the consensus */
	char     cons[80];
} dacc_window_result;
int  dacc_debug_windows(dacc_ctx *ctx, dacc_window_result *out, uint64_t cap, uint64_t *nwin);

/* Debug/parity hook: canonical serialisation of the OffsetLikely / KmerLimit tables built by
 * dacc_set_error_profile (compared bit for bit with the oracle's). */
int  dacc_debug_tables(dacc_ctx *ctx, uint64_t *out, uint64_t cap, uint64_t *n, uint64_t klimit_n);

/* Profiling hook: 32 per-phase shader-cycle counters of the window kernel (zero unless the library was
 * built with -DDACC_PROFILE). */
int  dacc_debug_profile(dacc_ctx *ctx, uint64_t *out32);
/* The fine sites of the same build (round 5 hot-spot ledger, scripts/prof_sites.py): out96[s] = shader cycles spent at site s,
 * out96[48+s] = visits, s < 48. */
int  dacc_debug_profile_fine(dacc_ctx *ctx, uint64_t *out96);

/* Debugging hook (environment DACC_DEBUG_RETRY=1 at dacc_create): quadruples (window, flags, strings, filter frequency)
 * of the windows the last LDS capacity tier handed to the generic engine in the last run. */
int  dacc_debug_retry(dacc_ctx *ctx, uint32_t *out, uint64_t cap, uint64_t *n);

#ifdef __cplusplus
}
#endif
#endif
