/*
 * daccord_io.h -- host-side readers / writers for the two on-disk inputs of `daccord <in.las> <in.db>`
 * (SURVEY.md section 8f row 2): the Dazzler read database (foo.db + .foo.idx + .foo.bps) and the DALIGNER
 * overlap file (foo.las).  The reference reads both through libmaus2 (src/daccord.cpp:1328-1375 DatabaseFile,
 * :2133-2160 OverlapParser / DalignerIndexDecoder), which is not part of the reference tree; the layouts below are
 * those of DAZZ_DB's DB.h and DALIGNER's align.h as restated in SURVEY.md section 10 -- FORMAT UNPINNED: no real
 * file was available to validate against, reader and writer are tested against each other only.
 *
 * Plain C, no device code: the arrays handed out are exactly the arguments of dacc_load_db / dacc_submit_piles
 * (include/daccord_hip.h).  All calls return 0 or a negative DACC_E* code.
 */
#ifndef DACCORD_IO_H
#define DACCORD_IO_H

#include <stdint.h>
#include "daccord_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dacc_db dacc_db;
typedef struct dacc_las dacc_las;

/* ---- read database: replaces libmaus2::dazzler::db::DatabaseFile + computeTrimVector (daccord.cpp:1328-1334) ---- */

/* Opens <path> = ".../foo.db" (the stub), ".../.foo.idx" and ".../.foo.bps"; the trimmed view (reads with
 * rlen >= cutoff and, unless the DB was built with -a, flagged best) is what read ids of a .las refer to. */
int  dacc_db_open(const char *path, dacc_db **db);
void dacc_db_close(dacc_db *db);
/* Arrays of the trimmed view, owned by the handle: the 2-bit payload, per read byte offset and length. */
int  dacc_db_arrays(dacc_db *db, const uint8_t **bps, uint64_t *bps_bytes,
                    const uint64_t **boff, const uint32_t **rlen, uint64_t *nreads);
const char *dacc_db_error(dacc_db *db);

/* Writes foo.db / .foo.idx / .foo.bps for the given reads (all reads kept: cutoff 0, DB_ALL). */
int  dacc_db_write(const char *path, const uint8_t *bps, uint64_t bps_bytes,
                   const uint64_t *boff, const uint32_t *rlen, uint64_t nreads);

/* ---- overlaps: replaces OverlapParser + the .las index (daccord.cpp:1075-1094, 2133-2160) ---- */

/* Opens the file and builds the A read -> byte offset table (one sequential pass over the record headers, or the sidecar
 * index <path>.daidx an earlier run left; DACC_LAS_INDEX=0 in the environment turns the sidecar off).  The records
 * themselves stay on disk: records must be sorted by A read, as LAsort leaves them. */
int  dacc_las_open(const char *path, dacc_las **las);
void dacc_las_close(dacc_las *las);
int  dacc_las_info(dacc_las *las, int64_t *novl, int32_t *tspace, int32_t *trace_bytes,
                   int64_t *min_aread, int64_t *max_aread);
/* Piles of the A reads in [afirst,alast) in .las order (not yet top-D selected: see dacc_pile_select): reads exactly the
 * byte range of these A reads from the file (daccord.cpp:2133-2181 does the same per pile through the .las index).
 * Output arrays are owned by the handle and valid until the next call / close. */
int  dacc_las_piles(dacc_las *las, int64_t afirst, int64_t alast,
                    const dacc_pile **piles, uint64_t *npiles,
                    const dacc_overlap **ovl, uint64_t *novl,
                    const void **trace, uint64_t *ntrace);
const char *dacc_las_error(dacc_las *las);

int  dacc_las_write(const char *path, int32_t tspace, const dacc_overlap *ovl, uint64_t novl,
                    const void *trace, uint64_t ntrace, int trace_bytes);

/* ---- the A reads of a run: -J part,parts or -I first,last applied to the A reads [las_min,las_max] of the overlap file
 * (src/daccord.cpp:1115-1227; -J is also how N processes / GPUs share one file, :1156-1183).  J / I: the option's text
 * ("2,8", "100,199") or NULL; J wins when both are given, as in the reference (else-if).  The run covers the A reads
 * [*minaread,*toparead); an empty part comes back as minaread = 0, toparead = -1 (the reference's "maxaread = -1").
 * DACC_EINVAL: text that does not parse as <int>,<int>, or a zero denominator over a non-empty span (message in err). ---- */
int  dacc_read_interval(int64_t las_min, int64_t las_max, const char *J, const char *I,
                        int64_t *minaread, int64_t *toparead, char *err, uint64_t errcap);

/* ---- truth-based accuracy check of a corrected fragment (the measurement of the package's checkconsensus tool,
 * src/checkconsensus.cpp:730-1075): the fragment (ASCII) is aligned completely to a window of the true sequence with free
 * ends, inside a band of +-band around the line from column c0 (fragment start) to c1 (fragment end).
 * stats = {matches, mismatches, insertions (fragment only), deletions (truth only)}; [rfrom,rto) = the part of the window
 * the fragment aligned to.  DACC_ENOTSUP: no alignment inside the band. ---- */
int  dacc_check_fragment(const uint8_t *frag, uint64_t n, const uint8_t *ref, uint64_t m, uint64_t c0, uint64_t c1, uint64_t band,
                         uint64_t *stats, uint64_t *rfrom, uint64_t *rto);

#ifdef __cplusplus
}
#endif
#endif
