/*
 * Device-side data layout shared by the kernels and the host driver (plain structs, HBM
 * resident).  See DESIGN.md "Data layout in HBM".
 */
#ifndef DACC_DEV_TYPES_HPP
#define DACC_DEV_TYPES_HPP
#include <stdint.h>

namespace dacc {

// flattened, zero padded OffsetLikely + KmerLimit tables (built on the host once, host_tables.cpp)
struct DevTables
{
	int32_t nrows;               // maxl+1 rows (reference positions), OffsetLikely::size()
	int32_t nsup;                // number of read positions covered (Vsupport.size())
	int32_t kln;                 // entries per k in klim
	int32_t pad;
	double const * dpnorm;       // [nrows][nsup]  DPnorm[i][pos], 0 outside the row's support
	double const * dpsq;         // [nrows][nsup]  DPnormSquare[i].V, addressed by absolute pos
	uint64_t const * dpsq_vs;    // [nrows][nsup]  DPnormSquare[i].VS (2^32 fixed point)
	uint16_t const * dpsq_first; // [nrows] firstsign of DPnormSquare[i]
	uint16_t const * dpsq_size;  // [nrows] V.size()
	uint16_t const * suplo;      // [nsup] Vsupport[pos].first
	uint16_t const * suphi;      // [nsup] Vsupport[pos].second
	uint32_t const * klim;       // [khigh-klow+1][kln] KmerLimit::getLimit(n)
};

struct DevParams
{
	uint32_t w, a, klow, khigh;
	int32_t minff, maxff;
	uint32_t minwindowcov;
	int32_t checklim;            // est_cor != 0 (DebruijnGraph::p, setupAddHeap)
	uint64_t maxalign, eminrate;
	int32_t tspace;
	int32_t producefull;
	uint64_t minlen;
};

// one overlap, device form
struct DevOvl
{
	int32_t bread; uint32_t flags;
	int32_t abpos, aepos, bbpos, bepos;
	uint32_t ekey;               // uint32(((erate-min)/ediv)*UINT32_MAX), HandleContext.hpp:1955
	int32_t y0, ny;              // windows [y0,y0+ny) in which the overlap is active
	int32_t nblk;                // number of trace blocks
	uint64_t wtoff;              // offset of its window table rows
	uint64_t blk0;               // id of its first trace block task
	uint64_t trace_off;
};

struct DevPile
{
	int32_t aread; uint32_t novl;
	uint64_t first_ovl;
	uint32_t l;                  // max aepos = window schedule length (HandleContext.hpp:1773-1776,1852)
	uint32_t nwin;
	uint64_t winbase;            // index of its first window
	uint64_t posbase;            // index of its first position slot (vote kernel)
	uint32_t rl;                 // A read length
	uint32_t pad;
};

// capacities of the per-wavefront scratch arena (chosen by the host from the batch)
struct ArenaCaps
{
	uint32_t maxs;      // strings per window (A included)
	uint32_t precap;    // prenodes (power of two)
	uint32_t nodecap;   // nodes
	uint32_t fcap;      // feasible-position entries (each direction)
	uint32_t strcap;    // stretches
	uint32_t linkcap;   // stretch link words
	uint32_t sfcap;     // stretch feasibility objects (each direction)
	uint32_t rlcap;     // reverse stretch links
	uint32_t poolcap;   // path pool entries (each direction)
	uint32_t blcap;     // distinct base lengths (heaps per length)
	uint32_t conscap;   // candidate text bytes
	uint32_t lstr;      // string stride of the generic engine: a multiple of 64, at least LSTR (host plan: longest B span a window can have)
	uint64_t bytes;     // total arena bytes per wavefront (filled by arena_layout)
};

enum { LSTR = 256 };       // least string stride of the generic engine (ArenaCaps::lstr); up to here the Myers column state stays in registers
enum { LPW = LSTR/64 };     // 64 bit words of such a column
enum { LSTRMAX = 4096 };   // largest string stride the host plan asks for (a longer B window string drops its pile)
enum { WREC = 256 };       // bytes per window output record (w <= 64)
enum { MAXCONS = 96 };     // max consensus length (w <= 64)
// Wide windows (w in 65..128; generic engine only): consensus up to MAXCONSW symbols, records of WRECW bytes with 16 bit group
// offsets: rec[0] = status, offset of group r (r = 0..w+1) little endian at rec[2+2r], symbols from rec[2+2(w+2)] on (at most
// w + MAXCONSW of them: one per A position plus the inserted ones).  The narrow layout (rec[1+r], symbols from rec[1+(w+2)]) is
// what the LDS tiers write and stays as it is.
enum { WRECW = 640 };
enum { MAXCONSW = 224 };
#define DACC_WIDE_W(w_) ((w_) > 64u)
#define DACC_WREC_OF(w_) (DACC_WIDE_W(w_) ? static_cast<uint32_t>(::dacc::WRECW) : static_cast<uint32_t>(::dacc::WREC))
#define DACC_MAXCONS_OF(w_) (DACC_WIDE_W(w_) ? static_cast<uint32_t>(::dacc::MAXCONSW) : static_cast<uint32_t>(::dacc::MAXCONS))
#define DACC_WMAX 128u     // largest window size (two 64 bit words of the consensus -> A alignment)

// window status / flags
enum { WS_INSUFFICIENT = 0, WS_OK = 1, WS_FAILED = 2, WS_OVERFLOW = 3 };

}
#endif
