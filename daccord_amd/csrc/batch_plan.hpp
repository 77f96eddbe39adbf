/*
 * Host-side planning of one batch of piles (pure C++): derives the device records
 * (DevPile / DevOvl), the window schedule of every pile (Windows, HandleContext.hpp:382-447),
 * the window range in which each overlap is active (HandleContext.hpp:1904-1977), the
 * normalised error keys that order the active set (:1780-1790, :1955-1957), task and table
 * offsets, and the scratch capacities.  No consensus arithmetic happens here.
 */
#ifndef DACC_BATCH_PLAN_HPP
#define DACC_BATCH_PLAN_HPP
#include <vector>
#include <string>
#include <limits>
#include <algorithm>
#include <thread>
#include <mutex>
#include <exception>
#include <atomic>
#include <cstdlib>
#include <cstdint>
#include "../../include/daccord_hip.h"
#include "dev_types.hpp"
#include "fast_window.hpp"

namespace dacc {

static inline uint32_t hostNextPow2(uint32_t v) { uint32_t p = 1; while ( p < v ) p <<= 1; return p; }

// Windows::computeN (HandleContext.hpp:390-408)
static inline uint32_t windowsN(uint64_t const l, uint64_t const a, uint64_t const w)
{
	uint64_t const npre = (l+a >= w) ? ((l+a-w)/a) : 0;
	if ( npre ) return ((npre-1)*a+w == l) ? npre : npre+1;
	return l >= w ? 1 : 0;
}
static inline void windowIv(uint64_t const l, uint64_t const a, uint64_t const w, uint64_t const y, uint64_t & s, uint64_t & e)
{
	if ( y*a+w <= l ) { s = y*a; e = s+w; } else { s = l-w; e = l; }
}

// larger scratch capacities for the generic engine after a window reported WS_OVERFLOW (dense graphs at small k)
static inline void growArenaCaps(ArenaCaps & c, uint32_t const w = 0)
{
	c.precap *= 2; c.nodecap *= 2;        // stays a power of two (bitonic sorts)
	c.fcap *= 4; c.strcap *= 2; c.linkcap *= 4; c.sfcap *= 4; c.rlcap *= 4; c.poolcap *= 4; c.conscap = 4*(c.conscap-DACC_MAXCONS_OF(w)) + DACC_MAXCONS_OF(w);
	c.blcap *= 2;                         // base lengths of the enumerated paths (long window strings make long paths)
}

struct BatchPlan
{
	std::vector<DevPile> piles;
	std::vector<DevOvl> ovl;
	std::vector<uint32_t> ovl_pile;
	std::vector<uint64_t> fragbase;
	uint64_t nwindows, nblocks, nwt, npos, nfragslots, algo_bytes;
	uint32_t maxdepth, maxcols;
	uint64_t maxspan;         // upper bound of the B window string lengths of the batch (from the trace values)
	std::vector<int32_t> pile_status;         // per submitted pile: DACC_OK or why it was dropped
	std::vector<std::string> pile_errors;     // first messages of dropped piles
	ArenaCaps caps;
	enum { NTIER = 3 };
	FastCaps ftier[NTIER];    // LDS fast path capacity tiers: 3, 2, 1 wavefronts per CU
	FastCaps ftier0;          // tier 0: small windows of shallow batches (size classes), runs in the first slot in front of tier 1
	FastCaps ftier7;          // tier 7: the middle size class (7 wavefronts per CU), between tier 0 and tier 1
	FastCaps ftierD;          // tier 10 (round 6): the dense-graph tier of shallow batches, between the second slot's tier 6 and tier 3
	FastCaps ftierL;          // tier 5: windows with a string of 65..128 bases (second stream, before the generic engine)
	uint64_t ndeepwin;        // windows with more strings / k-mer instances than the first tier of shallow batches holds
	bool deep;                // most windows are deep: the first tier is FastTier<4> (many strings, small graph) instead of FastTier<1>
	bool wide;                // window size 64 ... 127: the LDS tiers of the batch are FastTier<8> (second slot) and FastTier<9> (third slot) in front of the generic engine (round 6)

	// Per pile results of the parallel pass of plan(): everything that does not depend on the piles in front of it
	struct PileTmp
	{
		char const * bad; uint32_t pnovl, nwin, rl, maxdepth, maxcols; uint64_t l, pilepos, nblk, nwt, maxspan, ndeep, algo, inoff;
	};
	struct OvlTmp { int32_t nblk, y0, ny; uint32_t ekey; };

	// One pile: validation, window schedule, per overlap trace block count / active window range / error key, depth statistics.
	// Reads the input only and writes T and OT[T.inoff ...]: piles are independent here, so any number of threads may run it.
	static void planPile(dacc_params const & par, dacc_pile const & p, dacc_overlap const * O, void const * trace, uint64_t const ntrace, int const trace_bytes,
		uint32_t const * rlen, uint64_t const nreads, PileTmp & T, OvlTmp * OT, std::vector<int32_t> & diff)
	{
		uint8_t const * tr8 = static_cast<uint8_t const *>(trace); uint16_t const * tr16 = static_cast<uint16_t const *>(trace);
		auto const tv = [&](uint64_t const i) -> uint32_t { return trace_bytes == 2 ? tr16[i] : tr8[i]; };
		// A malformed pile is dropped (no overlaps -> no windows -> no fragments) and reported through pile_status, like
		// the reference's per-read try/catch (src/daccord.cpp:2464-2478); the batch goes on.
		dacc_overlap const * ita = O + p.first_ovl;
		char const * bad = 0;
		int64_t const tsv = par.tspace;
		for ( uint32_t z = 0; z < p.novl && !bad; ++z )
		{
			dacc_overlap const & o = ita[z];
			if ( o.aread != p.aread || o.bread < 0 || static_cast<uint64_t>(o.bread) >= nreads || o.abpos < 0 || o.aepos <= o.abpos ||
			     static_cast<uint32_t>(o.aepos) > rlen[o.aread] || o.bbpos < 0 || o.bepos < o.bbpos || static_cast<uint32_t>(o.bepos) > rlen[o.bread] ||
			     (z && ita[z-1].abpos > o.abpos) )
			{ bad = "malformed overlap record (ranges / not sorted by abpos)"; break; }
			int64_t const nblk = (o.aepos + tsv - 1)/tsv - o.abpos/tsv;
			if ( o.tlen != 2*nblk || o.trace_off + o.tlen > ntrace ) { bad = "trace length does not match the overlap's tspace blocks"; break; }
			uint64_t bsum = 0; uint32_t bmax = 0;
			for ( int64_t b = 0; b < nblk; ++b ) { uint32_t const bl = tv(o.trace_off+2*b+1); bsum += bl; if ( bl > bmax ) bmax = bl; }
			if ( static_cast<int64_t>(bsum) != o.bepos-o.bbpos ) { bad = "trace B lengths do not sum to bepos-bbpos"; break; }
			// what the LDS column stores of the trace kernels hold at 8 lanes per wavefront (capi.hip: k_trace_wide<4> for
			// tspace <= 256, <8> beyond); only two byte trace values can say more, and such a block is no alignment: the
			// pile is dropped
			if ( bmax > (tsv <= 256 ? 4096u : 2048u) ) { bad = "a trace block spans too many B bases"; break; }
		}
		uint32_t const pnovl = bad ? 0u : p.novl;
		T.bad = bad; T.pnovl = pnovl; T.maxdepth = 0; T.maxcols = 0; T.maxspan = 0; T.ndeep = 0; T.nblk = 0; T.nwt = 0; T.algo = 0;
		uint64_t maxaepos = 0;
		double maxerate = 0.0, minerate = 1.0;
		for ( uint32_t z = 0; z < pnovl; ++z )
		{
			dacc_overlap const & o = ita[z];
			if ( static_cast<uint64_t>(o.aepos) > maxaepos ) maxaepos = o.aepos;
			double const erate = static_cast<double>(o.diffs) / static_cast<double>(o.aepos-o.abpos);
			if ( erate > maxerate ) maxerate = erate;
			if ( erate < minerate ) minerate = erate;
		}
		double const ediv = (maxerate > minerate) ? (maxerate-minerate) : 1.0;
		T.l = maxaepos; T.nwin = pnovl ? windowsN(maxaepos,par.a,par.w) : 0; T.rl = rlen[p.aread];
		T.pilepos = std::max<uint64_t>(T.l,T.rl)+1;
		diff.assign(T.nwin+2,0);
		T.algo += (T.rl+3)/4;
		for ( uint32_t z = 0; z < pnovl; ++z )
		{
			dacc_overlap const & o = ita[z];
			OvlTmp & v = OT[T.inoff+z];
			double const erate = static_cast<double>(o.diffs) / static_cast<double>(o.aepos-o.abpos);
			uint64_t const escore = static_cast<uint64_t>(((erate-minerate)/ediv) * std::numeric_limits<uint32_t>::max());
			v.ekey = static_cast<uint32_t>(escore);
			int64_t const ts = par.tspace;
			int64_t const nblk = (o.aepos + ts - 1)/ts - o.abpos/ts;
			v.nblk = nblk;
			{
				// widest block, and the longest B span a window can have: a window of w bases touches at most nbw consecutive tspace blocks
				int64_t const nbw = (static_cast<int64_t>(par.w) + ts - 2)/ts + 1;
				uint64_t run = 0;
				for ( int64_t b = 0; b < nblk; ++b )
				{
					uint32_t const bl = tv(o.trace_off+2*b+1);
					if ( bl > T.maxcols ) T.maxcols = bl;
					run += bl;
					if ( b >= nbw ) run -= tv(o.trace_off+2*(b-nbw)+1);
					if ( run > T.maxspan ) T.maxspan = run;
				}
			}
			T.nblk += nblk;
			T.algo += 40 + static_cast<uint64_t>(o.tlen)*trace_bytes + (o.bepos-o.bbpos+3)/4;
			// active window range [y0,y0+ny): start(y) >= abpos and end(y) <= aepos
			uint32_t y0 = T.nwin, ny = 0;
			if ( T.nwin )
			{
				uint64_t y = std::min<uint64_t>((static_cast<uint64_t>(o.abpos)+par.a-1)/par.a,T.nwin-1);
				uint64_t s, e;
				while ( y > 0 ) { windowIv(T.l,par.a,par.w,y-1,s,e); if ( s >= static_cast<uint64_t>(o.abpos) ) --y; else break; }
				windowIv(T.l,par.a,par.w,y,s,e);
				if ( s >= static_cast<uint64_t>(o.abpos) )
				{
					uint64_t yl = y; bool any = false;
					for ( uint64_t q = y; q < T.nwin; ++q )
					{
						windowIv(T.l,par.a,par.w,q,s,e);
						if ( e <= static_cast<uint64_t>(o.aepos) ) { yl = q; any = true; } else break;
					}
					if ( any ) { y0 = y; ny = yl-y+1; }
				}
			}
			v.y0 = y0; v.ny = ny; T.nwt += ny;
			if ( ny ) { diff[y0] += 1; diff[y0+ny] -= 1; }
		}
		int32_t cur = 0;
		for ( uint32_t y = 0; y < T.nwin; ++y )
		{
			cur += diff[y]; if ( static_cast<uint32_t>(cur) > T.maxdepth ) T.maxdepth = cur;
			// strings of the window (A + active overlaps, capped by -d) against what FastTier<1> holds
			uint64_t const nb = par.maxalign > 0 ? static_cast<uint64_t>(par.maxalign-1) : 0;
			uint64_t const mao = 1 + std::min<uint64_t>(static_cast<uint64_t>(cur),nb);
			uint64_t const perstr = par.w >= par.klow ? static_cast<uint64_t>(par.w-par.klow+1) : 1;
			if ( mao > FastTier<1>::maxs || mao*perstr > FastTier<1>::precap ) ++T.ndeep;
		}
	}

	// threads of the planner: DACC_PLAN_THREADS, else as many as the host has, at most 16 and not more than one per 64 piles
	static unsigned planThreads(uint64_t const np)
	{
		unsigned n = 0;
		//  the environment value is a request within [1,64]; zero, negative or unparsable counts as unset
		if ( char const * e = getenv("DACC_PLAN_THREADS") ) { long const v = strtol(e,0,10); if ( v > 0 ) n = static_cast<unsigned>(v > 64 ? 64 : v); }
		if ( !n ) { n = std::thread::hardware_concurrency(); if ( !n ) n = 1; if ( n > 16 ) n = 16; }
		// DACC_PLAN_PILES_PER_THREAD (tests): piles a thread must have before another one is started, default 64
		uint64_t per = 64;
		if ( char const * e = getenv("DACC_PLAN_PILES_PER_THREAD") ) { long const v = strtol(e,0,10); if ( v > 0 ) per = static_cast<uint64_t>(v); }
		uint64_t const byload = np/per + 1;
		if ( n > byload ) n = static_cast<unsigned>(byload);
		return n ? n : 1;
	}
	// Runs f(tid,lo,hi) over [0,np) in chunks on nthreads threads (the caller is thread 0).  Nothing escapes a worker thread: an exception
	// there (bad_alloc of a per-thread buffer) is kept and rethrown in the caller after all threads have been joined, so that the C ABI's
	// catch blocks see it as they saw the serial planner's; a thread that cannot be started (system limit) is done without.
	template<typename F> static void planParallel(uint64_t const np, unsigned const nthreads, F const & f)
	{
		std::atomic<uint64_t> next(0);
		std::atomic<bool> failed(false);
		std::exception_ptr first; std::mutex firstlock;
		uint64_t const chunk = 32;
		auto const work = [&](unsigned const tid) {
			try
			{
				while ( !failed.load(std::memory_order_relaxed) ) { uint64_t const lo = next.fetch_add(chunk); if ( lo >= np ) break; f(tid,lo,std::min<uint64_t>(np,lo+chunk)); }
			}
			catch ( ... )
			{
				std::lock_guard<std::mutex> g(firstlock);
				if ( !first ) first = std::current_exception();
				failed.store(true);
			}
		};
		std::vector<std::thread> T;
		if ( nthreads > 1 )
		{
			try { T.reserve(nthreads-1); for ( unsigned t = 1; t < nthreads; ++t ) T.emplace_back(work,t); }
			catch ( ... ) {}      // fewer threads than asked for: the chunks are handed out dynamically, the plan is the same
		}
		work(0);
		for ( size_t t = 0; t < T.size(); ++t ) T[t].join();
		if ( first ) std::rethrow_exception(first);
	}

	// Three passes: (1) every pile on its own (planPile, threads), (2) the running offsets of the device arrays, serial and in pile
	// order (overlaps, trace blocks, window table rows, windows, positions, fragment slots; the first 64 messages of dropped piles),
	// (3) the device records at their offsets (threads).  Same bytes as the serial planner of rounds 1-4 (tests/test_plan.py).
	int plan(dacc_params const & par, dacc_pile const * P, uint64_t const np, dacc_overlap const * O, uint64_t const no,
		void const * trace, uint64_t const ntrace, int const trace_bytes, uint32_t const * rlen, uint64_t const nreads, std::string & err,
		uint32_t const tab_nrows = 0, uint32_t const tab_nsup = 0)
	{
		piles.clear(); ovl.clear(); ovl_pile.clear(); fragbase.clear(); pile_status.assign(np,DACC_OK); pile_errors.clear();
		nwindows = nblocks = nwt = npos = nfragslots = algo_bytes = 0; maxdepth = 0; maxcols = 0; maxspan = 0; ndeepwin = 0; deep = false; wide = false;
		if ( trace_bytes != 1 && trace_bytes != 2 ) { err = "trace values are 1 byte (tspace <= 125) or 2 bytes"; return DACC_EINVAL; }
		if ( par.tspace <= 0 || par.tspace > 512 ) { err = "tspace must be in [1,512] (column vectors of the trace kernels: 2, 4 or 8 64-bit words)"; return DACC_ENOTSUP; }
		std::vector<PileTmp> PT(np);
		uint64_t intot = 0;
		for ( uint64_t pi = 0; pi < np; ++pi )
		{
			dacc_pile const & p = P[pi];
			if ( p.aread < 0 || static_cast<uint64_t>(p.aread) >= nreads || p.first_ovl + p.novl > no ) { err = "pile out of range"; return DACC_EINVAL; }
			PT[pi].inoff = intot; intot += p.novl;
		}
		std::vector<OvlTmp> OT(intot);
		unsigned const nthreads = planThreads(np);
		{
			std::vector< std::vector<int32_t> > diffs(nthreads);
			planParallel(np,nthreads,[&](unsigned const tid, uint64_t const lo, uint64_t const hi) {
				for ( uint64_t pi = lo; pi < hi; ++pi ) planPile(par,P[pi],O,trace,ntrace,trace_bytes,rlen,nreads,PT[pi],OT.data(),diffs[tid]);
			});
		}
		// running offsets, in pile order
		piles.resize(np); fragbase.resize(np);
		std::vector<uint64_t> blk0(np), wt0(np);
		uint64_t novlout = 0;
		for ( uint64_t pi = 0; pi < np; ++pi )
		{
			PileTmp const & T = PT[pi];
			if ( T.bad )
			{
				pile_status[pi] = DACC_EINVAL;
				if ( pile_errors.size() < 64 ) pile_errors.push_back("read " + std::to_string(P[pi].aread) + ": " + T.bad);
			}
			DevPile & d = piles[pi];
			d.aread = P[pi].aread; d.novl = T.pnovl; d.first_ovl = novlout; d.l = T.l; d.nwin = T.nwin;
			d.winbase = nwindows; d.posbase = npos; d.rl = T.rl; d.pad = 0;
			blk0[pi] = nblocks; wt0[pi] = nwt;
			novlout += T.pnovl; nblocks += T.nblk; nwt += T.nwt; nwindows += T.nwin; npos += T.pilepos;
			fragbase[pi] = nfragslots; nfragslots += T.pilepos/100 + 2;
			algo_bytes += T.algo; ndeepwin += T.ndeep;
			if ( T.maxdepth > maxdepth ) maxdepth = T.maxdepth;
			if ( T.maxcols > maxcols ) maxcols = T.maxcols;
			if ( T.maxspan > maxspan ) maxspan = T.maxspan;
		}
		// device records of the overlaps at their offsets
		ovl.resize(novlout); ovl_pile.resize(novlout);
		planParallel(np,nthreads,[&](unsigned, uint64_t const lo, uint64_t const hi) {
			for ( uint64_t pi = lo; pi < hi; ++pi )
			{
				PileTmp const & T = PT[pi];
				dacc_overlap const * ita = O + P[pi].first_ovl;
				uint64_t b = blk0[pi], wt = wt0[pi];
				for ( uint32_t z = 0; z < T.pnovl; ++z )
				{
					dacc_overlap const & o = ita[z]; OvlTmp const & t = OT[T.inoff+z];
					DevOvl v;
					v.bread = o.bread; v.flags = o.flags; v.abpos = o.abpos; v.aepos = o.aepos; v.bbpos = o.bbpos; v.bepos = o.bepos;
					v.ekey = t.ekey; v.nblk = t.nblk; v.blk0 = b; v.trace_off = o.trace_off; v.y0 = t.y0; v.ny = t.ny; v.wtoff = wt;
					b += t.nblk; wt += t.ny;
					ovl[piles[pi].first_ovl+z] = v; ovl_pile[piles[pi].first_ovl+z] = pi;
				}
			}
		});
		// scratch capacities
		uint64_t const depthcap = std::min<uint64_t>(static_cast<uint64_t>(maxdepth)+1, par.maxalign ? par.maxalign : 1);
		// strings per window of the generic engine: as deep as the batch is, up to 8192 (the default -D keeps 5000 overlaps
		// per read; the number of wavefronts is bounded by the arena budget, capi.hip: boundByArena)
		caps.maxs = std::max<uint32_t>(2,static_cast<uint32_t>(std::min<uint64_t>(depthcap,8192)));
		// k-mer instances per string: 72 covers the windows the LDS tiers take (w <= 63 and the usual B strings); a wide window
		// (w in 65..128) has w-k+1 instances in its A string and about as many per B string (maxspan bounds them), so its first
		// capacities are sized for that instead of being grown by the overflow retry for most windows of the batch
		uint32_t const perstr = DACC_WIDE_W(par.w) ? static_cast<uint32_t>(std::min<uint64_t>(LSTRMAX,std::max<uint64_t>(72,std::max<uint64_t>(par.w,maxspan)))) : 72u;
		caps.precap = hostNextPow2(std::max<uint32_t>(256,std::max<uint32_t>(caps.maxs*perstr,maxdepth+1)));
		caps.nodecap = caps.precap;
		caps.fcap = caps.nodecap*24;
		caps.strcap = 2*caps.precap;
		caps.linkcap = 8*caps.precap;
		caps.sfcap = 8*caps.strcap;
		caps.rlcap = 2*caps.strcap;
		caps.poolcap = 8192;
		caps.blcap = 256;
		caps.conscap = 32768 + DACC_MAXCONS_OF(par.w);
		// string stride of the generic engine: what the longest possible B window string needs (LSTR for ordinary data:
		// two blocks of a hundred bases; more only for badly aligned blocks), a multiple of 64, at most LSTRMAX
		caps.lstr = static_cast<uint32_t>(std::min<uint64_t>(LSTRMAX,std::max<uint64_t>(LSTR,(maxspan+63)&~static_cast<uint64_t>(63))));
		if ( caps.lstr > LSTR ) caps.blcap = caps.lstr + 128;    // a stretch runs as far as a string does, a path half a window further
		caps.bytes = 0;
		// LDS fast path capacity tiers (compile time, fast_window.hpp); windows beyond them are re-run by the generic engine
		// A batch whose windows are mostly too deep for tier 1 (coverage of 40x and more) starts in the deep tier instead
		deep = 2*ndeepwin > nwindows;
		ftier[0] = deep ? fastCapsOf< FastTier<4> >(tab_nrows,tab_nsup) : fastCapsOf< FastTier<1> >(tab_nrows,tab_nsup);
		// (round 6) wide windows: tier 8 (second slot, no hand-over list in front of it = all windows), tier 9, then the generic engine
		wide = par.w > 63 && par.w <= 127;
		ftier[1] = wide ? fastCapsOf< FastTier<8> >(tab_nrows,tab_nsup) : (deep ? fastCapsOf< FastTier<2> >(tab_nrows,tab_nsup) : fastCapsOf< FastTier<6> >(tab_nrows,tab_nsup));
		ftier0 = fastCapsOf< FastTier<0> >(tab_nrows,tab_nsup);
		ftier7 = fastCapsOf< FastTier<7> >(tab_nrows,tab_nsup);
		ftierL = fastCapsOf< FastTier<5> >(tab_nrows,tab_nsup);
		ftierD = deep ? fastCapsOf< FastTier<11> >(tab_nrows,tab_nsup) : fastCapsOf< FastTier<10> >(tab_nrows,tab_nsup);      // (deep batches: tier 11)
		ftier[2] = wide ? fastCapsOf< FastTier<9> >(tab_nrows,tab_nsup) : fastCapsOf< FastTier<3> >(tab_nrows,tab_nsup);
		return DACC_OK;
	}
};

}
#endif
