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
	FastCaps ftierL;          // tier 5: windows with a string of 65..128 bases (second stream, before the generic engine)
	uint64_t ndeepwin;        // windows with more strings / k-mer instances than the first tier of shallow batches holds
	bool deep;                // most windows are deep: the first tier is FastTier<4> (many strings, small graph) instead of FastTier<1>

	int plan(dacc_params const & par, dacc_pile const * P, uint64_t const np, dacc_overlap const * O, uint64_t const no,
		void const * trace, uint64_t const ntrace, int const trace_bytes, uint32_t const * rlen, uint64_t const nreads, std::string & err,
		uint32_t const tab_nrows = 0, uint32_t const tab_nsup = 0)
	{
		piles.clear(); ovl.clear(); ovl_pile.clear(); fragbase.clear(); pile_status.assign(np,DACC_OK); pile_errors.clear();
		nwindows = nblocks = nwt = npos = nfragslots = algo_bytes = 0; maxdepth = 0; maxcols = 0; maxspan = 0; ndeepwin = 0; deep = false;
		if ( trace_bytes != 1 && trace_bytes != 2 ) { err = "trace values are 1 byte (tspace <= 125) or 2 bytes"; return DACC_EINVAL; }
		if ( par.tspace <= 0 || par.tspace > 512 ) { err = "tspace must be in [1,512] (column vectors of the trace kernels: 2, 4 or 8 64-bit words)"; return DACC_ENOTSUP; }
		uint8_t const * tr8 = static_cast<uint8_t const *>(trace); uint16_t const * tr16 = static_cast<uint16_t const *>(trace);
		auto const tv = [&](uint64_t const i) -> uint32_t { return trace_bytes == 2 ? tr16[i] : tr8[i]; };
		std::vector<int32_t> diff;
		for ( uint64_t pi = 0; pi < np; ++pi )
		{
			dacc_pile const & p = P[pi];
			if ( p.aread < 0 || static_cast<uint64_t>(p.aread) >= nreads || p.first_ovl + p.novl > no ) { err = "pile out of range"; return DACC_EINVAL; }
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
			if ( bad )
			{
				pile_status[pi] = DACC_EINVAL;
				if ( pile_errors.size() < 64 ) pile_errors.push_back("read " + std::to_string(p.aread) + ": " + bad);
			}
			DevPile d; d.aread = p.aread; d.novl = pnovl; d.first_ovl = ovl.size();
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
			d.l = maxaepos; d.nwin = pnovl ? windowsN(maxaepos,par.a,par.w) : 0;
			d.winbase = nwindows; d.posbase = npos; d.rl = rlen[p.aread]; d.pad = 0;
			uint64_t const pilepos = std::max<uint64_t>(d.l,d.rl)+1;
			diff.assign(d.nwin+2,0);
			algo_bytes += (d.rl+3)/4;
			for ( uint32_t z = 0; z < pnovl; ++z )
			{
				dacc_overlap const & o = ita[z];
				DevOvl v;
				v.bread = o.bread; v.flags = o.flags; v.abpos = o.abpos; v.aepos = o.aepos; v.bbpos = o.bbpos; v.bepos = o.bepos;
				double const erate = static_cast<double>(o.diffs) / static_cast<double>(o.aepos-o.abpos);
				uint64_t const escore = static_cast<uint64_t>(((erate-minerate)/ediv) * std::numeric_limits<uint32_t>::max());
				v.ekey = static_cast<uint32_t>(escore);
				int64_t const ts = par.tspace;
				int64_t const nblk = (o.aepos + ts - 1)/ts - o.abpos/ts;
				v.nblk = nblk; v.blk0 = nblocks; v.trace_off = o.trace_off;
				for ( int64_t b = 0; b < nblk; ++b ) { uint32_t const bl = tv(o.trace_off+2*b+1); if ( bl > maxcols ) maxcols = bl; }
				{
					// longest B span a window can have: a window of w bases touches at most nbw consecutive tspace blocks
					int64_t const nbw = (static_cast<int64_t>(par.w) + ts - 2)/ts + 1;
					uint64_t run = 0;
					for ( int64_t b = 0; b < nblk; ++b )
					{
						run += tv(o.trace_off+2*b+1);
						if ( b >= nbw ) run -= tv(o.trace_off+2*(b-nbw)+1);
						if ( run > maxspan ) maxspan = run;
					}
				}
				nblocks += nblk;
				algo_bytes += 40 + static_cast<uint64_t>(o.tlen)*trace_bytes + (o.bepos-o.bbpos+3)/4;
				// active window range [y0,y0+ny): start(y) >= abpos and end(y) <= aepos
				uint32_t y0 = d.nwin, ny = 0;
				if ( d.nwin )
				{
					uint64_t y = std::min<uint64_t>((static_cast<uint64_t>(o.abpos)+par.a-1)/par.a,d.nwin-1);
					uint64_t s, e;
					while ( y > 0 ) { windowIv(d.l,par.a,par.w,y-1,s,e); if ( s >= static_cast<uint64_t>(o.abpos) ) --y; else break; }
					windowIv(d.l,par.a,par.w,y,s,e);
					if ( s >= static_cast<uint64_t>(o.abpos) )
					{
						uint64_t yl = y; bool any = false;
						for ( uint64_t q = y; q < d.nwin; ++q )
						{
							windowIv(d.l,par.a,par.w,q,s,e);
							if ( e <= static_cast<uint64_t>(o.aepos) ) { yl = q; any = true; } else break;
						}
						if ( any ) { y0 = y; ny = yl-y+1; }
					}
				}
				v.y0 = y0; v.ny = ny; v.wtoff = nwt; nwt += ny;
				if ( ny ) { diff[y0] += 1; diff[y0+ny] -= 1; }
				ovl.push_back(v); ovl_pile.push_back(pi);
			}
			int32_t cur = 0;
			for ( uint32_t y = 0; y < d.nwin; ++y )
			{
				cur += diff[y]; if ( static_cast<uint32_t>(cur) > maxdepth ) maxdepth = cur;
				// strings of the window (A + active overlaps, capped by -d) against what FastTier<1> holds
				uint64_t const nb = par.maxalign > 0 ? static_cast<uint64_t>(par.maxalign-1) : 0;
				uint64_t const mao = 1 + std::min<uint64_t>(static_cast<uint64_t>(cur),nb);
				uint64_t const perstr = par.w >= par.klow ? static_cast<uint64_t>(par.w-par.klow+1) : 1;
				if ( mao > FastTier<1>::maxs || mao*perstr > FastTier<1>::precap ) ++ndeepwin;
			}
			nwindows += d.nwin; npos += pilepos;
			fragbase.push_back(nfragslots); nfragslots += pilepos/100 + 2;
			piles.push_back(d);
		}
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
		ftier[1] = deep ? fastCapsOf< FastTier<2> >(tab_nrows,tab_nsup) : fastCapsOf< FastTier<6> >(tab_nrows,tab_nsup);
		ftier0 = fastCapsOf< FastTier<0> >(tab_nrows,tab_nsup);
		ftierL = fastCapsOf< FastTier<5> >(tab_nrows,tab_nsup);
		ftier[2] = fastCapsOf< FastTier<3> >(tab_nrows,tab_nsup);
		return DACC_OK;
	}
};

}
#endif
