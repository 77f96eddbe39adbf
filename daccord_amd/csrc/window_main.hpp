/*
 * One window of one pile, end to end on one wavefront: active-set selection, B-window
 * extraction from the 2-bit read store via the per-overlap window tables written by the trace
 * kernel, length estimate, the k / filter-frequency / retry loop around the graph engine, and
 * the consensus->A alignment that yields the window's pile columns.
 * Mirrors the window body of HandleContext::operator() (src/HandleContext.hpp:1879-2500).
 */
#ifndef DACC_WINDOW_MAIN_HPP
#define DACC_WINDOW_MAIN_HPP
#include "dbg_window.hpp"

namespace dacc {

struct WindowBatch
{
	DevParams P;
	DevTables T;
	ArenaCaps C;
	// resident read store (.bps layout)
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen;
	// batch
	DevPile const * piles; uint32_t npiles;
	DevOvl const * ovl;
	uint32_t const * wt_b; uint32_t const * wt_e;   // per (overlap, active window): B start / end offset
	uint64_t nwindows;
	// outputs
	uint8_t * wrec;            // [nwindows][DACC_WREC_OF(w)]
	WindowOut * wout;          // [nwindows]
	uint8_t * arena;           // [gridDim][C.bytes]
	uint64_t * prof;           // optional per-phase cycle counters (profiling builds)
	uint32_t const * pregen;   // optional bit per window: already handed to the generic engine by the pre-scan (the LDS tiers skip it)
};

// base i of read r in the orientation the overlap uses (HandleContext.hpp:1910, 1952)
DEV uint8_t readBase(uint8_t const * bps, uint64_t const off, uint32_t const len, bool const inv, uint32_t const p)
{
	uint32_t const q = inv ? (len-1-p) : p;
	uint8_t const b = (bps[off+(q>>2)] >> (6-2*(q&3))) & 3;
	return inv ? (3-b) : b;
}

// window y of a pile schedule of length l (Windows::operator[], HandleContext.hpp:421-429)
DEV void windowInterval(uint32_t const l, uint32_t const a, uint32_t const w, uint32_t const y, uint32_t & s, uint32_t & e)
{
	if ( static_cast<uint64_t>(y)*a + w <= l ) { s = y*a; e = s+w; }
	else { s = l-w; e = l; }
}

DEV void processWindow(WindowBatch const & B, uint64_t const widx, uint8_t * arenabase)
{
	WindowEngine E;
	E.C = B.C; E.T = B.T; E.P = B.P;
	E.lane = wv_lane();
	E.flags = 0;
	// no member may carry a value over from the previous window (early returns on overflow skip assignments)
	E.mao = 0; E.k = 0; E.kmask = 0; E.npre = E.nlast = E.nn = 0; E.nmfirst = E.nmlast = 0; E.nstretch = E.nlinks = E.nsf = E.ncsf = E.nrl = 0;
	E.nrp = E.narp = E.np = E.nsiq = E.ncdh = E.nacc = E.conso = 0; E.maxsiq = E.maxlinks = 0;
	E.prof = B.prof;
#if defined(DACC_STATS)
	for ( int i = 0; i < 16; ++i ) E.st[i] = 0;
#endif
	PROF_T0
	arena_carve(E.A,arenabase,B.C,B.P.w);
	Arena & A = E.A;
	int const lane = E.lane;
	// the per-length path heaps are drained by every enumeration that completes, so their fill counts are zero from one pair to
	// the next -- but an enumeration that stops on a capacity overflow leaves them as they are, and the arena itself is whatever
	// the allocation or the previous layout (scratch retry with grown capacities) left there: start every window from zero
	for ( uint32_t i = lane; i < B.C.blcap; i += WSZ ) A.apq_n[i] = 0;

	// pile of this window: binary search over winbase
	uint32_t lo = 0, hi = B.npiles;
	while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( B.piles[mid].winbase <= widx ) lo = mid; else hi = mid; }
	DevPile const pile = B.piles[lo];
	uint32_t const y = static_cast<uint32_t>(widx - pile.winbase);
	uint32_t astart, aend;
	windowInterval(pile.l,B.P.a,B.P.w,y,astart,aend);

	WindowOut out; out.status = WS_INSUFFICIENT; out.mao = 0; out.elength = 0; out.k = 0; out.filterfreq = -1; out.conslen = 0; out.minrate = 0; out.flags = 0;
	uint8_t * rec = B.wrec + widx*DACC_WREC_OF(B.P.w);
	if ( lane == 0 ) rec[0] = 0;

	// ---- active set: overlaps with abpos <= astart and aepos >= aend (HandleContext.hpp:1904-1977) ----
	DevOvl const * ov = B.ovl + pile.first_ovl;
	uint32_t nact = 0;
	for ( uint32_t c = 0; c < pile.novl; c += WSZ )
	{
		uint32_t const z = c + lane;
		uint32_t act = 0;
		if ( z < pile.novl ) act = (ov[z].abpos <= static_cast<int32_t>(astart)) && (ov[z].aepos >= static_cast<int32_t>(aend));
		uint32_t tot; uint32_t const pre = wv_scan_excl(act,tot);
		if ( act && nact+pre < B.C.precap ) A.akeys[nact+pre] = (static_cast<uint64_t>(ov[z].ekey)<<32) | z;
		nact += tot;
	}
	if ( nact > B.C.precap ) { E.setOverflow(0x10000); nact = 0; }
	// order by (normalised error rate, z): the std::map iteration order of the reference (:1955-1959)
	uint32_t const ap2 = next_pow2(nact < 2 ? 2 : nact);
	for ( uint32_t i = nact + lane; i < ap2; i += WSZ ) A.akeys[i] = ~0ull;
	wv_sync();
	wv_bitonic_sort(A.akeys,ap2);
	// MA[0] = A window, then B windows while MAo < maxalign (:2033-2043)
	uint32_t mao = 0;
	if ( nact )
	{
		uint64_t const nb = (B.P.maxalign > 0) ? (B.P.maxalign-1) : 0;
		mao = 1 + static_cast<uint32_t>(nact < nb ? nact : nb);
	}
	if ( mao > B.C.maxs ) { E.setOverflow(0x20000); mao = 0; }
	E.mao = mao;
	out.mao = mao;

	// ---- strings ----
	if ( mao )
	{
		uint64_t const aoff = B.boff[pile.aread];
		for ( uint32_t p = lane; p < B.P.w; p += WSZ )
			A.str[p] = readBase(B.bps,aoff,B.rlen[pile.aread],false,astart+p);
		if ( lane == 0 ) A.slen[0] = B.P.w;
		for ( uint32_t j = 1; j < mao; ++j )
		{
			uint32_t const z = static_cast<uint32_t>(A.akeys[j-1] & 0xFFFFFFFFu);
			DevOvl const & o = ov[z];
			uint64_t const row = o.wtoff + (y - o.y0);
			uint32_t const bs = B.wt_b[row], be = B.wt_e[row];
			uint32_t const len = be-bs;
			if ( len > B.C.lstr ) { E.setOverflow(0x40000); if ( lane == 0 ) A.slen[j] = 0; continue; }
			uint64_t const off = B.boff[o.bread]; uint32_t const rl = B.rlen[o.bread]; bool const inv = o.flags & 1;
			for ( uint32_t p = lane; p < len; p += WSZ )
				A.str[static_cast<uint64_t>(j)*B.C.lstr+p] = readBase(B.bps,off,rl,inv,bs+p);
			if ( lane == 0 ) A.slen[j] = len;
		}
	}
	wv_sync();
	E.flags = wv_or(E.flags);
	PROF(E,0)

	int32_t elength = 0;
	if ( mao && !E.flags )
	{
		E.buildPeq();
		elength = E.estimateLength() + 1;
	}
	out.elength = elength;
	PROF(E,1)

	if ( mao >= B.P.minwindowcov && !E.flags )
	{
		bool pathfailed = true;
		uint64_t minrate = B.P.eminrate;
		bool haveMin = false;
		uint32_t bestlen = 0;
		// the accepted consensus is kept at the tail of the candidate text buffer
		uint8_t * best = A.cons + (B.C.conscap - DACC_MAXCONS_OF(B.P.w));
		for ( uint32_t k = B.P.klow; k <= B.P.khigh && !E.flags; ++k )
		{
			E.k = k; E.kmask = (k >= 32) ? ~0ull : ((1ull<<(2*k))-1);
			for ( int32_t ff = B.P.maxff; ff >= B.P.minff && !E.flags; --ff )
			{
				// setup + filterFreq + computeFeasibleKmerPositions (:2211-2228)
				PROF_T0
				E.buildInstances();
				E.flags = wv_or(E.flags); if ( E.flags ) break;
				PROF(E,2)
				E.buildNodes(ff > 1 ? ff : 1);
				E.flags = wv_or(E.flags); if ( E.flags ) break;
				PROF(E,3)
				E.buildSuccessors(mao);
				PROF(E,4)
				E.computeFeasible();
				E.flags = wv_or(E.flags); if ( E.flags ) break;
				PROF(E,5)
				if ( ff == 0 )
				{
					// gap filling (:2233-2268)
					E.levelSuccessors2();
					E.flags = wv_or(E.flags); if ( E.flags ) break;
					E.buildNodes(1);
					E.flags = wv_or(E.flags); if ( E.flags ) break;
					E.buildSuccessors(mao);
					E.computeFeasible();
				}
				PROF(E,6)
				E.buildFirstLast();
				PROF(E,7)
				E.flags = wv_or(E.flags);
				if ( E.flags ) break;
				uint32_t mintry = 0; bool lconsok = false;
				while ( true )
				{
					bool const consok = E.traverse(static_cast<int64_t>(elength)-4,static_cast<int64_t>(elength)+4);
					E.flags = wv_or(E.flags);
					if ( E.flags ) break;
					if ( consok )
					{
						// checkCandidatesU: summed edit distance of candidate 0 (:5476-5482) == its T9 error
						uint64_t const err = static_cast<uint64_t>(A.accerr[0]);
						if ( err < minrate )
						{
							lconsok = true; minrate = err; haveMin = true;
							bestlen = A.acc[0].l;
							if ( bestlen > DACC_MAXCONS_OF(B.P.w) ) { E.setOverflow(0x80000); break; }
							for ( uint32_t i = lane; i < bestlen; i += WSZ ) best[i] = A.cons[A.acc[0].o+i];
							out.k = k; out.filterfreq = ff;
							wv_sync();
						}
						else if ( haveMin ) lconsok = true;
						break;
					}
					else
					{
						if ( ++mintry >= 3 ) break;
					}
					if ( !E.addNextFromHeap() ) break;
				}
				E.flags = wv_or(E.flags);
				if ( lconsok ) { pathfailed = false; break; }
			}
		}
		if ( !E.flags )
		{
			if ( !pathfailed )
			{
				out.status = WS_OK; out.conslen = bestlen; out.minrate = minrate;
				{ PROF_T0 if ( lane == 0 ) E.alignAndEmit(best,bestlen,rec); PROF(E,14) }
			}
			else out.status = WS_FAILED;
		}
	}
	E.flags = wv_or(E.flags);
	if ( E.flags ) { out.status = WS_OVERFLOW; out.flags = E.flags; if ( lane == 0 ) rec[0] = 0; }
	if ( lane == 0 ) B.wout[widx] = out;
#if defined(DACC_STATS)
	for ( int i = 0; i < 16; ++i ) dacc_stats_sink(widx,i,E.st[i]);
#endif
	wv_sync();
}

}
#endif
