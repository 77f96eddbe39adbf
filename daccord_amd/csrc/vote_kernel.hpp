/*
 * Pile vote: per A position, majority vote over the columns contributed by the windows that
 * cover it, run segmentation and fragment emission.
 * Replaces src/HandleContext.hpp:2541-2724 (sort of PileElements, -f fill, runs with gaps <= 1 and
 * last-first >= 100, right-to-left column vote with 'D' padding, emission).  The reference sorts
 * ~4*L PileElements per read; here nothing is sorted: every window record is already ordered by
 * (apos,apre) and position p is covered by at most w/a+1 windows, so one thread per position
 * gathers its columns directly from the records of those windows.
 *
 * Symbol codes: 0..3 = A,C,G,T, 4 = 'D', 5..8 = a,c,g,t (producefull fill).
 */
#ifndef DACC_VOTE_KERNEL_HPP
#define DACC_VOTE_KERNEL_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "window_main.hpp"

namespace dacc {

struct VoteFragment { uint32_t first, last, len; uint32_t pad; uint64_t off; };

struct VoteBatch
{
	DevParams P;
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen;
	DevPile const * piles; uint32_t npiles;
	uint8_t const * wrec;
	// per position slot (pile.posbase + p), p in [0,npos)
	uint8_t * has; uint16_t * ld0; uint8_t * oc; uint32_t * ocs;
	// per pile symbol stream and fragments
	uint8_t * outsym;         // capacity 2*npos+64 per pile at 2*posbase + 64*pileindex
	VoteFragment * frags;     // capacity npos/100+2 per pile at fragbase
	uint64_t const * fragbase;
	uint32_t * nfrag;         // per pile
	uint32_t * errflag;
};

DEV uint32_t pileNpos(DevPile const & pile) { return (pile.l > pile.rl ? pile.l : pile.rl) + 1; }

// ASCII code used as the vote tie-break (std::greater<pair<count,char>>, HandleContext.hpp:2695)
DEV uint32_t symAscii(uint32_t const s)
{
	switch ( s ) { case 0: return 'A'; case 1: return 'C'; case 2: return 'G'; case 3: return 'T'; case 4: return 'D';
		case 5: return 'a'; case 6: return 'c'; case 7: return 'g'; default: return 't'; }
}

struct Cover { uint8_t const * rec; uint32_t r; };

// windows of the pile whose records contribute at position p; returns count (<= 8 kept)
DEV uint32_t coveringWindows(VoteBatch const & B, DevPile const & pile, uint32_t const p, Cover * cov, uint32_t const cap)
{
	uint32_t const a = B.P.a, w = B.P.w, nwin = pile.nwin;
	uint32_t n = 0;
	if ( ! nwin ) return 0;
	uint32_t const ylo = (p > w) ? ((p-w + a-1)/a) : 0;
	uint32_t yhi = p / a; if ( yhi > nwin-1 ) yhi = nwin-1;
	bool lastseen = false;
	for ( uint32_t y = ylo; y <= yhi; ++y )
	{
		uint32_t s, e; windowInterval(pile.l,a,w,y,s,e);
		if ( y == nwin-1 ) lastseen = true;
		if ( s <= p && p <= e )
		{
			uint8_t const * rec = B.wrec + (pile.winbase+y)*WREC;
			if ( rec[0] == 1 ) { if ( n < cap ) { cov[n].rec = rec; cov[n].r = p-s; ++n; } else atomicOrFlag(B.errflag); }
		}
	}
	if ( !lastseen )
	{
		uint32_t s, e; windowInterval(pile.l,a,w,nwin-1,s,e);
		if ( s <= p && p <= e )
		{
			uint8_t const * rec = B.wrec + (pile.winbase+nwin-1)*WREC;
			if ( rec[0] == 1 ) { if ( n < cap ) { cov[n].rec = rec; cov[n].r = p-s; ++n; } else atomicOrFlag(B.errflag); }
		}
	}
	return n;
}

// pass 1: has[p], ld0[p]
DEV void votePass1(VoteBatch const & B, DevPile const & pile, uint32_t const p)
{
	Cover cov[24];
	uint32_t const w = B.P.w;
	uint32_t const nc = coveringWindows(B,pile,p,cov,24);
	uint32_t l0 = 0, T = 0;
	for ( uint32_t c = 0; c < nc; ++c )
	{
		uint8_t const * off = cov[c].rec+1;
		uint32_t const r = cov[c].r;
		uint32_t const sz = off[r+1]-off[r];
		uint32_t const nins = sz - (r < w ? 1 : 0);
		if ( r < w ) ++l0;
		T = nins > T ? nins : T;
	}
	bool has = (l0 || T);
	// -f fills uncovered positions with the lower-case A base, but only for a pile that has overlaps (ita != ite, HandleContext.hpp:2543)
	if ( !has && B.P.producefull && pile.novl && p < pile.rl ) { has = true; l0 = 1; }
	B.has[pile.posbase+p] = has;
	B.ld0[pile.posbase+p] = l0;
}

// pass 2: vote the columns of position p; if out != 0 write the emitted symbols; returns their number
DEV uint32_t votePass2(VoteBatch const & B, DevPile const & pile, uint32_t const p, uint8_t * out)
{
	if ( ! B.has[pile.posbase+p] ) return 0;
	Cover cov[24];
	uint32_t const w = B.P.w;
	uint32_t const nc = coveringWindows(B,pile,p,cov,24);
	uint32_t const npos = pileNpos(pile);
	// depth: count of the apre==0 column of p, else carried from the nearest such column to the right in the run
	int32_t depth = -1;
	for ( uint32_t q = p; q < npos && B.has[pile.posbase+q]; ++q )
		if ( B.ld0[pile.posbase+q] ) { depth = B.ld0[pile.posbase+q]; break; }
	uint32_t T = 0;
	for ( uint32_t c = 0; c < nc; ++c )
	{
		uint8_t const * off = cov[c].rec+1; uint32_t const r = cov[c].r;
		uint32_t const nins = (off[r+1]-off[r]) - (r < w ? 1 : 0);
		T = nins > T ? nins : T;
	}
	uint32_t no = 0;
	// insertion columns apre = -T .. -1
	for ( uint32_t t = T; t >= 1; --t )
	{
		uint32_t cnt[9] = {0,0,0,0,0,0,0,0,0}; uint32_t ld = 0;
		for ( uint32_t c = 0; c < nc; ++c )
		{
			uint8_t const * off = cov[c].rec+1; uint32_t const r = cov[c].r;
			uint8_t const * sym = cov[c].rec + 1 + (w+2);
			uint32_t const nins = (off[r+1]-off[r]) - (r < w ? 1 : 0);
			if ( nins >= t ) { ++cnt[sym[off[r]+nins-t]]; ++ld; }
		}
		if ( depth > static_cast<int32_t>(ld) ) cnt[4] += depth-ld;
		uint32_t best = 0, bestkey = 0;
		for ( uint32_t s = 0; s < 9; ++s )
		{
			uint32_t const key = (cnt[s]<<8) | symAscii(s);
			if ( key > bestkey ) { bestkey = key; best = s; }
		}
		if ( cnt[best] && best != 4 ) { if ( out ) out[no] = best; ++no; }
	}
	// apre == 0 column
	uint32_t const l0 = B.ld0[pile.posbase+p];
	if ( l0 )
	{
		uint32_t cnt[9] = {0,0,0,0,0,0,0,0,0};
		uint32_t real = 0;
		for ( uint32_t c = 0; c < nc; ++c )
		{
			uint8_t const * off = cov[c].rec+1; uint32_t const r = cov[c].r;
			uint8_t const * sym = cov[c].rec + 1 + (w+2);
			if ( r < w ) { ++cnt[sym[off[r+1]-1]]; ++real; }
		}
		if ( !real )
		{
			// producefull fill: lower-case A base (HandleContext.hpp:2556-2573)
			uint8_t const b = readBase(B.bps,B.boff[pile.aread],B.rlen[pile.aread],false,p);
			++cnt[5+b];
		}
		uint32_t best = 0, bestkey = 0;
		for ( uint32_t s = 0; s < 9; ++s )
		{
			uint32_t const key = (cnt[s]<<8) | symAscii(s);
			if ( key > bestkey ) { bestkey = key; best = s; }
		}
		if ( cnt[best] && best != 4 ) { if ( out ) out[no] = best; ++no; }
	}
	return no;
}

// runs of consecutive positions with elements, kept if last-first >= 100 (HandleContext.hpp:2590-2612);
// ocs = exclusive prefix of the per-position emitted counts; single thread per pile
DEV void voteRuns(VoteBatch const & B, DevPile const & pile, uint32_t const pileindex)
{
	uint32_t const npos = pileNpos(pile);
	uint64_t const symbase = 2*pile.posbase + 64ull*pileindex;
	VoteFragment * F = B.frags + B.fragbase[pileindex];
	uint32_t nf = 0;
	uint32_t p = 0;
	while ( p < npos )
	{
		if ( ! B.has[pile.posbase+p] ) { ++p; continue; }
		uint32_t q = p;
		while ( q+1 < npos && B.has[pile.posbase+q+1] ) ++q;
		if ( q-p >= 100 )
		{
			uint32_t const so = B.ocs[pile.posbase+p];
			uint32_t const eo = B.ocs[pile.posbase+q] + B.oc[pile.posbase+q];
			uint32_t const len = eo-so;
			if ( B.P.producefull || len >= B.P.minlen )
			{
				F[nf].first = p; F[nf].last = q; F[nf].len = len; F[nf].pad = 0; F[nf].off = symbase + so;
				++nf;
			}
		}
		p = q+1;
	}
	B.nfrag[pileindex] = nf;
}

}
#endif
