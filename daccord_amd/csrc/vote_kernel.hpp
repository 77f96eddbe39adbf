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
	uint8_t const * pilebad;  // per pile: dropped (a window could not be processed), 0 = no such list
};

// output letter of a vote winner: 0-3 = ACGT, 4 = D, 5-8 = acgt (lower case: producefull fill).  The symbol stream is
// ASCII on the device so that the host copies fragments instead of translating 10^8 symbols per batch.
DEV uint8_t voteSym(uint32_t const best)
{
	return static_cast<uint8_t>(best < 4 ? ((0x54474341u >> (8*best)) & 0xFFu) : (best == 4 ? 0x44u : ((0x74676361u >> (8*(best-5))) & 0xFFu)));
}

DEV uint32_t pileNpos(DevPile const & pile) { return (pile.l > pile.rl ? pile.l : pile.rl) + 1; }

// ASCII code used as the vote tie-break (std::greater<pair<count,char>>, HandleContext.hpp:2695)
DEV uint32_t symAscii(uint32_t const s)
{
	switch ( s ) { case 0: return 'A'; case 1: return 'C'; case 2: return 'G'; case 3: return 'T'; case 4: return 'D';
		case 5: return 'a'; case 6: return 'c'; case 7: return 'g'; default: return 't'; }
}

// The windows of the pile whose records contribute at position p, in window order (the snapped last window comes
// last): an iterator instead of a list, so that nothing of it lives in scratch memory.
struct CoverIt { uint32_t y, yhi; uint32_t state; };     // state 0: regular windows, 1: the last window is still due, 2: done
// records of windows [y0,y0+n) of the pile staged in LDS by the workgroup (n == 0: none, read them from HBM/L2)
struct VoteTile { LDSQ uint8_t const * stage; uint32_t y0, n; };
// a window record, either in the LDS stage or in HBM (no generic pointers: the two address spaces stay apart)
struct VRec { uint8_t const * g; LDSQ uint8_t const * l; bool lds; };     // an explicit flag: LDS address 0 is a valid address
DEV uint32_t vrb(VRec const & R, uint32_t const i) { return R.lds ? static_cast<uint32_t>(R.l[i]) : static_cast<uint32_t>(R.g[i]); }
// offset of group r and symbol o of a record: narrow layout (w <= 64) or wide (dev_types.hpp)
DEV uint32_t vroff(VRec const & R, uint32_t const w, uint32_t const r) { return DACC_WIDE_W(w) ? (vrb(R,2+2*r) | (vrb(R,3+2*r)<<8)) : vrb(R,1+r); }
DEV uint32_t vrsym(VRec const & R, uint32_t const w, uint32_t const o) { return vrb(R,(DACC_WIDE_W(w) ? 2+2*(w+2) : 1+(w+2)) + o); }
DEV VRec voteRecord(VoteBatch const & B, DevPile const & pile, VoteTile const & VT, uint32_t const y)
{
	uint32_t const d = y - VT.y0;
	VRec R;
	if ( d < VT.n ) { R.l = VT.stage + d*WREC; R.g = B.wrec; R.lds = true; } else { R.l = VT.stage; R.g = B.wrec + (pile.winbase+y)*DACC_WREC_OF(B.P.w); R.lds = false; }     // (wide records are never staged)
	return R;
}
DEV void coverBegin(VoteBatch const & B, DevPile const & pile, uint32_t const p, CoverIt & it)
{
	uint32_t const a = B.P.a, w = B.P.w, nwin = pile.nwin;
	if ( ! nwin ) { it.y = 1; it.yhi = 0; it.state = 2; return; }
	it.y = (p > w) ? ((p-w + a-1)/a) : 0;
	it.yhi = p / a; if ( it.yhi > nwin-1 ) it.yhi = nwin-1;
	it.state = (it.y <= it.yhi && it.yhi == nwin-1) ? 2 : 1;     // the regular range already ends with the last window
	if ( it.y > it.yhi ) it.state = 1;
}
DEV bool coverNext(VoteBatch const & B, DevPile const & pile, VoteTile const & VT, uint32_t const p, CoverIt & it, VRec & rec, uint32_t & r)
{
	uint32_t const a = B.P.a, w = B.P.w, nwin = pile.nwin;
	while ( it.y <= it.yhi )
	{
		uint32_t const y = it.y++;
		uint32_t s, e; windowInterval(pile.l,a,w,y,s,e);
		if ( s <= p && p <= e )
		{
			VRec const rc = voteRecord(B,pile,VT,y);
			if ( vrb(rc,0) == 1 ) { rec = rc; r = p-s; return true; }
		}
	}
	if ( it.state == 1 )
	{
		it.state = 2;
		uint32_t s, e; windowInterval(pile.l,a,w,nwin-1,s,e);
		if ( s <= p && p <= e )
		{
			VRec const rc = voteRecord(B,pile,VT,nwin-1);
			if ( vrb(rc,0) == 1 ) { rec = rc; r = p-s; return true; }
		}
	}
	return false;
}

// pass 1: has[p], ld0[p]
DEV void votePass1(VoteBatch const & B, DevPile const & pile, uint32_t const p, VoteTile const VT = VoteTile{0,0,0})
{
	uint32_t const w = B.P.w;
	uint32_t l0 = 0, T = 0;
	CoverIt it; coverBegin(B,pile,p,it);
	VRec rec; uint32_t r;
	while ( coverNext(B,pile,VT,p,it,rec,r) )
	{
		uint32_t const sz = vroff(rec,w,r+1)-vroff(rec,w,r);
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

// winner of a column: maximum (count, ASCII code), HandleContext.hpp:2695; counts packed 7 bits per symbol would
// overflow for deep piles, so nine scalars (registers, the loops below are fully unrolled)
struct ColCnt { uint32_t c0, c1, c2, c3, c4, c5, c6, c7, c8; };
DEV void colClear(ColCnt & C) { C.c0 = C.c1 = C.c2 = C.c3 = C.c4 = C.c5 = C.c6 = C.c7 = C.c8 = 0; }
DEV void colAdd(ColCnt & C, uint32_t const s, uint32_t const v)
{
	C.c0 += (s == 0) ? v : 0; C.c1 += (s == 1) ? v : 0; C.c2 += (s == 2) ? v : 0; C.c3 += (s == 3) ? v : 0; C.c4 += (s == 4) ? v : 0;
	C.c5 += (s == 5) ? v : 0; C.c6 += (s == 6) ? v : 0; C.c7 += (s == 7) ? v : 0; C.c8 += (s >= 8) ? v : 0;
}
DEV bool colWinner(ColCnt const & C, uint32_t & best)
{
	uint32_t bk = 0; best = 0;
	#define DACC_CW(cnt_,s_,ch_) { uint32_t const key = ((cnt_)<<8) | (ch_); if ( key > bk ) { bk = key; best = s_; } }
	DACC_CW(C.c0,0,'A') DACC_CW(C.c1,1,'C') DACC_CW(C.c2,2,'G') DACC_CW(C.c3,3,'T') DACC_CW(C.c4,4,'D')
	DACC_CW(C.c5,5,'a') DACC_CW(C.c6,6,'c') DACC_CW(C.c7,7,'g') DACC_CW(C.c8,8,'t')
	#undef DACC_CW
	return (bk>>8) != 0 && best != 4;
}

// pass 2: vote the columns of position p; if out != 0 write the emitted symbols; returns their number
DEV uint32_t votePass2(VoteBatch const & B, DevPile const & pile, uint32_t const p, uint8_t * out, VoteTile const VT = VoteTile{0,0,0})
{
	if ( ! B.has[pile.posbase+p] ) return 0;
	uint32_t const w = B.P.w;
	uint32_t const npos = pileNpos(pile);
	// depth: count of the apre==0 column of p, else carried from the nearest such column to the right in the run
	int32_t depth = -1;
	for ( uint32_t q = p; q < npos && B.has[pile.posbase+q]; ++q )
		if ( B.ld0[pile.posbase+q] ) { depth = B.ld0[pile.posbase+q]; break; }
	uint32_t T = 0;
	{
		CoverIt it; coverBegin(B,pile,p,it);
		VRec rec; uint32_t r;
		while ( coverNext(B,pile,VT,p,it,rec,r) )
		{
			uint32_t const nins = (vroff(rec,w,r+1)-vroff(rec,w,r)) - (r < w ? 1 : 0);
			T = nins > T ? nins : T;
		}
	}
	uint32_t no = 0;
	// insertion columns apre = -T .. -1
	for ( uint32_t t = T; t >= 1; --t )
	{
		ColCnt C; colClear(C); uint32_t ld = 0;
		CoverIt it; coverBegin(B,pile,p,it);
		VRec rec; uint32_t r;
		while ( coverNext(B,pile,VT,p,it,rec,r) )
		{
			uint32_t const o0 = vroff(rec,w,r);
			uint32_t const nins = (vroff(rec,w,r+1)-o0) - (r < w ? 1 : 0);
			if ( nins >= t ) { colAdd(C,vrsym(rec,w,o0+nins-t),1); ++ld; }
		}
		if ( depth > static_cast<int32_t>(ld) ) C.c4 += depth-ld;
		uint32_t best;
		if ( colWinner(C,best) ) { if ( out ) out[no] = voteSym(best); ++no; }
	}
	// apre == 0 column
	uint32_t const l0 = B.ld0[pile.posbase+p];
	if ( l0 )
	{
		ColCnt C; colClear(C);
		uint32_t real = 0;
		CoverIt it; coverBegin(B,pile,p,it);
		VRec rec; uint32_t r;
		while ( coverNext(B,pile,VT,p,it,rec,r) )
		{
			if ( r < w ) { colAdd(C,vrsym(rec,w,vroff(rec,w,r+1)-1),1); ++real; }
		}
		if ( !real )
		{
			// producefull fill: lower-case A base (HandleContext.hpp:2556-2573)
			uint8_t const b = readBase(B.bps,B.boff[pile.aread],B.rlen[pile.aread],false,p);
			colAdd(C,5+b,1);
		}
		uint32_t best;
		if ( colWinner(C,best) ) { if ( out ) out[no] = voteSym(best); ++no; }
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
