/*
 * Trace-point expansion, one thread per tspace block of one overlap.
 *
 * Replaces libmaus2's OverlapDataInterface::computeTrace as used by the reference at
 * src/HandleContext.hpp:1904-1917 (one global alignment of <= tspace A bases against the B span
 * given by the trace point, per block), followed by the advanceA / getStringLengthUsed
 * bookkeeping of :1936-1949 and :2005-2029.  The reference materialises the whole edit script
 * and walks it window by window; here each block is aligned bit-parallel (Myers, 128-bit
 * column vectors for tspace <= 128) with the vertical delta vectors of every column kept in a
 * per-thread HBM slab, and the traceback emits only what the windows need: the B offset P(x)
 * at every A position x that is a window start or end.  P(x) = bbpos + number of B symbols
 * consumed up to and including the (x-abpos)-th A-consuming step (trailing insertions belong
 * to the next window, exactly advanceA's stopping rule).
 *
 * Traceback priority: diagonal > DEL (A only) > INS (B only) -- the product's aligner
 * definition, identical to the oracle's (oracle/o_align.hpp).
 */
#ifndef DACC_TRACE_KERNEL_HPP
#define DACC_TRACE_KERNEL_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "window_main.hpp"

namespace dacc {

struct TraceBatch
{
	DevParams P;
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen;
	DevPile const * piles; DevOvl const * ovl; uint32_t const * ovl_pile;
	uint8_t const * trace;
	uint32_t const * blk_ovl; uint32_t const * blk_b0;
	uint64_t nblocks;
	uint32_t * wt_b; uint32_t * wt_e;
	uint64_t * colv;      // [maxcols+1][nthreads][4]  Pv0,Mv0,Pv1,Mv1 per column
	uint16_t * colbot;    // [maxcols+1][nthreads]
	uint32_t maxcols;
	uint32_t nthreads;
	uint32_t * errflag;
};

// write P(x) into the window tables of overlap o where x is a window start / end
DEV void emitBoundary(TraceBatch const & B, DevPile const & pile, DevOvl const & o, uint32_t const x, uint32_t const bpos)
{
	uint32_t const a = B.P.a, w = B.P.w, nwin = pile.nwin;
	if ( ! o.ny ) return;
	uint32_t const y0 = o.y0, y1 = o.y0 + o.ny; // [y0,y1)
	// window starts at x
	{
		uint32_t cand[2]; uint32_t nc = 0;
		if ( x % a == 0 ) cand[nc++] = x / a;
		cand[nc++] = nwin-1;
		for ( uint32_t c = 0; c < nc; ++c )
		{
			uint32_t const y = cand[c];
			if ( c == 1 && nc == 2 && cand[0] == y ) continue;
			if ( y >= y0 && y < y1 )
			{
				uint32_t s, e; windowInterval(pile.l,a,w,y,s,e);
				if ( s == x ) B.wt_b[o.wtoff + (y-y0)] = bpos;
			}
		}
	}
	// window ends at x
	if ( x >= w )
	{
		uint32_t const xs = x - w;
		uint32_t cand[2]; uint32_t nc = 0;
		if ( xs % a == 0 ) cand[nc++] = xs / a;
		cand[nc++] = nwin-1;
		for ( uint32_t c = 0; c < nc; ++c )
		{
			uint32_t const y = cand[c];
			if ( c == 1 && nc == 2 && cand[0] == y ) continue;
			if ( y >= y0 && y < y1 )
			{
				uint32_t s, e; windowInterval(pile.l,a,w,y,s,e);
				if ( e == x ) B.wt_e[o.wtoff + (y-y0)] = bpos;
			}
		}
	}
}

// tid = global thread id (slab index), task = block id
DEV void traceBlock(TraceBatch const & B, uint64_t const task, uint32_t const tid)
{
	uint32_t const oi = B.blk_ovl[task];
	DevOvl const o = B.ovl[oi];
	DevPile const pile = B.piles[B.ovl_pile[oi]];
	uint32_t const bi = static_cast<uint32_t>(task - o.blk0);
	int32_t const ts = B.P.tspace;
	int32_t const ai = (o.abpos/ts)*ts + static_cast<int32_t>(bi)*ts;
	uint32_t const a0 = ai > o.abpos ? ai : o.abpos;
	uint32_t const a1 = (ai+ts) < o.aepos ? (ai+ts) : o.aepos;
	uint32_t const m = a1-a0;
	uint32_t const b0 = B.blk_b0[task];
	uint32_t const n = B.trace[o.trace_off + 2*bi + 1];
	if ( m > 128 || n > B.maxcols || m == 0 ) { if ( m ) atomicOrFlag(B.errflag); return; }

	uint64_t const aoff = B.boff[pile.aread]; uint32_t const arl = B.rlen[pile.aread];
	uint64_t const boffs = B.boff[o.bread]; uint32_t const brl = B.rlen[o.bread];
	bool const inv = o.flags & 1;

	// pattern masks
	uint64_t peq[8] = {0,0,0,0,0,0,0,0};
	for ( uint32_t i = 0; i < m; ++i )
	{
		uint8_t const c = readBase(B.bps,aoff,arl,false,a0+i);
		peq[2*c + (i>>6)] |= 1ull<<(i&63);
	}
	uint64_t const mask0 = (m >= 64) ? ~0ull : ((1ull<<m)-1);
	uint64_t const mask1 = (m <= 64) ? 0ull : ((m == 128) ? ~0ull : ((1ull<<(m-64))-1));
	bool const two = m > 64;
	uint64_t Pv0 = mask0, Mv0 = 0, Pv1 = mask1, Mv1 = 0;
	uint32_t score = m;
	uint64_t const top = two ? (1ull<<(m-65)) : (1ull<<(m-1));
	uint64_t const stride = B.nthreads;
	for ( uint32_t c = 0; c < n; ++c )
	{
		uint8_t const tc = readBase(B.bps,boffs,brl,inv,b0+c);
		uint64_t const Eq0 = peq[2*tc], Eq1 = peq[2*tc+1];
		uint64_t const Xv0 = Eq0 | Mv0;
		uint64_t const Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
		uint64_t Ph0 = Mv0 | ~(Xh0 | Pv0);
		uint64_t Mh0 = Pv0 & Xh0;
		uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
		if ( !two ) { if ( Ph0 & top ) ++score; else if ( Mh0 & top ) --score; }
		Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
		Pv0 = (Mh0 | ~(Xv0 | Ph0)) & mask0;
		Mv0 = (Ph0 & Xv0) & mask0;
		if ( two )
		{
			uint64_t const Eq1c = Eq1 | mhc;
			uint64_t const Xv1 = Eq1 | Mv1;
			uint64_t const Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
			uint64_t Ph1 = Mv1 | ~(Xh1 | Pv1);
			uint64_t Mh1 = Pv1 & Xh1;
			if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
			Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
			Pv1 = (Mh1 | ~(Xv1 | Ph1)) & mask1;
			Mv1 = (Ph1 & Xv1) & mask1;
		}
		uint64_t * col = B.colv + ((static_cast<uint64_t>(c+1))*stride + tid)*4;
		col[0] = Pv0; col[1] = Mv0; col[2] = Pv1; col[3] = Mv1;
		B.colbot[(static_cast<uint64_t>(c+1))*stride + tid] = score;
	}
	// traceback
	uint32_t i = m, j = n, d = score;
	while ( i )
	{
		// column j vectors
		uint64_t cp0, cm0, cp1, cm1;
		if ( j ) { uint64_t const * col = B.colv + (static_cast<uint64_t>(j)*stride + tid)*4; cp0 = col[0]; cm0 = col[1]; cp1 = col[2]; cm1 = col[3]; }
		else { cp0 = mask0; cm0 = 0; cp1 = mask1; cm1 = 0; }
		bool done = false;
		if ( j )
		{
			// D[i-1][j-1] = bottom(j-1) - sum of vertical deltas of rows i..m in column j-1
			uint64_t qp0, qm0, qp1, qm1; uint32_t bot;
			if ( j-1 ) { uint64_t const * col = B.colv + (static_cast<uint64_t>(j-1)*stride + tid)*4; qp0 = col[0]; qm0 = col[1]; qp1 = col[2]; qm1 = col[3]; bot = B.colbot[static_cast<uint64_t>(j-1)*stride + tid]; }
			else { qp0 = mask0; qm0 = 0; qp1 = mask1; qm1 = 0; bot = m; }
			uint32_t const sh = i-1;
			int32_t sum;
			if ( sh < 64 ) sum = dacc_popc64(qp0>>sh) + dacc_popc64(qp1) - dacc_popc64(qm0>>sh) - dacc_popc64(qm1);
			else sum = dacc_popc64(qp1>>(sh-64)) - dacc_popc64(qm1>>(sh-64));
			uint32_t const dd = bot - sum;
			uint8_t const ca = readBase(B.bps,aoff,arl,false,a0+i-1);
			uint8_t const cb = readBase(B.bps,boffs,brl,inv,b0+j-1);
			uint32_t const neq = (ca != cb);
			if ( dd + neq == d )
			{
				emitBoundary(B,pile,o,a0+i,b0+j);
				--i; --j; d = dd; done = true;
			}
		}
		if ( !done )
		{
			uint32_t const r = i-1;
			bool const plus = (r < 64) ? ((cp0>>r)&1) : ((cp1>>(r-64))&1);
			if ( plus )
			{
				emitBoundary(B,pile,o,a0+i,b0+j);
				--i; d = d-1; done = true;
			}
		}
		if ( !done ) { --j; d = d-1; }
		(void)cm0; (void)cm1;
	}
	// the overlap's first A position: no step consumed yet, P(abpos) = bbpos
	if ( a0 == static_cast<uint32_t>(o.abpos) ) emitBoundary(B,pile,o,a0,b0);
}

}
#endif
