/*
 * Trace-point expansion, one thread per tspace block of one overlap.
 *
 * Replaces libmaus2's OverlapDataInterface::computeTrace as used by the reference at
 * src/HandleContext.hpp:1904-1917 (one global alignment of <= tspace A bases against the B span
 * given by the trace point, per block), followed by the advanceA / getStringLengthUsed
 * bookkeeping of :1936-1949 and :2005-2029.  The reference materialises the whole edit script
 * and walks it window by window; here each block is aligned bit-parallel (Myers, 128-bit
 * column vectors for tspace <= 128); the forward pass keeps every 16th column in LDS and the
 * traceback recomputes the 16 columns of a segment when it enters it, so the alignment matrix
 * never leaves the CU.  The traceback emits only what the windows need: the B offset P(x)
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
	uint32_t maxcols;     // largest B span of a block in the batch
	uint32_t trace_bytes; // 1 (tspace <= 125) or 2 bytes per trace value
	uint32_t * errflag;
	uint64_t * slab;      // two word kernel: checkpoint slabs, traceSlabWords(maxcols) 64 bit words per workgroup
	uint32_t * work;      // two word kernel: counter the workgroups draw their rounds of 64 blocks from (0: grid stride)
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

// One column of the bit-parallel alignment: vertical delta vectors (two 64-bit words for up to 128 A rows).
struct TCol { uint64_t pv0, mv0, pv1, mv1; };

// Column stores.  The forward pass keeps only every T2C-th column (checkpoints); the traceback, which walks the columns
// downwards, recomputes the columns of a segment from the checkpoint at or before it when it enters the segment.
//
// Two word kernel (k_trace, tspace <= 128; round 3): the CHECKPOINTS of a lane go to a global scratch slab of its workgroup
// (coalesced 512 byte rows, written once and read once per block; the slabs of all resident workgroups stay in the L2 /
// Infinity Cache), only the segment being walked is in LDS.  With checkpoints AND segment in LDS (58.7 KB, rounds 1-2) two
// wavefronts shared a CU -- two of its four SIMDs had no wavefront at all.
//
// (round 6) What a segment column holds is what the traceback DECIDES on, not the delta vectors: with D = the edit distance matrix,
// vd(i,j) = D[i][j]-D[i-1][j] and hd(i,j) = D[i][j]-D[i][j-1],
//     D[i-1][j-1] = D[i][j] - vd(i,j) - hd(i-1,j),
// so the diagonal step (priority 1) is taken iff A[i-1] == B[j-1] (then D[i-1][j-1] == D[i][j] always) or vd(i,j) + hd(i-1,j) == 1, i.e.
// (vd = +1 and hd = 0) or (vd = 0 and hd = +1); DEL (priority 2) iff vd(i,j) = +1.  Both are bits of the column step's own vectors (the
// shifted horizontal deltas Ph / Mh ARE hd(i-1,j) at bit i-1), so a recomputed column stores DIAG = Eq | (Pv & ~(Ph|Mh)) | (~Pv & ~Mv & Ph)
// and Pv: 32 bytes per lane and column as before, but a traceback step is two 32 bit LDS reads and two bit tests instead of the five reads,
// two 64 bit shifts and twelve population counts that rebuilt D[i-1][j-1] from the bottom row's score (rounds 1-5); no score is carried at
// all, and the segment's first column (slot 0 of the old layout) is not needed.  The decisions are the same by construction: the old test
// `D[i-1][j-1] + (A[i-1] != B[j-1]) == D[i][j]` is the DIAG bit.  k_trace was the one kernel of the step near its instruction bound
// (VALU 55 % of the SIMDs' slots at 8 wavefronts per CU): 64 k -> 45 k VALU instructions per wavefront and block round.
enum { TRS = 16 };      // wide kernels (k_trace_wide): checkpoints and segment in LDS
// two word kernel: T2C columns between checkpoints (global slab), T2S columns per segment (LDS).  The traceback enters a segment by
// recomputing from the checkpoint at or before it: the columns in front of the segment are stepped over without being stored.
// T2S < T2C trades recomputed columns for LDS bytes = resident wavefronts.
// Measured with the old column format (profiles/r06p, 3000 reads of config 2, 5.4 M blocks): segments of 8 / 4 / 2 columns = 8 / 14 / 16
// wavefronts per CU: 16.3 / 13.9 / 17.7 ms (checkpoints every 4 columns with segments of 4: 15.9 ms -- twice the slab traffic for the same LDS).
#if !defined(DACC_T2C)
#define DACC_T2C 8
#endif
#if !defined(DACC_T2S)
#define DACC_T2S 4
#endif
enum { T2C = DACC_T2C, T2S = DACC_T2S };
static_assert(T2C % T2S == 0 && T2C <= 16,"segments tile the checkpoint groups; a group's B symbols are kept packed in 32 bits");
// segment slots of the 64 lanes of a wavefront interleaved in 32 bit units: field f (0-3: DIAG bits 0-31, 32-63, 64-95, 96-127; 4-7: Pv) of
// slot u of lane l at (u*8+f)*64 + l (no bank conflicts whatever slot and row a lane is at); slot u = column g*T2S+1+u
struct TraceStoreGL
{
	enum : uint32_t { SEG = T2S, CPS = T2C };
	uint64_t * g;                 // this workgroup's slab: word q (Pv0, Mv0, Pv1, Mv1) of checkpoint e of lane l at (e*4+q)*64 + l
	LDSQ uint32_t * w; uint32_t lane;
	DEV void putCp(uint32_t const e, TCol const & c) const
	{
		uint64_t * p = g + (e*4)*64 + lane;
		p[0] = c.pv0; p[64] = c.mv0; p[128] = c.pv1; p[192] = c.mv1;
	}
	DEV TCol getCp(uint32_t const e) const
	{
		uint64_t const * p = g + (e*4)*64 + lane;
		TCol c; c.pv0 = p[0]; c.mv0 = p[64]; c.pv1 = p[128]; c.mv1 = p[192]; return c;
	}
	DEV void putSeg(uint32_t const u, uint64_t const d0, uint64_t const d1, uint64_t const p0, uint64_t const p1) const
	{
		LDSQ uint32_t * p = w + (u*8)*64 + lane;
		p[0] = static_cast<uint32_t>(d0); p[64] = static_cast<uint32_t>(d0>>32); p[128] = static_cast<uint32_t>(d1); p[192] = static_cast<uint32_t>(d1>>32);
		p[256] = static_cast<uint32_t>(p0); p[320] = static_cast<uint32_t>(p0>>32); p[384] = static_cast<uint32_t>(p1); p[448] = static_cast<uint32_t>(p1>>32);
	}
	// bits 32*(r>>5) .. +31 of the DIAG / the Pv vector of slot u
	DEV uint32_t segDiag(uint32_t const u, uint32_t const r) const { return w[(u*8 + (r>>5))*64 + lane]; }
	DEV uint32_t segPv(uint32_t const u, uint32_t const r) const { return w[(u*8 + 4 + (r>>5))*64 + lane]; }
};
// host emulation: plain arrays of one thread (cp: traceCheckpoints(maxcols) columns, seg: 4 words (DIAG, Pv) per segment column)
struct TraceStoreMem
{
	enum : uint32_t { SEG = T2S, CPS = T2C };
	TCol * cp; uint64_t * seg;
	void putCp(uint32_t const e, TCol const & c) const { cp[e] = c; }
	TCol getCp(uint32_t const e) const { return cp[e]; }
	void putSeg(uint32_t const u, uint64_t const d0, uint64_t const d1, uint64_t const p0, uint64_t const p1) const { seg[4*u] = d0; seg[4*u+1] = d1; seg[4*u+2] = p0; seg[4*u+3] = p1; }
	uint32_t segDiag(uint32_t const u, uint32_t const r) const { return static_cast<uint32_t>(seg[4*u + (r>>6)] >> (32*((r>>5)&1))); }
	uint32_t segPv(uint32_t const u, uint32_t const r) const { return static_cast<uint32_t>(seg[4*u + 2 + (r>>6)] >> (32*((r>>5)&1))); }
};
HDEV uint32_t traceSlots(uint32_t const maxcols) { return maxcols/TRS + 1 + TRS; }     // wide kernels: checkpoints 0,TRS,2*TRS,.. + one segment
HDEV uint32_t traceCheckpoints(uint32_t const maxcols) { return maxcols/T2C + 1; }     // two word kernel: checkpoints per lane (global slab)
HDEV uint32_t traceSlabWords(uint32_t const maxcols) { return traceCheckpoints(maxcols)*4u*64u; }      // 64 bit words per workgroup
enum : uint32_t { TRACE2_LDS = T2S*64*32 };      // LDS bytes of a wavefront of k_trace

// Myers / Hyyro column step: C = column c -> column c+1 for B symbol tc.  DIAGS: also the traceback's decision bits of the new column
// (dg0 / dg1, see above).  No masking of the rows beyond the block's m: carries only travel upwards and no row >= m is ever read.
template<bool DIAGS>
DEV void traceStep(uint64_t const * peq, uint8_t const tc, TCol & C, bool const two, uint64_t & dg0, uint64_t & dg1)
{
	uint64_t const Eq0 = peq[2*tc], Eq1 = peq[2*tc+1];
	uint64_t const Xv0 = Eq0 | C.mv0;
	uint64_t const Xh0 = (((Eq0 & C.pv0) + C.pv0) ^ C.pv0) | Eq0;
	uint64_t Ph0 = C.mv0 | ~(Xh0 | C.pv0);
	uint64_t Mh0 = C.pv0 & Xh0;
	uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
	Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;      // bit r: horizontal delta of row r (row 0 = the top boundary: +1, global alignment)
	C.pv0 = Mh0 | ~(Xv0 | Ph0);
	C.mv0 = Ph0 & Xv0;
	if ( DIAGS ) dg0 = Eq0 | (C.pv0 & ~(Ph0|Mh0)) | (~(C.pv0|C.mv0) & Ph0);
	if ( two )
	{
		uint64_t const Eq1c = Eq1 | mhc;
		uint64_t const Xv1 = Eq1 | C.mv1;
		uint64_t const Xh1 = (((Eq1c & C.pv1) + C.pv1) ^ C.pv1) | Eq1c;
		uint64_t Ph1 = C.mv1 | ~(Xh1 | C.pv1);
		uint64_t Mh1 = C.pv1 & Xh1;
		Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
		C.pv1 = Mh1 | ~(Xv1 | Ph1);
		C.mv1 = Ph1 & Xv1;
		if ( DIAGS ) dg1 = Eq1 | (C.pv1 & ~(Ph1|Mh1)) | (~(C.pv1|C.mv1) & Ph1);
	}
}

// task = block id; ST = column store of this thread (traceCheckpoints(B.maxcols) checkpoints, ST::SEG segment columns)
template<typename ST>
DEV void traceBlock(TraceBatch const & B, uint64_t const task, ST const & st)
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
	uint32_t const n = B.trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(B.trace)[o.trace_off + 2*bi + 1] : B.trace[o.trace_off + 2*bi + 1];
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
	constexpr uint32_t TRS = ST::SEG;      // columns per segment (shadows the wide kernels' constant)
	constexpr uint32_t CPS = ST::CPS;      // columns between checkpoints
	TCol C; C.pv0 = mask0; C.mv0 = 0; C.pv1 = mask1; C.mv1 = 0;
	st.putCp(0,C);
	uint64_t dg0 = 0, dg1 = 0;
	// B symbols are fetched CPS at a time (independent loads, one wait) and kept packed 2 bits each
	#define DACC_LOADB(c0_,cnt_,dst_) { dst_ = 0; _Pragma("unroll") for ( uint32_t u = 0; u < CPS; ++u ) if ( u < (cnt_) ) dst_ |= static_cast<uint32_t>(readBase(B.bps,boffs,brl,inv,b0+(c0_)+u)) << (2*u); }
	for ( uint32_t c0 = 0; c0 < n; c0 += CPS )
	{
		uint32_t const cnt = (n-c0 < CPS) ? (n-c0) : static_cast<uint32_t>(CPS);
		uint32_t bb; DACC_LOADB(c0,cnt,bb)
		for ( uint32_t u = 0; u < cnt; ++u ) traceStep<false>(peq,(bb>>(2*u))&3,C,two,dg0,dg1);
		if ( cnt == CPS ) st.putCp(c0/CPS+1,C);
	}
	// window boundaries are rare among the A positions: x is a window start iff x % a == 0 or x == l-w, a window end iff
	// (x-w) % a == 0 or x == l; x % a is tracked incrementally (no division per step)
	uint32_t const wa = B.P.w % B.P.a, lw = pile.l >= B.P.w ? pile.l - B.P.w : 0xFFFFFFFFu;
	uint32_t xa = (a0+m) % B.P.a;
	#define DACC_EMIT(x_,b_) { uint32_t const xx_ = (x_); if ( xa == 0 || xa == wa || xx_ == lw || xx_ == pile.l ) emitBoundary(B,pile,o,xx_,b_); }
	#define DACC_XDEC { xa = xa ? xa-1 : B.P.a-1; }
	// traceback.  Step (i,j) with j >= 1 reads the decision bits of column j, which lies in segment (j-1)/TRS.  The segments are visited in
	// a loop that is uniform over the wavefront (top segment of the batch downwards), so the lanes recompute their segments in lock step
	// instead of one lane at a time.
	uint32_t i = m, j = n;
	for ( int32_t g = static_cast<int32_t>((B.maxcols ? B.maxcols-1 : 0)/TRS); g >= 0; --g )
	{
		if ( !(i && j && (j-1)/TRS == static_cast<uint32_t>(g)) ) continue;
		{
			// checkpoint at or before the segment's first column; the `skip` columns between them are stepped over
			uint32_t const cpi = (static_cast<uint32_t>(g)*TRS)/CPS, cp0 = cpi*CPS, skip = static_cast<uint32_t>(g)*TRS - cp0;
			TCol R = st.getCp(cpi);
			uint32_t const cntall = (n-cp0 < CPS) ? (n-cp0) : static_cast<uint32_t>(CPS);
			uint32_t bcp; DACC_LOADB(cp0,cntall,bcp)
			if ( CPS != TRS ) for ( uint32_t u = 0; u < skip; ++u ) traceStep<false>(peq,(bcp>>(2*u))&3,R,two,dg0,dg1);
			uint32_t const bseg = bcp >> (2*skip);
			uint32_t const c0 = g*TRS, cnt = (n-c0 < TRS) ? (n-c0) : static_cast<uint32_t>(TRS);
			for ( uint32_t u = 0; u < cnt; ++u )
			{
				traceStep<true>(peq,(bseg>>(2*u))&3,R,two,dg0,dg1);
				st.putSeg(u,dg0,dg1,R.pv0,R.pv1);      // column c0+u+1
			}
		}
		while ( i && j && (j-1)/TRS == static_cast<uint32_t>(g) )
		{
			uint32_t const u = (j-1) - g*TRS, r = i-1;
			uint32_t const dw = st.segDiag(u,r), pw = st.segPv(u,r);
			bool const diag = (dw >> (r&31)) & 1, del = (pw >> (r&31)) & 1;
			if ( diag || del ) { DACC_EMIT(a0+i,b0+j) --i; DACC_XDEC }
			if ( diag || !del ) --j;
		}
	}
	// column 0: only A symbols are left (vertical deltas of the first column are all +1)
	while ( i ) { DACC_EMIT(a0+i,b0) --i; DACC_XDEC }
	#undef DACC_EMIT
	#undef DACC_XDEC
	#undef DACC_LOADB
	// the overlap's first A position: no step consumed yet, P(abpos) = bbpos
	if ( a0 == static_cast<uint32_t>(o.abpos) ) emitBoundary(B,pile,o,a0,b0);
}


// ---------------------------------------------------------------------------------------------------------------------
// Wide blocks: tspace in (128, 64*NW].  The reference takes whatever trace spacing the LAS file has (daccord.cpp:1375);
// DALIGNER writes two byte trace values from tspace 126 on.  Same algorithm as traceBlock with NW 64-bit words per column
// vector (block-wise Myers: the horizontal deltas of the top row of a word are the carries of the word below); kept apart
// from the two word version above, which is the one every default run uses.  The column store is interleaved over the
// `nl` lanes of the workgroup that take blocks (fewer than 64 when 64 lanes' checkpoints would not fit LDS).
template<int NW> struct TColW { uint64_t pv[NW], mv[NW]; uint32_t score; };
template<int NW>
struct TraceStoreW
{
	LDSQ uint64_t * w; LDSQ uint16_t * sc; uint32_t lane, nl;
	DEV void put(uint32_t const e, TColW<NW> const & c) const
	{
		LDSQ uint64_t * p = w + static_cast<size_t>(e)*(2*NW)*nl + lane;
		#pragma unroll
		for ( int q = 0; q < NW; ++q ) { p[(2*q)*nl] = c.pv[q]; p[(2*q+1)*nl] = c.mv[q]; }
		sc[e*nl+lane] = static_cast<uint16_t>(c.score);
	}
	DEV TColW<NW> get(uint32_t const e) const
	{
		LDSQ uint64_t const * p = w + static_cast<size_t>(e)*(2*NW)*nl + lane;
		TColW<NW> c;
		#pragma unroll
		for ( int q = 0; q < NW; ++q ) { c.pv[q] = p[(2*q)*nl]; c.mv[q] = p[(2*q+1)*nl]; }
		c.score = sc[e*nl+lane]; return c;
	}
};
template<int NW> HDEV uint32_t traceWideBytesPerLane(uint32_t const maxcols) { return traceSlots(maxcols)*(16u*NW+2u); }
// word `idx` (run time) of a register array: a select chain over the compile time indices (no scratch)
template<int NW> DEV uint64_t traceWord(uint64_t const (&a)[NW], uint32_t const idx)
{
	uint64_t r = 0;
	#pragma unroll
	for ( int q = 0; q < NW; ++q ) r = (idx == static_cast<uint32_t>(q)) ? a[q] : r;
	return r;
}
template<int NW> struct TracePeqW { uint64_t e0[NW], e1[NW], e2[NW], e3[NW]; };
template<int NW> DEV uint64_t tracePeqWord(TracePeqW<NW> const & Q, uint32_t const c, int const q)
{
	// masks instead of selects: a select chain over the four arrays is turned back into an indexed load from a copy of Q in
	// scratch memory by the compiler
	uint64_t const m0 = 0ull - static_cast<uint64_t>(c == 0), m1 = 0ull - static_cast<uint64_t>(c == 1), m2 = 0ull - static_cast<uint64_t>(c == 2), m3 = 0ull - static_cast<uint64_t>(c == 3);
	return (Q.e0[q] & m0) | (Q.e1[q] & m1) | (Q.e2[q] & m2) | (Q.e3[q] & m3);
}
template<int NW>
DEV void traceStepW(TracePeqW<NW> const & Q, uint32_t const tc, TColW<NW> & C, uint64_t const (&mask)[NW], uint32_t const topblk, uint64_t const top)
{
	uint64_t phc = 1, mhc = 0;      // word 0: the top row of the matrix has horizontal delta +1 (global alignment)
	#pragma unroll
	for ( int q = 0; q < NW; ++q )
	{
		uint64_t const Eq = tracePeqWord<NW>(Q,tc,q);
		uint64_t const Eqc = Eq | mhc;
		uint64_t const Xv = Eq | C.mv[q];
		uint64_t const Xh = (((Eqc & C.pv[q]) + C.pv[q]) ^ C.pv[q]) | Eqc;
		uint64_t Ph = C.mv[q] | ~(Xh | C.pv[q]);
		uint64_t Mh = C.pv[q] & Xh;
		if ( static_cast<uint32_t>(q) == topblk ) { if ( Ph & top ) ++C.score; else if ( Mh & top ) --C.score; }
		uint64_t const nphc = Ph>>63, nmhc = Mh>>63;
		Ph = (Ph<<1) | phc; Mh = (Mh<<1) | mhc;
		C.pv[q] = (Mh | ~(Xv | Ph)) & mask[q];
		C.mv[q] = (Ph & Xv) & mask[q];
		phc = nphc; mhc = nmhc;
	}
}

template<int NW, typename ST>
DEV void traceBlockWide(TraceBatch const & B, uint64_t const task, ST const & st)
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
	uint32_t const n = B.trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(B.trace)[o.trace_off + 2*bi + 1] : B.trace[o.trace_off + 2*bi + 1];
	if ( m > 64u*NW || n > B.maxcols || m == 0 ) { if ( m ) atomicOrFlag(B.errflag); return; }

	uint64_t const aoff = B.boff[pile.aread]; uint32_t const arl = B.rlen[pile.aread];
	uint64_t const boffs = B.boff[o.bread]; uint32_t const brl = B.rlen[o.bread];
	bool const inv = o.flags & 1;

	TracePeqW<NW> Q; uint64_t mask[NW];
	#pragma unroll
	for ( int q = 0; q < NW; ++q )
	{
		Q.e0[q] = Q.e1[q] = Q.e2[q] = Q.e3[q] = 0;
		uint32_t const lo = 64u*q;
		mask[q] = m > lo ? ( m >= lo+64 ? ~0ull : ((1ull << (m-lo))-1) ) : 0ull;
	}
	// pattern masks: the word of a base is picked by selects over the compile time word indices (a run time index into
	// the register arrays would put them into scratch memory)
	for ( uint32_t i = 0; i < m; ++i )
	{
		uint32_t const c = readBase(B.bps,aoff,arl,false,a0+i); uint64_t const bit = 1ull << (i&63); uint32_t const wi = i>>6;
		#pragma unroll
		for ( int q = 0; q < NW; ++q )
		{
			uint64_t const b = (wi == static_cast<uint32_t>(q)) ? bit : 0ull;
			Q.e0[q] |= (c == 0) ? b : 0ull; Q.e1[q] |= (c == 1) ? b : 0ull; Q.e2[q] |= (c == 2) ? b : 0ull; Q.e3[q] |= (c == 3) ? b : 0ull;
		}
	}
	uint32_t const topblk = (m-1)>>6; uint64_t const top = 1ull << ((m-1)&63);
	uint32_t const ncp = B.maxcols/TRS + 1;
	TColW<NW> C;
	#pragma unroll
	for ( int q = 0; q < NW; ++q ) { C.pv[q] = mask[q]; C.mv[q] = 0; }
	C.score = m;
	st.put(0,C);
	#define DACC_LOADB(c0_,cnt_,dst_) { dst_ = 0; _Pragma("unroll") for ( uint32_t u = 0; u < TRS; ++u ) if ( u < (cnt_) ) dst_ |= static_cast<uint32_t>(readBase(B.bps,boffs,brl,inv,b0+(c0_)+u)) << (2*u); }
	for ( uint32_t c0 = 0; c0 < n; c0 += TRS )
	{
		uint32_t const cnt = (n-c0 < TRS) ? (n-c0) : static_cast<uint32_t>(TRS);
		uint32_t bb; DACC_LOADB(c0,cnt,bb)
		for ( uint32_t u = 0; u < cnt; ++u ) traceStepW<NW>(Q,(bb>>(2*u))&3,C,mask,topblk,top);
		if ( cnt == TRS ) st.put(c0/TRS+1,C);
	}
	uint32_t const wa = B.P.w % B.P.a, lw = pile.l >= B.P.w ? pile.l - B.P.w : 0xFFFFFFFFu;
	uint32_t xa = (a0+m) % B.P.a;
	#define DACC_EMIT(x_,b_) { uint32_t const xx_ = (x_); if ( xa == 0 || xa == wa || xx_ == lw || xx_ == pile.l ) emitBoundary(B,pile,o,xx_,b_); }
	#define DACC_XDEC { xa = xa ? xa-1 : B.P.a-1; }
	uint32_t i = m, j = n, d = C.score;
	for ( int32_t g = static_cast<int32_t>((B.maxcols ? B.maxcols-1 : 0)/TRS); g >= 0; --g )
	{
		if ( !(i && j && (j-1)/TRS == static_cast<uint32_t>(g)) ) continue;
		uint32_t bseg;
		{
			TColW<NW> R = st.get(g);
			uint32_t const c0 = g*TRS, cnt = (n-c0 < TRS) ? (n-c0) : static_cast<uint32_t>(TRS);
			DACC_LOADB(c0,cnt,bseg)
			for ( uint32_t u = 0; u < cnt; ++u )
			{
				traceStepW<NW>(Q,(bseg>>(2*u))&3,R,mask,topblk,top);
				st.put(ncp + u,R);
			}
		}
		while ( i && j && (j-1)/TRS == static_cast<uint32_t>(g) )
		{
			bool done = false;
			{
				// D[i-1][j-1] = bottom(j-1) - sum of the vertical deltas of rows i..m in column j-1
				TColW<NW> const q = st.get(((j-1) % TRS) ? (ncp + (j-2) % TRS) : ((j-1)/TRS));
				uint32_t const sh = i-1, shw = sh>>6, shb = sh&63;
				int32_t sum = 0;
				#pragma unroll
				for ( int b = 0; b < NW; ++b )
				{
					uint64_t const pvb = static_cast<uint32_t>(b) < shw ? 0ull : ( static_cast<uint32_t>(b) == shw ? (q.pv[b] >> shb) : q.pv[b] );
					uint64_t const mvb = static_cast<uint32_t>(b) < shw ? 0ull : ( static_cast<uint32_t>(b) == shw ? (q.mv[b] >> shb) : q.mv[b] );
					sum += dacc_popc64(pvb) - dacc_popc64(mvb);
				}
				uint32_t const dd = q.score - sum;
				uint32_t const cb = (bseg >> (2*((j-1) - g*TRS))) & 3;
				uint64_t pw = 0;
				#pragma unroll
				for ( int b = 0; b < NW; ++b ) pw = (static_cast<uint32_t>(b) == shw) ? tracePeqWord<NW>(Q,cb,b) : pw;
				uint32_t const neq = ((pw >> shb) & 1) ? 0u : 1u;
				if ( dd + neq == d )
				{
					DACC_EMIT(a0+i,b0+j)
					--i; --j; d = dd; done = true; DACC_XDEC
				}
			}
			if ( !done )
			{
				uint32_t const r = i-1;
				TColW<NW> const cj = st.get((j % TRS) ? (ncp + (j-1) % TRS) : (j/TRS));
				bool const plus = (traceWord<NW>(cj.pv,r>>6) >> (r&63)) & 1;
				if ( plus )
				{
					DACC_EMIT(a0+i,b0+j)
					--i; d = d-1; done = true; DACC_XDEC
				}
			}
			if ( !done ) { --j; d = d-1; }
		}
	}
	while ( i ) { DACC_EMIT(a0+i,b0) --i; DACC_XDEC }
	#undef DACC_EMIT
	#undef DACC_XDEC
	#undef DACC_LOADB
	if ( a0 == static_cast<uint32_t>(o.abpos) ) emitBoundary(B,pile,o,a0,b0);
}

}
#endif
