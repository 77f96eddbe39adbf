/*
 * LDS-resident fast path of the per-window de Bruijn consensus (one wavefront = one window).
 *
 * Same algorithm and exactly the same results as the generic engine (dbg_window.hpp, a close
 * restatement of the reference's control flow); what changes is where the state lives and how
 * the work is organised for CDNA4:
 *
 *  - all per-window working state sits in the workgroup's LDS slice with 8/16-bit fields; the
 *    sort buffer of the build phase is overlaid by the traversal structures;
 *  - node successors need no table: the <= 4 successor k-mers of a node are adjacent in the sorted
 *    node-key array, one lower bound per node is kept;
 *  - weights are exact 64-bit fixed point: every weight the reference forms is an integer multiple of
 *    2^-32 (w = sum of VS entries / 2^32, DotProduct.hpp:54-60) and all its partial sums stay below
 *    2^21, so the reference's FP64 sums/differences are exact and equal our integer sums times 2^-32.
 *    Thresholds become integer compares (>= 1e-3 <=> >= 4294968, > 0.1 / >= 0.1 <=> >= 429496730,
 *    >= 0.5 <=> >= 2^31), heap orders are unchanged;
 *  - k-mer feasibility is never materialised: stretch feasibility is evaluated with lanes = candidate
 *    start positions straight from a 32-bit copy of the fixed-point table held in LDS, results kept as
 *    a 64-bit position mask per stretch plus a compact weight list; stretch links are mask
 *    intersections evaluated on demand;
 *  - the reference recomputes stretches, feasibility and both path enumerations for every
 *    (first k-mer, last k-mer) candidate pair (traverse, DebruijnGraph.hpp:4774-5097; hundreds of
 *    pairs per window at k=14).  Here stretches and feasibility are computed once per activation
 *    state; the split pieces of every candidate first / last k-mer are added to the pool once;
 *    the reverse enumeration is cached per last k-mer and the forward enumeration per first k-mer,
 *    and reused for a pair whenever the other k-mer's split provably cannot touch it (no scan
 *    target of the enumeration equals a node that identifies the split stretch or its pieces);
 *    otherwise the pair is enumerated on its exact stretch set.
 *
 * Windows that do not fit the LDS capacities (deep piles, strings > 64, gap filling at filter
 * frequency 0, w > 63, rare shapes) return false and are re-run by the generic engine.
 */
#ifndef DACC_FAST_WINDOW_HPP
#define DACC_FAST_WINDOW_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "arena.hpp"
#include "window_main.hpp"

namespace dacc {

enum { WS_RETRY = 4 };
enum { FNC = 48 };          // max first / last k-mer candidates on the fast path
enum { FNOPAR = 0xFF };

struct FastCaps
{
	uint32_t maxs, precap, ncap, scap, lcap, wcap, rccap, fcap, siqcap, blcap, conscap, pad;
	uint32_t nrows, nsup;        // dimensions of the fixed-point table copy held in LDS
	uint32_t ldsbytes, pad2;
	uint64_t gbytes;
};

struct FSI { uint64_t w; uint8_t left, right, current, path; uint32_t pad; };   // ScoreInterval
struct FCC { uint64_t w; uint32_t o, l; };                                       // ConsensusCandidate

struct FastLds
{
	uint8_t * str; uint8_t * slen; uint64_t * peq; uint8_t * ipos; uint8_t * irpos;
	uint32_t * nv; uint16_t * nps; uint8_t * nfreq; uint16_t * succ0; uint16_t * sinfo; uint8_t * npred;
	uint8_t * pfrom; uint8_t * pto; uint8_t * cpfrom; uint8_t * cpto;
	uint64_t * mfirst; uint64_t * mlast;
	uint32_t * tab; uint8_t * suplo8; uint8_t * suphi8;
	// overlay, build phase
	uint64_t * pre; uint64_t * lastk; uint64_t * gfbuf; uint32_t gfcap;   // gfbuf: scratch behind the build overlay (gap filling)
	// overlay, traversal phase
	uint16_t * sfirst; uint16_t * slast; uint16_t * sslen; uint16_t * slink; uint64_t * maskF; uint64_t * maskR; uint16_t * woffF; uint16_t * woffR;
	uint16_t * links; uint64_t * wuF; uint64_t * wuR; uint8_t * ord;
	uint32_t * fkmer; uint32_t * lkmer; uint16_t * fnode; uint16_t * lnode; uint8_t * parF; uint8_t * posF; uint8_t * parL; uint8_t * posL;
	uint8_t * pieF; uint8_t * pieL;          // first piece id of candidate i (second = +1), FNOPAR if none
	// reverse cache
	uint64_t * rc_w; uint8_t * rc_parent; uint8_t * rc_stretch; uint8_t * rc_pos; uint8_t * rc_len; uint8_t * rc_baselen; uint8_t * rc_ord; uint8_t * rc_arw;
	uint16_t * rbase; uint8_t * rn; uint8_t * rnpool; uint8_t * rvalid;
	// forward pool
	uint64_t * f_w; uint8_t * f_parent; uint8_t * f_stretch; uint8_t * f_pos; uint8_t * f_baselen; uint8_t * f_len; uint8_t * fpop;
	// heaps
	uint8_t * rpst; uint8_t * hbl; uint8_t * hbl_n; FSI * siq;
	FCC * cdh; FCC * ch; FCC * acc; uint32_t * accerr; uint16_t * canderr; uint8_t * prevstr; uint8_t * curstr;
	// raw stretches (overlay of the caches)
	uint16_t * tfirst; uint16_t * tlast; uint16_t * tslen; uint16_t * tlink; uint64_t * skey;
	// final alignment (overlay of the caches)
	uint64_t * alpv; uint64_t * almv; uint16_t * albot; uint8_t * alops;
};

struct FastGlobal { uint8_t * cons; };

#define FCARVE(field,type,count) L.field = reinterpret_cast<type *>(base + o); o = (o + sizeof(type)*static_cast<uint64_t>(count) + 7) & ~static_cast<uint64_t>(7);

HDEV uint32_t fast_lds_carve(FastLds & L, uint8_t * base, FastCaps const & C)
{
	uint64_t o = 0;
	uint32_t const keycap = next_pow2(C.maxs < 2 ? 2 : C.maxs);
	FCARVE(str,uint8_t,C.maxs*64)
	FCARVE(slen,uint8_t,C.maxs)
	FCARVE(peq,uint64_t,C.maxs*4)
	FCARVE(ipos,uint8_t,C.precap)
	FCARVE(irpos,uint8_t,C.precap)
	FCARVE(nv,uint32_t,C.ncap)
	FCARVE(nps,uint16_t,C.ncap+1)
	FCARVE(nfreq,uint8_t,C.ncap)
	FCARVE(succ0,uint16_t,C.ncap)
	FCARVE(sinfo,uint16_t,C.ncap)
	FCARVE(npred,uint8_t,C.ncap)
	FCARVE(pfrom,uint8_t,C.ncap)
	FCARVE(pto,uint8_t,C.ncap)
	FCARVE(cpfrom,uint8_t,C.ncap)
	FCARVE(cpto,uint8_t,C.ncap)
	FCARVE(mfirst,uint64_t,keycap)
	FCARVE(mlast,uint64_t,keycap)
	FCARVE(tab,uint32_t,C.nrows*C.nsup)
	FCARVE(suplo8,uint8_t,C.nsup)
	FCARVE(suphi8,uint8_t,C.nsup)
	uint64_t const ubase = o;
	FCARVE(pre,uint64_t,C.precap)
	FCARVE(lastk,uint64_t,keycap)
	uint64_t const uA = o;
	o = ubase;
	FCARVE(sfirst,uint16_t,C.scap)
	FCARVE(slast,uint16_t,C.scap)
	FCARVE(sslen,uint16_t,C.scap)
	FCARVE(slink,uint16_t,C.scap)
	FCARVE(maskF,uint64_t,C.scap)
	FCARVE(maskR,uint64_t,C.scap)
	FCARVE(woffF,uint16_t,C.scap)
	FCARVE(woffR,uint16_t,C.scap)
	FCARVE(links,uint16_t,C.lcap)
	FCARVE(wuF,uint64_t,C.wcap)
	FCARVE(wuR,uint64_t,C.wcap)
	FCARVE(ord,uint8_t,C.scap)
	FCARVE(fkmer,uint32_t,FNC)
	FCARVE(lkmer,uint32_t,FNC)
	FCARVE(fnode,uint16_t,FNC)
	FCARVE(lnode,uint16_t,FNC)
	FCARVE(parF,uint8_t,FNC)
	FCARVE(posF,uint8_t,FNC)
	FCARVE(parL,uint8_t,FNC)
	FCARVE(posL,uint8_t,FNC)
	FCARVE(pieF,uint8_t,FNC)
	FCARVE(pieL,uint8_t,FNC)
	FCARVE(cdh,FCC,16)
	FCARVE(ch,FCC,16)
	FCARVE(acc,FCC,16)
	FCARVE(accerr,uint32_t,16)
	FCARVE(canderr,uint16_t,16*C.maxs)
	FCARVE(prevstr,uint8_t,MAXCONS)
	FCARVE(curstr,uint8_t,MAXCONS)
	uint64_t const pbase = o;
	FCARVE(rc_w,uint64_t,C.rccap)
	FCARVE(rc_parent,uint8_t,C.rccap)
	FCARVE(rc_stretch,uint8_t,C.rccap)
	FCARVE(rc_pos,uint8_t,C.rccap)
	FCARVE(rc_len,uint8_t,C.rccap)
	FCARVE(rc_baselen,uint8_t,C.rccap)
	FCARVE(rc_ord,uint8_t,C.rccap)
	FCARVE(rc_arw,uint8_t,C.rccap)
	FCARVE(rbase,uint16_t,FNC+1)
	FCARVE(rn,uint8_t,FNC+1)
	FCARVE(rnpool,uint8_t,FNC+1)
	FCARVE(rvalid,uint8_t,FNC+1)
	FCARVE(f_w,uint64_t,C.fcap)
	FCARVE(f_parent,uint8_t,C.fcap)
	FCARVE(f_stretch,uint8_t,C.fcap)
	FCARVE(f_pos,uint8_t,C.fcap)
	FCARVE(f_baselen,uint8_t,C.fcap)
	FCARVE(f_len,uint8_t,C.fcap)
	FCARVE(fpop,uint8_t,C.fcap)
	FCARVE(rpst,uint8_t,256)
	FCARVE(hbl,uint8_t,C.blcap*12)
	FCARVE(hbl_n,uint8_t,C.blcap)
	FCARVE(siq,FSI,C.siqcap)
	uint64_t const upool = o;
	o = pbase;
	FCARVE(tfirst,uint16_t,C.scap)
	FCARVE(tlast,uint16_t,C.scap)
	FCARVE(tslen,uint16_t,C.scap)
	FCARVE(tlink,uint16_t,C.scap)
	FCARVE(skey,uint64_t,next_pow2(C.scap))
	uint64_t const uraw = o;
	o = pbase;
	FCARVE(alpv,uint64_t,MAXCONS+1)
	FCARVE(almv,uint64_t,MAXCONS+1)
	FCARVE(albot,uint16_t,MAXCONS+1)
	FCARVE(alops,uint8_t,2*MAXCONS+2*64+8)
	uint64_t const ualn = o;
	uint64_t m = uA;
	if ( uraw > m ) m = uraw;
	if ( upool > m ) m = upool;
	if ( ualn > m ) m = ualn;
	L.gfbuf = reinterpret_cast<uint64_t *>(base + uA); L.gfcap = static_cast<uint32_t>((m-uA)/8);
	return static_cast<uint32_t>(m);
}

HDEV uint64_t fast_global_carve(FastGlobal & G, uint8_t * base, FastCaps const & C)
{
	G.cons = base;
	return (C.conscap+255)&~255ull;
}

struct FastBatch
{
	WindowBatch W;
	FastCaps F;
	uint64_t const * dpsq_vst;  // [nsup][nrows] transposed fixed-point table (HBM); copied to LDS as 32-bit
	uint8_t * garena;           // [gridDim][F.gbytes]
	uint32_t * retry;           // [0] = count, [1..] = window indices to re-run generically
};

DEV void fast_load_tables(FastLds & L, FastCaps const & C, DevTables const & T, uint64_t const * vst)
{
	int const lane = wv_lane();
	for ( uint32_t i = lane; i < C.nrows*C.nsup; i += WSZ ) L.tab[i] = static_cast<uint32_t>(vst[i]);
	for ( uint32_t i = lane; i < C.nsup; i += WSZ ) { L.suplo8[i] = T.suplo[i]; L.suphi8[i] = T.suphi[i]; }
	wv_sync();
}

#define FW_THRES_FEAS 4294968ull        /* weight >= 1e-3 */
#define FW_THRES_01   429496730ull      /* weight > 0.1 and weight >= 0.1 (no integer lies between) */
#define FW_THRES_05   2147483648ull     /* weight >= 0.5 */

struct FastEngine
{
	FastLds L; FastGlobal G; FastCaps C; DevTables T; DevParams P;
	int lane; uint32_t flags;
	uint64_t * prof;
	uint32_t mao, k; uint64_t kmask;
	uint32_t npre, nlast, nn, nmfirst, nmlast;
	uint32_t n0, npool, nlinks, nwF, nwR, nview;
	uint32_t nF, nL;
	uint32_t rctop;                      // used entries of the reverse cache
	uint32_t np, nfpop, nsiq, ncdh, nacc, conso;
	int32_t fcur_fi; int32_t fcur_li;    // what the forward pool holds: F(fi) on view(fi) (li = -1) or an exact pair view
	uint32_t prevlen;

	DEV void over(uint32_t b) { flags |= b; }
#if defined(DACC_EMUL) && defined(DACC_FSTATS)
	void fstat(int i, uint32_t v) { extern uint32_t g_fstat[16]; if ( v > g_fstat[i] ) g_fstat[i] = v; }
#endif

	DEV int32_t findNode(uint32_t const v) const
	{
		int32_t lo = 0, hi = static_cast<int32_t>(nn)-1;
		while ( lo <= hi )
		{
			int32_t const mid = (lo+hi)>>1;
			uint32_t const x = L.nv[mid];
			if ( x == v ) return mid;
			if ( x < v ) lo = mid+1; else hi = mid-1;
		}
		return -1;
	}
	DEV uint32_t lowerNode(uint32_t const v) const
	{
		uint32_t lo = 0, hi = nn;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( L.nv[mid] < v ) lo = mid+1; else hi = mid; }
		return lo;
	}
	DEV int32_t succNode(uint32_t const z, uint32_t const i) const
	{
		uint32_t const sym = (L.sinfo[z]>>(2*i))&3;
		uint32_t const target = static_cast<uint32_t>((static_cast<uint64_t>(L.nv[z])<<2) & kmask) | sym;
		for ( uint32_t q = L.succ0[z]; q < nn; ++q )
		{
			uint32_t const x = L.nv[q];
			if ( x == target ) return q;
			if ( x > target ) break;
		}
		return -1;
	}
	DEV uint32_t nsucc(uint32_t z) const { return (L.sinfo[z]>>8)&7; }
	DEV uint32_t nsuccact(uint32_t z) const { return (L.sinfo[z]>>11)&7; }

	// ================= build: instances, nodes, successors =================
	DEV void buildInstances()
	{
		uint32_t base = 0;
		for ( uint32_t j = 0; j < mao; ++j ) { uint32_t const len = L.slen[j]; base += (len >= k) ? (len-k+1) : 0; }
		npre = base;
		if ( npre > C.precap ) { over(1); npre = 0; return; }
		uint32_t o = 0, lo = 0;
		for ( uint32_t j = 0; j < mao; ++j )
		{
			uint32_t const len = L.slen[j];
			if ( len < k ) continue;
			uint32_t const numk = len-k+1;
			uint8_t const * s = L.str + j*64;
			for ( uint32_t i = lane; i < numk; i += WSZ )
			{
				uint64_t v = 0;
				for ( uint32_t q = 0; q < k; ++q ) v = (v<<2) | s[i+q];
				uint64_t const word = (v<<32) | (static_cast<uint64_t>(i)<<16) | j;
				L.pre[o+i] = word;
				if ( i == numk-1 ) L.lastk[lo] = word;
			}
			o += numk; ++lo;
		}
		nlast = lo;
		uint32_t const lp2 = next_pow2(nlast < 2 ? 2 : nlast);
		for ( uint32_t i = nlast + lane; i < lp2; i += WSZ ) L.lastk[i] = ~0ull;
		uint32_t const p2 = next_pow2(npre < 2 ? 2 : npre);
		for ( uint32_t i = npre + lane; i < p2; i += WSZ ) L.pre[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(L.lastk,lp2);
		wv_bitonic_sort(L.pre,p2);
	}

	DEV void buildNodes(uint32_t const f)
	{
		uint32_t base = 0;
		for ( uint32_t c = 0; c < npre; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t keep = 0, e = i;
			if ( i < npre && (i == 0 || (L.pre[i]>>32) != (L.pre[i-1]>>32)) )
			{
				uint64_t const km = L.pre[i]>>32;
				e = i+1;
				while ( e < npre && (L.pre[e]>>32) == km ) ++e;
				keep = (e-i) >= f;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(keep,tot);
			if ( keep )
			{
				uint32_t const z = base+pre;
				if ( z < C.ncap ) { L.nv[z] = static_cast<uint32_t>(L.pre[i]>>32); L.nps[z] = i; L.nfreq[z] = (e-i) > 255 ? 255 : (e-i); if ( (e-i) > 255 ) over(2); }
			}
			base += tot;
		}
		nn = base;
		if ( nn > C.ncap ) { over(2); nn = 0; }
		for ( uint32_t i = lane; i < npre; i += WSZ )
		{
			uint32_t const pos = (L.pre[i]>>16)&0xFFFF, seq = L.pre[i]&0xFFFF;
			L.ipos[i] = pos; L.irpos[i] = L.slen[seq]-pos-k;
		}
		wv_sync();
		if ( lane == 0 ) L.nps[nn] = npre;
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const s0 = L.nps[z], f2 = L.nfreq[z];
			uint32_t const lo = L.ipos[s0], hi = L.ipos[s0+f2-1];
			uint32_t rlo = 255, rhi = 0;
			for ( uint32_t q = 0; q < f2; ++q ) { uint32_t const r = L.irpos[s0+q]; rlo = r < rlo ? r : rlo; rhi = r > rhi ? r : rhi; }
			L.pfrom[z] = lo < C.nsup ? L.suplo8[lo] : C.nrows;
			L.pto[z] = hi < C.nsup ? L.suphi8[hi] : C.nrows;
			L.cpfrom[z] = rlo < C.nsup ? L.suplo8[rlo] : C.nrows;
			L.cpto[z] = rhi < C.nsup ? L.suphi8[rhi] : C.nrows;
		}
		uint32_t const kp2 = next_pow2(C.maxs < 2 ? 2 : C.maxs);
		base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t c0 = 0;
			if ( z < nn ) { uint32_t const s = L.nps[z]; for ( uint32_t q = 0; q < L.nfreq[z] && L.ipos[s+q] == 0; ++q ) ++c0; }
			uint32_t tot; uint32_t const pre = wv_scan_excl(c0 ? 1 : 0,tot);
			if ( c0 && base+pre < kp2 ) L.mfirst[base+pre] = ~((static_cast<uint64_t>(c0)<<32) | L.nv[z]);
			base += tot;
		}
		nmfirst = base;
		if ( nmfirst > kp2 ) { over(8); nmfirst = 0; }
		uint32_t const p2 = next_pow2(nmfirst < 2 ? 2 : nmfirst);
		for ( uint32_t i = nmfirst + lane; i < p2; i += WSZ ) L.mfirst[i] = ~0ull;
		base = 0;
		for ( uint32_t c = 0; c < nlast; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t const head = (i < nlast) && (i == 0 || (L.lastk[i]>>32) != (L.lastk[i-1]>>32));
			uint32_t tot; uint32_t const pre = wv_scan_excl(head,tot);
			if ( head )
			{
				uint32_t e = i+1;
				while ( e < nlast && (L.lastk[e]>>32) == (L.lastk[i]>>32) ) ++e;
				L.mlast[base+pre] = ~((static_cast<uint64_t>(e-i)<<32) | (L.lastk[i]>>32));
			}
			base += tot;
		}
		nmlast = base;
		uint32_t const q2 = next_pow2(nmlast < 2 ? 2 : nmlast);
		for ( uint32_t i = nmlast + lane; i < q2; i += WSZ ) L.mlast[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(L.mfirst,p2);
		wv_bitonic_sort(L.mlast,q2);
	}

	DEV void buildSuccessors(uint32_t const no)
	{
		uint32_t const lim = T.klim[(k-P.klow)*T.kln + (no < static_cast<uint32_t>(T.kln) ? no : T.kln-1)];
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const masked = static_cast<uint32_t>((static_cast<uint64_t>(L.nv[z])<<2) & kmask);
			uint32_t const s0 = lowerNode(masked);
			uint32_t Lk[4]; uint32_t n = 0;
			for ( uint32_t q = s0; q < nn && L.nv[q] <= (masked|3); ++q )
				Lk[n++] = (static_cast<uint32_t>(L.nfreq[q])<<8) | (L.nv[q]&3);
			for ( uint32_t a = 1; a < n; ++a )
			{
				uint32_t const kv = Lk[a]; int32_t b = a;
				while ( b > 0 && Lk[b-1] < kv ) { Lk[b] = Lk[b-1]; --b; }
				Lk[b] = kv;
			}
			uint32_t act = 0;
			if ( n )
			{
				act = 1;
				while ( act < n && ( ((Lk[act]>>8) >= (Lk[0]>>8)/2) || (P.checklim && ((Lk[act]>>8) >= lim)) ) ) ++act;
			}
			uint32_t order = 0;
			for ( uint32_t a = 0; a < n; ++a ) order |= (Lk[a]&3) << (2*a);
			L.succ0[z] = s0;
			L.sinfo[z] = order | (n<<8) | (act<<11);
		}
		wv_sync();
	}
	DEV bool addNextFromHeap()
	{
		uint32_t best = 0;
		for ( uint32_t z = lane; z < nn; z += WSZ )
			if ( nsuccact(z) < nsucc(z) )
			{
				uint32_t const fq = L.nfreq[succNode(z,nsuccact(z))];
				best = fq > best ? fq : best;
			}
		best = wv_max(best);
		if ( ! best ) return false;
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t a = nsuccact(z); uint32_t const n = nsucc(z);
			while ( a < n && L.nfreq[succNode(z,a)] == best ) ++a;
			L.sinfo[z] = (L.sinfo[z] & 0x7FF) | (a<<11);
		}
		wv_sync();
		return true;
	}

	// ================= gap filling at filter frequency 0 (getLevelSuccessors(2), :1016-1161) =================
	// feasible position list of node z is implicit: p in [pfrom,pto) with nodeU(z,p) >= 1e-3
	DEV void levelSuccessors2()
	{
		uint32_t const s = 2;
		uint64_t * REC = L.gfbuf;          // records (from<<48 | to<<32 | cv)
		uint32_t const cap = L.gfcap/2;    // second half holds the (cv,pos) candidates
		uint64_t * ANE = L.gfbuf + cap;
		// pass 1: missing intermediate k-mers between nodes two steps apart
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t cnt = 0; uint32_t lo = 0; uint32_t low = 0, vhigh = 0;
			if ( i < nn )
			{
				uint32_t const v = L.nv[i];
				low = static_cast<uint32_t>((static_cast<uint64_t>(v)<<(2*s)) & kmask);
				vhigh = static_cast<uint32_t>((static_cast<uint64_t>(v)<<2) & kmask);
				lo = lowerNode(low);
				for ( uint32_t q = lo; q < nn && L.nv[q] <= (low|0xF); ++q )
					if ( findNode((L.nv[q]>>2)|vhigh) < 0 ) ++cnt;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( i < nn && cnt && base+pre+cnt <= cap )
			{
				uint32_t o = base+pre;
				for ( uint32_t q = lo; q < nn && L.nv[q] <= (low|0xF); ++q )
				{
					uint32_t const cv = (L.nv[q]>>2)|vhigh;
					if ( findNode(cv) < 0 ) REC[o++] = (static_cast<uint64_t>(i)<<48) | (static_cast<uint64_t>(q)<<32) | cv;
				}
			}
			base += tot;
		}
		uint32_t const nrec = base;
		if ( nrec > cap ) { over(8192); return; }
		wv_sync();
		// pass 2: best common feasible position of `from` (shifted by s) and `to`
		base = 0;
		for ( uint32_t c = 0; c < nrec; c += WSZ )
		{
			uint32_t const r = c + lane;
			uint32_t have = 0; uint64_t cand = 0;
			if ( r < nrec )
			{
				uint32_t const from = REC[r]>>48, to = (REC[r]>>32)&0xFFFF; uint32_t const cv = static_cast<uint32_t>(REC[r]);
				uint64_t mweight = 0; uint32_t mp = 0;
				// common position pp: from feasible at pp-s, to feasible at pp; ascending pp, strict > keeps the first maximum
				for ( uint32_t pp = L.pfrom[to]; pp < L.pto[to]; ++pp )
				{
					if ( pp < s ) continue;
					uint32_t const pf = pp-s;
					if ( pf < L.pfrom[from] || pf >= L.pto[from] ) continue;
					uint64_t const ua = nodeU(from,pf,false), ub = nodeU(to,pp,false);
					if ( ua < FW_THRES_FEAS || ub < FW_THRES_FEAS ) continue;
					uint64_t const weight = ua+ub;
					if ( !have || weight > mweight ) { have = 1; mweight = weight; mp = pp - s + 1; }
				}
				cand = (static_cast<uint64_t>(cv)<<8) | mp;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(have,tot);
			if ( have ) ANE[base+pre] = cand;
			base += tot;
		}
		uint32_t const nane = base;
		uint32_t const p2 = next_pow2(nane < 2 ? 2 : nane);
		if ( p2 > cap ) { over(8192); return; }
		for ( uint32_t i = nane + lane; i < p2; i += WSZ ) ANE[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(ANE,p2);       // (v,pos) ascending, NodeAddElement::operator<
		// pass 3: attach each candidate to the first string that is long enough and append the instance
		base = 0;
		for ( uint32_t c = 0; c < nane; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t ok = 0; uint32_t seqid = 0, pos = 0; uint64_t cv = 0;
			if ( i < nane )
			{
				pos = ANE[i]&0xFF; cv = ANE[i]>>8;
				for ( uint32_t j = 0; j < mao; ++j ) if ( pos + k <= L.slen[j] ) { seqid = j; ok = 1; break; }
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(ok,tot);
			if ( ok && npre+base+pre < C.precap ) L.pre[npre+base+pre] = (cv<<32) | (static_cast<uint64_t>(pos)<<16) | seqid;
			base += tot;
		}
		if ( npre + base > C.precap ) { over(1); return; }
		npre += base;
		uint32_t const q2 = next_pow2(npre < 2 ? 2 : npre);
		for ( uint32_t i = npre + lane; i < q2; i += WSZ ) L.pre[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(L.pre,q2);
	}

	// ================= stretches, once per activation state =================
	DEV void computePredCounts()
	{
		uint32_t const shift = 2*(k-1);
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const v = L.nv[z];
			uint32_t const masked = v>>2, sym = v&3;
			uint32_t cnt = 0;
			for ( uint32_t s = 0; s < 4; ++s )
			{
				int32_t const u = findNode(masked | (s<<shift));
				if ( u >= 0 )
				{
					uint32_t const info = L.sinfo[u]; uint32_t const na = (info>>11)&7;
					for ( uint32_t i = 0; i < na; ++i ) if ( ((info>>(2*i))&3) == sym ) { ++cnt; break; }
				}
			}
			L.npred[z] = cnt;
		}
		wv_sync();
	}
	DEV uint32_t walkStretch(uint32_t const z, uint32_t const i, uint16_t * out, uint32_t & lastnode)
	{
		int32_t cur = succNode(z,i);
		uint32_t len = 2;
		if ( out ) { out[0] = z; out[1] = cur; }
		bool loop = (cur == static_cast<int32_t>(z));
		while ( !loop && nsuccact(cur) == 1 && L.npred[cur] == 1 )
		{
			cur = succNode(cur,0);
			if ( out ) out[len] = cur;
			++len;
			if ( cur == static_cast<int32_t>(z) ) loop = true;
			if ( len > nn+1 ) { over(16); break; }
		}
		lastnode = cur;
		return len;
	}
	// (first, ext, len desc, last) packed into 56 bits: Stretch::operator< (node ids ascend with k-mer value)
	DEV static uint64_t sortKey(uint32_t first, uint32_t ext, uint32_t len, uint32_t last)
	{
		return (static_cast<uint64_t>(first)<<42) | (static_cast<uint64_t>(ext)<<28) | (static_cast<uint64_t>(0x3FFF-len)<<14) | last;
	}
	DEV uint64_t poolKey(uint32_t const s) const { return sortKey(L.sfirst[s],L.links[L.slink[s]+1],L.sslen[s],L.slast[s]); }

	// raw stretches of the current activation state, sorted by Stretch::operator< (pool ids 0..n0-1)
	DEV void computeBaseStretches()
	{
		computePredCounts();
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t cnt = 0;
			if ( z < nn ) { uint32_t const ns = nsuccact(z); if ( ns && (L.npred[z] != 1 || ns > 1) ) cnt = ns; }
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( base+pre+cnt <= C.scap )
				for ( uint32_t i = 0; i < cnt; ++i ) { L.tfirst[base+pre+i] = z; L.tlast[base+pre+i] = i; }
			base += tot;
		}
		uint32_t const ns = base;
		if ( ns + 2 > C.scap || ns > 250 || nn >= 0x3FFF ) { over(32); n0 = 0; return; }
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ ) { uint32_t ln; L.tslen[q] = walkStretch(L.tfirst[q],L.tlast[q],0,ln); }
		wv_sync();
		base = 0;
		for ( uint32_t c = 0; c < ns; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const len = q < ns ? L.tslen[q] : 0;
			uint32_t tot; uint32_t const pre = wv_scan_excl(len,tot);
			if ( q < ns ) L.tlink[q] = base+pre;
			base += tot;
		}
		nlinks = base;
		if ( nlinks > C.lcap ) { over(64); n0 = 0; return; }
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ ) { uint32_t ln; walkStretch(L.tfirst[q],L.tlast[q],L.links+L.tlink[q],ln); L.tlast[q] = ln; }
		wv_sync();
		uint32_t const p2 = next_pow2(ns < 2 ? 2 : ns);
		for ( uint32_t q = lane; q < p2; q += WSZ )
			L.skey[q] = q < ns ? ((sortKey(L.tfirst[q],L.links[L.tlink[q]+1],L.tslen[q],L.tlast[q])<<8) | q) : ~0ull;
		wv_sync();
		wv_bitonic_sort(L.skey,p2);
		// distinct (first,ext) by construction; a duplicate would need stretchesUnique's tie handling -> generic engine
		uint32_t dup = 0;
		for ( uint32_t q = lane; q < ns; q += WSZ )
		{
			if ( q && (L.skey[q]>>36) == (L.skey[q-1]>>36) ) dup = 1;
			uint32_t const raw = L.skey[q]&0xFF;
			L.sfirst[q] = L.tfirst[raw]; L.slast[q] = L.tlast[raw]; L.sslen[q] = L.tslen[raw]; L.slink[q] = L.tlink[raw];
		}
		if ( wv_any(dup) ) { over(32); n0 = 0; return; }
		n0 = ns;
		wv_sync();
	}

	// pool stretch id = nodes [a,b] of parent stretch par
	DEV void makePiece(uint32_t const id, uint32_t const par, uint32_t const a, uint32_t const b)
	{
		uint16_t const * Lk = L.links + L.slink[par];
		L.sfirst[id] = Lk[a]; L.slast[id] = Lk[b]; L.sslen[id] = b-a+1; L.slink[id] = L.slink[par]+a;
	}
	// candidates + the pool stretch each one splits (splitStretches :2772-2841: first occurrence strictly inside)
	DEV void findCandidatesAndPieces()
	{
		uint32_t const firstthres = nmfirst ? ((static_cast<uint32_t>((~L.mfirst[0])>>32))*3)/4 : 0;
		uint32_t const lastthres = nmlast ? ((static_cast<uint32_t>((~L.mlast[0])>>32))*3)/4 : 0;
		uint32_t cf = 0, cl = 0;
		while ( cf < nmfirst && static_cast<uint32_t>((~L.mfirst[cf])>>32) >= firstthres ) ++cf;
		while ( cl < nmlast && static_cast<uint32_t>((~L.mlast[cl])>>32) >= lastthres ) ++cl;
		nF = cf; nL = cl; npool = n0;
		if ( nF > FNC || nL > FNC ) { over(8); return; }
		for ( uint32_t i = lane; i < nF; i += WSZ ) { uint32_t const km = static_cast<uint32_t>(~L.mfirst[i]); L.fkmer[i] = km; int32_t const z = findNode(km); L.fnode[i] = z < 0 ? 0xFFFF : z; L.parF[i] = FNOPAR; L.pieF[i] = FNOPAR; }
		for ( uint32_t i = lane; i < nL; i += WSZ ) { uint32_t const km = static_cast<uint32_t>(~L.mlast[i]); L.lkmer[i] = km; int32_t const z = findNode(km); L.lnode[i] = z < 0 ? 0xFFFF : z; L.parL[i] = FNOPAR; L.pieL[i] = FNOPAR; }
		wv_sync();
		// parents: lanes over base stretches.  An interior node has a unique active predecessor and successor, so it
		// lies strictly inside at most one stretch; anything else goes to the generic engine
		uint32_t multi = 0;
		for ( uint32_t s = lane; s < n0; s += WSZ )
		{
			uint32_t const len = L.sslen[s]; uint16_t const * Lk = L.links + L.slink[s];
			for ( uint32_t i = 1; i+1 < len; ++i )
			{
				uint32_t const z = Lk[i];
				for ( uint32_t c = 0; c < nF; ++c ) if ( L.fnode[c] == z ) { if ( L.parF[c] == FNOPAR ) { L.parF[c] = s; L.posF[c] = i; } else if ( L.parF[c] != s ) multi = 1; }
				for ( uint32_t c = 0; c < nL; ++c ) if ( L.lnode[c] == z ) { if ( L.parL[c] == FNOPAR ) { L.parL[c] = s; L.posL[c] = i; } else if ( L.parL[c] != s ) multi = 1; }
			}
		}
		if ( wv_any(multi) ) { over(32); return; }
		wv_sync();
		{
			uint32_t cnt = 0;
			for ( uint32_t c = 0; c < nF; ++c ) cnt += (L.parF[c] != FNOPAR);
			for ( uint32_t c = 0; c < nL; ++c ) cnt += (L.parL[c] != FNOPAR);
			if ( n0 + 2*cnt + 1 > C.scap || n0 + 2*cnt + 1 > 250 ) { over(32); return; }
		}
		if ( lane == 0 )
		{
			uint32_t id = n0;
			for ( uint32_t c = 0; c < nF; ++c ) if ( L.parF[c] != FNOPAR ) { L.pieF[c] = id; makePiece(id,L.parF[c],0,L.posF[c]); makePiece(id+1,L.parF[c],L.posF[c],L.sslen[L.parF[c]]-1); id += 2; }
			for ( uint32_t c = 0; c < nL; ++c ) if ( L.parL[c] != FNOPAR ) { L.pieL[c] = id; makePiece(id,L.parL[c],0,L.posL[c]); makePiece(id+1,L.parL[c],L.posL[c],L.sslen[L.parL[c]]-1); id += 2; }
			npool = id;
		}
		wv_sync();
		npool = wv_bcast(npool,0);
	}

	// ---- stretch feasibility for pool ids [sfrom,sto), lanes = candidate positions ----
	DEV void computeStretchFeas(uint32_t const sfrom, uint32_t const sto)
	{
		uint32_t const nrows = C.nrows;
		for ( uint32_t s = sfrom; s < sto; ++s )
		{
			uint32_t const len = L.sslen[s];
			uint16_t const * Lk = L.links + L.slink[s];
			uint64_t mF = 0, mR = 0;
			uint32_t bF = nwF, bR = nwR;
			for ( uint32_t c = 0; c < nrows; c += WSZ )
			{
				uint32_t const Pp = c + lane;
				bool ok = Pp < nrows, okr = ok;
				uint64_t sum = 0, rsum = 0;
				for ( uint32_t j = 0; j < len; ++j )
				{
					uint32_t const p = Pp+j;
					bool const in = p < nrows;
					{
						uint32_t const z = Lk[j]; uint32_t const i0 = L.nps[z], f = L.nfreq[z];
						uint64_t u = 0;
						for ( uint32_t q = 0; q < f; ++q ) u += in ? L.tab[static_cast<uint32_t>(L.ipos[i0+q])*nrows + p] : 0u;
						ok = ok && in && p >= L.pfrom[z] && p < L.pto[z] && u >= FW_THRES_FEAS;
						sum += u;
					}
					{
						uint32_t const z = Lk[len-1-j]; uint32_t const i0 = L.nps[z], f = L.nfreq[z];
						uint64_t u = 0;
						for ( uint32_t q = 0; q < f; ++q ) u += in ? L.tab[static_cast<uint32_t>(L.irpos[i0+q])*nrows + p] : 0u;
						okr = okr && in && p >= L.cpfrom[z] && p < L.cpto[z] && u >= FW_THRES_FEAS;
						rsum += u;
					}
				}
				uint32_t tot; uint32_t const pre = wv_scan_excl(ok ? 1 : 0,tot);
				if ( ok && bF+pre < C.wcap ) L.wuF[bF+pre] = sum;
				mF |= wv_or64(ok ? (1ull<<Pp) : 0ull);
				bF += tot;
				uint32_t const prer = wv_scan_excl(okr ? 1 : 0,tot);
				if ( okr && bR+prer < C.wcap ) L.wuR[bR+prer] = rsum;
				mR |= wv_or64(okr ? (1ull<<Pp) : 0ull);
				bR += tot;
			}
			if ( bF > C.wcap || bR > C.wcap ) { over(128); return; }
			if ( lane == 0 ) { L.maskF[s] = mF; L.maskR[s] = mR; L.woffF[s] = nwF; L.woffR[s] = nwR; }
			nwF = bF; nwR = bR;
		}
		wv_sync();
	}
	// fixed-point weight of a single node at (reverse) position p
	DEV uint64_t nodeU(uint32_t const z, uint32_t const p, bool const rev) const
	{
		if ( p >= C.nrows ) return 0;
		uint32_t const i0 = L.nps[z], f = L.nfreq[z];
		uint8_t const * IP = rev ? L.irpos : L.ipos;
		uint64_t u = 0;
		for ( uint32_t q = 0; q < f; ++q ) u += L.tab[static_cast<uint32_t>(IP[i0+q])*C.nrows + p];
		return u;
	}
	DEV int32_t sfFind(uint32_t const s, uint32_t const p) const
	{
		if ( p >= 64 ) return -1;
		uint64_t const m = L.maskF[s];
		if ( !((m>>p)&1) ) return -1;
		return L.woffF[s] + dacc_popc64(m & ((1ull<<p)-1));
	}
	DEV int32_t csfFind(uint32_t const s, uint32_t const p) const
	{
		if ( p >= 64 ) return -1;
		uint64_t const m = L.maskR[s];
		if ( !((m>>p)&1) ) return -1;
		return L.woffR[s] + dacc_popc64(m & ((1ull<<p)-1));
	}
	// getReverseStretchLinkWeight(A=i,B=b) >= 0.1 (computeStretchLinks :3388-3480), evaluated on demand
	DEV bool linkOk(uint32_t const i, uint32_t const b) const
	{
		uint32_t const shift = L.sslen[b]-1;
		if ( shift >= 64 ) return false;
		uint64_t const mA = L.maskR[i], mB = L.maskR[b];
		uint64_t common = mA & (mB<<shift);
		uint64_t weight = 0;
		uint32_t const alast = L.slast[i];
		while ( common )
		{
			uint32_t const pa = __builtin_ctzll(common); common &= common-1;
			uint32_t const ia = L.woffR[i] + dacc_popc64(mA & ((1ull<<pa)-1));
			uint32_t const pb = pa-shift;
			uint32_t const ib = L.woffR[b] + dacc_popc64(mB & ((1ull<<pb)-1));
			uint64_t const lweight = L.wuR[ib] + (L.wuR[ia] - nodeU(alast,pa,true));
			weight = lweight > weight ? lweight : weight;
		}
		return weight >= FW_THRES_01;
	}

	// ================= views: the stretch set of a pair in sorted order =================
	// view = base stretches minus the split parents plus the pieces; ord[] lists pool ids in Stretch::operator< order
	DEV void buildView(uint32_t const * rem, uint32_t const nrem, uint32_t const * add, uint32_t const nadd)
	{
		uint64_t akey[4];
		for ( uint32_t a = 0; a < nadd; ++a ) akey[a] = poolKey(add[a]);
		for ( uint32_t s = lane; s < n0; s += WSZ )
		{
			bool removed = false; uint32_t before = 0;
			for ( uint32_t r = 0; r < nrem; ++r ) { if ( rem[r] == s ) removed = true; else if ( rem[r] < s ) ++before; }
			if ( !removed )
			{
				uint64_t const key = poolKey(s);
				uint32_t ins = 0;
				for ( uint32_t a = 0; a < nadd; ++a ) if ( akey[a] < key ) ++ins;
				L.ord[s-before+ins] = s;
			}
		}
		if ( lane == 0 )
			for ( uint32_t a = 0; a < nadd; ++a )
			{
				uint32_t lo = 0, hi = n0;
				while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( poolKey(mid) < akey[a] ) lo = mid+1; else hi = mid; }
				uint32_t pos = lo;
				for ( uint32_t r = 0; r < nrem; ++r ) if ( rem[r] < lo ) --pos;
				for ( uint32_t b = 0; b < nadd; ++b ) if ( akey[b] < akey[a] ) ++pos;
				L.ord[pos] = add[a];
			}
		nview = n0 - nrem + nadd;
		wv_sync();
	}
	DEV uint32_t viewLowerBound(uint32_t const node) const
	{
		uint32_t lo = 0, hi = nview;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( L.sfirst[L.ord[mid]] < node ) lo = mid+1; else hi = mid; }
		return lo;
	}

	// ---- heaps (same sift algorithm as oracle/o_heap.hpp) ----
	template<bool MINHEAP> DEV static bool hless(uint64_t a, uint64_t b) { return MINHEAP ? (a < b) : (a > b); }
	template<bool MINHEAP>
	DEV void ipush(uint8_t * H, uint32_t & f, uint8_t const id, uint64_t const * W)
	{
		uint32_t i = f++; H[i] = id;
		while ( i )
		{
			uint32_t const p = (i-1)>>1;
			if ( hless<MINHEAP>(W[H[i]],W[H[p]]) ) { uint8_t const t = H[i]; H[i] = H[p]; H[p] = t; i = p; }
			else break;
		}
	}
	template<bool MINHEAP>
	DEV void ipop(uint8_t * H, uint32_t & f, uint64_t const * W)
	{
		H[0] = H[--f];
		uint32_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			uint32_t const m = hless<MINHEAP>(W[H[r-1]],W[H[r]]) ? (r-1) : r;
			if ( hless<MINHEAP>(W[H[i]],W[H[m]]) ) return;
			uint8_t const t = H[i]; H[i] = H[m]; H[m] = t; i = m;
		}
		uint32_t const l = 2*i+1;
		if ( l < f && !hless<MINHEAP>(W[H[i]],W[H[l]]) ) { uint8_t const t = H[i]; H[i] = H[l]; H[l] = t; }
	}
	template<typename TT, bool MINHEAP>
	DEV void spush(TT * H, uint32_t & f, TT const & e)
	{
		uint32_t i = f++; H[i] = e;
		while ( i )
		{
			uint32_t const p = (i-1)>>1;
			if ( hless<MINHEAP>(H[i].w,H[p].w) ) { TT const t = H[i]; H[i] = H[p]; H[p] = t; i = p; }
			else break;
		}
	}
	template<typename TT, bool MINHEAP>
	DEV void spop(TT * H, uint32_t & f)
	{
		H[0] = H[--f];
		uint32_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			uint32_t const m = hless<MINHEAP>(H[r-1].w,H[r].w) ? (r-1) : r;
			if ( hless<MINHEAP>(H[i].w,H[m].w) ) return;
			TT const t = H[i]; H[i] = H[m]; H[m] = t; i = m;
		}
		uint32_t const l = 2*i+1;
		if ( l < f && !hless<MINHEAP>(H[i].w,H[l].w) ) { TT const t = H[i]; H[i] = H[l]; H[l] = t; }
	}

	// ================= reverse enumeration on the current view (lane 0) =================
	uint32_t rb, nrp, narp, rlastk;
	DEV int32_t extendReversePath(uint32_t const parent, uint32_t const s)
	{
		if ( rb+nrp >= C.rccap || nrp >= 250 ) { over(512); return -1; }
		uint32_t const id = nrp++;
		uint32_t const ppos = L.rc_pos[rb+parent], plen = L.rc_len[rb+parent];
		int32_t const sfo = csfFind(s,ppos);
		uint64_t weight = L.rc_w[rb+parent]; uint32_t baselen = L.rc_baselen[rb+parent];
		if ( plen == 0 ) { baselen = L.sslen[s]+k-1; weight = sfo >= 0 ? L.wuR[sfo] : 0; }
		else { baselen += L.sslen[s]-1; if ( sfo >= 0 ) weight += L.wuR[sfo] - nodeU(L.slast[s],ppos,true); }
		uint32_t const npos = ppos + L.sslen[s]-1;
		if ( baselen > 255 || npos > 255 || plen+1 > 255 ) { over(2048); --nrp; return -1; }
		L.rc_parent[rb+id] = parent; L.rc_stretch[rb+id] = s; L.rc_len[rb+id] = plen+1; L.rc_pos[rb+id] = npos;
		L.rc_w[rb+id] = weight; L.rc_baselen[rb+id] = baselen;
		return id;
	}
	DEV bool checkReversePathFeasiblePosition(uint32_t const id) const
	{
		uint32_t const s = L.rc_stretch[rb+id];
		uint32_t const checkpos = L.rc_pos[rb+id] - (L.sslen[s]-1);
		int32_t const f = csfFind(s,checkpos);
		return f >= 0 && L.wuR[f] >= FW_THRES_05;
	}
	DEV uint32_t rpFront(uint32_t const id) const { return L.rc_len[rb+id] ? L.nv[L.sfirst[L.rc_stretch[rb+id]]] : rlastk; }
	DEV bool arpLess(uint8_t const a, uint8_t const b) const
	{
		uint32_t const fa = rpFront(a), fb = rpFront(b);
		if ( fa != fb ) return fa < fb;
		return L.rc_baselen[rb+a] < L.rc_baselen[rb+b];
	}
	DEV void arpULI(uint8_t * last) { uint8_t const val = *last; uint8_t * next = last-1; while ( arpLess(val,*next) ) { *last = *next; last = next; --next; } *last = val; }
	DEV void arpIns(uint8_t * first, uint8_t * last)
	{
		if ( first == last ) return;
		for ( uint8_t * i = first+1; i != last; ++i )
		{
			if ( arpLess(*i,*first) ) { uint8_t const val = *i; for ( uint8_t * q = i; q != first; --q ) *q = *(q-1); *first = val; }
			else arpULI(i);
		}
	}
	// libstdc++ std::sort permutation (introsort + final insertion sort), see dbg_window.hpp arpSort
	DEV void arpSort(uint8_t * first, uint8_t * last)
	{
		if ( first == last ) return;
		int32_t const n = last-first;
		if ( n > 16 )
		{
			int depth = 0; { int32_t t = n; while ( t > 1 ) { t >>= 1; ++depth; } depth *= 2; }
			uint8_t stF[24], stL[24], stD[24]; int sp = 0;   // offsets from first
			stF[0] = 0; stL[0] = n; stD[0] = depth; sp = 1;
			while ( sp )
			{
				--sp;
				uint8_t * f = first+stF[sp]; uint8_t * l = first+stL[sp]; int d = stD[sp];
				while ( l-f > 16 )
				{
					if ( d == 0 ) { over(1024); return; }
					--d;
					uint8_t * mid = f + (l-f)/2; uint8_t * a = f+1; uint8_t * b = mid; uint8_t * c = l-1;
					if ( arpLess(*a,*b) )
					{
						if ( arpLess(*b,*c) ) { uint8_t t = *f; *f = *b; *b = t; }
						else if ( arpLess(*a,*c) ) { uint8_t t = *f; *f = *c; *c = t; }
						else { uint8_t t = *f; *f = *a; *a = t; }
					}
					else if ( arpLess(*a,*c) ) { uint8_t t = *f; *f = *a; *a = t; }
					else if ( arpLess(*b,*c) ) { uint8_t t = *f; *f = *c; *c = t; }
					else { uint8_t t = *f; *f = *b; *b = t; }
					uint8_t * lo = f+1; uint8_t * hi = l;
					while ( true )
					{
						while ( arpLess(*lo,*f) ) ++lo;
						--hi;
						while ( arpLess(*f,*hi) ) --hi;
						if ( !(lo < hi) ) break;
						uint8_t t = *lo; *lo = *hi; *hi = t;
						++lo;
					}
					if ( sp < 24 ) { stF[sp] = lo-first; stL[sp] = l-first; stD[sp] = d; ++sp; } else { over(1024); return; }
					l = lo;
				}
			}
			arpIns(first,first+16);
			for ( uint8_t * i = first+16; i != last; ++i ) arpULI(i);
		}
		else arpIns(first,last);
	}

	// prepareTraverse :3582-3765 on the current view; result block at rc_*[rb ..rb+nrp), sorted order rc_ord, ranks rc_arw
	DEV void reverseEnumerate(uint32_t const lastkmer, int32_t const lastnode, int64_t const lmax)
	{
		nrp = 0; narp = 0; rlastk = lastkmer;
		for ( uint32_t i = 0; i < C.blcap; ++i ) L.hbl_n[i] = 0;
		uint32_t nrpst = 0;
		if ( lastnode >= 0 )
		{
			if ( rb >= C.rccap ) { over(512); return; }
			uint32_t const id = nrp++;
			L.rc_parent[rb+id] = 0xFF; L.rc_stretch[rb+id] = 0xFF; L.rc_len[rb+id] = 0; L.rc_pos[rb+id] = 0; L.rc_w[rb+id] = 0; L.rc_baselen[rb+id] = k;
			L.rpst[nrpst++] = id;
		}
		uint64_t const * W = L.rc_w + rb;
		while ( nrpst )
		{
			uint32_t const rp = L.rpst[0];
			ipop<false>(L.rpst,nrpst,W);
			uint32_t const bl = L.rc_baselen[rb+rp];
			if ( bl >= C.blcap ) { over(2048); return; }
			uint8_t * H = L.hbl + 12*bl; uint32_t hn = L.hbl_n[bl];
			if ( hn == 12 )
			{
				if ( W[rp] <= W[H[0]] ) continue;
				else ipop<true>(H,hn,W);
			}
			ipush<true>(H,hn,rp,W);
			L.hbl_n[bl] = hn;
			if ( narp >= 250 ) { over(512); return; }
			L.rc_ord[rb+narp++] = rp;
			if ( L.rc_len[rb+rp] == 0 )
			{
				for ( uint32_t t = 0; t < nview; ++t )
				{
					uint32_t const s = L.ord[t];
					if ( L.slast[s] == lastnode )
					{
						int32_t const rpe = extendReversePath(rp,s);
						if ( rpe < 0 ) return;
						if ( checkReversePathFeasiblePosition(rpe) ) { if ( nrpst >= 250 ) { over(512); return; } ipush<false>(L.rpst,nrpst,rpe,W); }
						else --nrp;
					}
				}
			}
			else if ( static_cast<int64_t>(L.rc_baselen[rb+rp]) < (lmax+1)/2 )
			{
				uint32_t const b = L.rc_stretch[rb+rp];
				uint32_t const bf = L.sfirst[b];
				for ( uint32_t t = 0; t < nview; ++t )
				{
					uint32_t const a = L.ord[t];
					if ( L.slast[a] == bf && linkOk(a,b) )
					{
						int32_t const rpe = extendReversePath(rp,a);
						if ( rpe < 0 ) return;
						if ( checkReversePathFeasiblePosition(rpe) ) { if ( nrpst >= 250 ) { over(512); return; } ipush<false>(L.rpst,nrpst,rpe,W); }
						else --nrp;
					}
				}
			}
		}
		arpSort(L.rc_ord+rb,L.rc_ord+rb+narp);
		for ( uint32_t i = 0; i < narp; ++i )
		{
			uint64_t const wi = W[L.rc_ord[rb+i]];
			uint32_t r = 0;
			for ( uint32_t j = 0; j < narp; ++j )
			{
				uint64_t const wj = W[L.rc_ord[rb+j]];
				if ( wj < wi || (wj == wi && j < i) ) ++r;
			}
			L.rc_arw[rb+i] = r;
		}
	}
	// can the cached reverse block (computed on the view of `last` alone) be used when stretch `par` is split at node `fn`?
	// the scans of the enumeration look for stretches whose last node equals a target; the split changes the answer
	// only for targets par.last (parent vs second piece) and fn (first piece)
	DEV bool reverseUnaffected(uint32_t const base, uint32_t const nacc2, int32_t const lastnode, uint32_t const par, uint32_t const fn, int64_t const lmax) const
	{
		uint32_t const plast = L.slast[par];
		if ( lastnode >= 0 && (static_cast<uint32_t>(lastnode) == plast || static_cast<uint32_t>(lastnode) == fn) ) return false;
		for ( uint32_t i = 0; i < nacc2; ++i )
		{
			uint32_t const rp = L.rc_ord[base+i];
			if ( L.rc_len[base+rp] && static_cast<int64_t>(L.rc_baselen[base+rp]) < (lmax+1)/2 )
			{
				uint32_t const target = L.sfirst[L.rc_stretch[base+rp]];
				if ( target == plast || target == fn ) return false;
			}
		}
		return true;
	}

	// ================= forward enumeration on the current view (lane 0) =================
	uint32_t apqlo, apqhi;
	DEV int32_t extendPath(int32_t const parent, uint32_t const s)
	{
		if ( np >= C.fcap || np >= 250 ) { over(512); return -1; }
		uint32_t const id = np++;
		uint32_t const ppos = parent >= 0 ? L.f_pos[parent] : 0;
		uint32_t const plen = parent >= 0 ? L.f_len[parent] : 0;
		uint64_t weight = parent >= 0 ? L.f_w[parent] : 0;
		uint32_t baselen = parent >= 0 ? L.f_baselen[parent] : 0;
		int32_t const sfo = sfFind(s,ppos);
		if ( plen == 0 ) { baselen = L.sslen[s]+k-1; weight = sfo >= 0 ? L.wuF[sfo] : 0; }
		else { baselen += L.sslen[s]-1; if ( sfo >= 0 ) weight += L.wuF[sfo] - nodeU(L.sfirst[s],ppos,false); }
		uint32_t const npos = ppos + (L.sslen[s]-1);
		if ( baselen > 255 || npos > 255 || plen+1 > 255 ) { over(2048); --np; return -1; }
		L.f_parent[id] = parent >= 0 ? parent : 0xFF; L.f_stretch[id] = s; L.f_len[id] = plen+1; L.f_pos[id] = npos;
		L.f_w[id] = weight; L.f_baselen[id] = baselen;
		return id;
	}
	DEV bool apqPush(uint32_t const id)
	{
		uint32_t const bl = L.f_baselen[id];
		if ( bl >= C.blcap ) { over(2048); return false; }
		uint8_t * H = L.hbl + 12*bl; uint32_t hn = L.hbl_n[bl];
		if ( hn == 12 )
		{
			if ( L.f_w[id] > L.f_w[H[0]] ) { ipop<true>(H,hn,L.f_w); ipush<true>(H,hn,id,L.f_w); }
		}
		else ipush<true>(H,hn,id,L.f_w);
		L.hbl_n[bl] = hn;
		if ( bl < apqlo ) apqlo = bl;
		if ( bl+1 > apqhi ) apqhi = bl+1;
		return true;
	}
	// path tree of traverse :4838-5041 without the score-interval part (that one depends on the reverse block)
	DEV void forwardEnumerate(int32_t const firstnode, int64_t const lmax)
	{
		np = 0; nfpop = 0;
		for ( uint32_t i = 0; i < C.blcap; ++i ) L.hbl_n[i] = 0;
		apqlo = C.blcap; apqhi = 0;
		for ( uint32_t t = 0; t < nview; ++t )
		{
			uint32_t const s = L.ord[t];
			if ( L.sfirst[s] == firstnode )
			{
				int32_t const id = extendPath(-1,s);
				if ( id < 0 || !apqPush(id) ) return;
			}
		}
		for ( uint32_t zz = apqlo; zz < apqhi; ++zz )
			while ( L.hbl_n[zz] )
			{
				uint8_t * H = L.hbl + 12*zz; uint32_t hn = L.hbl_n[zz];
				uint32_t const path = H[0];
				ipop<true>(H,hn,L.f_w);
				L.hbl_n[zz] = hn;
				if ( nfpop >= C.fcap ) { over(512); return; }
				L.fpop[nfpop++] = path;
				uint32_t const pbl = L.f_baselen[path];
				if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
				{
					uint32_t const lastn = L.slast[L.f_stretch[path]];
					for ( uint32_t t = viewLowerBound(lastn); t < nview && L.sfirst[L.ord[t]] == lastn; ++t )
					{
						uint32_t const s = L.ord[t];
						int32_t const sfo = sfFind(s,L.f_pos[path]);
						uint64_t const eweight = sfo >= 0 ? L.wuF[sfo] : 0;
						if ( eweight >= FW_THRES_01 )
						{
							int32_t const ep = extendPath(path,s);
							if ( ep < 0 ) return;
							if ( L.f_w[ep] >= FW_THRES_01 && static_cast<int64_t>(L.f_pos[ep]) + k <= lmax )
							{
								if ( !apqPush(ep) ) return;
							}
							else --np;
						}
					}
				}
			}
	}
	// scans of the forward enumeration look for stretches whose first node equals a target; splitting `par` at `ln`
	// changes the answer only for targets par.first (parent vs first piece) and ln (second piece)
	DEV bool forwardUnaffected(int32_t const firstnode, uint32_t const par, uint32_t const ln, int64_t const lmax) const
	{
		uint32_t const pfirst = L.sfirst[par];
		if ( static_cast<uint32_t>(firstnode) == pfirst || static_cast<uint32_t>(firstnode) == ln ) return false;
		for ( uint32_t i = 0; i < nfpop; ++i )
		{
			uint32_t const path = L.fpop[i];
			uint32_t const pbl = L.f_baselen[path];
			if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
			{
				uint32_t const target = L.slast[L.f_stretch[path]];
				if ( target == pfirst || target == ln ) return false;
			}
		}
		return true;
	}

	// ================= combining a forward tree with a reverse block (score intervals + pair loop) =================
	DEV uint64_t getPairScore(uint32_t const path, uint32_t const base, uint32_t const rp) const
	{
		uint32_t const s = L.f_stretch[path];
		uint32_t const spos = L.f_pos[path] - (L.sslen[s]-1);
		int32_t const sfo = sfFind(s,spos);
		if ( sfo >= 0 ) return L.f_w[path] + L.rc_w[base+rp] - nodeU(L.slast[s],L.f_pos[path],false);
		else return L.f_w[path] + L.rc_w[base+rp];
	}
	DEV uint32_t rcFront(uint32_t const base, uint32_t const rp, uint32_t const lastkmer) const { return L.rc_len[base+rp] ? L.nv[L.sfirst[L.rc_stretch[base+rp]]] : lastkmer; }
	// decodePathPair :4267-4300 into dst (2-bit codes); returns length or ~0
	DEV uint32_t decodePathPair(uint32_t const path, uint32_t const base, uint32_t const rp, uint8_t * dst)
	{
		uint8_t chain[64]; uint32_t cl = 0;
		for ( uint32_t q = path; q != 0xFF; q = L.f_parent[q] ) { if ( cl >= 64 ) { over(4096); return ~0u; } chain[cl++] = L.f_stretch[q]; }
		uint32_t need = k;
		for ( uint32_t i = 0; i < cl; ++i ) need += L.sslen[chain[i]]-1;
		for ( uint32_t q = rp; L.rc_len[base+q]; q = L.rc_parent[base+q] ) need += L.sslen[L.rc_stretch[base+q]]-1;
		if ( need > MAXCONS ) { over(4096); return ~0u; }
		uint32_t o = 0;
		uint32_t const firstv = L.nv[L.sfirst[chain[cl-1]]];
		for ( uint32_t i = 0; i < k; ++i ) dst[o++] = (firstv >> (2*(k-1-i))) & 3;
		for ( uint32_t ii = 0; ii < cl; ++ii )
		{
			uint32_t const s = chain[cl-1-ii]; uint16_t const * Lk = L.links + L.slink[s];
			for ( uint32_t j = 1; j < L.sslen[s]; ++j ) dst[o++] = L.nv[Lk[j]] & 3;
		}
		for ( uint32_t q = rp; L.rc_len[base+q]; q = L.rc_parent[base+q] )
		{
			uint32_t const s = L.rc_stretch[base+q]; uint16_t const * Lk = L.links + L.slink[s];
			for ( uint32_t j = 1; j < L.sslen[s]; ++j ) dst[o++] = L.nv[Lk[j]] & 3;
		}
		return o;
	}
	DEV void combinePair(uint32_t const base, uint32_t const nacc2, uint32_t const lastkmer, int64_t const lmin, int64_t const lmax, uint32_t const maxfullpath)
	{
		nsiq = 0;
		for ( uint32_t pi = 0; pi < nfpop; ++pi )
		{
			uint32_t const path = L.fpop[pi];
			int64_t const candlen = static_cast<int64_t>(L.f_pos[path]) + k;
			uint32_t const front = L.nv[L.slast[L.f_stretch[path]]];
			uint32_t lo = 0, hi = nacc2;
			while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( rcFront(base,L.rc_ord[base+mid],lastkmer) < front ) lo = mid+1; else hi = mid; }
			uint32_t e = lo;
			while ( e < nacc2 && rcFront(base,L.rc_ord[base+e],lastkmer) == front ) ++e;
			int64_t bllo = lmin + static_cast<int64_t>(k) - candlen; if ( bllo < 0 ) bllo = 0;
			int64_t blhi = lmax + static_cast<int64_t>(k) - candlen; if ( blhi < 0 ) blhi = 0;
			uint32_t const bllo16 = static_cast<uint16_t>(bllo), blhi16 = static_cast<uint16_t>(blhi);
			uint32_t sub = lo;
			while ( sub < e && L.rc_baselen[base+L.rc_ord[base+sub]] < bllo16 ) ++sub;
			uint32_t sup = sub;
			while ( sup < e && !(blhi16 < L.rc_baselen[base+L.rc_ord[base+sup]]) ) ++sup;
			if ( sub != sup )
			{
				uint32_t mi = sub;
				for ( uint32_t i = sub+1; i < sup; ++i ) if ( L.rc_arw[base+i] > L.rc_arw[base+mi] ) mi = i;
				if ( nsiq >= C.siqcap ) { over(512); return; }
				FSI si; si.left = sub; si.right = sup; si.current = mi; si.path = path; si.pad = 0; si.w = getPairScore(path,base,L.rc_ord[base+mi]);
				spush<FSI,false>(L.siq,nsiq,si);
			}
		}
		prevlen = ~0u;
		for ( uint32_t numfullpath = 0; nsiq && numfullpath < maxfullpath; ++numfullpath )
		{
			FSI const si = L.siq[0];
			// the score intervals leave the heap in non increasing weight order and everything still inside is not
			// heavier, so once the candidate heap is full and its top cannot be beaten, nothing of this pair can enter
			if ( ncdh == 16 && si.w <= L.cdh[0].w ) break;
			spop<FSI,false>(L.siq,nsiq);
			{
				uint32_t const v = L.rc_arw[base+si.current];
				if ( v )
				{
					bool found = false; uint32_t bu = 0, bi = 0;
					for ( uint32_t i = si.left; i < si.right; ++i )
					{
						uint32_t const r = L.rc_arw[base+i];
						if ( r <= v-1 && (!found || r > bu) ) { found = true; bu = r; bi = i; }
					}
					if ( found )
					{
						FSI sic = si; sic.current = bi; sic.w = getPairScore(si.path,base,L.rc_ord[base+bi]);
						if ( nsiq >= C.siqcap ) { over(512); return; }
						spush<FSI,false>(L.siq,nsiq,sic);
					}
				}
			}
			uint64_t const weight = si.w;
			if ( ncdh == 16 ) spop<FCC,true>(L.cdh,ncdh);   // weight > top here
			uint32_t const conslen = decodePathPair(si.path,base,L.rc_ord[base+si.current],L.curstr);
			if ( conslen == ~0u ) return;
			if ( conslen == prevlen )
			{
				bool eq = true;
				for ( uint32_t i = 0; i < conslen; ++i ) if ( L.prevstr[i] != L.curstr[i] ) { eq = false; break; }
				if ( eq ) continue;
			}
			for ( uint32_t i = 0; i < conslen; ++i ) L.prevstr[i] = L.curstr[i];
			prevlen = conslen;
			if ( conso + conslen > C.conscap - MAXCONS ) { over(4096); return; }
			for ( uint32_t i = 0; i < conslen; ++i ) G.cons[conso+i] = L.curstr[i];
			FCC cc; cc.w = weight; cc.o = conso; cc.l = conslen;
			conso += conslen;
			spush<FCC,true>(L.cdh,ncdh,cc);
		}
	}

	DEV uint32_t myersDistance(uint32_t const j, uint8_t const * text, uint32_t const n) const
	{
		uint32_t const m = L.slen[j];
		if ( m == 0 ) return n;
		uint64_t const * PEQ = L.peq + 4*j;
		uint32_t score = m;
		uint64_t Pv = ~0ull, Mv = 0;
		uint64_t const top = 1ull<<(m-1);
		for ( uint32_t c = 0; c < n; ++c )
		{
			uint64_t const Eq = PEQ[text[c]];
			uint64_t const Xv = Eq | Mv;
			uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			if ( Ph & top ) ++score; else if ( Mh & top ) --score;
			Ph = (Ph<<1) | 1ull; Mh <<= 1;
			Pv = Mh | ~(Xv | Ph);
			Mv = Ph & Xv;
		}
		return score;
	}
	DEV void buildPeq()
	{
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint64_t e[4] = {0,0,0,0};
			uint32_t const m = L.slen[j];
			uint8_t const * s = L.str + j*64;
			for ( uint32_t i = 0; i < m; ++i ) e[s[i]] |= 1ull<<i;
			for ( uint32_t i = 0; i < 4; ++i ) L.peq[4*j+i] = e[i];
		}
		wv_sync();
	}

	// ================= traverse (:4496-5170) for one activation state =================
	DEV bool traverse(int64_t const lmin, int64_t const lmax)
	{
		PROF_T0
		conso = 0; ncdh = 0; nacc = 0; nwF = 0; nwR = 0; rctop = 0; fcur_fi = -1; fcur_li = -1;
		computeBaseStretches();
		flags = wv_or(flags); if ( flags ) return false;
		PROF(*this,8)
		findCandidatesAndPieces();
		flags = wv_or(flags); if ( flags ) return false;
		computeStretchFeas(0,npool);
		flags = wv_or(flags); if ( flags ) return false;
		for ( uint32_t i = lane; i < FNC+1; i += WSZ ) L.rvalid[i] = 0;
		wv_sync();
		PROF(*this,9)
#if defined(DACC_EMUL) && defined(DACC_FSTATS)
		fstat(0,nwF); fstat(1,nwR); fstat(2,npool); fstat(3,nF); fstat(4,nL); fstat(5,nlinks); fstat(6,nn); fstat(7,npre);
#endif
		for ( uint32_t fi = 0; fi < nF; ++fi )
		{
			int32_t const firstnode = L.fnode[fi] == 0xFFFF ? -1 : L.fnode[fi];
			uint32_t const sf = L.parF[fi];
			for ( uint32_t li = 0; li < nL; ++li )
			{
				uint32_t const lastk = L.lkmer[li];
				int32_t const lastnode = L.lnode[li] == 0xFFFF ? -1 : L.lnode[li];
				uint32_t const sl = L.parL[li];
				bool const same = (sf != FNOPAR && sf == sl);
				// ---- the pair's exact stretch set (split at first, then at last)
				uint32_t remX[2], addX[4]; uint32_t nremX = 0, naddX = 0; bool needMid = false; uint32_t midA = 0, midB = 0;
				if ( same )
				{
					uint32_t const pf = L.posF[fi], pl = L.posL[li];
					remX[nremX++] = sf;
					if ( pf == pl ) { addX[naddX++] = L.pieF[fi]; addX[naddX++] = L.pieF[fi]+1; }
					else if ( pf < pl ) { addX[naddX++] = L.pieF[fi]; addX[naddX++] = L.pieL[li]+1; needMid = true; midA = pf; midB = pl; }
					else { addX[naddX++] = L.pieL[li]; addX[naddX++] = L.pieF[fi]+1; needMid = true; midA = pl; midB = pf; }
				}
				else
				{
					if ( sf != FNOPAR ) { remX[nremX++] = sf; addX[naddX++] = L.pieF[fi]; addX[naddX++] = L.pieF[fi]+1; }
					if ( sl != FNOPAR ) { remX[nremX++] = sl; addX[naddX++] = L.pieL[li]; addX[naddX++] = L.pieL[li]+1; }
				}
				// ---- reverse block: cached per last k-mer when the first k-mer's split cannot touch it
				bool rcached = false;
				if ( !same )
				{
					if ( ! L.rvalid[li] )
					{
						uint32_t remL[1], addL[2]; uint32_t nr = 0, na = 0;
						if ( sl != FNOPAR ) { remL[nr++] = sl; addL[na++] = L.pieL[li]; addL[na++] = L.pieL[li]+1; }
						buildView(remL,nr,addL,na);
						if ( lane == 0 )
						{
							rb = rctop;
							reverseEnumerate(lastk,lastnode,lmax);
							L.rbase[li] = rb; L.rn[li] = narp; L.rnpool[li] = nrp; L.rvalid[li] = 1;
							rctop = rb + nrp;
						}
						wv_sync();
						flags = wv_bcast(flags,0); if ( flags ) return false;
						rctop = wv_bcast(rctop,0);
					}
					rcached = (sf == FNOPAR);
					if ( !rcached )
					{
						uint32_t ok = 0;
						if ( lane == 0 ) ok = reverseUnaffected(L.rbase[li],L.rn[li],lastnode,sf,L.fnode[fi],lmax) ? 1 : 0;
						rcached = wv_bcast(ok,0) != 0;
					}
				}
				// ---- forward tree: cached for the current first k-mer when the last k-mer's split cannot touch it
				bool fcached = false;
				if ( !same )
				{
					if ( fcur_fi != static_cast<int32_t>(fi) || fcur_li != -1 )
					{
						uint32_t remF[1], addF[2]; uint32_t nr = 0, na = 0;
						if ( sf != FNOPAR ) { remF[nr++] = sf; addF[na++] = L.pieF[fi]; addF[na++] = L.pieF[fi]+1; }
						buildView(remF,nr,addF,na);
						if ( lane == 0 ) forwardEnumerate(firstnode,lmax);
						wv_sync();
						flags = wv_bcast(flags,0); if ( flags ) return false;
						fcur_fi = fi; fcur_li = -1;
					}
					fcached = (sl == FNOPAR);
					if ( !fcached )
					{
						uint32_t ok = 0;
						if ( lane == 0 ) ok = forwardUnaffected(firstnode,sl,L.lnode[li],lmax) ? 1 : 0;
						fcached = wv_bcast(ok,0) != 0;
					}
				}
				// ---- exact enumeration where a cache cannot be used
				uint32_t swF = nwF, swR = nwR;
				if ( !rcached || !fcached )
				{
					if ( needMid )
					{
						// middle piece of a stretch split twice: a temporary pool entry with its own feasibility
						if ( npool+1 > C.scap ) { over(32); return false; }
						if ( lane == 0 ) makePiece(npool,sf,midA,midB);
						wv_sync();
						computeStretchFeas(npool,npool+1);
						flags = wv_or(flags); if ( flags ) return false;
						addX[naddX++] = npool;
					}
					buildView(remX,nremX,addX,naddX);
					if ( lane == 0 )
					{
						if ( !rcached ) { rb = rctop; reverseEnumerate(lastk,lastnode,lmax); }
						if ( !fcached && !flags ) forwardEnumerate(firstnode,lmax);
					}
					wv_sync();
					flags = wv_bcast(flags,0); if ( flags ) return false;
					if ( !fcached ) { fcur_fi = fi; fcur_li = li; }
				}
				if ( lane == 0 )
				{
					uint32_t base, nacc2;
					if ( rcached ) { base = L.rbase[li]; nacc2 = L.rn[li]; } else { base = rb; nacc2 = narp; }
					combinePair(base,nacc2,lastk,lmin,lmax,16);
				}
				wv_sync();
				flags = wv_bcast(flags,0); if ( flags ) return false;
				nwF = swF; nwR = swR;   // drop the weights of a temporary middle piece
			}
		}
		PROF(*this,12)
#if defined(DACC_EMUL) && defined(DACC_FSTATS)
		fstat(8,rctop); fstat(9,np); fstat(10,conso);
#endif
		if ( lane == 0 )
		{
			uint32_t nch = 0;
			while ( ncdh ) { FCC const c = L.cdh[0]; spop<FCC,true>(L.cdh,ncdh); spush<FCC,false>(L.ch,nch,c); }
			while ( nch ) { L.acc[nacc++] = L.ch[0]; spop<FCC,false>(L.ch,nch); }
		}
		wv_sync();
		uint32_t const nc = wv_bcast(nacc,0);
		nacc = nc;
		for ( uint32_t t = lane; t < nc*mao; t += WSZ )
		{
			uint32_t const c = t / mao, j = t - c*mao;
			L.canderr[t] = myersDistance(j,G.cons + L.acc[c].o,L.acc[c].l);
		}
		wv_sync();
		if ( lane == 0 )
		{
			for ( uint32_t c = 0; c < nc; ++c )
			{
				uint32_t s = 0;
				for ( uint32_t j = 0; j < mao; ++j ) s += L.canderr[c*mao+j];
				L.accerr[c] = s;
			}
			for ( uint32_t i = 1; i < nc; ++i )
			{
				FCC const v = L.acc[i]; uint32_t const e = L.accerr[i];
				if ( e < L.accerr[0] )
				{
					for ( uint32_t q = i; q > 0; --q ) { L.acc[q] = L.acc[q-1]; L.accerr[q] = L.accerr[q-1]; }
					L.acc[0] = v; L.accerr[0] = e;
				}
				else
				{
					uint32_t q = i;
					while ( e < L.accerr[q-1] ) { L.acc[q] = L.acc[q-1]; L.accerr[q] = L.accerr[q-1]; --q; }
					L.acc[q] = v; L.accerr[q] = e;
				}
			}
		}
		wv_sync();
		PROF(*this,13)
		return nc != 0;
	}

	DEV int32_t estimateLength()
	{
		int32_t maxvprodindex = -1;
		uint32_t mn = 0xFFFFFFFFu, mx = 0;
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t const len = L.slen[j];
			uint32_t const lastpos = len ? len-1 : 0;
			mn = lastpos < mn ? lastpos : mn; mx = lastpos > mx ? lastpos : mx;
		}
		mn = ~wv_max(~mn); mx = wv_max(mx);
		uint32_t const supStart = mn < static_cast<uint32_t>(T.nsup) ? T.suplo[mn] : T.nrows;
		uint32_t const supEnd = mx < static_cast<uint32_t>(T.nsup) ? T.suphi[mx] : T.nrows;
		uint64_t bestbits = 0; uint32_t besti = 0xFFFFFFFFu;
		for ( uint32_t c = supStart; c < supEnd; c += WSZ )
		{
			uint32_t const i = c + lane;
			double vprod = 0.0;
			if ( i < supEnd )
			{
				double const * row = T.dpnorm + static_cast<uint64_t>(i)*T.nsup;
				vprod = 1.0;
				for ( uint32_t j = 0; j < mao; ++j )
				{
					uint32_t const len = L.slen[j];
					if ( len ) vprod *= ((len-1) < static_cast<uint32_t>(T.nsup) ? row[len-1] : 0.0);
				}
			}
			union { double d; uint64_t u; } cv; cv.d = vprod;
			uint64_t const mb = wv_max64(cv.u);
			if ( mb > bestbits )
			{
				uint64_t const fi = wv_min64(cv.u == mb ? i : 0xFFFFFFFFull);
				bestbits = mb; besti = static_cast<uint32_t>(fi);
			}
		}
		union { double d; uint64_t u; } dm; dm.d = DACC_DBL_MIN;
		if ( bestbits > dm.u ) maxvprodindex = besti;
		if ( maxvprodindex == -1 )
		{
			// density fallback (HandleContext.hpp:2103-2155): histogram of lengths (count-1) against the DPnormSquare rows
			int32_t res = -1;
			if ( lane == 0 )
			{
				uint32_t maxlen = 0;
				for ( uint32_t j = 0; j < mao; ++j ) maxlen = L.slen[j] > maxlen ? L.slen[j] : maxlen;
				int32_t maxoff = -1; double maxoffv = DACC_DBL_MIN;
				for ( int32_t i = 0; i < T.nrows; ++i )
				{
					uint32_t const fs = T.dpsq_first[i], sz = T.dpsq_size[i];
					double const * V = T.dpsq + static_cast<uint64_t>(i)*T.nsup;
					double sdot = 0;
					for ( uint32_t t = 0; t < sz; ++t )
					{
						uint32_t const jj = fs+t;
						if ( jj < maxlen+1 )
						{
							uint32_t cnt = 0;
							for ( uint32_t j = 0; j < mao; ++j ) cnt += (L.slen[j] == jj);
							double const o = cnt ? static_cast<double>(cnt-1) : 0.0;
							sdot += V[jj] * o;
						}
						else break;
					}
					if ( sdot > maxoffv ) { maxoff = i; maxoffv = sdot; }
				}
				if ( maxoff != -1 && maxoffv >= 1e-3 ) res = maxoff;
			}
			maxvprodindex = static_cast<int32_t>(wv_bcast(static_cast<uint32_t>(res),0));
		}
		return maxvprodindex;
	}

	DEV void alignAndEmit(uint8_t const * cons, uint32_t const n, uint8_t * rec)
	{
		uint32_t const m = P.w;
		uint8_t const * a = L.str;
		uint64_t const * PEQ = L.peq;
		uint64_t const mask = (m == 64) ? ~0ull : ((1ull<<m)-1);
		uint64_t Pv = mask, Mv = 0; uint32_t score = m;
		L.alpv[0] = Pv; L.almv[0] = Mv; L.albot[0] = m;
		uint64_t const top = 1ull<<(m-1);
		for ( uint32_t c = 0; c < n; ++c )
		{
			uint64_t const Eq = PEQ[cons[c]];
			uint64_t const Xv = Eq | Mv;
			uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			if ( Ph & top ) ++score; else if ( Mh & top ) --score;
			Ph = (Ph<<1) | 1ull; Mh <<= 1;
			Pv = (Mh | ~(Xv | Ph)) & mask;
			Mv = (Ph & Xv) & mask;
			L.alpv[c+1] = Pv; L.almv[c+1] = Mv; L.albot[c+1] = score;
		}
		uint32_t i = m, j = n; uint32_t d = score; uint32_t nops = 0;
		while ( i || j )
		{
			uint32_t op = 2; bool done = false;
			if ( i && j )
			{
				uint64_t const sh = i-1;
				uint32_t const dd = L.albot[j-1] - dacc_popc64(L.alpv[j-1]>>sh) + dacc_popc64(L.almv[j-1]>>sh);
				uint32_t const neq = (a[i-1] != cons[j-1]);
				if ( dd + neq == d ) { op = neq ? 1 : 0; --i; --j; d = dd; done = true; }
			}
			if ( !done && i )
			{
				uint64_t const bit = 1ull<<(i-1);
				if ( L.alpv[j] & bit ) { op = 3; --i; d = d-1; done = true; }
			}
			if ( !done ) { op = 2; --j; d = d-1; }
			L.alops[nops++] = op;
		}
		uint8_t * off = rec+1; uint8_t * sym = rec + 1 + (m+2);
		rec[0] = 1;
		uint32_t so = 0, cpos = 0, t = nops;
		for ( uint32_t r = 0; r <= m; ++r )
		{
			off[r] = so;
			while ( t && L.alops[t-1] == 2 ) { sym[so++] = cons[cpos++]; --t; }
			if ( r < m )
			{
				uint32_t const op = L.alops[--t];
				sym[so++] = (op == 3) ? 4 : cons[cpos++];
			}
		}
		off[m+1] = so;
	}
};

// returns true if the window was completed on the fast path, false if it must be re-run generically
DEV bool processWindowFast(FastBatch const & FB, uint64_t const widx, uint8_t * lds, uint8_t * garena)
{
	WindowBatch const & B = FB.W;
	FastEngine E;
	E.C = FB.F; E.T = B.T; E.P = B.P;
	E.lane = wv_lane(); E.flags = 0; E.prof = B.prof;
	fast_lds_carve(E.L,lds,FB.F);
	fast_global_carve(E.G,garena,FB.F);
	FastLds & L = E.L;
	int const lane = E.lane;
	PROF_T0
	#define FFAIL(code) { if ( lane == 0 ) { B.wout[widx].status = WS_RETRY; B.wout[widx].flags = ((code)<<24) | (E.flags & 0xFFFFFF); } return false; }

	uint32_t lo = 0, hi = B.npiles;
	while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( B.piles[mid].winbase <= widx ) lo = mid; else hi = mid; }
	DevPile const pile = B.piles[lo];
	uint32_t const y = static_cast<uint32_t>(widx - pile.winbase);
	uint32_t astart, aend;
	windowInterval(pile.l,B.P.a,B.P.w,y,astart,aend);

	WindowOut out; out.status = WS_INSUFFICIENT; out.mao = 0; out.elength = 0; out.k = 0; out.filterfreq = -1; out.conslen = 0; out.minrate = 0; out.flags = 0;
	uint8_t * rec = B.wrec + widx*WREC;
	if ( lane == 0 ) rec[0] = 0;
	if ( B.P.w > 63 ) { FFAIL(1) }

	DevOvl const * ov = B.ovl + pile.first_ovl;
	uint32_t nact = 0;
	for ( uint32_t c = 0; c < pile.novl; c += WSZ )
	{
		uint32_t const z = c + lane;
		uint32_t act = 0;
		if ( z < pile.novl ) act = (ov[z].abpos <= static_cast<int32_t>(astart)) && (ov[z].aepos >= static_cast<int32_t>(aend));
		uint32_t tot; uint32_t const pre = wv_scan_excl(act,tot);
		if ( act && nact+pre < FB.F.precap ) L.pre[nact+pre] = (static_cast<uint64_t>(ov[z].ekey)<<32) | z;
		nact += tot;
	}
	if ( nact > FB.F.precap ) { FFAIL(2) }
	uint32_t const ap2 = next_pow2(nact < 2 ? 2 : nact);
	for ( uint32_t i = nact + lane; i < ap2; i += WSZ ) L.pre[i] = ~0ull;
	wv_sync();
	wv_bitonic_sort(L.pre,ap2);
	uint32_t mao = 0;
	if ( nact )
	{
		uint64_t const nb = (B.P.maxalign > 0) ? (B.P.maxalign-1) : 0;
		mao = 1 + static_cast<uint32_t>(nact < nb ? nact : nb);
	}
	if ( mao > FB.F.maxs ) { FFAIL(3) }
	E.mao = mao; out.mao = mao;

	uint32_t toolong = 0;
	if ( mao )
	{
		uint64_t const aoff = B.boff[pile.aread];
		for ( uint32_t p = lane; p < B.P.w; p += WSZ ) L.str[p] = readBase(B.bps,aoff,B.rlen[pile.aread],false,astart+p);
		if ( lane == 0 ) L.slen[0] = B.P.w;
		for ( uint32_t j = 1; j < mao; ++j )
		{
			uint32_t const z = static_cast<uint32_t>(L.pre[j-1] & 0xFFFFFFFFu);
			DevOvl const & o = ov[z];
			uint64_t const row = o.wtoff + (y - o.y0);
			uint32_t const bs = B.wt_b[row], be = B.wt_e[row];
			uint32_t const len = be-bs;
			if ( len > 64 ) { toolong = 1; continue; }
			uint64_t const off = B.boff[o.bread]; uint32_t const rl = B.rlen[o.bread]; bool const inv = o.flags & 1;
			for ( uint32_t p = lane; p < len; p += WSZ ) L.str[j*64+p] = readBase(B.bps,off,rl,inv,bs+p);
			if ( lane == 0 ) L.slen[j] = len;
		}
	}
	wv_sync();
	if ( toolong ) { FFAIL(4) }
	PROF(E,0)

	int32_t elength = 0;
	if ( mao )
	{
		E.buildPeq();
		elength = E.estimateLength()+1;
	}
	out.elength = elength;
	PROF(E,1)

	if ( mao >= B.P.minwindowcov )
	{
		bool pathfailed = true;
		uint64_t minrate = B.P.eminrate;
		bool haveMin = false;
		uint32_t bestlen = 0;
		uint8_t * best = E.G.cons + (FB.F.conscap - MAXCONS);
		for ( uint32_t k = B.P.klow; k <= B.P.khigh; ++k )
		{
			E.k = k; E.kmask = (1ull<<(2*k))-1;
			for ( int32_t ff = B.P.maxff; ff >= B.P.minff; --ff )
			{
				PROF_T0
				E.buildInstances();
				PROF(E,2)
				E.buildNodes(ff > 1 ? ff : 1);
				PROF(E,3)
				E.buildSuccessors(mao);
				PROF(E,4)
				if ( ff == 0 )
				{
					// gap filling (HandleContext.hpp:2233-2268)
					E.levelSuccessors2();
					E.flags = wv_or(E.flags);
					if ( E.flags ) { FFAIL(6) }
					E.buildNodes(1);
					E.buildSuccessors(mao);
					PROF(E,6)
				}
				E.flags = wv_or(E.flags);
				if ( E.flags ) { FFAIL(7) }
				uint32_t mintry = 0; bool lconsok = false;
				while ( true )
				{
					bool const consok = E.traverse(static_cast<int64_t>(elength)-4,static_cast<int64_t>(elength)+4);
					E.flags = wv_or(E.flags);
					if ( E.flags ) { FFAIL(8) }
					if ( consok )
					{
						uint64_t const err = L.accerr[0];
						if ( err < minrate )
						{
							lconsok = true; minrate = err; haveMin = true;
							bestlen = L.acc[0].l;
							if ( bestlen > MAXCONS ) { FFAIL(9) }
							for ( uint32_t i = lane; i < bestlen; i += WSZ ) best[i] = E.G.cons[L.acc[0].o+i];
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
				if ( lconsok ) { pathfailed = false; break; }
			}
		}
		if ( !pathfailed )
		{
			out.status = WS_OK; out.conslen = bestlen; out.minrate = minrate;
			PROF_T0
			if ( lane == 0 ) E.alignAndEmit(best,bestlen,rec);
			PROF(E,14)
		}
		else out.status = WS_FAILED;
	}
	if ( lane == 0 ) B.wout[widx] = out;
	wv_sync();
	return true;
	#undef FFAIL
}

}
#endif
