/*
 * LDS-resident fast path of the per-window de Bruijn consensus (one wavefront = one window).
 *
 * Same algorithm and exactly the same results as the generic engine (dbg_window.hpp); what
 * changes is where the state lives and how the heavy phases are laid out for CDNA4:
 *   - all per-window working state (strings, Myers masks, k-mer instances, nodes, stretches,
 *     path pools, bounded heaps) sits in the workgroup's LDS slice with 8/16-bit fields,
 *     ~40 KB per wavefront => 4 resident wavefronts per CU; the sort buffer of the build
 *     phase is overlaid by the traversal structures,
 *   - node successors are found without a table: the <=4 successor k-mers of a node are adjacent in the
 *     sorted node-key array, so one lower bound per node is kept,
 *   - k-mer feasibility is never materialised: stretch feasibility is evaluated with
 *     lanes = candidate start positions (<= 64) straight from the fixed-point table, every lane
 *     walking the same (wave-uniform) node/instance loop, results kept as a 64-bit position
 *     mask per stretch plus a compact weight table,
 *   - stretch links are mask intersections.
 * Windows that do not fit the LDS capacities (deep piles, strings > 64, gap filling at filter
 * frequency 0, w > 63) are flagged WS_RETRY and re-run by the generic engine; they are a small
 * minority and the result is identical either way.
 */
#ifndef DACC_FAST_WINDOW_HPP
#define DACC_FAST_WINDOW_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "arena.hpp"
#include "window_main.hpp"

namespace dacc {

enum { WS_RETRY = 4 };

struct FastCaps
{
	uint32_t maxs, precap, ncap, scap, lcap, pcapr, pcapf, siqcap, blcap, sfcap, conscap, pad;
	uint32_t nrows, nsup;        // dimensions of the fixed-point table copy held in LDS
	uint32_t ldsbytes, pad2;
	uint64_t gbytes;
};

struct FastLds
{
	uint8_t * str; uint8_t * slen; uint64_t * peq; uint8_t * ipos; uint8_t * irpos;
	uint32_t * nv; uint16_t * nps; uint8_t * nfreq; uint16_t * succ0; uint16_t * sinfo; uint8_t * npred;
	uint64_t * mfirst; uint64_t * mlast;
	uint8_t * pfrom; uint8_t * pto; uint8_t * cpfrom; uint8_t * cpto;
	uint32_t * tab; uint8_t * suplo8; uint8_t * suphi8;   // per-workgroup copies of the model tables
	// overlay, build phase
	uint64_t * pre; uint64_t * lastk;
	// overlay, traversal phase
	uint16_t * sfirst; uint16_t * slast; uint16_t * sslen; uint16_t * slink; uint64_t * maskF; uint64_t * maskR; uint16_t * woffF; uint16_t * woffR;
	uint16_t * links;
	uint16_t * tfirst; uint16_t * tlast; uint16_t * tslen; uint16_t * tlink; uint64_t * skey;
	uint16_t * rp_parent; uint16_t * rp_stretch; double * rp_weight; uint16_t * rp_pos; uint16_t * rp_len; uint16_t * rp_baselen;
	uint16_t * p_parent; uint16_t * p_stretch; double * p_weight; uint16_t * p_pos; uint16_t * p_len; uint16_t * p_baselen;
	uint16_t * arp; uint16_t * arw; uint16_t * arwr;
	double * rpst_w; uint16_t * rpst_i; HeapSI * siq;
	uint16_t * hbl; uint8_t * hbl_n;
	HeapCC * cdh; HeapCC * ch; HeapCC * acc; double * accerr; uint16_t * canderr;
	uint64_t * alpv; uint64_t * almv; uint16_t * albot; uint8_t * alops;
};

struct FastGlobal { double * wF; double * wR; uint8_t * cons; };

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
	FCARVE(mfirst,uint64_t,keycap)
	FCARVE(mlast,uint64_t,keycap)
	FCARVE(pfrom,uint8_t,C.ncap)
	FCARVE(pto,uint8_t,C.ncap)
	FCARVE(cpfrom,uint8_t,C.ncap)
	FCARVE(cpto,uint8_t,C.ncap)
	FCARVE(tab,uint32_t,C.nrows*C.nsup)
	FCARVE(suplo8,uint8_t,C.nsup)
	FCARVE(suphi8,uint8_t,C.nsup)
	uint64_t const ubase = o;
	// build phase
	FCARVE(pre,uint64_t,C.precap)
	FCARVE(lastk,uint64_t,keycap)
	uint64_t const uA = o;
	// traversal phase (overlays the build phase)
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
	FCARVE(cdh,HeapCC,16)
	FCARVE(ch,HeapCC,16)
	FCARVE(acc,HeapCC,16)
	FCARVE(accerr,double,16)
	FCARVE(canderr,uint16_t,16*C.maxs)
	uint64_t const pbase = o;
	// raw stretches (until the final stretch arrays exist), then pools / heaps, then the final alignment
	FCARVE(tfirst,uint16_t,C.scap)
	FCARVE(tlast,uint16_t,C.scap)
	FCARVE(tslen,uint16_t,C.scap)
	FCARVE(tlink,uint16_t,C.scap)
	FCARVE(skey,uint64_t,next_pow2(C.scap))
	uint64_t const uraw = o;
	o = pbase;
	FCARVE(rp_parent,uint16_t,C.pcapr)
	FCARVE(rp_stretch,uint16_t,C.pcapr)
	FCARVE(rp_weight,double,C.pcapr)
	FCARVE(rp_pos,uint16_t,C.pcapr)
	FCARVE(rp_len,uint16_t,C.pcapr)
	FCARVE(rp_baselen,uint16_t,C.pcapr)
	FCARVE(p_parent,uint16_t,C.pcapf)
	FCARVE(p_stretch,uint16_t,C.pcapf)
	FCARVE(p_weight,double,C.pcapf)
	FCARVE(p_pos,uint16_t,C.pcapf)
	FCARVE(p_len,uint16_t,C.pcapf)
	FCARVE(p_baselen,uint16_t,C.pcapf)
	FCARVE(arp,uint16_t,C.pcapr)
	FCARVE(arw,uint16_t,C.pcapr)
	FCARVE(arwr,uint16_t,C.pcapr)
	uint64_t const hbase = o;
	FCARVE(rpst_w,double,C.pcapr)
	FCARVE(rpst_i,uint16_t,C.pcapr)
	uint64_t const h1 = o;
	o = hbase;
	FCARVE(siq,HeapSI,C.siqcap)
	if ( h1 > o ) o = h1;
	FCARVE(hbl,uint16_t,C.blcap*12)
	FCARVE(hbl_n,uint8_t,C.blcap)
	uint64_t const upool = o;
	o = pbase;
	FCARVE(alpv,uint64_t,MAXCONS+1)
	FCARVE(almv,uint64_t,MAXCONS+1)
	FCARVE(albot,uint16_t,MAXCONS+1)
	FCARVE(alops,uint8_t,2*MAXCONS+2*64+8)
	uint64_t ualn = o;
	uint64_t m = uA;
	if ( uraw > m ) m = uraw;
	if ( upool > m ) m = upool;
	if ( ualn > m ) m = ualn;
	return static_cast<uint32_t>(m);
}

HDEV uint64_t fast_global_carve(FastGlobal & G, uint8_t * base, FastCaps const & C)
{
	uint64_t o = 0;
	G.wF = reinterpret_cast<double *>(base+o); o += 3ull*8*C.sfcap;
	G.wR = reinterpret_cast<double *>(base+o); o += 3ull*8*C.sfcap;
	G.cons = base+o; o += (C.conscap+15)&~15u;
	return o;
}

struct FastBatch
{
	WindowBatch W;              // shared inputs / outputs (arena unused here)
	FastCaps F;
	uint64_t const * dpsq_vst;  // [nsup][nrows] transposed fixed-point table
	uint8_t * garena;           // [gridDim][F.gbytes]
	uint32_t * retry;           // [0] = count, [1..] = window indices to re-run generically
};

// copy the model tables into the workgroup's LDS (once per workgroup)
DEV void fast_load_tables(FastLds & L, FastCaps const & C, DevTables const & T, uint64_t const * vst)
{
	int const lane = wv_lane();
	for ( uint32_t i = lane; i < C.nrows*C.nsup; i += WSZ ) L.tab[i] = static_cast<uint32_t>(vst[i]);
	for ( uint32_t i = lane; i < C.nsup; i += WSZ ) { L.suplo8[i] = T.suplo[i]; L.suphi8[i] = T.suphi[i]; }
	wv_sync();
}

struct FastEngine
{
	FastLds L; FastGlobal G; FastCaps C; DevTables T; DevParams P;
	uint64_t const * vst;
	int lane; uint32_t flags;
	uint64_t * prof;
	uint32_t mao, k; uint64_t kmask;
	uint32_t npre, nlast, nn, nmfirst, nmlast, nstretch, nlinks, nwF, nwR;
	uint32_t nrp, narp, np, nsiq, ncdh, nacc, conso;

	DEV void over(uint32_t b) { flags |= b; }

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
	// i-th successor (descending (freq,sym) order) of node z
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

	// ---- instances + sort (setupPreNodes) ----
	DEV void buildInstances()
	{
		uint32_t base = 0;
		// offsets: per string exclusive scan (maxs <= 64 on the fast path => one chunk on the device)
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

	// ---- nodes (setupNodes + filterFreq) + first/last lists ----
	DEV void buildNodes(uint32_t const f)
	{
		// run heads of the sorted instance array -> kept runs (freq >= f)
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
		// compressed instance arrays (pos, reverse pos)
		for ( uint32_t i = lane; i < npre; i += WSZ )
		{
			uint32_t const pos = (L.pre[i]>>16)&0xFFFF, seq = L.pre[i]&0xFFFF;
			L.ipos[i] = pos; L.irpos[i] = L.slen[seq]-pos-k;
		}
		wv_sync();
		if ( lane == 0 ) L.nps[nn] = npre;
		// per node: rows in which it can be feasible at all (getSupportLow/High of the extreme instance positions)
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const s0 = L.nps[z], f = L.nfreq[z];
			uint32_t lo = L.ipos[s0], hi = L.ipos[s0+f-1];       // instances are position sorted
			uint32_t rlo = 255, rhi = 0;
			for ( uint32_t q = 0; q < f; ++q ) { uint32_t const r = L.irpos[s0+q]; rlo = r < rlo ? r : rlo; rhi = r > rhi ? r : rhi; }
			L.pfrom[z] = lo < C.nsup ? L.suplo8[lo] : C.nrows;
			L.pto[z] = hi < C.nsup ? L.suphi8[hi] : C.nrows;
			L.cpfrom[z] = rlo < C.nsup ? L.suplo8[rlo] : C.nrows;
			L.cpto[z] = rhi < C.nsup ? L.suphi8[rhi] : C.nrows;
		}
		// maxFirst: (count at read position 0, kmer) descending (maxForPosList(0))
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
		uint32_t p2 = next_pow2(nmfirst < 2 ? 2 : nmfirst);
		for ( uint32_t i = nmfirst + lane; i < p2; i += WSZ ) L.mfirst[i] = ~0ull;
		// maxLast: runs of the sorted last k-mers (maxLastList)
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

	// ---- successors + activation (setupAddHeap / setNodesActive) ----
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
				int32_t const t = succNode(z,nsuccact(z));
				uint32_t const fq = L.nfreq[t];
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

	// ---- stretches (computeStretches + splitStretches x2 + stretchesUnique) ----
	DEV void computeStretches(int32_t const firstnode, int32_t const lastsplit)
	{
		PROF_T0
		computePredCounts();
		PROF(*this,16)
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
		uint32_t ns = base;
		if ( ns + 2 > C.scap ) { over(32); nstretch = 0; return; }
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ ) { uint32_t ln; L.tslen[q] = walkStretch(L.tfirst[q],L.tlast[q],0,ln); }
		wv_sync();
		PROF(*this,17)
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
		if ( nlinks > C.lcap ) { over(64); nstretch = 0; return; }
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ ) { uint32_t ln; walkStretch(L.tfirst[q],L.tlast[q],L.links+L.tlink[q],ln); L.tlast[q] = ln; }
		wv_sync();
		PROF(*this,18)
		// splits at `first` and `last`: an interior node belongs to exactly one stretch, so at most one split per round
		for ( int round = 0; round < 2; ++round )
		{
			int32_t const v = round ? lastsplit : firstnode;
			if ( v < 0 ) continue;
			// stretches containing v strictly inside, in index order (an interior node has a unique active
			// predecessor and successor, so normally there is at most one)
			uint32_t const ns0 = ns;
			uint32_t lastdone = 0; bool first = true;
			while ( true )
			{
				uint32_t found = 0xFFFFFFFFu;
				for ( uint32_t q = lane; q < ns0; q += WSZ )
				{
					uint32_t const len = L.tslen[q]; uint16_t const * Lk = L.links + L.tlink[q];
					for ( uint32_t i = 1; i+1 < len; ++i )
						if ( Lk[i] == v )
						{
							uint32_t const key = (q<<16)|i;
							if ( (first || (key>>16) > (lastdone>>16)) && key < found ) found = key;
							break;
						}
				}
				uint32_t const mn = ~wv_max(~found);
				if ( mn == 0xFFFFFFFFu ) break;
				uint32_t const q = mn>>16, split = mn&0xFFFF;
				if ( ns >= C.scap ) { over(32); nstretch = 0; return; }
				if ( lane == 0 )
				{
					uint32_t const len = L.tslen[q]; uint32_t const lo = L.tlink[q];
					L.tfirst[ns] = v; L.tlast[ns] = L.tlast[q]; L.tslen[ns] = len-split; L.tlink[ns] = lo+split;
					L.tlast[q] = v; L.tslen[q] = split+1;
				}
				++ns; lastdone = mn; first = false;
				wv_sync();
			}
			wv_sync();
		}
		PROF(*this,19)
		uint32_t const p2 = next_pow2(ns < 2 ? 2 : ns);
		for ( uint32_t q = lane; q < p2; q += WSZ )
			L.skey[q] = q < ns ?
				((static_cast<uint64_t>(L.tfirst[q])<<50) | (static_cast<uint64_t>(L.links[L.tlink[q]+1])<<36) | (static_cast<uint64_t>(0x3FFF-L.tslen[q])<<22) | (static_cast<uint64_t>(L.tlast[q])<<8) | q)
				: ~0ull;
		wv_sync();
		if ( nn >= 0x3FFF || ns > 255 ) { over(32); nstretch = 0; return; }
		wv_bitonic_sort(L.skey,p2);
		base = 0;
		for ( uint32_t c = 0; c < ns; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const keep = (q < ns) && (q == 0 || (L.skey[q]>>36) != (L.skey[q-1]>>36));
			uint32_t tot; uint32_t const pre = wv_scan_excl(keep,tot);
			if ( keep )
			{
				uint32_t const raw = L.skey[q]&0xFF; uint32_t const s = base+pre;
				// final arrays alias nothing of the raw arrays (separate LDS regions)
				L.sfirst[s] = L.tfirst[raw]; L.slast[s] = L.tlast[raw]; L.sslen[s] = L.tslen[raw]; L.slink[s] = L.tlink[raw];
			}
			base += tot;
		}
		nstretch = base;
		wv_sync();
		PROF(*this,20)
	}

	// ---- stretch feasibility, lanes = candidate positions (computeFeasibleStretchPositions) ----
	// weight(node,p) = (sum over instances of VS[p][pos]) / 2^32 is an exact multiple of 2^-32, and so is every
	// partial sum the reference forms (< 2^21), hence the FP64 sums of the reference are exact and equal the
	// integer sums below times 2^-32, in any order.  weight >= 1e-3  <=>  integer sum >= 4294968.
	DEV void computeStretchFeas()
	{
		nwF = 0; nwR = 0;
		uint32_t const nrows = C.nrows;
		for ( uint32_t s = 0; s < nstretch; ++s )
		{
			uint32_t const len = L.sslen[s];
			uint16_t const * Lk = L.links + L.slink[s];
			uint64_t mF = 0, mR = 0;
			uint32_t bF = nwF, bR = nwR;
			for ( uint32_t c = 0; c < nrows; c += WSZ )
			{
				uint32_t const Pp = c + lane;
				bool ok = Pp < nrows, okr = ok;
				uint64_t sum = 0, rsum = 0, wf = 0, wl = 0, rwf = 0, rwl = 0;
				for ( uint32_t j = 0; j < len; ++j )
				{
					// forward: node j at position Pp+j
					{
						uint32_t const z = Lk[j]; uint32_t const s0 = L.nps[z], f = L.nfreq[z];
						uint32_t const p = Pp+j;
						bool const in = p < nrows;
						uint64_t u = 0;
						for ( uint32_t q = 0; q < f; ++q ) u += in ? L.tab[static_cast<uint32_t>(L.ipos[s0+q])*nrows + p] : 0u;
						ok = ok && in && p >= L.pfrom[z] && p < L.pto[z] && u >= 4294968ull;
						sum += u; if ( j == 0 ) wf = u; wl = u;
					}
					// reverse: node len-1-j at reverse position Pp+j
					{
						uint32_t const z = Lk[len-1-j]; uint32_t const s0 = L.nps[z], f = L.nfreq[z];
						uint32_t const p = Pp+j;
						bool const in = p < nrows;
						uint64_t u = 0;
						for ( uint32_t q = 0; q < f; ++q ) u += in ? L.tab[static_cast<uint32_t>(L.irpos[s0+q])*nrows + p] : 0u;
						okr = okr && in && p >= L.cpfrom[z] && p < L.cpto[z] && u >= 4294968ull;
						rsum += u; if ( j == 0 ) rwf = u; rwl = u;
					}
				}
				double const sc = 2.3283064365386962890625e-10; // 2^-32
				uint32_t tot; uint32_t const pre = wv_scan_excl(ok ? 1 : 0,tot);
				if ( ok && bF+pre < C.sfcap ) { G.wF[3*(bF+pre)] = static_cast<double>(sum)*sc; G.wF[3*(bF+pre)+1] = static_cast<double>(wf)*sc; G.wF[3*(bF+pre)+2] = static_cast<double>(wl)*sc; }
				mF |= wv_or64(ok ? (1ull<<Pp) : 0ull);
				bF += tot;
				uint32_t const prer = wv_scan_excl(okr ? 1 : 0,tot);
				if ( okr && bR+prer < C.sfcap ) { G.wR[3*(bR+prer)] = static_cast<double>(rsum)*sc; G.wR[3*(bR+prer)+1] = static_cast<double>(rwf)*sc; G.wR[3*(bR+prer)+2] = static_cast<double>(rwl)*sc; }
				mR |= wv_or64(okr ? (1ull<<Pp) : 0ull);
				bR += tot;
			}
			if ( bF > C.sfcap || bR > C.sfcap || bF > 0xFFFF || bR > 0xFFFF ) { over(128); return; }
			if ( lane == 0 ) { L.maskF[s] = mF; L.maskR[s] = mR; L.woffF[s] = nwF; L.woffR[s] = nwR; }
			nwF = bF; nwR = bR;
		}
		wv_sync();
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
	DEV uint32_t stretchLowerBound(uint32_t const node) const
	{
		uint32_t lo = 0, hi = nstretch;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( L.sfirst[mid] < node ) lo = mid+1; else hi = mid; }
		return lo;
	}
	// getReverseStretchLinkWeight(A=i,B=b) >= 0.1 ?  (computeStretchLinks); links are evaluated on demand
	DEV bool linkOk(uint32_t const i, uint32_t const b) const
	{
		uint32_t const shift = L.sslen[b]-1;
		if ( shift >= 64 ) return false;
		uint64_t const mA = L.maskR[i], mB = L.maskR[b];
		uint64_t common = mA & (mB<<shift);
		double weight = 0.0;
		while ( common )
		{
			uint32_t const pa = __builtin_ctzll(common); common &= common-1;
			uint32_t const ia = L.woffR[i] + dacc_popc64(mA & ((1ull<<pa)-1));
			uint32_t const pb = pa-shift;
			uint32_t const ib = L.woffR[b] + dacc_popc64(mB & ((1ull<<pb)-1));
			double const lweight = G.wR[3*ib] + (G.wR[3*ia] - G.wR[3*ia+1]);
			weight = lweight > weight ? lweight : weight;
		}
		return weight >= 1e-1;
	}

	// ---- index heaps over pool weights (same sift algorithm as oracle/o_heap.hpp) ----
	template<bool MINHEAP>
	DEV static bool hless(double a, double b) { return MINHEAP ? (a < b) : (a > b); }
	template<bool MINHEAP>
	DEV void ipush(uint16_t * H, uint32_t & f, uint16_t const id, double const * W)
	{
		uint32_t i = f++; H[i] = id;
		while ( i )
		{
			uint32_t const p = (i-1)>>1;
			if ( hless<MINHEAP>(W[H[i]],W[H[p]]) ) { uint16_t const t = H[i]; H[i] = H[p]; H[p] = t; i = p; }
			else break;
		}
	}
	template<bool MINHEAP>
	DEV void ipop(uint16_t * H, uint32_t & f, double const * W)
	{
		H[0] = H[--f];
		uint32_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			uint32_t const m = hless<MINHEAP>(W[H[r-1]],W[H[r]]) ? (r-1) : r;
			if ( hless<MINHEAP>(W[H[i]],W[H[m]]) ) return;
			uint16_t const t = H[i]; H[i] = H[m]; H[m] = t; i = m;
		}
		uint32_t const l = 2*i+1;
		if ( l < f && !hless<MINHEAP>(W[H[i]],W[H[l]]) ) { uint16_t const t = H[i]; H[i] = H[l]; H[l] = t; }
	}

	DEV int32_t extendReversePath(uint32_t const parent, uint32_t const s)
	{
		if ( nrp >= C.pcapr ) { over(512); return -1; }
		uint32_t const id = nrp++;
		uint32_t const ppos = L.rp_pos[parent], plen = L.rp_len[parent];
		int32_t const sfo = csfFind(s,ppos);
		double weight = L.rp_weight[parent]; uint32_t baselen = L.rp_baselen[parent];
		if ( plen == 0 ) { baselen = L.sslen[s]+k-1; weight = sfo >= 0 ? G.wR[3*sfo] : 0.0; }
		else { baselen += L.sslen[s]-1; if ( sfo >= 0 ) weight += G.wR[3*sfo] - G.wR[3*sfo+1]; }
		L.rp_parent[id] = parent; L.rp_stretch[id] = s; L.rp_len[id] = plen+1; L.rp_pos[id] = ppos + L.sslen[s]-1;
		L.rp_weight[id] = weight; L.rp_baselen[id] = baselen;
		return id;
	}
	DEV bool checkReversePathFeasiblePosition(uint32_t const id) const
	{
		uint32_t const s = L.rp_stretch[id];
		uint32_t const checkpos = L.rp_pos[id] - (L.sslen[s]-1);
		int32_t const f = csfFind(s,checkpos);
		return f >= 0 && G.wR[3*f] >= 0.5;
	}
	DEV uint32_t rpFront(uint32_t const id, uint32_t const lastkmer) const { return L.rp_len[id] ? L.nv[L.sfirst[L.rp_stretch[id]]] : lastkmer; }

	// std::sort permutation on ARP (see dbg_window.hpp arpSort); comparator (front, baselen)
	uint32_t sortlastk;
	DEV bool arpLess(uint16_t const a, uint16_t const b) const
	{
		uint32_t const fa = rpFront(a,sortlastk), fb = rpFront(b,sortlastk);
		if ( fa != fb ) return fa < fb;
		return L.rp_baselen[a] < L.rp_baselen[b];
	}
	DEV void arpULI(uint16_t * last) { uint16_t const val = *last; uint16_t * next = last-1; while ( arpLess(val,*next) ) { *last = *next; last = next; --next; } *last = val; }
	DEV void arpIns(uint16_t * first, uint16_t * last)
	{
		if ( first == last ) return;
		for ( uint16_t * i = first+1; i != last; ++i )
		{
			if ( arpLess(*i,*first) ) { uint16_t const val = *i; for ( uint16_t * q = i; q != first; --q ) *q = *(q-1); *first = val; }
			else arpULI(i);
		}
	}
	DEV void arpSort(uint16_t * first, uint16_t * last)
	{
		if ( first == last ) return;
		int64_t const n = last-first;
		if ( n > 16 )
		{
			int depth = 0; { int64_t t = n; while ( t > 1 ) { t >>= 1; ++depth; } depth *= 2; }
			uint16_t * stF[40]; uint16_t * stL[40]; int stD[40]; int sp = 0;
			stF[0] = first; stL[0] = last; stD[0] = depth; sp = 1;
			while ( sp )
			{
				--sp;
				uint16_t * f = stF[sp]; uint16_t * l = stL[sp]; int d = stD[sp];
				while ( l-f > 16 )
				{
					if ( d == 0 ) { over(1024); return; }
					--d;
					uint16_t * mid = f + (l-f)/2; uint16_t * a = f+1; uint16_t * b = mid; uint16_t * c = l-1;
					if ( arpLess(*a,*b) )
					{
						if ( arpLess(*b,*c) ) { uint16_t t = *f; *f = *b; *b = t; }
						else if ( arpLess(*a,*c) ) { uint16_t t = *f; *f = *c; *c = t; }
						else { uint16_t t = *f; *f = *a; *a = t; }
					}
					else if ( arpLess(*a,*c) ) { uint16_t t = *f; *f = *a; *a = t; }
					else if ( arpLess(*b,*c) ) { uint16_t t = *f; *f = *c; *c = t; }
					else { uint16_t t = *f; *f = *b; *b = t; }
					uint16_t * lo = f+1; uint16_t * hi = l;
					while ( true )
					{
						while ( arpLess(*lo,*f) ) ++lo;
						--hi;
						while ( arpLess(*f,*hi) ) --hi;
						if ( !(lo < hi) ) break;
						uint16_t t = *lo; *lo = *hi; *hi = t;
						++lo;
					}
					if ( sp < 40 ) { stF[sp] = lo; stL[sp] = l; stD[sp] = d; ++sp; } else { over(1024); return; }
					l = lo;
				}
			}
			arpIns(first,first+16);
			for ( uint16_t * i = first+16; i != last; ++i ) arpULI(i);
		}
		else arpIns(first,last);
	}

	DEV void reverseEnumerate(uint32_t const lastkmer, int32_t const lastnode, int64_t const lmax)
	{
		nrp = 0; narp = 0;
		for ( uint32_t i = 0; i < C.blcap; ++i ) L.hbl_n[i] = 0;
		uint32_t nrpst = 0;
		if ( lastnode >= 0 )
		{
			uint32_t const id = nrp++;
			L.rp_parent[id] = 0xFFFF; L.rp_stretch[id] = 0xFFFF; L.rp_len[id] = 0; L.rp_pos[id] = 0; L.rp_weight[id] = 0.0; L.rp_baselen[id] = k;
			L.rpst_i[nrpst] = id; ++nrpst;
		}
		while ( nrpst )
		{
			uint32_t const rp = L.rpst_i[0];
			ipop<false>(L.rpst_i,nrpst,L.rp_weight);
			uint32_t const bl = L.rp_baselen[rp];
			if ( bl >= C.blcap ) { over(2048); return; }
			uint16_t * H = L.hbl + 12*bl; uint32_t hn = L.hbl_n[bl];
			if ( hn == 12 )
			{
				if ( L.rp_weight[rp] <= L.rp_weight[H[0]] ) continue;
				else ipop<true>(H,hn,L.rp_weight);
			}
			ipush<true>(H,hn,rp,L.rp_weight);
			L.hbl_n[bl] = hn;
			L.arp[narp++] = rp;
			if ( L.rp_len[rp] == 0 )
			{
				for ( uint32_t s = 0; s < nstretch; ++s )
					if ( L.slast[s] == lastnode )
					{
						int32_t const rpe = extendReversePath(rp,s);
						if ( rpe < 0 ) return;
						if ( checkReversePathFeasiblePosition(rpe) ) ipush<false>(L.rpst_i,nrpst,rpe,L.rp_weight);
						else --nrp; // an infeasible extension is never referenced again: recycle its slot
					}
			}
			else if ( static_cast<int64_t>(L.rp_baselen[rp]) < (lmax+1)/2 )
			{
				uint32_t const b = L.rp_stretch[rp];
				// candidates A with A.last == B.first, ascending A (the sorted (to,from) pairs of the reference)
				uint32_t const bf = L.sfirst[b];
				for ( uint32_t a = 0; a < nstretch; ++a )
					if ( L.slast[a] == bf && linkOk(a,b) )
					{
						int32_t const rpe = extendReversePath(rp,a);
						if ( rpe < 0 ) return;
						if ( checkReversePathFeasiblePosition(rpe) ) ipush<false>(L.rpst_i,nrpst,rpe,L.rp_weight);
						else --nrp;
					}
			}
		}
		sortlastk = lastkmer;
		arpSort(L.arp,L.arp+narp);
		for ( uint32_t i = 0; i < narp; ++i )
		{
			double const wi = L.rp_weight[L.arp[i]];
			uint32_t r = 0;
			for ( uint32_t j = 0; j < narp; ++j )
			{
				double const wj = L.rp_weight[L.arp[j]];
				if ( wj < wi || (wj == wi && j < i) ) ++r;
			}
			L.arw[i] = r; L.arwr[i] = narp-r-1;
		}
	}

	DEV int32_t extendPath(int32_t const parent, uint32_t const s)
	{
		if ( np >= C.pcapf ) { over(512); return -1; }
		uint32_t const id = np++;
		uint32_t const ppos = parent >= 0 ? L.p_pos[parent] : 0;
		uint32_t const plen = parent >= 0 ? L.p_len[parent] : 0;
		double weight = parent >= 0 ? L.p_weight[parent] : 0.0;
		uint32_t baselen = parent >= 0 ? L.p_baselen[parent] : 0;
		int32_t const sfo = sfFind(s,ppos);
		if ( plen == 0 ) { baselen = L.sslen[s]+k-1; weight = sfo >= 0 ? G.wF[3*sfo] : 0; }
		else { baselen += L.sslen[s]-1; if ( sfo >= 0 ) weight += G.wF[3*sfo] - G.wF[3*sfo+1]; }
		L.p_parent[id] = parent >= 0 ? parent : 0xFFFF; L.p_stretch[id] = s; L.p_len[id] = plen+1; L.p_pos[id] = ppos + (L.sslen[s]-1);
		L.p_weight[id] = weight; L.p_baselen[id] = baselen;
		return id;
	}
	uint32_t apqlo, apqhi;
	DEV bool apqPush(uint32_t const id)
	{
		uint32_t const bl = L.p_baselen[id];
		if ( bl >= C.blcap ) { over(2048); return false; }
		uint16_t * H = L.hbl + 12*bl; uint32_t hn = L.hbl_n[bl];
		if ( hn == 12 )
		{
			if ( L.p_weight[id] > L.p_weight[H[0]] ) { ipop<true>(H,hn,L.p_weight); ipush<true>(H,hn,id,L.p_weight); }
		}
		else ipush<true>(H,hn,id,L.p_weight);
		L.hbl_n[bl] = hn;
		if ( bl < apqlo ) apqlo = bl;
		if ( bl+1 > apqhi ) apqhi = bl+1;
		return true;
	}
	DEV double getPairScore(uint32_t const path, uint32_t const rp) const
	{
		uint32_t const s = L.p_stretch[path];
		uint32_t const spos = L.p_pos[path] - (L.sslen[s]-1);
		int32_t const sfo = sfFind(s,spos);
		if ( sfo >= 0 ) return L.p_weight[path] + L.rp_weight[rp] - G.wF[3*sfo+2];
		else return L.p_weight[path] + L.rp_weight[rp];
	}
	DEV uint32_t decodePathPair(uint32_t const path, uint32_t const rp, uint32_t o)
	{
		uint16_t chain[64]; uint32_t cl = 0;
		for ( uint32_t q = path; q != 0xFFFF; q = L.p_parent[q] ) { if ( cl >= 64 ) { over(4096); return ~0u; } chain[cl++] = L.p_stretch[q]; }
		uint32_t need = k;
		for ( uint32_t i = 0; i < cl; ++i ) need += L.sslen[chain[i]]-1;
		for ( uint32_t q = rp; L.rp_len[q]; q = L.rp_parent[q] ) need += L.sslen[L.rp_stretch[q]]-1;
		if ( o + need > C.conscap - MAXCONS ) { over(4096); return ~0u; }
		uint32_t const firstv = L.nv[L.sfirst[chain[cl-1]]];
		for ( uint32_t i = 0; i < k; ++i ) G.cons[o++] = (firstv >> (2*(k-1-i))) & 3;
		for ( uint32_t ii = 0; ii < cl; ++ii )
		{
			uint32_t const s = chain[cl-1-ii]; uint16_t const * Lk = L.links + L.slink[s];
			for ( uint32_t j = 1; j < L.sslen[s]; ++j ) G.cons[o++] = L.nv[Lk[j]] & 3;
		}
		for ( uint32_t q = rp; L.rp_len[q]; q = L.rp_parent[q] )
		{
			uint32_t const s = L.rp_stretch[q]; uint16_t const * Lk = L.links + L.slink[s];
			for ( uint32_t j = 1; j < L.sslen[s]; ++j ) G.cons[o++] = L.nv[Lk[j]] & 3;
		}
		return o;
	}

	DEV void forwardAndPairs(int32_t const firstnode, uint32_t const lastkmer, int64_t const lmin, int64_t const lmax, uint32_t const maxfullpath)
	{
		np = 0; nsiq = 0;
		for ( uint32_t i = 0; i < C.blcap; ++i ) L.hbl_n[i] = 0;
		apqlo = C.blcap; apqhi = 0;
		for ( uint32_t s = 0; s < nstretch; ++s )
			if ( L.sfirst[s] == firstnode )
			{
				int32_t const id = extendPath(-1,s);
				if ( id < 0 || !apqPush(id) ) return;
			}
		for ( uint32_t zz = apqlo; zz < apqhi; ++zz )
			while ( L.hbl_n[zz] )
			{
				uint16_t * H = L.hbl + 12*zz; uint32_t hn = L.hbl_n[zz];
				uint32_t const path = H[0];
				ipop<true>(H,hn,L.p_weight);
				L.hbl_n[zz] = hn;
				int64_t const candlen = static_cast<int64_t>(L.p_pos[path]) + k;
				uint32_t const laststretch = L.p_stretch[path];
				uint32_t const front = L.nv[L.slast[laststretch]];
				uint32_t lo = 0, hi = narp;
				while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( rpFront(L.arp[mid],lastkmer) < front ) lo = mid+1; else hi = mid; }
				uint32_t e = lo;
				while ( e < narp && rpFront(L.arp[e],lastkmer) == front ) ++e;
				int64_t bllo = lmin + static_cast<int64_t>(k) - candlen; if ( bllo < 0 ) bllo = 0;
				int64_t blhi = lmax + static_cast<int64_t>(k) - candlen; if ( blhi < 0 ) blhi = 0;
				uint32_t const bllo16 = static_cast<uint16_t>(bllo), blhi16 = static_cast<uint16_t>(blhi);
				uint32_t sub = lo;
				while ( sub < e && L.rp_baselen[L.arp[sub]] < bllo16 ) ++sub;
				uint32_t sup = sub;
				while ( sup < e && !(blhi16 < L.rp_baselen[L.arp[sup]]) ) ++sup;
				if ( sub != sup )
				{
					uint32_t mi = sub;
					for ( uint32_t i = sub+1; i < sup; ++i ) if ( L.arwr[i] < L.arwr[mi] ) mi = i;
					if ( nsiq >= C.siqcap ) { over(512); return; }
					HeapSI si; si.left = sub; si.right = sup; si.current = mi; si.path = path; si.w = getPairScore(path,L.arp[mi]);
					heap_push<HeapSI,CmpWGreater>(L.siq,nsiq,si);
				}
				uint32_t const pbl = L.p_baselen[path];
				if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
				{
					uint32_t const lastn = L.slast[laststretch];
					for ( uint32_t s = stretchLowerBound(lastn); s < nstretch && L.sfirst[s] == lastn; ++s )
					{
						int32_t const sfo = sfFind(s,L.p_pos[path]);
						double const eweight = sfo >= 0 ? G.wF[3*sfo] : 0.0;
						if ( eweight > 0.1 )
						{
							int32_t const ep = extendPath(path,s);
							if ( ep < 0 ) return;
							if ( L.p_weight[ep] > 0.1 && static_cast<int64_t>(L.p_pos[ep]) + k <= lmax )
							{
								if ( !apqPush(ep) ) return;
							}
							else --np; // never referenced again
						}
					}
				}
			}
		uint32_t prevo = 0, prevlen = ~0u;
		for ( uint32_t numfullpath = 0; nsiq && numfullpath < maxfullpath; ++numfullpath )
		{
			HeapSI const si = L.siq[0];
			heap_popvoid<HeapSI,CmpWGreater>(L.siq,nsiq);
			{
				uint32_t const v = L.arw[si.current];
				if ( v )
				{
					bool found = false; uint32_t bu = 0, bi = 0;
					for ( uint32_t i = si.left; i < si.right; ++i )
					{
						uint32_t const r = L.arw[i];
						if ( r <= v-1 && (!found || r > bu) ) { found = true; bu = r; bi = i; }
					}
					if ( found )
					{
						HeapSI sic = si; sic.current = bi; sic.w = getPairScore(si.path,L.arp[bi]);
						if ( nsiq >= C.siqcap ) { over(512); return; }
						heap_push<HeapSI,CmpWGreater>(L.siq,nsiq,sic);
					}
				}
			}
			double const weight = si.w;
			if ( ncdh == 16 )
			{
				if ( weight <= L.cdh[0].w ) continue;
				else heap_popvoid<HeapCC,CmpWLess>(L.cdh,ncdh);
			}
			uint32_t const consstart = conso;
			uint32_t const nc = decodePathPair(si.path,L.arp[si.current],conso);
			if ( nc == ~0u ) return;
			conso = nc;
			uint32_t const conslen = conso-consstart;
			if ( conslen == prevlen )
			{
				bool eq = true;
				for ( uint32_t i = 0; i < conslen; ++i ) if ( G.cons[prevo+i] != G.cons[consstart+i] ) { eq = false; break; }
				if ( eq ) continue;
			}
			prevo = consstart; prevlen = conslen;
			HeapCC cc; cc.w = weight; cc.o = consstart; cc.l = conslen;
			heap_push<HeapCC,CmpWLess>(L.cdh,ncdh,cc);
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

	DEV bool traverse(int64_t const lmin, int64_t const lmax)
	{
		if ( lane == 0 ) { conso = 0; ncdh = 0; nacc = 0; }
		uint32_t const firstthres = nmfirst ? ((static_cast<uint32_t>((~L.mfirst[0])>>32))*3)/4 : 0;
		uint32_t const lastthres = nmlast ? ((static_cast<uint32_t>((~L.mlast[0])>>32))*3)/4 : 0;
		for ( uint32_t fi = 0; fi < nmfirst; ++fi )
		{
			uint64_t const fkey = ~L.mfirst[fi];
			if ( static_cast<uint32_t>(fkey>>32) < firstthres ) break;
			for ( uint32_t li = 0; li < nmlast; ++li )
			{
				uint64_t const lkey = ~L.mlast[li];
				if ( static_cast<uint32_t>(lkey>>32) < lastthres ) break;
				uint32_t const firstk = static_cast<uint32_t>(fkey), lastk = static_cast<uint32_t>(lkey);
				int32_t const firstnode = findNode(firstk);
				int32_t const lastnode = findNode(lastk);
				PROF_T0
				computeStretches(firstnode,lastnode);
				PROF(*this,8)
				flags = wv_or(flags);
				if ( flags ) return false;
				computeStretchFeas();
				PROF(*this,9)
				flags = wv_or(flags);
				if ( flags ) return false;
				if ( lane == 0 )
				{
					reverseEnumerate(lastk,lastnode,lmax);
					PROF(*this,11)
					if ( ! flags ) forwardAndPairs(firstnode,lastk,lmin,lmax,16);
					PROF(*this,12)
				}
				wv_sync();
				flags = wv_bcast(flags,0);
				if ( flags ) return false;
			}
		}
		PROF_T0
		if ( lane == 0 )
		{
			uint32_t nch = 0;
			while ( ncdh ) { HeapCC const c = L.cdh[0]; heap_popvoid<HeapCC,CmpWLess>(L.cdh,ncdh); heap_push<HeapCC,CmpWGreater>(L.ch,nch,c); }
			while ( nch ) { L.acc[nacc++] = L.ch[0]; heap_popvoid<HeapCC,CmpWGreater>(L.ch,nch); }
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
				uint64_t s = 0;
				for ( uint32_t j = 0; j < mao; ++j ) s += L.canderr[c*mao+j];
				L.accerr[c] = static_cast<double>(s);
			}
			for ( uint32_t i = 1; i < nc; ++i )
			{
				HeapCC const v = L.acc[i]; double const e = L.accerr[i];
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
		return maxvprodindex; // -1: the density fallback is left to the generic engine
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
	E.C = FB.F; E.T = B.T; E.P = B.P; E.vst = FB.dpsq_vst;
	E.lane = wv_lane(); E.flags = 0; E.prof = B.prof;
	fast_lds_carve(E.L,lds,FB.F);
	fast_global_carve(E.G,garena,FB.F);
	FastLds & L = E.L;
	int const lane = E.lane;
	PROF_T0

	uint32_t lo = 0, hi = B.npiles;
	while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( B.piles[mid].winbase <= widx ) lo = mid; else hi = mid; }
	DevPile const pile = B.piles[lo];
	uint32_t const y = static_cast<uint32_t>(widx - pile.winbase);
	uint32_t astart, aend;
	windowInterval(pile.l,B.P.a,B.P.w,y,astart,aend);

	WindowOut out; out.status = WS_INSUFFICIENT; out.mao = 0; out.elength = 0; out.k = 0; out.filterfreq = -1; out.conslen = 0; out.minrate = 0; out.flags = 0;
	uint8_t * rec = B.wrec + widx*WREC;
	if ( lane == 0 ) rec[0] = 0;
	if ( B.P.w > 63 ) return false;

	// active set -> keys in the (still unused) instance buffer
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
	if ( nact > FB.F.precap ) return false;
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
	if ( mao > FB.F.maxs ) return false;
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
	if ( toolong ) return false;
	PROF(E,0)

	int32_t elength = 0;
	if ( mao )
	{
		E.buildPeq();
		int32_t const idx = E.estimateLength();
		if ( idx < 0 ) return false;
		elength = idx+1;
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
				if ( ff == 0 ) return false; // gap filling: generic engine
				PROF_T0
				E.buildInstances();
				PROF(E,2)
				E.buildNodes(ff > 1 ? ff : 1);
				PROF(E,3)
				E.buildSuccessors(mao);
				PROF(E,4)
				E.flags = wv_or(E.flags);
				if ( E.flags ) return false;
				uint32_t mintry = 0; bool lconsok = false;
				while ( true )
				{
					bool const consok = E.traverse(static_cast<int64_t>(elength)-4,static_cast<int64_t>(elength)+4);
					E.flags = wv_or(E.flags);
					if ( E.flags ) return false;
					if ( consok )
					{
						uint64_t const err = static_cast<uint64_t>(L.accerr[0]);
						if ( err < minrate )
						{
							lconsok = true; minrate = err; haveMin = true;
							bestlen = L.acc[0].l;
							if ( bestlen > MAXCONS ) return false;
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
}

}
#endif
