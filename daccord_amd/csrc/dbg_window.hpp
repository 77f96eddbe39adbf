/*
 * Per-window local de Bruijn graph consensus for one wavefront (gfx950, 64 lanes).
 *
 * This is the MI355X-native re-design of the reference's DebruijnGraph<k>
 * (src/DebruijnGraph.hpp:671-5483) and of the window body of HandleContext::operator()
 * (src/HandleContext.hpp:2051-2344).  It is not a translation: the reference's per-thread
 * AutoArrays, 4^k node cache, PF/RPF/SP lists, bucket/bit-vector scans, RMQ and wavelet
 * tree are replaced by
 *   - one sorted array of packed k-mer instances (bitonic sort across the wavefront),
 *     nodes = runs of that array, k-mer -> node by binary search over the sorted node keys,
 *   - feasibility as integer fixed-point dot products evaluated lane-per-node,
 *   - stretches built lane-per-(node,successor), keyed and sorted by node rank,
 *   - parent-pointer path pools instead of copied stretch lists,
 *   - bit-parallel (Myers) edit distances, lane-per-(candidate,string).
 * What IS kept exactly is everything that decides the result: orderings, thresholds,
 * FP64 summation order, bounded-heap mechanics (same sift algorithm as the oracle's
 * o_heap.hpp) and libstdc++'s std::sort permutation where the reference sorts with ties
 * (ARP, DebruijnGraph.hpp:3742).  Compile with -ffp-contract=off.
 */
#ifndef DACC_DBG_WINDOW_HPP
#define DACC_DBG_WINDOW_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "arena.hpp"

namespace dacc {

#define DACC_DBL_MIN 2.2250738585072014e-308

// optional per-phase cycle accounting (profiling builds only: -DDACC_PROFILE)
#define DACC_PROFW 128      // counters per workgroup row: 32 phases + 48 fine sites (cycles, visits)
#if defined(DACC_PROFILE) && !defined(DACC_EMUL)
  #define PROF_T0 uint64_t _pt = clock64();
  #define PROF(E_,id) { uint64_t const _n = clock64(); if ( (E_).lane == 0 && (E_).prof ) atomicAdd(reinterpret_cast<unsigned long long *>((E_).prof+(id)),static_cast<unsigned long long>(_n-_pt)); _pt = _n; }
#else
  #define PROF_T0
  #define PROF(E_,id)
#endif

struct WindowOut
{
	int32_t status, mao, elength, k, filterfreq, conslen;
	uint64_t minrate;
	uint32_t flags;
};

// ---------- small generic heaps (same algorithm as oracle/o_heap.hpp) ----------
struct CmpWLess { DEV bool operator()(double a, double b) const { return a < b; } };
struct CmpWGreater { DEV bool operator()(double a, double b) const { return a > b; } };

template<typename T, typename Cmp>
DEV void heap_push(T * H, uint32_t & f, T const & e)
{
	Cmp cmp;
	uint32_t i = f++;
	H[i] = e;
	while ( i )
	{
		uint32_t const p = (i-1)>>1;
		if ( cmp(H[i].w,H[p].w) ) { T const t = H[i]; H[i] = H[p]; H[p] = t; i = p; }
		else break;
	}
}
template<typename T, typename Cmp>
DEV void heap_popvoid(T * H, uint32_t & f)
{
	Cmp cmp;
	H[0] = H[--f];
	uint32_t i = 0, r;
	while ( (r = 2*i+2) < f )
	{
		uint32_t const m = cmp(H[r-1].w,H[r].w) ? (r-1) : r;
		if ( cmp(H[i].w,H[m].w) ) return;
		T const t = H[i]; H[i] = H[m]; H[m] = t;
		i = m;
	}
	uint32_t const l = 2*i+1;
	if ( l < f && !cmp(H[i].w,H[l].w) ) { T const t = H[i]; H[i] = H[l]; H[l] = t; }
}

// ---------- the per-wavefront engine ----------
struct WindowEngine
{
	Arena A;
	ArenaCaps C;
	DevTables T;
	DevParams P;
	int lane;
	uint64_t * prof;
	uint32_t flags;      // overflow flags (wave-uniform after wv_any)

	// window strings
	uint32_t mao;
	// graph state (wave-uniform scalars)
	uint32_t k; uint64_t kmask;
	uint32_t npre, nlast, nn;     // instances, last-kmers, nodes
	uint32_t nmfirst, nmlast;
	uint32_t nstretch, nlinks, nsf, ncsf, nrl;
	uint32_t nrp, narp, np, nsiq, ncdh, nacc, conso;
	uint32_t maxsiq, maxlinks;

	DEV void setOverflow(uint32_t bit) { flags |= bit; }
#if defined(DACC_STATS)
	uint32_t st[16];
	DEV void stat(int i, uint32_t v) { if ( v > st[i] ) st[i] = v; }
#else
	DEV void stat(int, uint32_t) {}
#endif

	// ---- k-mer -> node id by binary search over the ascending node keys ----
	DEV int32_t findNode(uint32_t const v) const
	{
		int32_t lo = 0, hi = static_cast<int32_t>(nn)-1;
		while ( lo <= hi )
		{
			int32_t const mid = (lo+hi)>>1;
			uint32_t const x = A.nv[mid];
			if ( x == v ) return mid;
			if ( x < v ) lo = mid+1; else hi = mid-1;
		}
		return -1;
	}

	// ================= G1: k-mer instances (setupPreNodes, DebruijnGraph.hpp:2018-2304) =================
	DEV void buildInstances()
	{
		// per string k-mer counts -> offsets (chunked exclusive scan)
		uint32_t base = 0;
		for ( uint32_t c = 0; c < mao; c += WSZ )
		{
			uint32_t const j = c + lane;
			uint32_t const len = (j < mao) ? A.slen[j] : 0;
			uint32_t const numk = (j < mao && len >= k) ? (len-k+1) : 0;
			uint32_t tot; uint32_t const pre = wv_scan_excl(numk,tot);
			if ( j < mao ) A.koff[j] = base + pre;
			base += tot;
		}
		npre = base;
		if ( lane == 0 ) A.koff[mao] = base;
		wv_sync();
		if ( npre > C.precap ) { setOverflow(1); npre = 0; return; }
		// instances: kmer<<32 | pos<<16 | seq  (DebruijnGraph.hpp:1209-1217)
		for ( uint32_t j = 0; j < mao; ++j )
		{
			uint32_t const len = A.slen[j];
			if ( len < k ) continue;
			uint32_t const numk = len-k+1;
			uint8_t const * s = A.str + static_cast<uint64_t>(j)*C.lstr;
			uint32_t const o = A.koff[j];
			for ( uint32_t i = lane; i < numk; i += WSZ )
			{
				uint64_t v = 0;
				for ( uint32_t q = 0; q < k; ++q ) v = (v<<2) | s[i+q];
				A.pre[o+i] = (v<<32) | (static_cast<uint64_t>(i)<<16) | j;
			}
		}
		// last k-mer of every string (DebruijnGraph.hpp:2113, sorted :2302)
		base = 0;
		for ( uint32_t c = 0; c < mao; c += WSZ )
		{
			uint32_t const j = c + lane;
			uint32_t const len = (j < mao) ? A.slen[j] : 0;
			uint32_t const has = (j < mao && len >= k) ? 1 : 0;
			uint32_t tot; uint32_t const pre = wv_scan_excl(has,tot);
			if ( has )
			{
				uint8_t const * s = A.str + static_cast<uint64_t>(j)*C.lstr;
				uint64_t v = 0;
				for ( uint32_t q = 0; q < k; ++q ) v = (v<<2) | s[len-k+q];
				A.lastk[base+pre] = (v<<32) | (static_cast<uint64_t>(len-k)<<16) | j;
			}
			base += tot;
		}
		nlast = base;
		uint32_t const lp2 = next_pow2(nlast < 2 ? 2 : nlast);
		for ( uint32_t i = nlast + lane; i < lp2; i += WSZ ) A.lastk[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(A.lastk,lp2);
		sortInstances();
	}

	DEV void sortInstances()
	{
		uint32_t const p2 = next_pow2(npre < 2 ? 2 : npre);
		for ( uint32_t i = npre + lane; i < p2; i += WSZ ) A.pre[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(A.pre,p2);
	}

	// ================= G2+G3: nodes (setupNodes :1918-2014, filterFreq :1181-1197) =================
	// nodes = runs of equal k-mer in the sorted instance array with frequency >= f, ascending by k-mer
	DEV void buildNodes(uint32_t const f)
	{
		// pass 1: run heads -> unfiltered run starts
		uint32_t base = 0;
		for ( uint32_t c = 0; c < npre; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t const head = (i < npre) && (i == 0 || (A.pre[i]>>32) != (A.pre[i-1]>>32));
			uint32_t tot; uint32_t const pre = wv_scan_excl(head,tot);
			if ( head ) A.nstart0[base+pre] = i;
			base += tot;
		}
		uint32_t const nrun = base;
		if ( lane == 0 ) A.nstart0[nrun] = npre;
		wv_sync();
		// pass 2: keep runs with freq >= f
		base = 0;
		for ( uint32_t c = 0; c < nrun; c += WSZ )
		{
			uint32_t const u = c + lane;
			uint32_t s = 0, e = 0;
			if ( u < nrun ) { s = A.nstart0[u]; e = A.nstart0[u+1]; }
			uint32_t const keep = (u < nrun) && ((e-s) >= f);
			uint32_t tot; uint32_t const pre = wv_scan_excl(keep,tot);
			if ( keep )
			{
				uint32_t const z = base+pre;
				if ( z < C.nodecap )
				{
					uint32_t const freq = e-s;
					A.nv[z] = static_cast<uint32_t>(A.pre[s]>>32);
					A.nps[z] = s;
					A.nfreq[z] = freq;
					// positions are ascending inside a run (sorted by kmer, pos, seq)
					A.plow[z] = (A.pre[s]>>16)&0xFFFF;
					A.phigh[z] = (A.pre[e-1]>>16)&0xFFFF;
					uint32_t c0 = 0, rlo = 0xFFFF, rhi = 0;
					for ( uint32_t q = s; q < e; ++q )
					{
						uint32_t const pos = (A.pre[q]>>16)&0xFFFF, seq = A.pre[q]&0xFFFF;
						if ( pos == 0 ) ++c0;
						uint32_t const rpos = A.slen[seq]-pos-k;   // RSP, DebruijnGraph.hpp:1955
						rlo = rpos < rlo ? rpos : rlo; rhi = rpos > rhi ? rpos : rhi;
					}
					A.cnt0[z] = c0; A.cplow[z] = rlo; A.cphigh[z] = rhi;
				}
			}
			base += tot;
		}
		nn = base;
		if ( nn > C.nodecap ) { setOverflow(2); nn = 0; }
		wv_sync();
	}

	// ================= G4: successors + activation (setupAddHeap/setNodesActive :1770-1859) =================
	DEV void buildSuccessors(uint32_t const no)
	{
		uint32_t const lim = T.klim[(k-P.klow)*T.kln + (no < static_cast<uint32_t>(T.kln) ? no : T.kln-1)];
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const v = A.nv[z];
			uint32_t const masked = static_cast<uint32_t>((static_cast<uint64_t>(v)<<2) & kmask);
			uint32_t L[4]; int32_t I[4]; uint32_t n = 0;
			for ( uint32_t s = 0; s < 4; ++s )
			{
				int32_t const id = findNode(masked|s);
				if ( id >= 0 ) { L[n] = (static_cast<uint32_t>(A.nfreq[id])<<8)|s; I[n] = id; ++n; }
			}
			// sort descending by (freq<<8|sym) (Links::sort, Links.hpp:47-53); keys are distinct
			for ( uint32_t a = 1; a < n; ++a )
			{
				uint32_t const kv = L[a]; int32_t const ki = I[a]; int32_t b = a;
				while ( b > 0 && L[b-1] < kv ) { L[b] = L[b-1]; I[b] = I[b-1]; --b; }
				L[b] = kv; I[b] = ki;
			}
			uint32_t act = 0;
			if ( n )
			{
				act = 1;
				while ( act < n && ( ((L[act]>>8) >= (L[0]>>8)/2) || (P.checklim && ((L[act]>>8) >= lim)) ) ) ++act;
			}
			A.nsucc[z] = n; A.nsuccact[z] = act;
			for ( uint32_t s = 0; s < 4; ++s ) { A.succ[4*z+s] = s < n ? L[s] : 0; A.succid[4*z+s] = s < n ? I[s] : -1; }
		}
		wv_sync();
	}

	// addNextFromHeap (:1861-1897): the edge activation heap orders (freq desc, node, edge); because each
	// node's successors are frequency sorted, its heap entries are exactly the edges i >= numsuccactive,
	// so "pop everything with the top frequency" is a max-reduce plus a per-node advance
	DEV bool addNextFromHeap()
	{
		uint32_t best = 0;
		for ( uint32_t z = lane; z < nn; z += WSZ )
			if ( A.nsuccact[z] < A.nsucc[z] )
			{
				uint32_t const fq = A.succ[4*z+A.nsuccact[z]]>>8;
				best = fq > best ? fq : best;
			}
		best = wv_max(best);
		if ( ! best ) return false;
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t a = A.nsuccact[z]; uint32_t const n = A.nsucc[z];
			while ( a < n && (A.succ[4*z+a]>>8) == best ) ++a;
			A.nsuccact[z] = a;
		}
		wv_sync();
		return true;
	}

	// ================= G6: feasible k-mer positions (:3117-3174, :3826-3904) =================
	DEV void computeFeasible()
	{
		// reserve pto-pfrom (resp. cpto-cpfrom) slots per node
		uint32_t base = 0, cbase = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t cap = 0, ccap = 0;
			if ( z < nn )
			{
				uint32_t const pf = A.plow[z] < static_cast<uint32_t>(T.nsup) ? T.suplo[A.plow[z]] : T.nrows;
				uint32_t const pt = A.phigh[z] < static_cast<uint32_t>(T.nsup) ? T.suphi[A.phigh[z]] : T.nrows;
				cap = pt > pf ? pt-pf : 0;
				uint32_t const cpf = A.cplow[z] < static_cast<uint32_t>(T.nsup) ? T.suplo[A.cplow[z]] : T.nrows;
				uint32_t const cpt = A.cphigh[z] < static_cast<uint32_t>(T.nsup) ? T.suphi[A.cphigh[z]] : T.nrows;
				ccap = cpt > cpf ? cpt-cpf : 0;
			}
			uint32_t tot; uint32_t pre = wv_scan_excl(cap,tot);
			if ( z < nn ) A.feasoff[z] = base+pre;
			base += tot;
			pre = wv_scan_excl(ccap,tot);
			if ( z < nn ) A.cfeasoff[z] = cbase+pre;
			cbase += tot;
		}
		if ( base > C.fcap || cbase > C.fcap ) { setOverflow(4); for ( uint32_t z = lane; z < nn; z += WSZ ) { A.nfeas[z] = 0; A.ncfeas[z] = 0; } wv_sync(); return; }
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const s = A.nps[z], e = s + A.nfreq[z];
			{
				uint32_t const pf = A.plow[z] < static_cast<uint32_t>(T.nsup) ? T.suplo[A.plow[z]] : T.nrows;
				uint32_t const pt = A.phigh[z] < static_cast<uint32_t>(T.nsup) ? T.suphi[A.phigh[z]] : T.nrows;
				uint32_t o = A.feasoff[z], cnt = 0;
				for ( uint32_t p = pf; p < pt; ++p )
				{
					uint32_t const fs = T.dpsq_first[p], sz = T.dpsq_size[p];
					uint64_t const * VS = T.dpsq_vs + static_cast<uint64_t>(p)*T.nsup;
					uint64_t uprr = 0;
					for ( uint32_t q = s; q < e; ++q )
					{
						uint32_t const pos = (A.pre[q]>>16)&0xFFFF;
						if ( pos >= fs && pos < fs+sz ) uprr += VS[pos];
					}
					double const wgt = static_cast<double>(uprr) / 4294967296.0;
					if ( wgt >= 1e-3 ) { A.fp_p[o+cnt] = p; A.fp_w[o+cnt] = wgt; ++cnt; }
				}
				A.nfeas[z] = cnt;
			}
			{
				uint32_t const pf = A.cplow[z] < static_cast<uint32_t>(T.nsup) ? T.suplo[A.cplow[z]] : T.nrows;
				uint32_t const pt = A.cphigh[z] < static_cast<uint32_t>(T.nsup) ? T.suphi[A.cphigh[z]] : T.nrows;
				uint32_t o = A.cfeasoff[z], cnt = 0;
				for ( uint32_t p = pf; p < pt; ++p )
				{
					uint32_t const fs = T.dpsq_first[p], sz = T.dpsq_size[p];
					uint64_t const * VS = T.dpsq_vs + static_cast<uint64_t>(p)*T.nsup;
					uint64_t uprr = 0;
					for ( uint32_t q = s; q < e; ++q )
					{
						uint32_t const pos = (A.pre[q]>>16)&0xFFFF, seq = A.pre[q]&0xFFFF;
						uint32_t const rpos = A.slen[seq]-pos-k;
						if ( rpos >= fs && rpos < fs+sz ) uprr += VS[rpos];
					}
					double const wgt = static_cast<double>(uprr) / 4294967296.0;
					if ( wgt >= 1e-3 ) { A.cfp_p[o+cnt] = p; A.cfp_w[o+cnt] = wgt; ++cnt; }
				}
				A.ncfeas[z] = cnt;
			}
		}
		wv_sync();
	}

	DEV bool feasLookup(int32_t const z, uint32_t const p, double & w) const
	{
		uint32_t const o = A.feasoff[z], n = A.nfeas[z];
		for ( uint32_t i = 0; i < n; ++i ) { uint32_t const q = A.fp_p[o+i]; if ( q == p ) { w = A.fp_w[o+i]; return true; } if ( q > p ) break; }
		return false;
	}
	DEV bool cfeasLookup(int32_t const z, uint32_t const p, double & w) const
	{
		uint32_t const o = A.cfeasoff[z], n = A.ncfeas[z];
		for ( uint32_t i = 0; i < n; ++i ) { uint32_t const q = A.cfp_p[o+i]; if ( q == p ) { w = A.cfp_w[o+i]; return true; } if ( q > p ) break; }
		return false;
	}

	// ================= T1: first / last k-mer candidates (:1280-1304, :1360-1391) =================
	DEV void buildFirstLast()
	{
		uint32_t const kp2 = next_pow2(C.maxs < 2 ? 2 : C.maxs);
		// maxFirst: (count at read position 0, kmer), descending
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t const has = (z < nn) && A.cnt0[z];
			uint32_t tot; uint32_t const pre = wv_scan_excl(has,tot);
			if ( has && base+pre < kp2 ) A.mfirst[base+pre] = ~((static_cast<uint64_t>(A.cnt0[z])<<32) | A.nv[z]);
			base += tot;
		}
		nmfirst = base;
		if ( nmfirst > kp2 ) { setOverflow(8); nmfirst = 0; }
		uint32_t p2 = next_pow2(nmfirst < 2 ? 2 : nmfirst);
		for ( uint32_t i = nmfirst + lane; i < p2; i += WSZ ) A.mfirst[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(A.mfirst,p2);   // ascending of the complement = descending (count,kmer)
		// maxLast: runs of the sorted last-kmer array
		base = 0;
		for ( uint32_t c = 0; c < nlast; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t const head = (i < nlast) && (i == 0 || (A.lastk[i]>>32) != (A.lastk[i-1]>>32));
			uint32_t tot; uint32_t const pre = wv_scan_excl(head,tot);
			if ( head )
			{
				uint32_t e = i+1;
				while ( e < nlast && (A.lastk[e]>>32) == (A.lastk[i]>>32) ) ++e;
				A.mlast[base+pre] = ~((static_cast<uint64_t>(e-i)<<32) | (A.lastk[i]>>32));
			}
			base += tot;
		}
		nmlast = base;
		p2 = next_pow2(nmlast < 2 ? 2 : nmlast);
		for ( uint32_t i = nmlast + lane; i < p2; i += WSZ ) A.mlast[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(A.mlast,p2);
	}

	// ================= T2/T3: stretches (:2844-2986, :2772-2841, :3087-3114) =================
	DEV void computePredCounts()
	{
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const v = A.nv[z];
			uint32_t const masked = v>>2;
			uint32_t const shift = 2*(k-1);
			uint32_t cnt = 0;
			for ( uint32_t s = 0; s < 4; ++s )
			{
				int32_t const u = findNode(masked | (s<<shift));
				if ( u >= 0 )
				{
					uint32_t const na = A.nsuccact[u];
					for ( uint32_t i = 0; i < na; ++i )
						if ( A.succid[4*u+i] == static_cast<int32_t>(z) ) { ++cnt; break; }
				}
			}
			A.npred[z] = cnt;
		}
		wv_sync();
	}

	// walk the unique-successor / unique-predecessor chain starting with edge (z -> succ i);
	// write node ids to out (if non null); returns length.  A revisit can only hit the start node
	// (every extended interior node has exactly one active predecessor), see DESIGN.md.
	DEV uint32_t walkStretch(uint32_t const z, uint32_t const i, int32_t * out, int32_t & lastnode)
	{
		int32_t cur = A.succid[4*z+i];
		uint32_t len = 2;
		if ( out ) { out[0] = z; out[1] = cur; }
		bool loop = (cur == static_cast<int32_t>(z));
		while ( !loop && A.nsuccact[cur] == 1 && A.npred[cur] == 1 )
		{
			cur = A.succid[4*cur];
			if ( out ) out[len] = cur;
			++len;
			if ( cur == static_cast<int32_t>(z) ) loop = true;
			if ( len > nn+1 ) { setOverflow(16); break; }
		}
		lastnode = cur;
		return len;
	}

	DEV void computeStretches(int32_t const firstnode, int32_t const lastnode_split)
	{
		computePredCounts();
		// start edges: nodes with >= 1 active successor and (active preds != 1 or active succs > 1)
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t cnt = 0;
			if ( z < nn )
			{
				uint32_t const ns = A.nsuccact[z];
				if ( ns && (A.npred[z] != 1 || ns > 1) ) cnt = ns;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( z < nn ) A.scnt[z] = base+pre;
			base += tot;
		}
		uint32_t const nstart = base;
		if ( lane == 0 ) A.scnt[nn] = nstart;
		wv_sync();
		// every stretch can be split at most twice (at `first` and at `last`) -> up to 3 pieces
		if ( 3*nstart > C.strcap ) { setOverflow(32); nstretch = 0; return; }
		// pass 1: lengths (tslen used as scratch for raw lengths)
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			if ( z < nn )
			{
				uint32_t const b = A.scnt[z], e = A.scnt[z+1];
				for ( uint32_t q = b; q < e; ++q ) { int32_t ln; A.tslen[q] = walkStretch(z,q-b,0,ln); }
			}
		}
		wv_sync();
		// offsets into links (chunked scan)
		base = 0;
		for ( uint32_t c = 0; c < nstart; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const len = q < nstart ? A.tslen[q] : 0;
			uint32_t tot; uint32_t const pre = wv_scan_excl(len,tot);
			if ( q < nstart ) A.tlink[q] = base+pre;
			base += tot;
		}
		if ( 3*base+16 > C.linkcap ) { setOverflow(64); nstretch = 0; return; }
		maxlinks = base;
		uint32_t rawlinks = base;
		wv_sync();
		// pass 2: write node id chains
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			if ( z < nn )
			{
				uint32_t const b = A.scnt[z], e = A.scnt[z+1];
				for ( uint32_t q = b; q < e; ++q )
				{
					int32_t ln; walkStretch(z,q-b,A.links + A.tlink[q],ln);
					A.tfirst[q] = z; A.text[q] = A.links[A.tlink[q]+1]; A.tlast[q] = ln;
				}
			}
		}
		wv_sync();
		// splitStretches(first) then splitStretches(last): a stretch containing v strictly inside is
		// replaced by [..v] and [v..] (first occurrence).  The pieces share the link array (overlapping by one).
		uint32_t ns = nstart;
		for ( int round = 0; round < 2; ++round )
		{
			int32_t const v = round ? lastnode_split : firstnode;
			if ( v < 0 ) continue;
			uint32_t addbase = ns;
			for ( uint32_t c = 0; c < ns; c += WSZ )
			{
				uint32_t const q = c + lane;
				int32_t split = -1;
				if ( q < ns )
				{
					uint32_t const len = A.tslen[q]; int32_t const * L = A.links + A.tlink[q];
					for ( uint32_t i = 1; i+1 < len; ++i ) if ( L[i] == v ) { split = i; break; }
				}
				uint32_t tot; uint32_t const pre = wv_scan_excl(split >= 0 ? 1 : 0,tot);
				if ( split >= 0 )
				{
					uint32_t const len = A.tslen[q]; uint32_t const lo = A.tlink[q];
					uint32_t const nq = addbase+pre;
					// second piece [split,len)
					A.tfirst[nq] = v; A.text[nq] = A.links[lo+split+1]; A.tlast[nq] = A.tlast[q]; A.tslen[nq] = len-split; A.tlink[nq] = lo+split;
					// first piece [0,split] in place
					A.tlast[q] = v; A.tslen[q] = split+1;
				}
				addbase += tot;
			}
			ns = addbase;
			wv_sync();
		}
		(void)rawlinks;
		// sort by (first, ext, len desc, last) == Stretch::operator< (node ids are ascending in k-mer value);
		// identical stretches are interchangeable, so any tie order is equivalent to the reference's
		uint32_t const p2 = next_pow2(ns < 2 ? 2 : ns);
		if ( p2 > C.strcap || nn > 0xFFFE ) { setOverflow(32); nstretch = 0; return; }
		for ( uint32_t q = lane; q < p2; q += WSZ )
		{
			if ( q < ns )
				A.skey[q] = (static_cast<uint64_t>(A.tfirst[q])<<48) | (static_cast<uint64_t>(A.text[q])<<32) | (static_cast<uint64_t>(0xFFFF-A.tslen[q])<<16) | static_cast<uint64_t>(A.tlast[q]);
			A.sidx[q] = q < ns ? q : 0xFFFFFFFFu;
		}
		wv_sync();
		wv_bitonic_sort_idx(A.sidx,A.skey,p2);
		// unique + keep the first (longest) per (first,ext) (stretchesUnique :3087-3114)
		base = 0;
		for ( uint32_t c = 0; c < ns; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const keep = (q < ns) && (q == 0 || (A.skey[A.sidx[q]]>>32) != (A.skey[A.sidx[q-1]]>>32));
			uint32_t tot; uint32_t const pre = wv_scan_excl(keep,tot);
			if ( keep )
			{
				uint32_t const raw = A.sidx[q];
				uint32_t const s = base+pre;
				A.sfirst[s] = A.tfirst[raw]; A.sext[s] = A.text[raw]; A.sslen[s] = A.tslen[raw]; A.slast[s] = A.tlast[raw]; A.slink[s] = A.tlink[raw];
			}
			base += tot;
		}
		nstretch = base;
		wv_sync();
	}

	// ================= T4: feasible stretch positions (:3176-3330) =================
	DEV void computeStretchFeas()
	{
		// capacity: forward <= nfeas(first node), reverse <= ncfeas(last node)
		uint32_t base = 0, cbase = 0;
		for ( uint32_t c = 0; c < nstretch; c += WSZ )
		{
			uint32_t const s = c + lane;
			uint32_t cap = 0, ccap = 0;
			if ( s < nstretch ) { cap = A.nfeas[A.sfirst[s]]; ccap = A.ncfeas[A.slast[s]]; }
			uint32_t tot; uint32_t pre = wv_scan_excl(cap,tot);
			if ( s < nstretch ) A.sfo[s] = base+pre;
			base += tot;
			pre = wv_scan_excl(ccap,tot);
			if ( s < nstretch ) A.scfo[s] = cbase+pre;
			cbase += tot;
		}
		nsf = base; ncsf = cbase;
		if ( base > C.sfcap || cbase > C.sfcap ) { setOverflow(128); for ( uint32_t s = lane; s < nstretch; s += WSZ ) { A.sfl[s] = 0; A.scfl[s] = 0; } wv_sync(); return; }
		for ( uint32_t s = lane; s < nstretch; s += WSZ )
		{
			uint32_t const len = A.sslen[s];
			int32_t const * L = A.links + A.slink[s];
			{
				// forward: all k-mers j feasible at start+j; weight summed in j order (:3232-3233)
				int32_t const z0 = L[0];
				uint32_t const o0 = A.feasoff[z0], n0 = A.nfeas[z0];
				uint32_t o = A.sfo[s], cnt = 0;
				for ( uint32_t i = 0; i < n0; ++i )
				{
					uint32_t const p0 = A.fp_p[o0+i];
					double const w0 = A.fp_w[o0+i];
					double weight = 0.0; weight += w0;
					double wl = w0; bool ok = true;
					for ( uint32_t j = 1; j < len; ++j )
					{
						double wj;
						if ( !feasLookup(L[j],p0+j,wj) ) { ok = false; break; }
						weight += wj; wl = wj;
					}
					if ( ok ) { A.sf_p[o+cnt] = p0; A.sf_w[o+cnt] = weight; A.sf_wf[o+cnt] = w0; A.sf_wl[o+cnt] = wl; ++cnt; }
				}
				A.sfl[s] = cnt;
			}
			{
				// reverse: positions counted from the read end; enumerated by the LAST k-mer's reverse position,
				// weights summed from the last k-mer back to the first (push order of :3262-3296)
				int32_t const zl = L[len-1];
				uint32_t const o0 = A.cfeasoff[zl], n0 = A.ncfeas[zl];
				uint32_t o = A.scfo[s], cnt = 0;
				for ( uint32_t i = 0; i < n0; ++i )
				{
					uint32_t const pl = A.cfp_p[o0+i];
					double const w0 = A.cfp_w[o0+i];
					double weight = 0.0; weight += w0;
					double wlast = w0; bool ok = true;
					for ( uint32_t jj = 1; jj < len; ++jj )
					{
						uint32_t const j = len-1-jj;
						double wj;
						if ( !cfeasLookup(L[j],pl+jj,wj) ) { ok = false; break; }
						weight += wj; wlast = wj;
					}
					if ( ok ) { A.csf_p[o+cnt] = pl; A.csf_w[o+cnt] = weight; A.csf_wf[o+cnt] = w0; A.csf_wl[o+cnt] = wlast; ++cnt; }
				}
				A.scfl[s] = cnt;
			}
		}
		wv_sync();
	}

	// first stretch index with sfirst >= node (stretches are sorted by first)
	DEV uint32_t stretchLowerBound(int32_t const node) const
	{
		uint32_t lo = 0, hi = nstretch;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( A.sfirst[mid] < node ) lo = mid+1; else hi = mid; }
		return lo;
	}

	// ================= T5: stretch links (:3388-3480) =================
	DEV void computeStretchLinks()
	{
		// count per stretch A: candidates B with B.first == A.last
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nstretch; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t cnt = 0;
			if ( i < nstretch )
			{
				uint32_t b = stretchLowerBound(A.slast[i]);
				while ( b < nstretch && A.sfirst[b] == A.slast[i] ) { ++cnt; ++b; }
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( i < nstretch ) A.sidx[i] = base+pre;
			base += tot;
		}
		uint32_t const ncand = base;
		uint32_t const p2 = next_pow2(ncand < 2 ? 2 : ncand);
		if ( p2 > C.rlcap ) { setOverflow(256); nrl = 0; return; }
		for ( uint32_t q = lane; q < p2; q += WSZ ) A.rlkey[q] = ~0ull;
		wv_sync();
		for ( uint32_t i = lane; i < nstretch; i += WSZ )
		{
			uint32_t o = A.sidx[i];
			uint32_t b = stretchLowerBound(A.slast[i]);
			for ( ; b < nstretch && A.sfirst[b] == A.slast[i]; ++b, ++o )
			{
				// getReverseStretchLinkWeight(A=i,B=b): max over common bucket of B.w + (A.w - A.wf)
				uint32_t const shift = A.sslen[b]-1;
				double weight = 0.0;
				uint32_t ia = 0, ib = 0;
				uint32_t const na = A.scfl[i], nb = A.scfl[b];
				uint32_t const oa = A.scfo[i], ob = A.scfo[b];
				while ( ia < na && ib < nb )
				{
					uint32_t const pa = A.csf_p[oa+ia], pb = A.csf_p[ob+ib]+shift;
					if ( pa == pb )
					{
						double const lweight = A.csf_w[ob+ib] + (A.csf_w[oa+ia] - A.csf_wf[oa+ia]);
						weight = lweight > weight ? lweight : weight;
						++ia; ++ib;
					}
					else if ( pa < pb ) ++ia; else ++ib;
				}
				if ( weight >= 1e-1 ) A.rlkey[o] = (static_cast<uint64_t>(b)<<32) | i;
			}
		}
		wv_sync();
		wv_bitonic_sort(A.rlkey,p2);
		// count valid
		uint32_t cnt = 0;
		for ( uint32_t q = lane; q < ncand; q += WSZ ) cnt += (A.rlkey[q] != ~0ull);
		nrl = wv_sum(cnt);
	}

	// ---- cached stretch position weights (:3906-3934) ----
	DEV int32_t sfFind(uint32_t const s, uint32_t const p) const
	{
		uint32_t const o = A.sfo[s], n = A.sfl[s];
		for ( uint32_t i = 0; i < n; ++i ) { uint32_t const q = A.sf_p[o+i]; if ( q == p ) return o+i; if ( q > p ) break; }
		return -1;
	}
	DEV int32_t csfFind(uint32_t const s, uint32_t const p) const
	{
		uint32_t const o = A.scfo[s], n = A.scfl[s];
		for ( uint32_t i = 0; i < n; ++i ) { uint32_t const q = A.csf_p[o+i]; if ( q == p ) return o+i; if ( q > p ) break; }
		return -1;
	}

	// ================= T6: reverse enumeration (prepareTraverse :3582-3739), lane 0 =================
	// returns pool index or -1 on overflow
	DEV int32_t extendReversePath(int32_t const parent, uint32_t const s)
	{
		if ( nrp >= C.poolcap ) { setOverflow(512); return -1; }
		int32_t const id = nrp++;
		uint32_t const ppos = A.rp_pos[parent];
		int32_t const sfo = csfFind(s,ppos);
		uint32_t const plen = A.rp_len[parent];
		double weight = A.rp_weight[parent];
		uint32_t baselen = A.rp_baselen[parent];
		if ( plen == 0 ) { baselen = A.sslen[s]+k-1; weight = sfo >= 0 ? A.csf_w[sfo] : 0.0; }
		else { baselen += A.sslen[s]-1; if ( sfo >= 0 ) weight += A.csf_w[sfo] - A.csf_wf[sfo]; }
		A.rp_parent[id] = parent; A.rp_stretch[id] = s; A.rp_len[id] = plen+1;
		A.rp_pos[id] = ppos + A.sslen[s]-1; A.rp_front[id] = A.nv[A.sfirst[s]];
		A.rp_weight[id] = weight; A.rp_baselen[id] = baselen;
		return id;
	}
	DEV bool checkReversePathFeasiblePosition(int32_t const id) const
	{
		if ( A.rp_len[id] )
		{
			uint32_t const s = A.rp_stretch[id];
			uint32_t const checkpos = A.rp_pos[id] - (A.sslen[s]-1);
			int32_t const f = csfFind(s,checkpos);
			return f >= 0 && A.csf_w[f] >= 0.5;
		}
		return true;
	}

	// libstdc++ std::sort (introsort + final insertion sort) on the ARP index array with
	// comparator (front, baselen) -- reproduces the permutation of DebruijnGraph.hpp:3742 exactly
	DEV bool arpLess(int32_t const a, int32_t const b) const
	{
		uint32_t const fa = A.rp_front[a], fb = A.rp_front[b];
		if ( fa != fb ) return fa < fb;
		return A.rp_baselen[a] < A.rp_baselen[b];
	}
	DEV void arpUnguardedLinearInsert(int32_t * last)
	{
		int32_t const val = *last;
		int32_t * next = last-1;
		while ( arpLess(val,*next) ) { *last = *next; last = next; --next; }
		*last = val;
	}
	DEV void arpInsertionSort(int32_t * first, int32_t * last)
	{
		if ( first == last ) return;
		for ( int32_t * i = first+1; i != last; ++i )
		{
			if ( arpLess(*i,*first) )
			{
				int32_t const val = *i;
				for ( int32_t * q = i; q != first; --q ) *q = *(q-1);
				*first = val;
			}
			else arpUnguardedLinearInsert(i);
		}
	}
	DEV void arpSort(int32_t * first, int32_t * last)
	{
		if ( first == last ) return;
		int64_t const n = last-first;
		int depth = 0; { int64_t t = n; while ( t > 1 ) { t >>= 1; ++depth; } depth *= 2; }
		// explicit stack replaces the recursion of __introsort_loop (recurse on [cut,last), loop on [first,cut))
		int32_t * stF[64]; int32_t * stL[64]; int stD[64]; int sp = 0;
		stF[0] = first; stL[0] = last; stD[0] = depth; sp = 1;
		while ( sp )
		{
			--sp;
			int32_t * f = stF[sp]; int32_t * l = stL[sp]; int d = stD[sp];
			while ( l-f > 16 )
			{
				if ( d == 0 ) { setOverflow(1024); return; } // heapsort fallback of introsort: not reproduced, fail loudly
				--d;
				// __unguarded_partition_pivot
				int32_t * mid = f + (l-f)/2;
				int32_t * a = f+1; int32_t * b = mid; int32_t * c = l-1;
				// __move_median_to_first(f,a,b,c)
				if ( arpLess(*a,*b) )
				{
					if ( arpLess(*b,*c) ) { int32_t t = *f; *f = *b; *b = t; }
					else if ( arpLess(*a,*c) ) { int32_t t = *f; *f = *c; *c = t; }
					else { int32_t t = *f; *f = *a; *a = t; }
				}
				else if ( arpLess(*a,*c) ) { int32_t t = *f; *f = *a; *a = t; }
				else if ( arpLess(*b,*c) ) { int32_t t = *f; *f = *c; *c = t; }
				else { int32_t t = *f; *f = *b; *b = t; }
				// __unguarded_partition(f+1,l,f)
				int32_t * lo = f+1; int32_t * hi = l;
				while ( true )
				{
					while ( arpLess(*lo,*f) ) ++lo;
					--hi;
					while ( arpLess(*f,*hi) ) --hi;
					if ( !(lo < hi) ) break;
					int32_t t = *lo; *lo = *hi; *hi = t;
					++lo;
				}
				int32_t * cut = lo;
				// recursion on [cut,l) happens FIRST in libstdc++, then the loop continues on [f,cut):
				// the two ranges are disjoint, so the order of processing does not change the result
				if ( sp < 64 ) { stF[sp] = cut; stL[sp] = l; stD[sp] = d; ++sp; } else { setOverflow(1024); return; }
				l = cut;
			}
		}
		// __final_insertion_sort
		if ( n > 16 )
		{
			arpInsertionSort(first,first+16);
			for ( int32_t * i = first+16; i != last; ++i ) arpUnguardedLinearInsert(i);
		}
		else arpInsertionSort(first,last);
	}

	DEV void reverseEnumerate(uint32_t const lastkmer, int32_t const lastnode, int64_t const lmax)
	{
		// lane 0 only
		nrp = 0; narp = 0;
		for ( uint32_t i = 0; i < C.blcap; ++i ) A.arph_n[i] = 0;
		uint32_t nrpst = 0;
		if ( lastnode >= 0 )
		{
			int32_t const id = nrp++;
			A.rp_parent[id] = -1; A.rp_stretch[id] = -1; A.rp_len[id] = 0; A.rp_pos[id] = 0; A.rp_front[id] = lastkmer;
			A.rp_weight[id] = 0.0; A.rp_baselen[id] = k;
			HeapWI e; e.w = 0.0; e.idx = id; e.pad = 0;
			heap_push<HeapWI,CmpWGreater>(A.rpst,nrpst,e);
		}
		while ( nrpst )
		{
			HeapWI const top = A.rpst[0];
			heap_popvoid<HeapWI,CmpWGreater>(A.rpst,nrpst);
			int32_t const rp = top.idx;
			uint32_t const srcbaselen = A.rp_baselen[rp];
			if ( srcbaselen >= C.blcap ) { setOverflow(2048); return; }
			HeapWI * H = A.arph + 12*srcbaselen;
			uint32_t hn = A.arph_n[srcbaselen];
			if ( hn == 12 )
			{
				if ( A.rp_weight[rp] <= H[0].w ) continue;
				else heap_popvoid<HeapWI,CmpWLess>(H,hn);
			}
			HeapWI e; e.w = A.rp_weight[rp]; e.idx = rp; e.pad = 0;
			heap_push<HeapWI,CmpWLess>(H,hn,e);
			A.arph_n[srcbaselen] = hn;
			A.arp[narp++] = rp;
			if ( A.rp_len[rp] == 0 )
			{
				for ( uint32_t s = 0; s < nstretch; ++s )
					if ( A.slast[s] == lastnode )
					{
						int32_t const rpe = extendReversePath(rp,s);
						if ( rpe < 0 ) return;
						if ( checkReversePathFeasiblePosition(rpe) )
						{
							if ( nrpst >= C.poolcap ) { setOverflow(512); return; }
							HeapWI x; x.w = A.rp_weight[rpe]; x.idx = rpe; x.pad = 0;
							heap_push<HeapWI,CmpWGreater>(A.rpst,nrpst,x);
						}
					}
			}
			else if ( static_cast<int64_t>(A.rp_baselen[rp]) < (lmax+1)/2 )
			{
				uint32_t const laststretch = A.rp_stretch[rp];
				// equal_range over the sorted (to,from) link pairs
				uint32_t lo = 0, hi = nrl;
				while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( (A.rlkey[mid]>>32) < laststretch ) lo = mid+1; else hi = mid; }
				for ( uint32_t q = lo; q < nrl && (A.rlkey[q]>>32) == laststretch; ++q )
				{
					int32_t const rpe = extendReversePath(rp,A.rlkey[q]&0xFFFFFFFFu);
					if ( rpe < 0 ) return;
					if ( checkReversePathFeasiblePosition(rpe) )
					{
						if ( nrpst >= C.poolcap ) { setOverflow(512); return; }
						HeapWI x; x.w = A.rp_weight[rpe]; x.idx = rpe; x.pad = 0;
						heap_push<HeapWI,CmpWGreater>(A.rpst,nrpst,x);
					}
				}
			}
		}
		// sort ARP by (front,baselen) with std::sort's permutation, then weight ranks (:3742-3757)
		arpSort(A.arp,A.arp+narp);
		// ARWT = (weight,index) ascending: ranks by counting (total order, so any method is identical)
		for ( uint32_t i = 0; i < narp; ++i )
		{
			double const wi = A.rp_weight[A.arp[i]];
			uint32_t r = 0;
			for ( uint32_t j = 0; j < narp; ++j )
			{
				double const wj = A.rp_weight[A.arp[j]];
				if ( wj < wi || (wj == wi && j < i) ) ++r;
			}
			A.arw[i] = r; A.arwr[i] = narp-r-1;
		}
	}

	// ================= T7: forward enumeration + pair loop (:4838-5097), lane 0 =================
	DEV int32_t extendPath(int32_t const parent, uint32_t const s)
	{
		if ( np >= C.poolcap ) { setOverflow(512); return -1; }
		int32_t const id = np++;
		uint32_t const ppos = parent >= 0 ? A.p_pos[parent] : 0;
		uint32_t const plen = parent >= 0 ? A.p_len[parent] : 0;
		double weight = parent >= 0 ? A.p_weight[parent] : 0.0;
		uint32_t baselen = parent >= 0 ? A.p_baselen[parent] : 0;
		int32_t const sfo = sfFind(s,ppos);
		if ( plen == 0 ) { baselen = A.sslen[s]+k-1; weight = sfo >= 0 ? A.sf_w[sfo] : 0; }
		else { baselen += A.sslen[s]-1; if ( sfo >= 0 ) weight += A.sf_w[sfo] - A.sf_wf[sfo]; }
		A.p_parent[id] = parent; A.p_stretch[id] = s; A.p_len[id] = plen+1; A.p_pos[id] = ppos + (A.sslen[s]-1);
		A.p_weight[id] = weight; A.p_baselen[id] = baselen;
		return id;
	}
	DEV bool apqPush(int32_t const id)
	{
		uint32_t const bl = A.p_baselen[id];
		if ( bl >= C.blcap ) { setOverflow(2048); return false; }
		HeapWI * H = A.apq + 12*bl; uint32_t hn = A.apq_n[bl];
		HeapWI e; e.w = A.p_weight[id]; e.idx = id; e.pad = 0;
		if ( hn == 12 )
		{
			if ( e.w > H[0].w ) { heap_popvoid<HeapWI,CmpWLess>(H,hn); heap_push<HeapWI,CmpWLess>(H,hn,e); }
		}
		else heap_push<HeapWI,CmpWLess>(H,hn,e);
		A.apq_n[bl] = hn;
		return true;
	}
	DEV double getPairScore(int32_t const path, int32_t const rp) const
	{
		uint32_t const s = A.p_stretch[path];
		uint32_t const spos = A.p_pos[path] - (A.sslen[s]-1);
		int32_t const sfo = sfFind(s,spos);
		if ( sfo >= 0 ) return A.p_weight[path] + A.rp_weight[rp] - A.sf_wl[sfo];
		else return A.p_weight[path] + A.rp_weight[rp];
	}
	// candidate text of (forward path, reverse path) (decodePathPair :4267-4300); returns new conso or ~0 on overflow
	DEV uint32_t decodePathPair(int32_t const path, int32_t const rp, uint32_t o)
	{
		// forward stretches root -> leaf
		int32_t chain[64]; uint32_t cl = 0;
		for ( int32_t q = path; q >= 0; q = A.p_parent[q] ) { if ( cl >= 64 ) { setOverflow(4096); return ~0u; } chain[cl++] = A.p_stretch[q]; }
		uint32_t need = k;
		for ( uint32_t i = 0; i < cl; ++i ) need += A.sslen[chain[i]]-1;
		for ( int32_t q = rp; q >= 0 && A.rp_len[q]; q = A.rp_parent[q] ) need += A.sslen[A.rp_stretch[q]]-1;
		if ( o + need > C.conscap - DACC_MAXCONS_OF(P.w) ) { setOverflow(4096); return ~0u; } // the tail holds the accepted consensus
		uint32_t const firstv = A.nv[A.sfirst[chain[cl-1]]];
		for ( uint32_t i = 0; i < k; ++i ) A.cons[o++] = (firstv >> (2*(k-1-i))) & 3;
		for ( uint32_t ii = 0; ii < cl; ++ii )
		{
			uint32_t const s = chain[cl-1-ii];
			int32_t const * L = A.links + A.slink[s];
			for ( uint32_t j = 1; j < A.sslen[s]; ++j ) A.cons[o++] = A.nv[L[j]] & 3;
		}
		for ( int32_t q = rp; q >= 0 && A.rp_len[q]; q = A.rp_parent[q] )
		{
			uint32_t const s = A.rp_stretch[q];
			int32_t const * L = A.links + A.slink[s];
			for ( uint32_t j = 1; j < A.sslen[s]; ++j ) A.cons[o++] = A.nv[L[j]] & 3;
		}
		return o;
	}

	DEV void forwardAndPairs(int32_t const firstnode, int64_t const lmin, int64_t const lmax, uint32_t const maxfullpath)
	{
		// lane 0 only
		np = 0; nsiq = 0; maxsiq = 0;
		for ( uint32_t s = 0; s < nstretch; ++s )
			if ( A.sfirst[s] == firstnode )
			{
				int32_t const id = extendPath(-1,s);
				if ( id < 0 || !apqPush(id) ) return;
			}
		for ( uint32_t zz = 0; zz < C.blcap; ++zz )
			while ( A.apq_n[zz] )
			{
				HeapWI * H = A.apq + 12*zz; uint32_t hn = A.apq_n[zz];
				int32_t const path = H[0].idx;
				heap_popvoid<HeapWI,CmpWLess>(H,hn);
				A.apq_n[zz] = hn;
				int64_t const candlen = static_cast<int64_t>(A.p_pos[path]) + k;
				uint32_t const laststretch = A.p_stretch[path];
				uint32_t const front = A.nv[A.slast[laststretch]];
				// equal_range on front, then [lower_bound(baselen lo), upper_bound(baselen hi)) (:4909-4943)
				uint32_t lo = 0, hi = narp;
				while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( A.rp_front[A.arp[mid]] < front ) lo = mid+1; else hi = mid; }
				uint32_t e = lo;
				while ( e < narp && A.rp_front[A.arp[e]] == front ) ++e;
				int64_t bllo = lmin + static_cast<int64_t>(k) - candlen; if ( bllo < 0 ) bllo = 0;
				int64_t blhi = lmax + static_cast<int64_t>(k) - candlen; if ( blhi < 0 ) blhi = 0;
				uint32_t const bllo16 = static_cast<uint16_t>(bllo), blhi16 = static_cast<uint16_t>(blhi);
				uint32_t sub = lo;
				while ( sub < e && A.rp_baselen[A.arp[sub]] < bllo16 ) ++sub;
				uint32_t sup = sub;
				while ( sup < e && !(blhi16 < A.rp_baselen[A.arp[sup]]) ) ++sup;
				if ( sub != sup )
				{
					// primary score interval: the reverse path of maximum weight rank in [sub,sup)
					uint32_t mi = sub;
					for ( uint32_t i = sub+1; i < sup; ++i ) if ( A.arwr[i] < A.arwr[mi] ) mi = i;
					if ( nsiq >= C.poolcap ) { setOverflow(512); return; }
					HeapSI si; si.left = sub; si.right = sup; si.current = mi; si.path = path; si.w = getPairScore(path,A.arp[mi]);
					heap_push<HeapSI,CmpWGreater>(A.siq,nsiq,si);
					if ( nsiq > maxsiq ) maxsiq = nsiq;
				}
				uint32_t const pbl = A.p_baselen[path];
				if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
				{
					int32_t const lastn = A.slast[laststretch];
					for ( uint32_t s = stretchLowerBound(lastn); s < nstretch && A.sfirst[s] == lastn; ++s )
					{
						int32_t const sfo = sfFind(s,A.p_pos[path]);
						double const eweight = sfo >= 0 ? A.sf_w[sfo] : 0.0;
						if ( eweight > 0.1 )
						{
							int32_t const ep = extendPath(path,s);
							if ( ep < 0 ) return;
							if ( A.p_weight[ep] > 0.1 && static_cast<int64_t>(A.p_pos[ep]) + k <= lmax )
								if ( !apqPush(ep) ) return;
						}
					}
				}
			}
		uint32_t prevo = 0, prevlen = ~0u;
		for ( uint32_t numfullpath = 0; nsiq && numfullpath < maxfullpath; ++numfullpath )
		{
			HeapSI const si = A.siq[0];
			heap_popvoid<HeapSI,CmpWGreater>(A.siq,nsiq);
			// nextScoreInterval (:3513-3534): next lower weight rank inside [left,right)
			{
				uint32_t const v = A.arw[si.current];
				if ( v )
				{
					bool found = false; uint32_t bu = 0, bi = 0;
					for ( uint32_t i = si.left; i < si.right; ++i )
					{
						uint32_t const r = A.arw[i];
						if ( r <= v-1 && (!found || r > bu) ) { found = true; bu = r; bi = i; }
					}
					if ( found )
					{
						HeapSI sic = si; sic.current = bi; sic.w = getPairScore(si.path,A.arp[bi]);
						if ( nsiq >= C.poolcap ) { setOverflow(512); return; }
						heap_push<HeapSI,CmpWGreater>(A.siq,nsiq,sic);
					}
				}
			}
			double const weight = si.w;
			if ( ncdh == 16 )
			{
				if ( weight <= A.cdh[0].w ) continue;
				else heap_popvoid<HeapCC,CmpWLess>(A.cdh,ncdh);
			}
			uint32_t const consstart = conso;
			uint32_t const nc = decodePathPair(si.path,A.arp[si.current],conso);
			if ( nc == ~0u ) return;
			conso = nc;
			uint32_t const conslen = conso-consstart;
			if ( conslen == prevlen )
			{
				bool eq = true;
				for ( uint32_t i = 0; i < conslen; ++i ) if ( A.cons[prevo+i] != A.cons[consstart+i] ) { eq = false; break; }
				if ( eq ) continue;
			}
			prevo = consstart; prevlen = conslen;
			HeapCC cc; cc.w = weight; cc.o = consstart; cc.l = conslen;
			heap_push<HeapCC,CmpWLess>(A.cdh,ncdh,cc);
		}
	}

	// ================= T9: candidate errors, lane per (candidate,string) (:5355-5363) =================
	// global edit distance, bit-parallel (Myers), pattern = window string (<= LSTR symbols: one, two or LPW words)
	DEV uint32_t myersDistance(uint32_t const j, uint8_t const * text, uint32_t const n) const
	{
		uint32_t const m = A.slen[j];
		if ( m == 0 ) return n;
		uint32_t const lpw = C.lstr>>6;
		uint64_t const * PEQ = A.peq + static_cast<uint64_t>(4*lpw)*j;
		uint32_t score = m;
		if ( m <= 64 )
		{
			uint64_t Pv = ~0ull, Mv = 0;
			uint64_t const top = 1ull<<(m-1);
			for ( uint32_t c = 0; c < n; ++c )
			{
				uint64_t const Eq = PEQ[lpw*text[c]];
				uint64_t const Xv = Eq | Mv;
				uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
				uint64_t Ph = Mv | ~(Xh | Pv);
				uint64_t Mh = Pv & Xh;
				if ( Ph & top ) ++score; else if ( Mh & top ) --score;
				Ph = (Ph<<1) | 1ull; Mh <<= 1;
				Pv = Mh | ~(Xv | Ph);
				Mv = Ph & Xv;
			}
		}
		else if ( m <= 128 )
		{
			uint64_t Pv0 = ~0ull, Mv0 = 0, Pv1 = ~0ull, Mv1 = 0;
			uint64_t const top = 1ull<<(m-65);
			for ( uint32_t c = 0; c < n; ++c )
			{
				uint64_t const Eq0 = PEQ[lpw*text[c]], Eq1 = PEQ[lpw*text[c]+1];
				// word 0
				uint64_t const Xv0 = Eq0 | Mv0;
				uint64_t const Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
				uint64_t Ph0 = Mv0 | ~(Xh0 | Pv0);
				uint64_t Mh0 = Pv0 & Xh0;
				uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
				Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
				Pv0 = Mh0 | ~(Xv0 | Ph0);
				Mv0 = Ph0 & Xv0;
				// word 1: the horizontal delta leaving word 0 enters here (a negative one acts like a match)
				uint64_t const Eq1c = Eq1 | mhc;
				uint64_t const Xv1 = Eq1 | Mv1;
				uint64_t const Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
				uint64_t Ph1 = Mv1 | ~(Xh1 | Pv1);
				uint64_t Mh1 = Pv1 & Xh1;
				if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
				Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
				Pv1 = Mh1 | ~(Xv1 | Ph1);
				Mv1 = Ph1 & Xv1;
			}
		}
		else if ( m <= LSTR )
		{
			// block-wise over LPW words (constant trip counts: the column state stays in registers); the score is read
			// at the top bit of the pattern's last word, the words behind it compute and are ignored
			uint64_t Pv[LPW], Mv[LPW];
			for ( uint32_t b = 0; b < LPW; ++b ) { Pv[b] = ~0ull; Mv[b] = 0; }
			uint32_t const lw = (m-1)>>6;
			uint64_t const top = 1ull<<((m-1)&63);
			for ( uint32_t c = 0; c < n; ++c )
			{
				uint64_t phc = 1, mhc = 0;   // horizontal delta entering row 0: +1 (global alignment)
				for ( uint32_t b = 0; b < LPW; ++b )
				{
					uint64_t const Eq = PEQ[lpw*text[c]+b];
					uint64_t const Eqc = Eq | mhc;
					uint64_t const Xv = Eq | Mv[b];
					uint64_t const Xh = (((Eqc & Pv[b]) + Pv[b]) ^ Pv[b]) | Eqc;
					uint64_t Ph = Mv[b] | ~(Xh | Pv[b]);
					uint64_t Mh = Pv[b] & Xh;
					if ( b == lw ) { if ( Ph & top ) ++score; else if ( Mh & top ) --score; }
					uint64_t const pho = Ph>>63, mho = Mh>>63;
					Ph = (Ph<<1) | phc; Mh = (Mh<<1) | mhc;
					Pv[b] = Mh | ~(Xv | Ph);
					Mv[b] = Ph & Xv;
					phc = pho; mhc = mho;
				}
			}
		}
		else
		{
			// beyond LSTR: the same over nw words with the column state of this lane in the arena
			uint32_t const nw = (m+63)>>6;
			uint64_t * const Pv = A.mst + static_cast<uint64_t>(lane)*2*lpw, * const Mv = Pv + lpw;
			for ( uint32_t b = 0; b < nw; ++b ) { Pv[b] = ~0ull; Mv[b] = 0; }
			uint64_t const top = 1ull<<((m-1)&63);
			for ( uint32_t c = 0; c < n; ++c )
			{
				uint64_t phc = 1, mhc = 0;
				for ( uint32_t b = 0; b < nw; ++b )
				{
					uint64_t const Eq = PEQ[lpw*text[c]+b];
					uint64_t const Eqc = Eq | mhc;
					uint64_t const pv = Pv[b], mv = Mv[b];
					uint64_t const Xv = Eq | mv;
					uint64_t const Xh = (((Eqc & pv) + pv) ^ pv) | Eqc;
					uint64_t Ph = mv | ~(Xh | pv);
					uint64_t Mh = pv & Xh;
					if ( b == nw-1 ) { if ( Ph & top ) ++score; else if ( Mh & top ) --score; }
					uint64_t const pho = Ph>>63, mho = Mh>>63;
					Ph = (Ph<<1) | phc; Mh = (Mh<<1) | mhc;
					Pv[b] = Mh | ~(Xv | Ph);
					Mv[b] = Ph & Xv;
					phc = pho; mhc = mho;
				}
			}
		}
		return score;
	}

	DEV void buildPeq()
	{
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t const lpw = C.lstr>>6;
			uint64_t * e = A.peq + static_cast<uint64_t>(4*lpw)*j;
			for ( uint32_t i = 0; i < 4*lpw; ++i ) e[i] = 0;
			uint32_t const m = A.slen[j];
			uint8_t const * s = A.str + static_cast<uint64_t>(j)*C.lstr;
			for ( uint32_t i = 0; i < m; ++i ) e[lpw*s[i] + (i>>6)] |= 1ull<<(i&63);
		}
		wv_sync();
	}

	// ================= traverse (:4496-5170) =================
	DEV bool traverse(int64_t const lmin, int64_t const lmax)
	{
		if ( lane == 0 ) { conso = 0; ncdh = 0; nacc = 0; }
		uint32_t const firstthres = nmfirst ? ((static_cast<uint32_t>((~A.mfirst[0])>>32))*3)/4 : 0;
		uint32_t const lastthres = nmlast ? ((static_cast<uint32_t>((~A.mlast[0])>>32))*3)/4 : 0;
		for ( uint32_t fi = 0; fi < nmfirst; ++fi )
		{
			uint64_t const fkey = ~A.mfirst[fi];
			if ( static_cast<uint32_t>(fkey>>32) < firstthres ) break;
			for ( uint32_t li = 0; li < nmlast; ++li )
			{
				uint64_t const lkey = ~A.mlast[li];
				if ( static_cast<uint32_t>(lkey>>32) < lastthres ) break;
				uint32_t const firstk = static_cast<uint32_t>(fkey), lastk = static_cast<uint32_t>(lkey);
				int32_t const firstnode = findNode(firstk);
				int32_t const lastnode = findNode(lastk);
				// a last k-mer that is not a node of the (filtered) graph has an empty reverse enumeration (it starts from the
				// node of `last`, :3582-3600): no score interval, no candidate, nothing of this pair is kept
				if ( lastnode < 0 ) continue;
				// prepareTraverse (:3541-3787)
				PROF_T0
				computeStretches(firstnode,lastnode);
				PROF(*this,8)
				computeStretchFeas();
				PROF(*this,9)
				computeStretchLinks();
				PROF(*this,10)
				if ( wv_any(flags != 0) ) return false;
				if ( lane == 0 )
				{
					reverseEnumerate(lastk,lastnode,lmax);
					PROF(*this,11)
					if ( ! flags ) forwardAndPairs(firstnode,lmin,lmax,16);
					PROF(*this,12)
				}
				wv_sync();
				flags = wv_bcast(flags,0);
				stat(0,nn); stat(1,npre); stat(2,nstretch); stat(3,nsf); stat(4,ncsf); stat(5,nrl); stat(6,nrp); stat(7,narp); stat(8,np); stat(9,maxsiq); stat(10,conso); stat(11,maxlinks);
				if ( flags ) return false;
			}
		}
		// CDH -> CH -> ACC in descending weight (:5099-5136), lane 0
		PROF_T0
		if ( lane == 0 )
		{
			uint32_t nch = 0;
			while ( ncdh ) { HeapCC const c = A.cdh[0]; heap_popvoid<HeapCC,CmpWLess>(A.cdh,ncdh); heap_push<HeapCC,CmpWGreater>(A.ch,nch,c); }
			while ( nch ) { A.acc[nacc++] = A.ch[0]; heap_popvoid<HeapCC,CmpWGreater>(A.ch,nch); }
		}
		wv_sync();
		uint32_t const nc = wv_bcast(nacc,0);
		nacc = nc;
		// errors: sum over strings of edit distance(candidate,string)
		for ( uint32_t t = lane; t < nc*mao; t += WSZ )
		{
			uint32_t const c = t / mao, j = t - c*mao;
			A.canderr[t] = myersDistance(j,A.cons + A.acc[c].o,A.acc[c].l);
		}
		wv_sync();
		if ( lane == 0 )
		{
			for ( uint32_t c = 0; c < nc; ++c )
			{
				uint64_t s = 0;
				for ( uint32_t j = 0; j < mao; ++j ) s += A.canderr[c*mao+j];
				A.accerr[c] = static_cast<double>(s);
			}
			// std::sort by error for <= 16 elements == libstdc++ insertion sort (stable) (:5156)
			for ( uint32_t i = 1; i < nc; ++i )
			{
				HeapCC const v = A.acc[i]; double const e = A.accerr[i];
				if ( e < A.accerr[0] )
				{
					for ( uint32_t q = i; q > 0; --q ) { A.acc[q] = A.acc[q-1]; A.accerr[q] = A.accerr[q-1]; }
					A.acc[0] = v; A.accerr[0] = e;
				}
				else
				{
					uint32_t q = i;
					while ( e < A.accerr[q-1] ) { A.acc[q] = A.acc[q-1]; A.accerr[q] = A.accerr[q-1]; --q; }
					A.acc[q] = v; A.accerr[q] = e;
				}
			}
		}
		wv_sync();
		PROF(*this,13)
		return nc != 0;
	}

	// ================= G7: gap filling at filterfreq 0 (getLevelSuccessors(2) :1016-1161), lane 0 =================
	DEV void levelSuccessors2()
	{
		if ( lane == 0 )
		{
			uint32_t const s = 2;
			uint32_t nls = 0, nane = 0;
			uint32_t const lcap = C.nodecap*4/4; // entries of 4 words
			for ( uint32_t i = 0; i < nn; ++i )
			{
				uint32_t const v = A.nv[i];
				uint32_t const low = static_cast<uint32_t>((static_cast<uint64_t>(v)<<(2*s)) & kmask);
				uint32_t const high = low | 0xF;
				// lower_bound / upper_bound over node keys
				uint32_t lo = 0, hi = nn;
				while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( A.nv[mid] < low ) lo = mid+1; else hi = mid; }
				for ( uint32_t npi = lo; npi < nn && A.nv[npi] <= high; ++npi )
				{
					uint32_t const nvv = A.nv[npi];
					// s == 2: the single intermediate word (i = 1)
					uint32_t const vhigh = static_cast<uint32_t>((static_cast<uint64_t>(v)<<2) & kmask);
					uint32_t const vlow = nvv >> 2;
					uint32_t const cv = vlow | vhigh;
					if ( findNode(cv) < 0 )
					{
						if ( nls >= lcap ) { setOverflow(8192); goto lsdone; }     // (lane 0 must reach the barrier below like the other lanes)
						A.ls[4*nls+0] = i; A.ls[4*nls+1] = npi; A.ls[4*nls+2] = cv; A.ls[4*nls+3] = 1; ++nls;
					}
				}
			}
			for ( uint32_t q = 0; q < nls; ++q )
			{
				uint32_t const from = A.ls[4*q], to = A.ls[4*q+1]; uint32_t const cv = A.ls[4*q+2]; uint32_t const off = A.ls[4*q+3];
				// merge feasible positions of `from` (shifted by s) and `to`; both ascending in position.
				// The reference sorts pairs (pos,weight) and takes runs of equal pos: with one entry from
				// each list the run is ordered by weight, and the sum of the two is what is compared.
				uint32_t ia = 0, ib = 0;
				uint32_t const na = A.nfeas[from], nb = A.nfeas[to];
				uint32_t const oa = A.feasoff[from], ob = A.feasoff[to];
				double mweight = DACC_DBL_MIN; uint32_t mp = 0; bool any = false;
				while ( ia < na && ib < nb )
				{
					uint32_t const pa = A.fp_p[oa+ia]+s, pb = A.fp_p[ob+ib];
					if ( pa == pb )
					{
						double const wa = A.fp_w[oa+ia], wb = A.fp_w[ob+ib];
						// sorted pair order: smaller weight first; sum = first + last
						double const weight = (wa < wb) ? (wa + wb) : (wb + wa);
						if ( weight > mweight ) { mweight = weight; mp = pa - s + off; any = true; }
						++ia; ++ib;
					}
					else if ( pa < pb ) ++ia; else ++ib;
				}
				if ( any ) { A.ane[nane++] = (static_cast<uint64_t>(cv)<<32) | mp; }
			}
			// sort (v,pos) ascending: insertion sort (duplicates are identical)
			for ( uint32_t i = 1; i < nane; ++i )
			{
				uint64_t const v = A.ane[i]; uint32_t q = i;
				while ( q > 0 && A.ane[q-1] > v ) { A.ane[q] = A.ane[q-1]; --q; }
				A.ane[q] = v;
			}
			uint32_t added = 0;
			for ( uint32_t i = 0; i < nane; ++i )
			{
				uint32_t const pos = A.ane[i] & 0xFFFFFFFFu;
				int32_t seqid = -1;
				for ( uint32_t j = 0; j < mao && seqid < 0; ++j )
					if ( pos + k <= A.slen[j] ) seqid = j;
				if ( seqid != -1 )
				{
					if ( npre + added >= C.precap ) { setOverflow(1); goto lsdone; }
					A.pre[npre+added] = ((A.ane[i]>>32)<<32) | (static_cast<uint64_t>(pos)<<16) | static_cast<uint32_t>(seqid);
					++added;
				}
			}
			npre += added;
		}
		lsdone:
		wv_sync();
		npre = wv_bcast(npre,0);
		flags = wv_bcast(flags,0);
		sortInstances();
	}

	// ================= H4: length estimate (HandleContext.hpp:2051-2155) =================
	DEV int32_t estimateLength()
	{
		int32_t maxvprodindex = -1;
		uint32_t mn = 0xFFFFFFFFu, mx = 0;
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t const len = A.slen[j];
			uint32_t const lastpos = len ? len-1 : 0;   // max(len-1,0)
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
					uint32_t const len = A.slen[j];
					if ( len ) vprod *= ((len-1) < static_cast<uint32_t>(T.nsup) ? row[len-1] : 0.0);
				}
			}
			union { double d; uint64_t u; } cv; cv.d = vprod;
			uint64_t const mb = wv_max64(cv.u);
			if ( mb > bestbits )
			{
				// first index achieving the new maximum in this chunk
				uint64_t const fi = wv_min64(cv.u == mb ? i : 0xFFFFFFFFull);
				bestbits = mb; besti = static_cast<uint32_t>(fi);
			}
		}
		union { double d; uint64_t u; } dm; dm.d = DACC_DBL_MIN;
		if ( bestbits > dm.u ) maxvprodindex = besti;
		if ( maxvprodindex == -1 )
		{
			// density fallback (:2103-2155): histogram of lengths, value count-1, dot product with DPnormSquare rows
			int32_t res = -1;
			if ( lane == 0 )
			{
				uint32_t maxlen = 0;
				for ( uint32_t j = 0; j < mao; ++j ) maxlen = A.slen[j] > maxlen ? A.slen[j] : maxlen;
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
							for ( uint32_t j = 0; j < mao; ++j ) cnt += (A.slen[j] == jj);
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

	// ================= H6: consensus -> A window alignment (HandleContext.hpp:2429-2493), lane 0 =================
	// global alignment of A window (pattern, w <= 64) vs consensus (text), traceback priority
	// diagonal > DEL (A only) > INS (consensus only) -- the product's aligner definition;
	// emits the window record: rec[0]=1, rec[1+r] = offset of group r (r = 0..w+1), then symbols
	DEV void alignAndEmit(uint8_t const * cons, uint32_t const n, uint8_t * rec)
	{
		if ( DACC_WIDE_W(P.w) ) { alignAndEmitWide(cons,n,rec); return; }
		uint32_t const m = P.w;
		uint8_t const * a = A.str; // string 0 = A window
		uint64_t const * PEQ = A.peq;
		uint64_t const mask = (m == 64) ? ~0ull : ((1ull<<m)-1);
		uint64_t Pv = mask, Mv = 0; uint32_t score = m;
		A.alpv[0] = Pv; A.almv[0] = Mv; A.albot[0] = m;
		uint64_t const top = 1ull<<(m-1);
		for ( uint32_t c = 0; c < n; ++c )
		{
			uint64_t const Eq = PEQ[(C.lstr>>6)*cons[c]];
			uint64_t const Xv = Eq | Mv;
			uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			uint64_t Ph = Mv | ~(Xh | Pv);
			uint64_t Mh = Pv & Xh;
			if ( Ph & top ) ++score; else if ( Mh & top ) --score;
			Ph = (Ph<<1) | 1ull; Mh <<= 1;
			Pv = (Mh | ~(Xv | Ph)) & mask;
			Mv = (Ph & Xv) & mask;
			A.alpv[c+1] = Pv; A.almv[c+1] = Mv; A.albot[c+1] = score;
		}
		// traceback
		uint32_t i = m, j = n; uint32_t d = score; uint32_t nops = 0;
		while ( i || j )
		{
			uint32_t op;
			bool done = false;
			if ( i && j )
			{
				// D[i-1][j-1] from column j-1: bottom minus the vertical deltas of rows i..m
				uint64_t const sh = i-1;
				uint32_t const dd = A.albot[j-1] - dacc_popc64(A.alpv[j-1]>>sh) + dacc_popc64(A.almv[j-1]>>sh);
				uint32_t const neq = (a[i-1] != cons[j-1]);
				if ( dd + neq == d ) { op = neq ? 1 : 0; --i; --j; d = dd; done = true; }
			}
			if ( !done && i )
			{
				// D[i-1][j] = d - vd_j(i)
				uint64_t const bit = 1ull<<(i-1);
				int32_t const vd = (A.alpv[j] & bit) ? 1 : ((A.almv[j] & bit) ? -1 : 0);
				if ( vd == 1 ) { op = 3; --i; d = d-1; done = true; }
			}
			if ( !done ) { op = 2; --j; d = d-1; }
			A.alops[nops++] = op;
		}
		// forward emission: group r (r = 0..m) = consensus symbols inserted before A position r,
		// followed (r < m) by the symbol aligned to position r ('D' = 4 for a deletion);
		// group m holds a trailing insertion run (attached to apos == aend, HandleContext.hpp:2451-2464)
		uint8_t * off = rec+1; uint8_t * sym = rec + 1 + (m+2);
		rec[0] = 1;
		uint32_t so = 0, cpos = 0, t = nops;
		for ( uint32_t r = 0; r <= m; ++r )
		{
			off[r] = so;
			while ( t && A.alops[t-1] == 2 ) { sym[so++] = cons[cpos++]; --t; }
			if ( r < m )
			{
				uint32_t const op = A.alops[--t];
				sym[so++] = (op == 3) ? 4 : cons[cpos++];
			}
		}
		off[m+1] = so;
	}
	// the same for w in 65..128: two words per column (rows 0..63 in word 0, 64..m-1 in word 1; column c of the stores at
	// [2c], [2c+1]), the wide record layout (dev_types.hpp: 16 bit group offsets)
	DEV void alignAndEmitWide(uint8_t const * cons, uint32_t const n, uint8_t * rec)
	{
		uint32_t const m = P.w;
		uint8_t const * a = A.str;
		uint64_t const * PEQ = A.peq;
		uint32_t const lpw = C.lstr>>6;
		uint64_t const mask1 = (m == 128) ? ~0ull : ((1ull<<(m-64))-1);
		uint64_t Pv0 = ~0ull, Mv0 = 0, Pv1 = mask1, Mv1 = 0; uint32_t score = m;
		A.alpv[0] = Pv0; A.alpv[1] = Pv1; A.almv[0] = Mv0; A.almv[1] = Mv1; A.albot[0] = m;
		uint64_t const top = 1ull<<(m-65);
		for ( uint32_t c = 0; c < n; ++c )
		{
			uint64_t const Eq0 = PEQ[lpw*cons[c]], Eq1 = PEQ[lpw*cons[c]+1];
			uint64_t const Xv0 = Eq0 | Mv0;
			uint64_t const Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
			uint64_t Ph0 = Mv0 | ~(Xh0 | Pv0);
			uint64_t Mh0 = Pv0 & Xh0;
			uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
			Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
			Pv0 = Mh0 | ~(Xv0 | Ph0);
			Mv0 = Ph0 & Xv0;
			// word 1: the horizontal delta leaving word 0 enters here (a negative one acts like a match)
			uint64_t const Eq1c = Eq1 | mhc;
			uint64_t const Xv1 = Eq1 | Mv1;
			uint64_t const Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
			uint64_t Ph1 = Mv1 | ~(Xh1 | Pv1);
			uint64_t Mh1 = Pv1 & Xh1;
			if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
			Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
			Pv1 = (Mh1 | ~(Xv1 | Ph1)) & mask1;
			Mv1 = (Ph1 & Xv1) & mask1;
			A.alpv[2*c+2] = Pv0; A.alpv[2*c+3] = Pv1; A.almv[2*c+2] = Mv0; A.almv[2*c+3] = Mv1; A.albot[c+1] = score;
		}
		uint32_t i = m, j = n; uint32_t d = score; uint32_t nops = 0;
		while ( i || j )
		{
			uint32_t op;
			bool done = false;
			if ( i && j )
			{
				// D[i-1][j-1] from column j-1: bottom minus the vertical deltas of rows i..m (rows sh.. of the two words)
				uint32_t const sh = i-1;
				uint64_t const p0 = A.alpv[2*(j-1)], p1 = A.alpv[2*(j-1)+1], q0 = A.almv[2*(j-1)], q1 = A.almv[2*(j-1)+1];
				uint32_t const np_ = sh < 64 ? (dacc_popc64(p0>>sh) + dacc_popc64(p1)) : dacc_popc64(p1>>(sh-64));
				uint32_t const nm_ = sh < 64 ? (dacc_popc64(q0>>sh) + dacc_popc64(q1)) : dacc_popc64(q1>>(sh-64));
				uint32_t const dd = A.albot[j-1] - np_ + nm_;
				uint32_t const neq = (a[i-1] != cons[j-1]);
				if ( dd + neq == d ) { op = neq ? 1 : 0; --i; --j; d = dd; done = true; }
			}
			if ( !done && i )
			{
				uint32_t const r = i-1;
				uint64_t const bit = 1ull<<(r&63);
				uint64_t const pv = A.alpv[2*j+(r>>6)], mv = A.almv[2*j+(r>>6)];
				int32_t const vd = (pv & bit) ? 1 : ((mv & bit) ? -1 : 0);
				if ( vd == 1 ) { op = 3; --i; d = d-1; done = true; }
			}
			if ( !done ) { op = 2; --j; d = d-1; }
			A.alops[nops++] = op;
		}
		uint8_t * off = rec+2; uint8_t * sym = rec + 2 + 2*(m+2);
		rec[0] = 1; rec[1] = 0;
		uint32_t so = 0, cpos = 0, t = nops;
		for ( uint32_t r = 0; r <= m; ++r )
		{
			off[2*r] = so & 0xFF; off[2*r+1] = so >> 8;
			while ( t && A.alops[t-1] == 2 ) { sym[so++] = cons[cpos++]; --t; }
			if ( r < m )
			{
				uint32_t const op = A.alops[--t];
				sym[so++] = (op == 3) ? 4 : cons[cpos++];
			}
		}
		off[2*(m+1)] = so & 0xFF; off[2*(m+1)+1] = so >> 8;
	}
};

}
#endif
