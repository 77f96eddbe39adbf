/*
 * Error profile estimation, product host code: the insertion / deletion / substitution rates the
 * model tables are built from (p_i, p_d, est_cor), estimated from the data like the reference does when <las>.eprof
 * is missing (src/daccord.cpp:1653-1878):
 *
 *   for the first 1024 piles: handleIndelEstimate<8> (:271-631) -- windows of 40 bases every 5 bases over the pile (no
 *   snapped last window, :457), the B windows of the overlaps that span a window (activation :484-523, retirement with
 *   `<=` :525-529, at most maxalign strings, the A window only in two database mode :562-566); a window with at least
 *   three strings none of which repeats a 7-mer (KmerRepeatDetector(k-1), :380, 583-590) gets a de Bruijn graph with
 *   k = 8 restricted to k-mers seen at least twice (:604-606) and the TRIVIAL traversal (DebruijnGraph.hpp:3794-3824:
 *   the most frequent k-mer at position 0 and the most frequent last k-mer must be joined by one unbranched stretch);
 *   every string is then aligned to that consensus and the alignment operations are counted (:619-640).
 *   rates (:1867-1878): len = matches+mismatches+deletions, p_i = insertions/len, p_d = deletions/len,
 *   est_cor = 1 - (mismatches+deletions+insertions)/len.
 *
 * This is a one-off sample (about 2 million small windows), run on host threads; it shares nothing with the device
 * path and nothing with oracle/.  Alignment definition (libmaus2's aligner is not in the reference tree): unit costs,
 * traceback from the end preferring diagonal, then the step that consumes the first string only, then the other --
 * the same definition the trace kernel and the window kernels use.
 */
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include <thread>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include "../../include/daccord_hip.h"

namespace {

enum { OP_MATCH = 0, OP_MISMATCH = 1, OP_INS = 2, OP_DEL = 3 };     // INS: second string only, DEL: first string only
enum { EK = 8, EW = 40, EA = 5 };

struct Store
{
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen; uint64_t nreads;
	// read r as 0..3 symbols, forward or reverse complement
	void decode(int64_t const r, bool const rc, std::vector<uint8_t> & out) const
	{
		uint32_t const n = rlen[r]; uint8_t const * p = bps + boff[r];
		out.resize(n);
		for ( uint32_t i = 0; i < n; ++i ) out[i] = (p[i>>2] >> (6-2*(i&3))) & 3;
		if ( rc ) { std::reverse(out.begin(),out.end()); for ( uint32_t i = 0; i < n; ++i ) out[i] = 3-out[i]; }
	}
};

// global alignment of a[0,m) and b[0,n); operations appended to ops in forward order; returns the distance.
// Up to 128 first-string symbols (every trace block at tspace <= 128 and every 40 base window) the matrix is computed bit
// parallel (Myers / Hyyro, columns = second string, two 64 bit words of vertical deltas per column, all columns kept) and the
// traceback reads D[i-1][j-1] and the vertical delta at (i,j) out of the column words -- the same recurrence, the same
// traceback rule and therefore the same operations as the cell-by-cell form below, which remains for longer first strings
// (the estimator spent 22 s of 8 host threads on 1024 piles in the cell-by-cell form, almost all of it here).
struct Dp
{
	std::vector<uint16_t> D; std::vector<uint8_t> rev;
	std::vector<uint64_t> col; std::vector<int32_t> sc;
	uint32_t runBits(uint8_t const * a, uint32_t const m, uint8_t const * b, uint32_t const n, std::vector<uint8_t> * ops, uint64_t * cnt)
	{
		uint64_t peq[4][2] = {{0,0},{0,0},{0,0},{0,0}};
		for ( uint32_t i = 0; i < m; ++i ) peq[a[i]&3][i>>6] |= 1ull<<(i&63);
		uint64_t const mask0 = (m >= 64) ? ~0ull : ((1ull<<m)-1);
		uint64_t const mask1 = (m <= 64) ? 0ull : ((m == 128) ? ~0ull : ((1ull<<(m-64))-1));
		bool const two = m > 64;
		uint64_t const top = two ? (1ull<<(m-65)) : (1ull<<(m-1));
		col.resize(static_cast<size_t>(n+1)*4); sc.resize(n+1);
		uint64_t pv0 = mask0, mv0 = 0, pv1 = mask1, mv1 = 0; int32_t score = m;
		col[0] = pv0; col[1] = mv0; col[2] = pv1; col[3] = mv1; sc[0] = score;
		for ( uint32_t j = 1; j <= n; ++j )
		{
			uint32_t const c = b[j-1]&3;
			uint64_t const Eq0 = peq[c][0], Eq1 = peq[c][1];
			uint64_t const Xv0 = Eq0 | mv0;
			uint64_t const Xh0 = (((Eq0 & pv0) + pv0) ^ pv0) | Eq0;
			uint64_t Ph0 = mv0 | ~(Xh0 | pv0);
			uint64_t Mh0 = pv0 & Xh0;
			uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
			if ( !two ) { if ( Ph0 & top ) ++score; else if ( Mh0 & top ) --score; }
			Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
			pv0 = (Mh0 | ~(Xv0 | Ph0)) & mask0; mv0 = (Ph0 & Xv0) & mask0;
			if ( two )
			{
				uint64_t const Eq1c = Eq1 | mhc;
				uint64_t const Xv1 = Eq1 | mv1;
				uint64_t const Xh1 = (((Eq1c & pv1) + pv1) ^ pv1) | Eq1c;
				uint64_t Ph1 = mv1 | ~(Xh1 | pv1);
				uint64_t Mh1 = pv1 & Xh1;
				if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
				Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
				pv1 = (Mh1 | ~(Xv1 | Ph1)) & mask1; mv1 = (Ph1 & Xv1) & mask1;
			}
			uint64_t * q = &col[static_cast<size_t>(j)*4]; q[0] = pv0; q[1] = mv0; q[2] = pv1; q[3] = mv1; sc[j] = score;
		}
		rev.clear();
		uint32_t i = m, j = n; int32_t d = score;
		while ( i || j )
		{
			bool done = false;
			if ( i && j )
			{
				// D[i-1][j-1] = bottom(j-1) - sum of the vertical deltas of rows i..m in column j-1
				uint64_t const * q = &col[static_cast<size_t>(j-1)*4];
				uint32_t const sh = i-1;
				int32_t sum;
				if ( sh < 64 ) sum = __builtin_popcountll(q[0]>>sh) + __builtin_popcountll(q[2]) - __builtin_popcountll(q[1]>>sh) - __builtin_popcountll(q[3]);
				else sum = __builtin_popcountll(q[2]>>(sh-64)) - __builtin_popcountll(q[3]>>(sh-64));
				int32_t const dd = sc[j-1] - sum;
				bool const eq = a[i-1] == b[j-1];
				if ( dd + (eq ? 0 : 1) == d ) { rev.push_back(eq ? OP_MATCH : OP_MISMATCH); --i; --j; d = dd; done = true; }
			}
			if ( !done && i )
			{
				uint64_t const * q = &col[static_cast<size_t>(j)*4];
				uint32_t const r = i-1;
				bool const plus = (r < 64) ? ((q[0]>>r)&1) : ((q[2]>>(r-64))&1);     // D[i][j] = D[i-1][j] + 1
				if ( plus ) { rev.push_back(OP_DEL); --i; --d; done = true; }
			}
			if ( !done ) { rev.push_back(OP_INS); --j; --d; }
		}
		if ( ops ) ops->insert(ops->end(),rev.rbegin(),rev.rend());
		if ( cnt ) for ( size_t x = 0; x < rev.size(); ++x ) ++cnt[rev[x]];
		return static_cast<uint32_t>(score);
	}
	uint32_t run(uint8_t const * a, uint32_t const m, uint8_t const * b, uint32_t const n, std::vector<uint8_t> * ops, uint64_t * cnt)
	{
		if ( m >= 1 && m <= 128 && n >= 1 ) return runBits(a,m,b,n,ops,cnt);
		uint32_t const W = n+1;
		D.resize(static_cast<size_t>(m+1)*W);
		for ( uint32_t j = 0; j <= n; ++j ) D[j] = j;
		for ( uint32_t i = 1; i <= m; ++i )
		{
			uint16_t * r = &D[static_cast<size_t>(i)*W]; uint16_t const * q = r-W;
			r[0] = i; uint8_t const c = a[i-1];
			for ( uint32_t j = 1; j <= n; ++j )
			{
				uint16_t v = q[j-1] + (c != b[j-1] ? 1 : 0);
				uint16_t const u = q[j]+1, l = r[j-1]+1;
				if ( u < v ) v = u;
				if ( l < v ) v = l;
				r[j] = v;
			}
		}
		rev.clear();
		uint32_t i = m, j = n;
		while ( i || j )
		{
			uint16_t const d = D[static_cast<size_t>(i)*W+j];
			if ( i && j && D[static_cast<size_t>(i-1)*W+(j-1)] + (a[i-1] != b[j-1] ? 1 : 0) == d ) { rev.push_back(a[i-1] == b[j-1] ? OP_MATCH : OP_MISMATCH); --i; --j; }
			else if ( i && D[static_cast<size_t>(i-1)*W+j] + 1 == d ) { rev.push_back(OP_DEL); --i; }
			else { rev.push_back(OP_INS); --j; }
		}
		if ( ops ) ops->insert(ops->end(),rev.rbegin(),rev.rend());
		if ( cnt ) for ( size_t x = 0; x < rev.size(); ++x ) ++cnt[rev[x]];
		return D[static_cast<size_t>(m)*W+n];
	}
};

// position in an operation string after `na` first-string symbols (stops right behind the na-th one); also the number
// of second-string symbols passed
static inline void advanceOps(std::vector<uint8_t> const & ops, size_t & pos, uint64_t const na, uint64_t & usedA, uint64_t & usedB)
{
	usedA = 0; usedB = 0;
	while ( pos < ops.size() && usedA < na )
	{
		uint8_t const o = ops[pos++];
		if ( o == OP_INS ) ++usedB; else { ++usedA; if ( o != OP_DEL ) ++usedB; }
	}
}

// does the string contain some q-mer twice?  (KmerRepeatDetector(q).detect)  One bit per q-mer (q = 7: 2 KB), cleared again
// through the list of the q-mers seen (a sort of the q-mers per string was a quarter of the estimator's time)
static bool repeatsQmer(uint8_t const * s, uint32_t const n, uint32_t const q, std::vector<uint32_t> & tmp, std::vector<uint64_t> & bits)
{
	if ( n < q+1 ) return false;
	if ( bits.size() != ((1ull<<(2*q))+63)/64 ) bits.assign(((1ull<<(2*q))+63)/64,0);
	tmp.clear(); uint32_t v = 0; uint32_t const mask = (1u<<(2*q))-1;
	bool rep = false;
	for ( uint32_t i = 0; i < n && !rep; ++i )
	{
		v = ((v<<2)|s[i]) & mask;
		if ( i+1 >= q )
		{
			uint64_t const b = 1ull<<(v&63);
			if ( bits[v>>6] & b ) rep = true; else { bits[v>>6] |= b; tmp.push_back(v); }
		}
	}
	for ( size_t i = 0; i < tmp.size(); ++i ) bits[tmp[i]>>6] = 0;
	return rep;
}

// The k = 8 graph of one window restricted to k-mers of frequency >= 2, and its trivial traversal.
struct TrivialGraph
{
	struct Node { uint32_t v, freq, nsa; uint8_t ord[4]; uint8_t ns; };     // successors by (freq,symbol) descending, nsa of them active
	std::vector<Node> N; std::vector<int32_t> id;                           // id[k-mer] = node or -1
	std::vector<uint32_t> lastk, touched, nodev; std::vector<uint16_t> cnt, cnt0;      // cnt / cnt0: occurrences (all / at position 0) per k-mer
	struct Str { uint32_t first, ext, last, len, off; };
	std::vector<Str> S; std::vector<uint32_t> L; std::vector<uint8_t> mark;
	TrivialGraph() : id(1u<<(2*EK),-1) {}
	static uint32_t const KM = (1u<<(2*EK))-1;

	int32_t node(uint32_t const v) const { return id[v]; }
	// active successor k-mers of node z
	uint32_t succ(uint32_t const z, uint32_t const i) const { return ((N[z].v<<2)&KM) | N[z].ord[i]; }
	uint32_t activePreds(uint32_t const v) const
	{
		uint32_t c = 0;
		for ( uint32_t s = 0; s < 4; ++s )
		{
			int32_t const u = node((v>>2) | (s<<(2*(EK-1))));
			if ( u >= 0 ) for ( uint32_t i = 0; i < N[u].nsa; ++i ) if ( N[u].ord[i] == (v&3) ) { ++c; break; }
		}
		return c;
	}
	void clearIds() { for ( size_t i = 0; i < N.size(); ++i ) id[N[i].v] = -1; }

	// strings -> nodes of frequency >= 2 with their active successors; first = most frequent k-mer at position 0 among
	// those nodes, last = most frequent final k-mer of a string (both: first maximum in ascending k-mer order)
	bool build(std::vector< std::pair<uint8_t const *,uint32_t> > const & M, uint32_t & first, uint32_t & last)
	{
		clearIds(); N.clear(); lastk.clear(); touched.clear();
		if ( cnt.empty() ) { cnt.assign(1u<<(2*EK),0); cnt0.assign(1u<<(2*EK),0); }
		// occurrences per k-mer and occurrences at position 0 (16 bit k-mers: direct tables, cleared through `touched`)
		for ( size_t j = 0; j < M.size(); ++j )
		{
			uint32_t const n = M[j].second; if ( n < EK ) continue;
			uint32_t v = 0;
			for ( uint32_t i = 0; i < n; ++i )
			{
				v = ((v<<2)|M[j].first[i]) & KM;
				if ( i+1 >= EK )
				{
					if ( cnt[v]++ == 0 ) touched.push_back(v);
					if ( i+1 == EK ) ++cnt0[v];
					if ( i+1 == n ) lastk.push_back(v);
				}
			}
		}
		std::sort(lastk.begin(),lastk.end());
		// nodes = k-mers seen at least twice, in ascending k-mer order
		nodev.clear();
		for ( size_t i = 0; i < touched.size(); ++i ) if ( cnt[touched[i]] >= 2 ) nodev.push_back(touched[i]);
		std::sort(nodev.begin(),nodev.end());
		uint32_t bestc = 0; first = 0;
		for ( size_t i = 0; i < nodev.size(); ++i )
		{
			Node x; x.v = nodev[i]; x.freq = cnt[x.v]; x.nsa = 0; x.ns = 0; id[x.v] = N.size(); N.push_back(x);
			uint32_t const c0 = cnt0[x.v];
			if ( c0 > bestc ) { bestc = c0; first = x.v; }
		}
		for ( size_t i = 0; i < touched.size(); ++i ) { cnt[touched[i]] = 0; cnt0[touched[i]] = 0; }
		uint32_t bestl = 0; last = 0;
		for ( size_t l = 0; l < lastk.size(); )
		{
			size_t h = l; while ( h < lastk.size() && lastk[h] == lastk[l] ) ++h;
			if ( h-l > bestl ) { bestl = h-l; last = lastk[l]; }
			l = h;
		}
		for ( size_t z = 0; z < N.size(); ++z )
		{
			uint32_t key[4]; uint32_t n = 0;
			for ( uint32_t s = 0; s < 4; ++s ) { int32_t const u = node(((N[z].v<<2)&KM)|s); if ( u >= 0 ) key[n++] = (N[u].freq<<8)|s; }
			std::sort(key,key+n,[](uint32_t const a, uint32_t const b){ return a > b; });
			N[z].ns = n; for ( uint32_t i = 0; i < n; ++i ) N[z].ord[i] = key[i]&3;
			uint32_t a = n ? 1 : 0;
			while ( a < n && (key[a]>>8) >= (key[0]>>8)/2 ) ++a;       // KmerLimit is off for this graph (p = 0, daccord.cpp:421-422)
			N[z].nsa = a;
		}
		return !N.empty();
	}
	void addStretch(uint32_t const lo, uint32_t const hi)       // copy of L[lo,hi) as a new stretch
	{
		Str s; s.first = L[lo]; s.ext = L[lo+1]; s.last = L[hi-1]; s.len = hi-lo; s.off = L.size();
		for ( uint32_t i = lo; i < hi; ++i ) { uint32_t const v = L[i]; L.push_back(v); }
		S.push_back(s);
	}
	void split(uint32_t const v)
	{
		size_t const n0 = S.size(); std::vector<uint8_t> gone(n0,0);
		for ( size_t z = 0; z < n0; ++z )
		{
			Str const s = S[z];
			for ( uint32_t i = 1; i+1 < s.len; ++i )
				if ( L[s.off+i] == v ) { addStretch(s.off,s.off+i+1); addStretch(s.off+i,s.off+s.len); gone[z] = 1; break; }
		}
		size_t o = 0;
		for ( size_t z = 0; z < S.size(); ++z ) if ( z >= n0 || !gone[z] ) S[o++] = S[z];
		S.resize(o);
	}
	// unbranched walks without the predecessor test (computeStretches(false)), split at first and last; is there a
	// stretch from first to last?  Its k-mers spell the consensus.
	bool trivial(uint32_t const first, uint32_t const last, std::vector<uint8_t> & cons)
	{
		S.clear(); L.clear(); mark.assign(N.size(),0);
		for ( size_t z = 0; z < N.size(); ++z )
		{
			uint32_t const nsa = N[z].nsa;
			if ( !(nsa && (activePreds(N[z].v) != 1 || nsa > 1)) ) continue;
			for ( uint32_t i = 0; i < nsa; ++i )
			{
				uint32_t const start = L.size(); uint32_t cur = succ(z,i);
				L.push_back(N[z].v); mark[z] = 1; L.push_back(cur); mark[node(cur)] = 1;
				uint32_t len = 2; bool loop = (cur == N[z].v);
				while ( !loop && N[node(cur)].nsa == 1 )
				{
					cur = succ(node(cur),0); L.push_back(cur); ++len;
					if ( mark[node(cur)] ) loop = true; else mark[node(cur)] = 1;
				}
				for ( uint32_t j = start; j < start+len; ++j ) mark[node(L[j])] = 0;
				if ( loop && N[z].v != cur )
				{
					uint32_t j = 0; while ( L[start+j] != cur ) ++j;
					len = j+1; L.resize(start+len);
				}
				Str s; s.first = N[z].v; s.ext = L[start+1]; s.last = cur; s.len = len; s.off = start; S.push_back(s);
			}
		}
		split(first); split(last);
		// one survivor per (first, ext): the longest, then the smallest last (Stretch::operator<, stretchesUnique)
		std::sort(S.begin(),S.end(),[](Str const & a, Str const & b){
			if ( a.first != b.first ) return a.first < b.first; if ( a.ext != b.ext ) return a.ext < b.ext;
			if ( a.len != b.len ) return a.len > b.len; return a.last < b.last; });
		size_t o = 0;
		for ( size_t l = 0; l < S.size(); ) { size_t h = l+1; while ( h < S.size() && S[h].first == S[l].first && S[h].ext == S[l].ext ) ++h; S[o++] = S[l]; l = h; }
		S.resize(o);
		for ( size_t i = 0; i < S.size(); ++i )
			if ( S[i].first == first && S[i].last == last )
			{
				cons.clear();
				for ( int32_t q = EK-1; q >= 0; --q ) cons.push_back((first>>(2*q))&3);
				for ( uint32_t j = 1; j < S[i].len; ++j ) cons.push_back(L[S[i].off+j]&3);
				return true;
			}
		return false;
	}
};

// deep: error rate of every window that got a consensus, as round(rate * (2^32-1)) (handleIndelEstimateDeep, daccord.cpp:634-995:
// the same function as handleIndelEstimate with this one output, :963-968); collected only when asked for (--deepprofileonly)
struct Acc { uint64_t cnt[4]; uint64_t usable, unusable; std::vector<double> eloc; bool wantdeep; std::vector<uint32_t> deep; Acc() : usable(0), unusable(0), wantdeep(false) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; } };

struct Worker
{
	Store const & R; int32_t tspace; bool twodb; uint64_t maxalign;
	Dp dp; TrivialGraph G; std::vector<uint32_t> tmp; std::vector<uint64_t> qbits; std::vector<uint8_t> ra, cons;
	std::vector< std::vector<uint8_t> > rb, ops;
	Worker(Store const & r, int32_t ts, bool two, uint64_t ma) : R(r), tspace(ts), twodb(two), maxalign(ma) {}

	static uint32_t tv(void const * trace, int const tb, uint64_t const i) { return tb == 2 ? static_cast<uint16_t const *>(trace)[i] : static_cast<uint8_t const *>(trace)[i]; }

	// one pile (handleIndelEstimate): adds to A, returns the pile's mean window error rate (0: no window)
	double pile(dacc_overlap const * ita, uint32_t const n, void const * trace, int const tb, Acc & A)
	{
		if ( !n ) return 0.0;
		double maxe = 0.0, mine = 1.0;
		for ( uint32_t z = 0; z < n; ++z ) { double const e = static_cast<double>(ita[z].diffs)/static_cast<double>(ita[z].aepos-ita[z].abpos); if ( e > maxe ) maxe = e; if ( e < mine ) mine = e; }
		double const ediv = (maxe > mine) ? (maxe-mine) : 1.0;
		R.decode(ita[0].aread,false,ra);
		rb.resize(n); ops.resize(n);
		uint64_t maxaepos = 0;
		for ( uint32_t z = 0; z < n; ++z )
		{
			dacc_overlap const & o = ita[z];
			if ( static_cast<uint64_t>(o.aepos) > maxaepos ) maxaepos = o.aepos;
			R.decode(o.bread,o.flags&1,rb[z]);
			ops[z].clear();
			int64_t ai = (o.abpos/tspace)*static_cast<int64_t>(tspace), bi = o.bbpos;
			for ( int32_t b = 0; b < o.tlen/2; ++b )
			{
				int64_t const ae = std::min<int64_t>(ai+tspace,o.aepos), be = bi + tv(trace,tb,o.trace_off+2*b+1), as = std::max<int64_t>(ai,o.abpos);
				dp.run(ra.data()+as,ae-as,rb[z].data()+bi,be-bi,&ops[z],0);
				ai = ae; bi = be;
			}
		}
		struct Active { size_t pos; uint64_t ub; uint64_t aepos; };
		std::map<uint64_t,Active> act;
		uint64_t const ylimit = (maxaepos + EA >= EW) ? ((maxaepos + EA - EW)/EA) : 0;
		uint32_t zin = 0; double esum = 0; uint64_t ecnt = 0;
		std::vector< std::pair<uint8_t const *,uint32_t> > M;
		for ( uint64_t y = 0; y < ylimit; ++y )
		{
			uint64_t const astart = y*EA, aend = astart+EW;
			while ( zin < n && static_cast<int64_t>(astart) >= ita[zin].abpos )
			{
				dacc_overlap const & o = ita[zin];
				if ( o.aepos >= static_cast<int64_t>(astart) )
				{
					Active a; a.pos = 0; uint64_t ua, ub; advanceOps(ops[zin],a.pos,astart-o.abpos,ua,ub); a.ub = o.bbpos + ub; a.aepos = o.aepos;
					double const er = static_cast<double>(o.diffs)/static_cast<double>(o.aepos-o.abpos);
					uint64_t const escore = static_cast<uint64_t>(((er-mine)/ediv) * 4294967295.0);
					act[(escore<<32)|zin] = a;
				}
				++zin;
			}
			for ( std::map<uint64_t,Active>::iterator it = act.begin(); it != act.end(); ) { if ( it->second.aepos <= aend ) act.erase(it++); else ++it; }
			M.clear();
			for ( std::map<uint64_t,Active>::iterator it = act.begin(); it != act.end(); ++it )
			{
				Active & a = it->second; uint32_t const z = it->first & 0xFFFFFFFFu;
				size_t p = a.pos; uint64_t ua, ub; advanceOps(ops[z],p,EW,ua,ub);
				// the reference's test `&RC != &RC2` (daccord.cpp:522) compares two local containers (daccord.cpp:1775-1776): always
				// true, so the A window always joins the window's strings, with one database as with two (found with oracle/_ref in round 4)
				if ( M.empty() ) M.push_back(std::make_pair(static_cast<uint8_t const *>(ra.data()+astart),static_cast<uint32_t>(EW)));
				if ( M.size() < maxalign ) M.push_back(std::make_pair(static_cast<uint8_t const *>(rb[z].data()+a.ub),static_cast<uint32_t>(ub)));
				uint64_t va, vb; advanceOps(ops[z],a.pos,EA,va,vb); a.ub += vb;
			}
			if ( M.size() < 3 ) continue;
			bool rep = false;
			for ( size_t i = 0; i < M.size(); ++i ) rep = rep || repeatsQmer(M[i].first,M[i].second,EK-1,tmp,qbits);
			if ( rep ) { ++A.unusable; continue; }
			++A.usable;
			uint32_t first, last;
			if ( !G.build(M,first,last) || !G.trivial(first,last,cons) ) continue;
			uint64_t c[4] = {0,0,0,0};
			for ( size_t i = 0; i < M.size(); ++i ) dp.run(cons.data(),cons.size(),M[i].first,M[i].second,0,c);
			for ( int q = 0; q < 4; ++q ) A.cnt[q] += c[q];
			uint64_t const tot = c[0]+c[1]+c[2]+c[3];
			double const er = tot ? static_cast<double>(c[1]+c[2]+c[3])/static_cast<double>(tot) : 0.0;
			esum += er; ++ecnt;
			if ( A.wantdeep )
			{
				uint64_t const v = static_cast<uint64_t>(4294967295.0 * er + 0.5);
				A.deep.push_back(static_cast<uint32_t>(v < 4294967295ull ? v : 4294967295ull));
			}
		}
		return ecnt ? esum/ecnt : 0.0;
	}
};

}

struct dacc_eprof
{
	Store R; int32_t tspace; bool twodb; Acc A; std::string err; uint64_t skipped = 0, seen = 0;
};

extern "C" {

int dacc_eprof_create(dacc_eprof ** out, int32_t tspace, uint8_t const * bps, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads, int two_databases)
{
	if ( !out || !bps || !boff || !rlen || tspace <= 0 ) return DACC_EINVAL;
	dacc_eprof * e = 0;
	try { e = new dacc_eprof; } catch ( ... ) { return DACC_ENOMEM; }
	e->R.bps = bps; e->R.boff = boff; e->R.rlen = rlen; e->R.nreads = nreads; e->tspace = tspace; e->twodb = two_databases != 0;
	*out = e; return DACC_OK;
}
void dacc_eprof_destroy(dacc_eprof * e) { delete e; }
// piles given to dacc_eprof_add so far and how many of them were left out for malformed overlap / trace records (the
// skip must be visible to the caller -- a .las that does not belong to the .db yields a profile from a small subset otherwise)
int dacc_eprof_skipped(dacc_eprof * e, uint64_t * skipped, uint64_t * seen)
{
	if ( !e || !skipped || !seen ) return DACC_EINVAL;
	*skipped = e->skipped; *seen = e->seen;
	return DACC_OK;
}

// A malformed pile (record outside the database, trace values that do not add up, overlaps of different A reads or not
// sorted by abpos) is skipped, as the correction path drops only that pile (batch_plan.hpp) and the reference logs one read
// and goes on (daccord.cpp:2464-2478); nothing is thrown across the ABI, worker exceptions end the call with DACC_ENOMEM.
int dacc_eprof_add(dacc_eprof * e, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl,
	void const * trace, uint64_t ntrace, int trace_bytes, uint64_t maxalign, int nthreads)
{
	if ( !e || (npiles && (!piles || !ovl || !trace)) || (trace_bytes != 1 && trace_bytes != 2) ) return DACC_EINVAL;
	try
	{
		std::vector<uint8_t> good(npiles,1);
		for ( uint64_t i = 0; i < npiles; ++i )
		{
			if ( piles[i].first_ovl + piles[i].novl > novl ) return DACC_EINVAL;     // the arrays themselves are inconsistent
			for ( uint32_t z = 0; z < piles[i].novl && good[i]; ++z )
			{
				dacc_overlap const & o = ovl[piles[i].first_ovl+z];
				bool ok = !( o.aread < 0 || static_cast<uint64_t>(o.aread) >= e->R.nreads || o.bread < 0 || static_cast<uint64_t>(o.bread) >= e->R.nreads || o.abpos < 0 || o.aepos <= o.abpos ||
				     static_cast<uint32_t>(o.aepos) > e->R.rlen[o.aread] || o.bbpos < 0 || o.bepos < o.bbpos || static_cast<uint32_t>(o.bepos) > e->R.rlen[o.bread] || o.tlen < 0 || o.trace_off + o.tlen > ntrace );
				ok = ok && o.aread == piles[i].aread && ( z == 0 || ovl[piles[i].first_ovl+z-1].abpos <= o.abpos );
				if ( ok ) { int64_t const ts = e->tspace; ok = o.tlen == 2*((o.aepos+ts-1)/ts - o.abpos/ts); }
				if ( ok )
				{
					uint64_t bs = 0; for ( int32_t b = 0; b < o.tlen/2; ++b ) bs += Worker::tv(trace,trace_bytes,o.trace_off+2*b+1);
					ok = static_cast<int64_t>(bs) == o.bepos-o.bbpos;
				}
				if ( !ok ) good[i] = 0;
			}
			if ( !good[i] ) e->skipped += 1;
			e->seen += 1;
		}
		if ( nthreads < 1 ) nthreads = 1;
		std::vector<Acc> part(nthreads); std::vector<double> eloc(npiles,0.0);
		for ( int t = 0; t < nthreads; ++t ) part[t].wantdeep = e->A.wantdeep;
		std::atomic<uint64_t> next(0);
		std::atomic<int> failed(0);
		auto body = [&](int const t)
		{
			try
			{
				Worker W(e->R,e->tspace,e->twodb,maxalign);
				for ( uint64_t i = next++; i < npiles; i = next++ )
					if ( good[i] ) eloc[i] = W.pile(ovl+piles[i].first_ovl,piles[i].novl,trace,trace_bytes,part[t]);
			}
			catch ( ... ) { failed = 1; }
		};
		std::vector<std::thread> T;
		for ( int t = 1; t < nthreads; ++t )
		{
			try { T.emplace_back(body,t); } catch ( ... ) { break; }     // fewer threads than asked for is not an error
		}
		body(0);
		for ( size_t t = 0; t < T.size(); ++t ) T[t].join();
		if ( failed ) return DACC_ENOMEM;
		for ( int t = 0; t < nthreads; ++t ) { for ( int q = 0; q < 4; ++q ) e->A.cnt[q] += part[t].cnt[q]; e->A.usable += part[t].usable; e->A.unusable += part[t].unusable; e->A.deep.insert(e->A.deep.end(),part[t].deep.begin(),part[t].deep.end()); }
		for ( uint64_t i = 0; i < npiles; ++i ) if ( eloc[i] != 0.0 ) e->A.eloc.push_back(eloc[i]);      // in pile order
	}
	catch ( std::bad_alloc const & ) { return DACC_ENOMEM; }
	catch ( ... ) { return DACC_ENOMEM; }
	return DACC_OK;
}

int dacc_eprof_finish(dacc_eprof * e, uint64_t counts[4], uint64_t * usable, uint64_t * unusable, double * eavg, double * edif, double prof[3])
{
	if ( !e || !prof ) return DACC_EINVAL;
	uint64_t const matches = e->A.cnt[OP_MATCH], mism = e->A.cnt[OP_MISMATCH], ins = e->A.cnt[OP_INS], del = e->A.cnt[OP_DEL];
	if ( counts ) { counts[0] = matches; counts[1] = mism; counts[2] = ins; counts[3] = del; }
	if ( usable ) *usable = e->A.usable;
	if ( unusable ) *unusable = e->A.unusable;
	double es = 0; for ( size_t i = 0; i < e->A.eloc.size(); ++i ) es += e->A.eloc[i];
	double const avg = e->A.eloc.size() ? es/e->A.eloc.size() : 0.0;
	double dif = 0; for ( size_t i = 0; i < e->A.eloc.size(); ++i ) dif += (avg-e->A.eloc[i])*(avg-e->A.eloc[i]);
	if ( e->A.eloc.size() ) dif = std::sqrt(dif/e->A.eloc.size());
	if ( eavg ) *eavg = avg;
	if ( edif ) *edif = dif;
	uint64_t const len = matches + mism + del, numerr = mism + del + ins;
	if ( !len ) return DACC_ENOTSUP;      // no usable window: the caller must supply a profile
	prof[0] = static_cast<double>(ins)/len; prof[1] = static_cast<double>(del)/len; prof[2] = 1.0 - static_cast<double>(numerr)/len;
	return DACC_OK;
}

// --deepprofileonly (daccord.cpp:1442-1650): the error rates of the windows with a consensus, ascending; `on` before the first
// dacc_eprof_add.  The values stay valid until the next call on e.
int dacc_eprof_set_deep(dacc_eprof * e, int on) { if ( !e ) return DACC_EINVAL; e->A.wantdeep = on != 0; return DACC_OK; }
int dacc_eprof_deep(dacc_eprof * e, uint32_t const ** values, uint64_t * n)
{
	if ( !e || !values || !n ) return DACC_EINVAL;
	std::sort(e->A.deep.begin(),e->A.deep.end());
	*values = e->A.deep.data(); *n = e->A.deep.size();
	return DACC_OK;
}

// the estimator's own pile selection (daccord.cpp:1705-1737): keep the maxinput overlaps with the LOWEST score
// (ties: the later record replaces the heap's maximum), each survivor in the slot of the record it replaced, then
// std::sort by abpos
int dacc_pile_select_lowest(dacc_overlap const * in, uint64_t n, int trace_bytes, uint64_t maxinput, dacc_overlap * out, uint64_t * nout)
{
	if ( !nout || (n && (!in || !out)) || (trace_bytes != 1 && trace_bytes != 2) ) return DACC_EINVAL;
	*nout = 0;
	if ( !maxinput || !n ) return DACC_OK;
	typedef std::pair<uint64_t,uint64_t> P;       // (score, slot), max-heap on score
	std::vector<P> H; uint64_t f = 0;
	auto less = [](P const & a, P const & b){ return a.first < b.first; };
	for ( uint64_t i = 0; i < n; ++i )
	{
		uint64_t const score = static_cast<uint64_t>(ldexp(static_cast<double>(in[i].diffs)/static_cast<double>(in[i].aepos-in[i].abpos),30));
		if ( H.size() == maxinput )
		{
			if ( score > H.front().first ) continue;
			uint64_t const p = H.front().second;
			std::pop_heap(H.begin(),H.end(),less); H.pop_back();
			out[p] = in[i]; H.push_back(P(score,p)); std::push_heap(H.begin(),H.end(),less);
		}
		else { uint64_t const p = f++; H.push_back(P(score,p)); std::push_heap(H.begin(),H.end(),less); out[p] = in[i]; }
	}
	std::sort(out,out+f,[](dacc_overlap const & A, dacc_overlap const & B){ return A.abpos < B.abpos; });
	*nout = f;
	return DACC_OK;
}

}
