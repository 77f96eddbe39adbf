/*
 * P1, the pile loader's selection step (product host code): keep at most maxinput overlaps of
 * a pile and order them by abpos, exactly as src/daccord.cpp:2120-2288 does before calling
 * HandleContext::operator().  Input records are in .las order.
 *
 * The reference keeps the survivors in a bounded min-heap on
 *   score = uint64(ldexp(diffs/(aepos-abpos),30))                      (daccord.cpp:2166-2167)
 * evicting the lowest score when full (:2169-2177), re-reads them grouped by the 64 KiB input
 * block they were parsed from -- the final block first and in descending entry order
 * (:2199-2214), earlier blocks ascending (:2216-2262) -- and then std::sorts by abpos
 * (:2284-2288).  std::sort is not stable, so the order it is handed matters for ties; we hand
 * it the same order.  (Attribution of a record that straddles a 64 KiB boundary is libmaus2
 * OverlapParser behaviour outside the reference tree: taken as the block holding its last byte.)
 */
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include "../../include/daccord_hip.h"

namespace {
struct Entry { uint64_t score, block, entry, idx; };
struct ScoreHeap
{
	std::vector<Entry> H; size_t f;
	explicit ScoreHeap(size_t n) : H(n), f(0) {}
	void push(Entry const & e)
	{
		size_t i = f++; H[i] = e;
		while ( i ) { size_t const p = (i-1)>>1; if ( H[i].score < H[p].score ) { std::swap(H[i],H[p]); i = p; } else break; }
	}
	void popvoid()
	{
		H[0] = H[--f];
		size_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			size_t const m = (H[r-1].score < H[r].score) ? r-1 : r;
			if ( H[i].score < H[m].score ) return;
			std::swap(H[i],H[m]); i = m;
		}
		size_t const l = 2*i+1;
		if ( l < f && !(H[i].score < H[l].score) ) std::swap(H[i],H[l]);
	}
};
}

extern "C" int dacc_pile_select(dacc_overlap const * in, uint64_t n, int trace_bytes, uint64_t maxinput, dacc_overlap * out, uint64_t * nout)
{
	if ( !nout || (n && (!in || !out)) || (trace_bytes != 1 && trace_bytes != 2) ) return DACC_EINVAL;
	*nout = 0;
	if ( !maxinput || !n ) return DACC_OK;
	uint64_t const cap = maxinput < n ? maxinput : n;
	ScoreHeap RHO(cap);
	uint64_t const blocksize = 64*1024;
	uint64_t bytepos = 0, curblock = 0, entry = 0, lastblock = 0;
	for ( uint64_t i = 0; i < n; ++i )
	{
		uint64_t const s = 40 + static_cast<uint64_t>(in[i].tlen)*trace_bytes;
		uint64_t const endb = (bytepos+s-1)/blocksize;
		if ( endb != curblock ) { curblock = endb; entry = 0; }
		bytepos += s;
		uint64_t const score = static_cast<uint64_t>(ldexp(static_cast<double>(in[i].diffs)/static_cast<double>(in[i].aepos-in[i].abpos),30));
		if ( RHO.f == maxinput && score > RHO.H[0].score ) RHO.popvoid();
		if ( RHO.f < maxinput )
		{
			Entry e; e.score = score; e.block = curblock; e.entry = entry; e.idx = i;
			RHO.push(e);
		}
		++entry; lastblock = curblock;
	}
	std::sort(RHO.H.begin(),RHO.H.begin()+RHO.f,[](Entry const & A, Entry const & B){ return A.block != B.block ? A.block < B.block : A.entry < B.entry; });
	uint64_t o = 0, f = RHO.f;
	while ( f && RHO.H[f-1].block == lastblock ) { out[o++] = in[RHO.H[f-1].idx]; --f; }
	for ( uint64_t i = 0; i < f; ++i ) out[o++] = in[RHO.H[i].idx];
	std::sort(out,out+o,[](dacc_overlap const & A, dacc_overlap const & B){ return A.abpos < B.abpos; });
	*nout = o;
	return DACC_OK;
}
