/*
 * Truth-based accuracy check of corrected fragments: the measurement of the reference package's
 * `checkconsensus` tool (src/checkconsensus.cpp:730-1075, output described in README.md:406-472) for data whose
 * ground truth is known: every corrected fragment is aligned to the stretch of the true sequence its read interval
 * came from (the fragment must align completely, the truth window has free ends) and the alignment operations are
 * counted.  checkconsensus finds that stretch through a BAM alignment of the raw read; here the caller passes a window
 * of the truth around the expected place, and the alignment is banded around the expected diagonal.
 * Host code, no device part; independent of oracle/.
 */
#include <vector>
#include <cstdint>
#include <algorithm>
#include "../../include/daccord_hip.h"

extern "C" int dacc_check_fragment(uint8_t const * frag, uint64_t n, uint8_t const * ref, uint64_t m, uint64_t c0, uint64_t c1, uint64_t band,
	uint64_t * stats /* matches, mismatches, insertions (fragment only), deletions (truth only) */, uint64_t * rfrom, uint64_t * rto)
{
	// rows: fragment symbols i = 0..n, columns: truth positions j = 0..m; row i may use columns [lo(i),hi(i)] around the
	// line from (0,c0) to (n,c1); D[0][j] = 0 (free start in the truth), end = cheapest column of row n (free end)
	if ( !frag || !ref || !stats || !rfrom || !rto || c1 < c0 || c1 > m ) return DACC_EINVAL;
	uint64_t const W = 2*band+1;
	std::vector<uint16_t> D((n+1)*W,0xFFFF);
	auto center = [&](uint64_t const i) -> int64_t { return static_cast<int64_t>(c0) + (n ? static_cast<int64_t>((static_cast<__int128>(c1-c0)*i)/n) : 0); };
	auto lo = [&](uint64_t const i) -> int64_t { return center(i) - static_cast<int64_t>(band); };
	auto at = [&](uint64_t const i, int64_t const j) -> uint16_t { int64_t const q = j-lo(i); return (q < 0 || q >= static_cast<int64_t>(W) || j < 0 || j > static_cast<int64_t>(m)) ? 0xFFFF : D[i*W+q]; };
	for ( uint64_t q = 0; q < W; ++q ) { int64_t const j = lo(0)+q; if ( j >= 0 && j <= static_cast<int64_t>(m) ) D[q] = 0; }
	for ( uint64_t i = 1; i <= n; ++i )
		for ( uint64_t q = 0; q < W; ++q )
		{
			int64_t const j = lo(i)+q; if ( j < 0 || j > static_cast<int64_t>(m) ) continue;
			uint32_t v = 0xFFFF;
			uint16_t const up = at(i-1,j); if ( up != 0xFFFF ) v = up+1u;                                   // fragment symbol without partner
			if ( j > 0 ) { uint16_t const dg = at(i-1,j-1); if ( dg != 0xFFFF ) { uint32_t const c = dg + (frag[i-1] != ref[j-1] ? 1u : 0u); if ( c < v ) v = c; }
			               uint16_t const lf = at(i,j-1); if ( lf != 0xFFFF && lf+1u < v ) v = lf+1u; }
			D[i*W+q] = v > 0xFFFE ? 0xFFFE : v;
		}
	int64_t bj = -1; uint16_t best = 0xFFFF;
	for ( uint64_t q = 0; q < W; ++q ) { int64_t const j = lo(n)+q; if ( j >= 0 && j <= static_cast<int64_t>(m) && D[n*W+q] < best ) { best = D[n*W+q]; bj = j; } }
	if ( bj < 0 ) return DACC_ENOTSUP;
	stats[0] = stats[1] = stats[2] = stats[3] = 0;
	uint64_t i = n; int64_t j = bj; *rto = bj;
	while ( i )
	{
		uint16_t const d = at(i,j);
		if ( j > 0 && at(i-1,j-1) != 0xFFFF && at(i-1,j-1) + (frag[i-1] != ref[j-1] ? 1 : 0) == d ) { ++stats[frag[i-1] == ref[j-1] ? 0 : 1]; --i; --j; }
		else if ( at(i-1,j) != 0xFFFF && at(i-1,j)+1 == d ) { ++stats[2]; --i; }
		else if ( j > 0 && at(i,j-1) != 0xFFFF && at(i,j-1)+1 == d ) { ++stats[3]; --j; }
		else return DACC_ENOTSUP;     // the band was too narrow for this fragment
	}
	*rfrom = j;
	return DACC_OK;
}
