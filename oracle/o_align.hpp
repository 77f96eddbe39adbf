/*
 * ORACLE (test infrastructure, not product code).
 *
 * Global unit-cost alignment with traceback, standing in for libmaus2::lcs::Aligner
 * (reference picks y256_8 / x128_8 / NP at run time, src/DebruijnGraphBase.hpp:26-41, so
 * the reference's own co-optimal traceback choice is machine dependent).  libmaus2 is
 * not in /root/reference: this is OUR documented definition (parity unpinned):
 *
 *   D[i][j] = edit distance of a[0,i) and b[0,j); traceback starts at (m,n) and at each
 *   cell takes the FIRST admissible move in the order
 *       1. diagonal  (D[i-1][j-1] + (a[i-1]!=b[j-1]) == D[i][j])  -> MATCH / MISMATCH
 *       2. up        (D[i-1][j]   + 1 == D[i][j])                 -> DEL  (consumes a only)
 *       3. left      (D[i][j-1]   + 1 == D[i][j])                 -> INS  (consumes b only)
 *
 * Step semantics follow the reference's use of AlignmentTraceContainer
 * (src/HandleContext.hpp:2446-2491): MATCH/MISMATCH consume a and b, DEL consumes a,
 * INS consumes b.  advanceA(n) stops immediately after the n-th a-consuming step
 * (src/HandleContext.hpp:1936-1949, 2005-2029).
 */
#ifndef ORACLE_ALIGN_HPP
#define ORACLE_ALIGN_HPP
#include <vector>
#include <cstdint>
#include <algorithm>
#include <utility>

namespace oracle {

enum Step : uint8_t { STEP_MATCH = 0, STEP_MISMATCH = 1, STEP_INS = 2, STEP_DEL = 3 };

struct Aligner
{
	std::vector<uint16_t> D;
	std::vector<uint8_t> trace; // forward order after align()

	// returns edit distance; trace holds the edit script
	uint64_t align(uint8_t const * a, uint64_t const m, uint8_t const * b, uint64_t const n)
	{
		uint64_t const W = n+1;
		D.resize((m+1)*W);
		for ( uint64_t j = 0; j <= n; ++j ) D[j] = j;
		for ( uint64_t i = 1; i <= m; ++i )
		{
			uint16_t * row = &D[i*W];
			uint16_t const * prow = &D[(i-1)*W];
			row[0] = i;
			uint8_t const ai = a[i-1];
			for ( uint64_t j = 1; j <= n; ++j )
			{
				uint16_t const diag = prow[j-1] + (ai != b[j-1]);
				uint16_t const up = prow[j] + 1;
				uint16_t const left = row[j-1] + 1;
				row[j] = std::min(diag,std::min(up,left));
			}
		}
		trace.clear();
		uint64_t i = m, j = n;
		while ( i || j )
		{
			uint16_t const d = D[i*W+j];
			if ( i && j && D[(i-1)*W+(j-1)] + (a[i-1] != b[j-1]) == d )
			{
				trace.push_back( (a[i-1] == b[j-1]) ? STEP_MATCH : STEP_MISMATCH );
				--i; --j;
			}
			else if ( i && D[(i-1)*W+j] + 1 == d )
			{
				trace.push_back(STEP_DEL);
				--i;
			}
			else
			{
				trace.push_back(STEP_INS);
				--j;
			}
		}
		std::reverse(trace.begin(),trace.end());
		return D[m*W+n];
	}
};

// edit distance only (libmaus2::lcs::AlignmentOneAgainstManyInterface::process, used at
// src/DebruijnGraph.hpp:5361) -- unique by mathematics, any correct implementation is identical
inline uint64_t editDistance(uint8_t const * a, uint64_t const m, uint8_t const * b, uint64_t const n, std::vector<uint32_t> & tmp)
{
	tmp.resize(n+1);
	for ( uint64_t j = 0; j <= n; ++j ) tmp[j] = j;
	for ( uint64_t i = 1; i <= m; ++i )
	{
		uint32_t diag = tmp[0];
		tmp[0] = i;
		uint8_t const ai = a[i-1];
		for ( uint64_t j = 1; j <= n; ++j )
		{
			uint32_t const nd = tmp[j];
			uint32_t const v = std::min( diag + (ai != b[j-1]), std::min(tmp[j]+1,tmp[j-1]+1) );
			tmp[j] = v;
			diag = nd;
		}
	}
	return tmp[n];
}

// AlignmentTraceContainer::advanceA (libmaus2, recalled): consume steps until n a-symbols used;
// returns (a-symbols consumed, steps consumed)
inline std::pair<uint64_t,uint64_t> advanceA(uint8_t const * ta, uint8_t const * te, uint64_t const n)
{
	uint8_t const * tc = ta;
	uint64_t c = 0;
	while ( tc != te && c < n )
	{
		switch ( *(tc++) )
		{
			case STEP_MATCH: case STEP_MISMATCH: case STEP_DEL: ++c; break;
			default: break;
		}
	}
	return std::pair<uint64_t,uint64_t>(c,tc-ta);
}

// AlignmentTraceContainer::getStringLengthUsed: (a used, b used)
inline std::pair<uint64_t,uint64_t> getStringLengthUsed(uint8_t const * ta, uint8_t const * te)
{
	uint64_t ua = 0, ub = 0;
	for ( ; ta != te; ++ta )
		switch ( *ta )
		{
			case STEP_MATCH: case STEP_MISMATCH: ++ua; ++ub; break;
			case STEP_DEL: ++ua; break;
			case STEP_INS: ++ub; break;
		}
	return std::pair<uint64_t,uint64_t>(ua,ub);
}

}
#endif
