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
#include <cstdlib>

namespace oracle {

/*
 * Exposure switches (scripts/exposure_report.py): every choice this restatement had to make where
 * libmaus2 defines the behaviour can be flipped through the environment, so that the share of the output that depends
 * on the choice can be MEASURED.  Defaults (all unset / 0) are the documented definitions above; nothing in the product
 * reads these.
 *   ORACLE_TB_BLOCK / ORACLE_TB_CONS  traceback priority of the block alignments (computeTrace) / of the consensus -> A
 *                                     alignment: 0 diag>del>ins (default), 1 diag>ins>del, 2 del>diag>ins, 3 ins>diag>del,
 *                                     4 del>ins>diag, 5 ins>del>diag
 *   ORACLE_HEAP_TIE                   0 default; 1: sift-up swaps on <= (equal keys rise); 2: sift-down prefers the right
 *                                     child among equal children; 3: both
 *   ORACLE_KLIM_DELTA                 added to binomRowUpperLimit (-1, 0, +1)
 *   ORACLE_CONV                       0 double, ascending index (default); 1 long double accumulation; 2 descending index
 */
struct Variant
{
	int tb_block, tb_cons, heap_tie, klim_delta, conv;
	static int envi(char const * n) { char const * e = std::getenv(n); return e ? std::atoi(e) : 0; }
	Variant() : tb_block(envi("ORACLE_TB_BLOCK")), tb_cons(envi("ORACLE_TB_CONS")), heap_tie(envi("ORACLE_HEAP_TIE")),
		klim_delta(envi("ORACLE_KLIM_DELTA")), conv(envi("ORACLE_CONV")) {}
};
inline Variant const & variant() { static Variant const V; return V; }

enum Step : uint8_t { STEP_MATCH = 0, STEP_MISMATCH = 1, STEP_INS = 2, STEP_DEL = 3 };

struct Aligner
{
	std::vector<uint16_t> D;
	std::vector<uint8_t> trace; // forward order after align()

	// returns edit distance; trace holds the edit script
	uint64_t align(uint8_t const * a, uint64_t const m, uint8_t const * b, uint64_t const n, int const order = 0)
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
		// move priority: 0 = diagonal, 1 = up (DEL), 2 = left (INS), tried in the order the variant names
		static int const ORD[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {2,0,1}, {1,2,0}, {2,1,0} };
		int const * const ord = ORD[(order >= 0 && order < 6) ? order : 0];
		while ( i || j )
		{
			uint16_t const d = D[i*W+j];
			bool const okd = i && j && D[(i-1)*W+(j-1)] + (a[i-1] != b[j-1]) == d;
			bool const oku = i && D[(i-1)*W+j] + 1 == d;
			bool const okl = j && D[i*W+(j-1)] + 1 == d;
			int mv = -1;
			for ( int q = 0; q < 3 && mv < 0; ++q )
				if ( (ord[q] == 0 && okd) || (ord[q] == 1 && oku) || (ord[q] == 2 && okl) ) mv = ord[q];
			if ( mv == 0 )
			{
				trace.push_back( (a[i-1] == b[j-1]) ? STEP_MATCH : STEP_MISMATCH );
				--i; --j;
			}
			else if ( mv == 1 )
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
