/*
 * ORACLE (test infrastructure, not product code).
 *
 * Bounded binary heap standing in for libmaus2::util::FiniteSizeHeap<T,Cmp>.
 * libmaus2 (github.com/gt1/libmaus2, pinned only as ">= 2.0.352" by the reference's
 * configure.ac:163) is NOT in /root/reference, so this is a restatement of its published
 * algorithm as recalled: an array-embedded binary heap whose top() is the minimum under Cmp,
 * push = append + sift-up with strict comparisons, pop = move last to root + sift-down
 * preferring the smaller child (left when the two children compare equal under "right < left"
 * being false).  Call sites in the reference that fix the semantics we need:
 *   top()==minimum under Cmp:  src/HandleContext.hpp:1969 (pops the smallest aepos),
 *                               src/DebruijnGraph.hpp:3626-3665 (drops RP.weight <= top().weight)
 *   pushBump grows the array:   src/DebruijnGraph.hpp:1841, 3671, 4943
 *   full()/empty()/clear():     src/DebruijnGraph.hpp:3626, 4851, 5063
 * Order among elements that compare EQUAL is defined by this sift order; it is "parity
 * unpinned" against real libmaus2 but the HIP path reproduces exactly
 * this algorithm, so oracle == device bit for bit.
 */
#ifndef ORACLE_HEAP_HPP
#define ORACLE_HEAP_HPP
#include <vector>
#include <functional>
#include <cstddef>
#include <cassert>
#include "o_align.hpp"   // variant()

namespace oracle {

template<typename T, typename Cmp = std::less<T> >
struct FiniteSizeHeap
{
	std::vector<T> H;
	size_t f;
	Cmp cmp;

	explicit FiniteSizeHeap(size_t n = 0, Cmp const & c = Cmp()) : H(n), f(0), cmp(c) {}

	bool empty() const { return f == 0; }
	bool full() const { return f == H.size(); }
	void clear() { f = 0; }
	T const & top() const { assert(f); return H[0]; }

	void push(T const & e)
	{
		assert(f < H.size());
		size_t i = f++;
		H[i] = e;
		while ( i )
		{
			size_t const p = (i-1) >> 1;
			if ( cmp(H[i],H[p]) || ((variant().heap_tie & 1) && !cmp(H[p],H[i])) )
			{
				std::swap(H[i],H[p]);
				i = p;
			}
			else
				break;
		}
	}

	void pushBump(T const & e)
	{
		if ( full() )
			H.resize(H.size() ? 2*H.size() : 1);
		push(e);
	}

	void popvoid()
	{
		assert(f);
		H[0] = H[--f];
		size_t i = 0;
		size_t r;
		while ( (r = 2*i+2) < f )
		{
			size_t const m = (variant().heap_tie & 2) ? (cmp(H[r],H[r-1]) ? r : (cmp(H[r-1],H[r]) ? (r-1) : r)) : (cmp(H[r-1],H[r]) ? (r-1) : r);
			if ( cmp(H[i],H[m]) )
				return;
			std::swap(H[i],H[m]);
			i = m;
		}
		size_t const l = 2*i+1;
		if ( l < f && !cmp(H[i],H[l]) )
			std::swap(H[i],H[l]);
	}

	T pop()
	{
		T const t = H[0];
		popvoid();
		return t;
	}
};

}
#endif
