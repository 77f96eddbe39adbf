/*
 * ORACLE SUPPORT (test infrastructure, not product code; nothing under daccord_amd/ includes, links or loads this).
 *
 * A stand-in for the handful of libmaus2 symbols that the reference's hot-path headers use, so that the UNMODIFIED
 * reference sources under /root/reference/src (HandleContext.hpp, DebruijnGraph.hpp, OffsetLikely.hpp, DotProduct.hpp,
 * ComputeOffsetLikely.hpp, Node.hpp, Links.hpp, ...) can be compiled where they lie (oracle/ref_shim/build.sh ->
 * oracle/_ref/libdaccord_ref.so) and run against oracle/'s restatement on the same inputs (tests/test_oracle_vs_ref.py).
 * libmaus2 itself (>= 2.0.352, configure.ac:163) is not in /root/reference and not on this system.
 *
 * What this pins and what it does not: the ~9 000 reference lines of window schedule, graph construction, traversal, candidate
 * scoring and pile vote run as written, so a restatement error in oracle/ shows up as a diff.  The libmaus2 primitives below
 * are OUR definitions (the same choices oracle/o_heap.hpp, o_align.hpp and o_offsetlikely.hpp document, one per function):
 * bounded heap sift order, aligner traceback priority, direct convolution, binomial terms in __float128.  Those stay
 * "parity unpinned" (DESIGN.md section 6 measures how much of the output depends on each).
 *
 * Every class below names the libmaus2 header it stands in for.  Members that only unused variants of the reference
 * headers touch (the Overlap-struct twin of HandleContext::operator(), the window-parallel variant, HANDLE_DEBUG code)
 * are declarations with trivial bodies: they must compile, they never run.
 */
#ifndef DACC_REF_SHIM_LIBMAUS2_HPP
#define DACC_REF_SHIM_LIBMAUS2_HPP

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#if defined(_OPENMP)
#include <omp.h>
#endif

#define UNIQUE_PTR_MOVE(x) std::move(x)
// libmaus2/types/types.hpp: branch prediction hints
#if !defined(expect_true)
#define expect_true(x) __builtin_expect(!!(x),1)
#define expect_false(x) __builtin_expect(!!(x),0)
#endif
#define LIBMAUS2_HAVE_SHIM 1

namespace libmaus2 {

// ---------------------------------------------------------------- libmaus2/util/unique_ptr.hpp, shared_ptr.hpp
namespace util {
template<typename T> struct unique_ptr { typedef std::unique_ptr<T> type; };
template<typename T> struct shared_ptr { typedef std::shared_ptr<T> type; };
}

// ---------------------------------------------------------------- libmaus2/exception/LibMausException.hpp
namespace exception {
struct LibMausException : public std::exception
{
	std::shared_ptr<std::ostringstream> postr;
	std::string s;
	LibMausException() : postr(new std::ostringstream) {}
	~LibMausException() throw() {}
	std::ostream & getStream() { return *postr; }
	void finish(bool = true) { s = postr->str(); }
	char const * what() const throw() { return s.c_str(); }
};
}

// ---------------------------------------------------------------- libmaus2/autoarray/AutoArray.hpp
namespace autoarray {
enum alloc_type { alloc_type_cxx = 0, alloc_type_c = 1, alloc_type_memalign_cacheline = 2, alloc_type_memalign_pagesize = 3 };

// An owning array; like libmaus2's, a copy TAKES the storage of its source (auto_ptr style), which the reference relies on
// in assignments from temporaries (HandleContext.hpp:1736-1737).  Elements are value-initialised when `erase` is true.
template<typename N, alloc_type atype = alloc_type_cxx>
struct AutoArray
{
	typedef N value_type;
	typedef AutoArray<N,atype> this_type;
	typedef typename ::libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;
	typedef typename ::libmaus2::util::shared_ptr<this_type>::type shared_ptr_type;
	typedef N * iterator;
	typedef N const * const_iterator;

	mutable N * array;
	mutable uint64_t n;

	AutoArray() : array(0), n(0) {}
	AutoArray(uint64_t const rn, bool const erase = true) : array(rn ? new N[rn] : 0), n(rn)
	{
		if ( erase ) for ( uint64_t i = 0; i < n; ++i ) array[i] = N();
	}
	AutoArray(AutoArray const & o) : array(o.array), n(o.n) { o.array = 0; o.n = 0; }
	AutoArray & operator=(AutoArray const & o)
	{
		if ( this != &o ) { delete [] array; array = o.array; n = o.n; o.array = 0; o.n = 0; }
		return *this;
	}
	~AutoArray() { delete [] array; }

	uint64_t size() const { return n; }
	uint64_t getN() const { return n; }
	bool empty() const { return n == 0; }
	uint64_t byteSize() const { return n * sizeof(N) + sizeof(*this); }
	N * get() { return array; }
	N const * get() const { return array; }
	N * begin() { return array; }
	N const * begin() const { return array; }
	N * end() { return array + n; }
	N const * end() const { return array + n; }
	N & operator[](uint64_t const i) { return array[i]; }
	N const & operator[](uint64_t const i) const { return array[i]; }
	N & at(uint64_t const i) { if ( i >= n ) throw std::out_of_range("AutoArray::at"); return array[i]; }
	void release() { delete [] array; array = 0; n = 0; }
	void swap(AutoArray & o) { std::swap(array,o.array); std::swap(n,o.n); }
	// keeps the first min(size(),rn) elements; new elements are value-initialised
	void resize(uint64_t const rn)
	{
		N * na = rn ? new N[rn] : 0;
		uint64_t const c = std::min(n,rn);
		for ( uint64_t i = 0; i < c; ++i ) na[i] = array[i];
		for ( uint64_t i = c; i < rn; ++i ) na[i] = N();
		delete [] array; array = na; n = rn;
	}
	void ensureSize(uint64_t const rn) { if ( n < rn ) resize(rn); }
	void bump() { resize(n ? 2*n : 1); }
	// append at o, doubling the array when it is full
	void push(uint64_t & o, N const & v)
	{
		if ( o == n ) bump();
		array[o++] = v;
	}
	AutoArray clone() const { AutoArray C(n,false); for ( uint64_t i = 0; i < n; ++i ) C.array[i] = array[i]; return C; }
};
}

// ---------------------------------------------------------------- libmaus2/math: numbits.hpp, lowbits.hpp, gpow, binom.hpp, GmpFloat.hpp, Convolution.hpp
namespace math {
inline unsigned int numbits(uint64_t v) { unsigned int c = 0; while ( v ) { ++c; v >>= 1; } return c; }
inline unsigned int numbits(uint32_t v) { return numbits(static_cast<uint64_t>(v)); }
inline uint64_t lowbits(unsigned int const b) { return b >= 64 ? ~0ull : ((1ull << b) - 1ull); }
// generic power by squaring (libmaus2/math/gpow.hpp, as recalled)
template<typename N> N gpow(N b, uint64_t e)
{
	N r = N(1);
	while ( e ) { if ( e & 1 ) r = r * b; b = b * b; e >>= 1; }
	return r;
}

// stands in for the 512 bit GMP float of Binom::binomVector: 113 bit __float128, rounded once to double
struct GmpFloat
{
	__float128 v;
	GmpFloat(double const d = 0.0, unsigned int = 64) : v(d) {}
	explicit GmpFloat(__float128 const q, int) : v(q) {}
	operator double() const { return static_cast<double>(v); }
	GmpFloat operator*(GmpFloat const & o) const { return GmpFloat(v*o.v,0); }
	GmpFloat operator+(GmpFloat const & o) const { return GmpFloat(v+o.v,0); }
	GmpFloat operator-(GmpFloat const & o) const { return GmpFloat(v-o.v,0); }
	GmpFloat operator/(GmpFloat const & o) const { return GmpFloat(v/o.v,0); }
	GmpFloat & operator*=(GmpFloat const & o) { v *= o.v; return *this; }
	GmpFloat & operator+=(GmpFloat const & o) { v += o.v; return *this; }
	bool operator<(GmpFloat const & o) const { return v < o.v; }
	bool operator>=(GmpFloat const & o) const { return v >= o.v; }
};
inline std::ostream & operator<<(std::ostream & out, GmpFloat const & G) { return out << static_cast<double>(G); }

struct Binom
{
	// C(n,d) p^d (1-p)^(n-d), d = 0..n (the arithmetic of oracle/o_offsetlikely.hpp::binomVector, term by term)
	static __float128 term(double const p, uint64_t const n, uint64_t const d)
	{
		__float128 const pp = p;
		__float128 const qq = static_cast<__float128>(1) - pp;
		__float128 c = 1;
		for ( uint64_t i = 1; i <= d; ++i )
		{
			c = c * static_cast<__float128>(n-d+i);
			c = c / static_cast<__float128>(i);
		}
		__float128 pw = 1;
		for ( uint64_t i = 0; i < d; ++i ) pw = pw * pp;
		__float128 qw = 1;
		for ( uint64_t i = 0; i < n-d; ++i ) qw = qw * qq;
		return c * pw * qw;
	}
	static std::vector<GmpFloat> binomVector(double const p, uint64_t const n, unsigned int const /* prec */)
	{
		std::vector<GmpFloat> V(n+1);
		for ( uint64_t d = 0; d <= n; ++d ) V[d] = GmpFloat(term(p,n,d),0);
		return V;
	}
	// smallest c with P(X <= c) >= lim for X ~ B(n,p)
	static uint64_t binomRowUpperLimit(double const p, uint64_t const n, double const lim)
	{
		__float128 s = 0;
		for ( uint64_t c = 0; c <= n; ++c )
		{
			s += term(p,n,c);
			if ( s >= static_cast<__float128>(lim) ) return c;
		}
		return n;
	}
	static uint64_t binomRowUpperGmpFloatLimit(double const p, uint64_t const n, unsigned int const, double const lim) { return binomRowUpperLimit(p,n,lim); }
};

struct Convolution
{
	// exact direct linear convolution in double, out[n] = sum_i x[i]*y[n-i], i ascending (oracle/o_offsetlikely.hpp::convolve)
	static std::vector<double> direct(std::vector<double> const & x, std::vector<double> const & y)
	{
		if ( x.empty() || y.empty() ) return std::vector<double>();
		std::vector<double> r(x.size()+y.size()-1);
		for ( uint64_t n = 0; n < r.size(); ++n )
		{
			uint64_t const ilow = (n >= y.size()-1) ? (n-(y.size()-1)) : 0;
			uint64_t const ihigh = std::min<uint64_t>(n,x.size()-1);
			double s = 0.0;
			for ( uint64_t i = ilow; i <= ihigh; ++i ) s += x[i]*y[n-i];
			r[n] = s;
		}
		return r;
	}
	template<typename A, typename B>
	static std::vector<double> convolutionFFTRef(std::vector<A> const & x, std::vector<B> const & y)
	{
		std::vector<double> dx(x.size()), dy(y.size());
		for ( uint64_t i = 0; i < x.size(); ++i ) dx[i] = static_cast<double>(x[i]);
		for ( uint64_t i = 0; i < y.size(); ++i ) dy[i] = static_cast<double>(y[i]);
		return direct(dx,dy);
	}
	template<typename A, typename B>
	static std::vector<double> convolutionFFT(std::vector<A> const & x, std::vector<B> const & y) { return convolutionFFTRef(x,y); }
};
}

// ---------------------------------------------------------------- libmaus2/hashing/hash.hpp (referenced in disabled code only)
namespace hashing { struct EvaHash { static uint32_t hash2(uint32_t const *, uint32_t) { return 0; } }; }

// ---------------------------------------------------------------- libmaus2/util/PrefixSums.hpp, TempFileRemovalContainer.hpp, FiniteSizeHeap.hpp
namespace util {
struct PrefixSums
{
	// exclusive prefix sums in place, returns the total
	template<typename It> static uint64_t prefixSums(It a, It e)
	{
		uint64_t s = 0;
		for ( ; a != e; ++a ) { uint64_t const t = *a; *a = s; s += t; }
		return s;
	}
};
struct TempFileRemovalContainer { static void addTempFile(std::string const &) {} static void setup() {} };

// Bounded binary heap, top() = minimum under the comparator: the sift order of oracle/o_heap.hpp (append + sift-up with
// strict comparisons; pop = last to the root + sift-down to the smaller child, the left one when the children are equal)
template<typename _element_type, typename _comparator_type = std::less<_element_type> >
struct FiniteSizeHeap
{
	typedef _element_type element_type;
	typedef _comparator_type comparator_type;
	typedef FiniteSizeHeap<element_type,comparator_type> this_type;
	typedef typename ::libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;
	typedef typename ::libmaus2::util::shared_ptr<this_type>::type shared_ptr_type;

	::libmaus2::autoarray::AutoArray<element_type> H;
	uint64_t f;
	comparator_type comp;

	FiniteSizeHeap(uint64_t const size, comparator_type const & rcomp = comparator_type()) : H(size,false), f(0), comp(rcomp) {}

	bool empty() const { return f == 0; }
	bool full() const { return f == H.size(); }
	uint64_t getFill() const { return f; }
	uint64_t size() const { return f; }
	uint64_t capacity() const { return H.size(); }
	uint64_t byteSize() const { return H.byteSize(); }
	void clear() { f = 0; }
	element_type const & top() const { assert(f); return H[0]; }
	element_type & top() { assert(f); return H[0]; }

	void push(element_type const & e)
	{
		assert ( f < H.size() );
		uint64_t i = f++;
		H[i] = e;
		while ( i )
		{
			uint64_t const p = (i-1) >> 1;
			if ( comp(H[i],H[p]) ) { std::swap(H[i],H[p]); i = p; }
			else break;
		}
	}
	void pushBump(element_type const & e)
	{
		if ( full() ) H.resize(H.size() ? 2*H.size() : 1);
		push(e);
	}
	void ensureSize(uint64_t const n) { if ( H.size() < n ) H.resize(n); }
	void popvoid()
	{
		assert ( f );
		H[0] = H[--f];
		uint64_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			uint64_t const m = comp(H[r-1],H[r]) ? (r-1) : r;
			if ( comp(H[i],H[m]) ) return;
			std::swap(H[i],H[m]);
			i = m;
		}
		uint64_t const l = 2*i+1;
		if ( l < f && !comp(H[i],H[l]) ) std::swap(H[i],H[l]);
	}
	element_type pop() { element_type const t = H[0]; popvoid(); return t; }
	void pop(element_type & t) { t = H[0]; popvoid(); }
};
}

// ---------------------------------------------------------------- libmaus2/bitio/BitVector.hpp
namespace bitio {
struct BitVector
{
	std::vector<uint64_t> W;
	uint64_t nbits, nset;
	BitVector(uint64_t const n = 0) : W((n+63)/64,0), nbits(n), nset(0) {}
	uint64_t size() const { return nbits; }
	uint64_t byteSize() const { return W.size()*8 + sizeof(*this); }
	void ensureSize(uint64_t const n) { if ( n > nbits ) { W.resize((n+63)/64,0); nbits = n; } }
	bool get(uint64_t const i) const { return (W[i>>6] >> (i&63)) & 1; }
	void set(uint64_t const i) { if ( !get(i) ) { W[i>>6] |= (1ull<<(i&63)); ++nset; } }
	void set(uint64_t const i, bool const b) { if ( b ) set(i); else erase(i); }
	void erase(uint64_t const i) { if ( get(i) ) { W[i>>6] &= ~(1ull<<(i&63)); --nset; } }
	// number of set bits
	uint64_t getRank() const { return nset; }
	// smallest j >= i with bit j set (the caller guarantees there is one)
	uint64_t next1(uint64_t i) const
	{
		uint64_t w = i>>6;
		uint64_t cur = W[w] & (~0ull << (i&63));
		while ( !cur ) { ++w; assert ( w < W.size() ); cur = W[w]; }
		return (w<<6) + __builtin_ctzll(cur);
	}
};
}

// ---------------------------------------------------------------- libmaus2/rank/ERank222B.hpp, wavelet/WaveletTree.hpp, rmq/QuickDynamicRMQ.hpp
// The reference uses them on a PERMUTATION (ranks of the accepted reverse paths, DebruijnGraph.hpp:3744-3765), so the
// answers are unique: position of the minimum of a range, largest value <= v in a range, position of a value.
namespace rank { struct ERank222B {}; }
namespace wavelet {
template<typename rank_type, typename value_type>
struct WaveletTree
{
	typedef WaveletTree<rank_type,value_type> this_type;
	typedef typename ::libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;
	struct ProduceBitsContext {};
	std::vector<value_type> A;
	std::map<value_type,std::vector<uint64_t> > pos;
	WaveletTree(value_type const * a, uint64_t const n) { ProduceBitsContext C; init(a,n,C); }
	void init(value_type const * a, uint64_t const n, ProduceBitsContext &)
	{
		A.assign(a,a+n); pos.clear();
		for ( uint64_t i = 0; i < n; ++i ) pos[A[i]].push_back(i);
	}
	uint64_t byteSize() const { return A.size()*sizeof(value_type); }
	// range previous value: largest value <= v in A[l,r), or max uint64 if there is none
	uint64_t rpv(uint64_t const l, uint64_t const r, value_type const v) const
	{
		bool have = false; value_type best = 0;
		for ( uint64_t i = l; i < r && i < A.size(); ++i )
			if ( A[i] <= v && (!have || A[i] > best) ) { have = true; best = A[i]; }
		return have ? static_cast<uint64_t>(best) : std::numeric_limits<uint64_t>::max();
	}
	// position of the (i+1)-th occurrence of value v
	uint64_t select(value_type const v, uint64_t const i) const
	{
		typename std::map<value_type,std::vector<uint64_t> >::const_iterator it = pos.find(v);
		assert ( it != pos.end() && i < it->second.size() );
		return it->second[i];
	}
};
}
namespace rmq {
template<typename iterator>
struct QuickDynamicRMQ
{
	typedef QuickDynamicRMQ<iterator> this_type;
	typedef typename ::libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;
	iterator A; uint64_t n;
	QuickDynamicRMQ() : A(), n(0) {}
	void init(iterator a, uint64_t const rn) { A = a; n = rn; }
	uint64_t byteSize() const { return sizeof(*this); }
	// position of the (leftmost) minimum of A[l..r], both inclusive
	uint64_t operator()(uint64_t const l, uint64_t const r) const
	{
		uint64_t m = l;
		for ( uint64_t i = l+1; i <= r; ++i ) if ( A[i] < A[m] ) m = i;
		return m;
	}
	uint64_t rmq(uint64_t const l, uint64_t const r) const { return (*this)(l,r); }
	void regressionTest() const {}
};
}

// ---------------------------------------------------------------- libmaus2/timing/RealTimeClock.hpp
namespace timing {
struct RealTimeClock
{
	std::chrono::steady_clock::time_point t0;
	RealTimeClock() : t0(std::chrono::steady_clock::now()) {}
	bool start() { t0 = std::chrono::steady_clock::now(); return true; }
	double getElapsedSeconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); }
	std::string formatTime(double const s) const { std::ostringstream o; o << s << "s"; return o.str(); }
};
inline std::ostream & operator<<(std::ostream & out, RealTimeClock const & R) { return out << R.getElapsedSeconds() << "s"; }
}

// ---------------------------------------------------------------- libmaus2/parallel: locks, SynchronousCounter.hpp, LockedGrowingFreeList.hpp
namespace parallel {
struct PosixSpinLock
{
	typedef ::libmaus2::util::unique_ptr<PosixSpinLock>::type unique_ptr_type;
	std::mutex m;
	void lock() { m.lock(); }
	void unlock() { m.unlock(); }
};
struct ScopePosixSpinLock
{
	PosixSpinLock & L;
	ScopePosixSpinLock(PosixSpinLock & rL) : L(rL) { L.lock(); }
	~ScopePosixSpinLock() { L.unlock(); }
};
typedef PosixSpinLock PosixMutex;
template<typename T>
struct SynchronousCounter
{
	T v; std::mutex m;
	SynchronousCounter(T const rv = T()) : v(rv) {}
	T operator++(int) { std::lock_guard<std::mutex> g(m); return v++; }
	T operator++() { std::lock_guard<std::mutex> g(m); return ++v; }
	T operator+=(T const a) { std::lock_guard<std::mutex> g(m); v += a; return v; }
	operator T() { std::lock_guard<std::mutex> g(m); return v; }
	T get() { std::lock_guard<std::mutex> g(m); return v; }
};
template<typename _element_type, typename _allocator_type, typename _type_info_type>
struct LockedGrowingFreeList
{
	typedef _element_type element_type;
	typedef _allocator_type allocator_type;
	typedef _type_info_type type_info_type;
	typedef typename type_info_type::pointer_type pointer_type;
	std::mutex m;
	std::deque<pointer_type> F;
	allocator_type alloc;
	LockedGrowingFreeList(allocator_type ralloc = allocator_type()) : alloc(ralloc) {}
	pointer_type get()
	{
		std::lock_guard<std::mutex> g(m);
		if ( F.empty() ) return alloc();
		pointer_type p = F.back(); F.pop_back(); return p;
	}
	void put(pointer_type p) { std::lock_guard<std::mutex> g(m); F.push_back(p); }
	bool empty() { std::lock_guard<std::mutex> g(m); return F.empty(); }
	uint64_t byteSize() { return 0; }
};
}

// ---------------------------------------------------------------- libmaus2/aio: StreamLock.hpp, InputStream, OutputStreamInstance.hpp
namespace aio {
struct StreamLock { static inline ::libmaus2::parallel::PosixSpinLock cerrlock; };
struct InputStream : public std::istringstream
{
	typedef ::libmaus2::util::unique_ptr<InputStream>::type unique_ptr_type;
	InputStream() {}
};
struct OutputStreamInstance : public std::ofstream
{
	typedef ::libmaus2::util::unique_ptr<OutputStreamInstance>::type unique_ptr_type;
	OutputStreamInstance(std::string const & fn) : std::ofstream(fn.c_str()) {}
};
}

// ---------------------------------------------------------------- libmaus2/fastx/acgtnMap.hpp
namespace fastx {
inline uint8_t mapChar(uint8_t const c)
{
	switch ( c ) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline uint8_t remapChar(uint8_t const c) { static char const M[] = "ACGTN"; return M[c < 4 ? c : 4]; }
inline char invertUnmapped(char const c)
{
	switch ( c ) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}
// libmaus2/fastx/KmerRepeatDetector.hpp (the estimator, daccord.cpp:384, :545): detect = some q-mer occurs twice in the string
// (the definition of oracle/o_eprof.hpp::kmerRepeatDetect)
struct KmerRepeatDetector
{
	unsigned int q;
	KmerRepeatDetector(unsigned int const rq) : q(rq) {}
	template<typename it> bool detect(it p, uint64_t const n)
	{
		std::set<std::string> S;
		for ( uint64_t i = 0; i+q <= n; ++i )
			if ( !S.insert(std::string(p+i,p+i+q)).second ) return true;
		return false;
	}
	void printRepeats() const {}
};
inline std::string reverseComplementUnmapped(std::string const & s)
{
	std::string r(s.rbegin(),s.rend());
	for ( uint64_t i = 0; i < r.size(); ++i ) r[i] = invertUnmapped(r[i]);
	return r;
}
}

// ---------------------------------------------------------------- libmaus2/lcs
namespace lcs {
struct BaseConstants
{
	enum step_type { STEP_MATCH, STEP_MISMATCH, STEP_INS, STEP_DEL, STEP_RESET };
};
inline std::ostream & operator<<(std::ostream & out, BaseConstants::step_type const s)
{
	switch ( s ) { case BaseConstants::STEP_MATCH: return out << "+"; case BaseConstants::STEP_MISMATCH: return out << "-"; case BaseConstants::STEP_INS: return out << "I"; case BaseConstants::STEP_DEL: return out << "D"; default: return out << "R"; }
}

struct AlignmentStatistics
{
	uint64_t matches, mismatches, insertions, deletions;
	AlignmentStatistics() : matches(0), mismatches(0), insertions(0), deletions(0) {}
	AlignmentStatistics(uint64_t const a, uint64_t const b, uint64_t const c, uint64_t const d) : matches(a), mismatches(b), insertions(c), deletions(d) {}
	double getErrorRate() const { uint64_t const t = matches+mismatches+insertions+deletions; return t ? static_cast<double>(mismatches+insertions+deletions)/t : 0.0; }
	uint64_t getEditDistance() const { return mismatches+insertions+deletions; }
	AlignmentStatistics & operator+=(AlignmentStatistics const & o) { matches += o.matches; mismatches += o.mismatches; insertions += o.insertions; deletions += o.deletions; return *this; }
};
inline std::ostream & operator<<(std::ostream & out, AlignmentStatistics const & A)
{
	return out << "AlignmentStatistics(matches=" << A.matches << ",mismatches=" << A.mismatches << ",insertions=" << A.insertions << ",deletions=" << A.deletions << ")";
}

// Edit script container: the script of an alignment occupies [ta,te) at the END of the buffer.
// MATCH / MISMATCH consume a symbol of a and of b, DEL a symbol of a only, INS a symbol of b only
// (the reference's use: HandleContext.hpp:2446-2491).
struct AlignmentTraceContainer : public BaseConstants
{
	typedef AlignmentTraceContainer this_type;
	typedef ::libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;
	typedef ::libmaus2::util::shared_ptr<this_type>::type shared_ptr_type;

	::libmaus2::autoarray::AutoArray<step_type> trace;
	step_type * te;
	step_type * ta;

	AlignmentTraceContainer(uint64_t const tracelen = 0) : trace(tracelen,false), te(trace.end()), ta(te) {}
	AlignmentTraceContainer(AlignmentTraceContainer const & o) : trace(o.trace.clone()), te(trace.end()), ta(te - (o.te-o.ta)) {}
	AlignmentTraceContainer & operator=(AlignmentTraceContainer const & o)
	{
		if ( this != &o ) { trace = o.trace.clone(); te = trace.end(); ta = te - (o.te-o.ta); }
		return *this;
	}
	uint64_t capacity() const { return trace.size(); }
	uint64_t getTraceLength() const { return te-ta; }
	void resize(uint64_t const n) { trace = ::libmaus2::autoarray::AutoArray<step_type>(n,false); te = trace.end(); ta = te; }
	void reset() { ta = te; }
	// replace the content by the steps [a,e)
	template<typename It> void assign(It a, It e)
	{
		uint64_t const n = e-a;
		if ( trace.size() < n ) resize(n);
		te = trace.end(); ta = te - n;
		std::copy(a,e,ta);
	}

	// consume steps until n symbols of a are used; returns (symbols of a used, steps consumed); stops right after the
	// n-th a-consuming step (oracle/o_align.hpp::advanceA)
	static std::pair<uint64_t,uint64_t> advanceA(step_type const * ta, step_type const * te, uint64_t const n)
	{
		step_type const * tc = ta; uint64_t c = 0;
		while ( tc != te && c < n )
			switch ( *(tc++) ) { case STEP_MATCH: case STEP_MISMATCH: case STEP_DEL: ++c; break; default: break; }
		return std::pair<uint64_t,uint64_t>(c,tc-ta);
	}
	static std::pair<uint64_t,uint64_t> advanceB(step_type const * ta, step_type const * te, uint64_t const n)
	{
		step_type const * tc = ta; uint64_t c = 0;
		while ( tc != te && c < n )
			switch ( *(tc++) ) { case STEP_MATCH: case STEP_MISMATCH: case STEP_INS: ++c; break; default: break; }
		return std::pair<uint64_t,uint64_t>(c,tc-ta);
	}
	std::pair<uint64_t,uint64_t> advanceA(uint64_t const n) const { return advanceA(ta,te,n); }
	std::pair<uint64_t,uint64_t> advanceB(uint64_t const n) const { return advanceB(ta,te,n); }
	// (symbols of a, symbols of b) the steps [ta,te) consume
	static std::pair<uint64_t,uint64_t> getStringLengthUsed(step_type const * ta, step_type const * te)
	{
		uint64_t ua = 0, ub = 0;
		for ( ; ta != te; ++ta )
			switch ( *ta ) { case STEP_MATCH: case STEP_MISMATCH: ++ua; ++ub; break; case STEP_DEL: ++ua; break; case STEP_INS: ++ub; break; default: break; }
		return std::pair<uint64_t,uint64_t>(ua,ub);
	}
	std::pair<uint64_t,uint64_t> getStringLengthUsed() const { return getStringLengthUsed(ta,te); }
	static AlignmentStatistics getAlignmentStatistics(step_type const * ta, step_type const * te)
	{
		AlignmentStatistics S;
		for ( ; ta != te; ++ta )
			switch ( *ta ) { case STEP_MATCH: ++S.matches; break; case STEP_MISMATCH: ++S.mismatches; break; case STEP_INS: ++S.insertions; break; case STEP_DEL: ++S.deletions; break; default: break; }
		return S;
	}
	AlignmentStatistics getAlignmentStatistics() const { return getAlignmentStatistics(ta,te); }
	static uint64_t getNumErrors(step_type const * ta, step_type const * te) { return getAlignmentStatistics(ta,te).getEditDistance(); }
};

struct AlignmentPrint
{
	// (diagnostics only: the reference prints alignments under verbosity / debug switches)
	template<typename ita, typename itb, typename itt>
	static std::ostream & printAlignmentLines(std::ostream & out, ita a, uint64_t const na, itb b, uint64_t const nb, uint64_t const /* cols */, itt ta, itt te)
	{
		std::string la, lb; uint64_t ia = 0, ib = 0;
		for ( ; ta != te; ++ta )
			switch ( *ta )
			{
				case BaseConstants::STEP_MATCH: case BaseConstants::STEP_MISMATCH: la += (ia < na ? static_cast<char>(a[ia]) : '?'); lb += (ib < nb ? static_cast<char>(b[ib]) : '?'); ++ia; ++ib; break;
				case BaseConstants::STEP_DEL: la += (ia < na ? static_cast<char>(a[ia]) : '?'); lb += '-'; ++ia; break;
				case BaseConstants::STEP_INS: la += '-'; lb += (ib < nb ? static_cast<char>(b[ib]) : '?'); ++ib; break;
				default: break;
			}
		return out << la << '\n' << lb << '\n';
	}
	template<typename ita, typename itb, typename itt, typename mapper>
	static std::ostream & printAlignmentLines(std::ostream & out, ita a, uint64_t const na, itb b, uint64_t const nb, uint64_t const cols, itt ta, itt te, mapper) { return printAlignmentLines(out,a,na,b,nb,cols,ta,te); }
};

// Global unit-cost aligner with traceback (stands in for the Aligner the factory hands out: y256_8 / x128_8 / NP,
// DebruijnGraphBase.hpp:26-41).  Traceback from (m,n) taking the first admissible move of diagonal > up (DEL, a only) >
// left (INS, b only): the definition of oracle/o_align.hpp, its default variant.
struct Aligner
{
	typedef ::libmaus2::util::unique_ptr<Aligner>::type unique_ptr_type;
	virtual ~Aligner() {}
	virtual void align(uint8_t const * a, size_t const l_a, uint8_t const * b, size_t const l_b) = 0;
	virtual AlignmentTraceContainer const & getTraceContainer() const = 0;
};
struct NP : public Aligner
{
	AlignmentTraceContainer ATC;
	std::vector<uint16_t> D;
	std::vector<BaseConstants::step_type> T;
	int order;      // traceback priority (0 = the default; see oracle/o_align.hpp ORACLE_TB_*); set by the harness only
	NP() : order(0) {}
	uint64_t np(uint8_t const * a, uint8_t const * ae, uint8_t const * b, uint8_t const * be) { align(a,ae-a,b,be-b); return ATC.getAlignmentStatistics().getEditDistance(); }
	void align(uint8_t const * a, size_t const m, uint8_t const * b, size_t const n)
	{
		uint64_t const W = n+1;
		D.resize((m+1)*W);
		for ( uint64_t j = 0; j <= n; ++j ) D[j] = j;
		for ( uint64_t i = 1; i <= m; ++i )
		{
			uint16_t * row = &D[i*W]; uint16_t const * prow = &D[(i-1)*W];
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
		T.clear();
		static int const ORD[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {2,0,1}, {1,2,0}, {2,1,0} };
		int const * const ord = ORD[(order >= 0 && order < 6) ? order : 0];
		uint64_t i = m, j = n;
		while ( i || j )
		{
			uint16_t const d = D[i*W+j];
			bool const okd = i && j && D[(i-1)*W+(j-1)] + (a[i-1] != b[j-1]) == d;
			bool const oku = i && D[(i-1)*W+j] + 1 == d;
			bool const okl = j && D[i*W+(j-1)] + 1 == d;
			int mv = -1;
			for ( int q = 0; q < 3 && mv < 0; ++q )
				if ( (ord[q] == 0 && okd) || (ord[q] == 1 && oku) || (ord[q] == 2 && okl) ) mv = ord[q];
			if ( mv == 0 ) { T.push_back((a[i-1] == b[j-1]) ? BaseConstants::STEP_MATCH : BaseConstants::STEP_MISMATCH); --i; --j; }
			else if ( mv == 1 ) { T.push_back(BaseConstants::STEP_DEL); --i; }
			else { T.push_back(BaseConstants::STEP_INS); --j; }
		}
		std::reverse(T.begin(),T.end());
		ATC.assign(T.begin(),T.end());
	}
	AlignmentTraceContainer const & getTraceContainer() const { return ATC; }
};
struct AlignerFactory
{
	enum aligner_type { libmaus2_lcs_AlignerFactory_x128_8, libmaus2_lcs_AlignerFactory_x128_16, libmaus2_lcs_AlignerFactory_y256_8, libmaus2_lcs_AlignerFactory_y256_16, libmaus2_lcs_AlignerFactory_NP, libmaus2_lcs_AlignerFactory_Dalign };
	static std::set<aligner_type> getSupportedAligners() { std::set<aligner_type> S; S.insert(libmaus2_lcs_AlignerFactory_NP); return S; }
	static Aligner::unique_ptr_type construct(aligner_type const) { return Aligner::unique_ptr_type(new NP); }
};

// edit distance of one query against many strings (unique by mathematics: any correct implementation agrees)
struct AlignmentOneAgainstManyInterface
{
	typedef ::libmaus2::util::unique_ptr<AlignmentOneAgainstManyInterface>::type unique_ptr_type;
	virtual ~AlignmentOneAgainstManyInterface() {}
	virtual void process(uint8_t const * qa, uint8_t const * qe, std::pair<uint8_t const *,uint64_t> const * MA, uint64_t const MAo, ::libmaus2::autoarray::AutoArray<uint64_t> & E) = 0;
};
struct AlignmentOneAgainstManyGeneric : public AlignmentOneAgainstManyInterface
{
	std::vector<uint32_t> tmp;
	void process(uint8_t const * qa, uint8_t const * qe, std::pair<uint8_t const *,uint64_t> const * MA, uint64_t const MAo, ::libmaus2::autoarray::AutoArray<uint64_t> & E)
	{
		E.ensureSize(MAo);
		uint64_t const m = qe-qa;
		for ( uint64_t s = 0; s < MAo; ++s )
		{
			uint8_t const * b = MA[s].first; uint64_t const n = MA[s].second;
			tmp.resize(n+1);
			for ( uint64_t j = 0; j <= n; ++j ) tmp[j] = j;
			for ( uint64_t i = 1; i <= m; ++i )
			{
				uint32_t diag = tmp[0]; tmp[0] = i;
				uint8_t const ai = qa[i-1];
				for ( uint64_t j = 1; j <= n; ++j )
				{
					uint32_t const nd = tmp[j];
					tmp[j] = std::min(diag + (ai != b[j-1]),std::min(tmp[j]+1,tmp[j-1]+1));
					diag = nd;
				}
			}
			E[s] = tmp[n];
		}
	}
};
struct AlignmentOneAgainstManyFactory
{
	static AlignmentOneAgainstManyInterface::unique_ptr_type uconstruct() { return AlignmentOneAgainstManyInterface::unique_ptr_type(new AlignmentOneAgainstManyGeneric); }
};

// (used by HANDLE_DEBUG / verbose diagnostics of the reference only: never run by the harness)
struct NNPAlignResult { uint64_t abpos, aepos, bbpos, bepos, dif; NNPAlignResult() : abpos(0), aepos(0), bbpos(0), bepos(0), dif(0) {} double getErrorRate() const { return 0.0; } };
inline std::ostream & operator<<(std::ostream & out, NNPAlignResult const &) { return out << "NNPAlignResult"; }
struct NNPTraceContainer
{
	void computeTrace(AlignmentTraceContainer & ATC) const { ATC.reset(); }
	template<typename it> static void computeTrace(it, it, AlignmentTraceContainer & ATC) { ATC.reset(); }
};
struct NNP
{
	template<typename it> NNPAlignResult align(it, it, uint64_t, it, it, uint64_t, NNPTraceContainer &, bool = true) { return NNPAlignResult(); }
};
struct SuffixArrayLCS
{
	struct LCSResult { uint32_t maxlcp, maxpos_a, maxpos_b; LCSResult() : maxlcp(0), maxpos_a(0), maxpos_b(0) {} };
	static LCSResult lcsmin(std::string const &, std::string const &) { return LCSResult(); }
};
}

// ---------------------------------------------------------------- libmaus2/bambam (HANDLE_DEBUG signatures only)
namespace bambam {
enum bam_cigar_ops { BAM_CIGAR_M };
typedef std::pair<int32_t,uint32_t> cigar_operation;
struct BamAlignment
{
	typedef ::libmaus2::util::shared_ptr<BamAlignment>::type shared_ptr_type;
	uint32_t getCigarOperations(::libmaus2::autoarray::AutoArray<cigar_operation> &) const { return 0; }
	bool isReverse() const { return false; }
	int64_t getPos() const { return 0; }
	uint64_t getFrontDel() const { return 0; }
	uint64_t getReferenceLength() const { return 0; }
	uint64_t getFrontSoftClipping() const { return 0; }
};
struct CigarStringParser
{
	template<typename it> static void cigarToTrace(it, it, ::libmaus2::lcs::AlignmentTraceContainer & ATC, bool = true) { ATC.reset(); }
};
}

// ---------------------------------------------------------------- libmaus2/util/U: number wrapper of the serialising sorters
namespace util { template<typename N> struct U { N u; U(N const ru = N()) : u(ru) {} bool operator<(U const & o) const { return u < o.u; } }; }
// ---------------------------------------------------------------- libmaus2/sorting/ParallelStableSort.hpp (window-parallel variant only),
// SerialisingSortingBufferedOutputFileArray (--deepprofileonly collects its window error rates in one: here a vector)
namespace sorting {
template<typename T>
struct SerialisingSortingBufferedOutputFileArray
{
	struct sorter_type { std::vector<T> V; void put(T const & v) { V.push_back(v); } };
};
struct ParallelStableSort
{
	template<typename iterator, typename order_type>
	static void parallelMerge(iterator aa, iterator ae, iterator ba, iterator be, iterator out, order_type order = order_type(), uint64_t const = 1)
	{
		std::merge(aa,ae,ba,be,out,order);
	}
};
}

// ---------------------------------------------------------------- libmaus2/dazzler
namespace dazzler {
namespace db {
// The read database as the harness holds it: forward read as letters at A[0,l), its reverse complement at A[l,2l)
// (DecodedReadContainer.hpp:180-192 relies on exactly this layout of DatabaseFile::decodeReadAndRC)
struct DatabaseFile
{
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen; uint64_t nreads;
	DatabaseFile() : bps(0), boff(0), rlen(0), nreads(0) {}
	::libmaus2::aio::InputStream::unique_ptr_type openBaseStream() const { return ::libmaus2::aio::InputStream::unique_ptr_type(new ::libmaus2::aio::InputStream); }
	::libmaus2::aio::InputStream::unique_ptr_type openIndexStream() const { return ::libmaus2::aio::InputStream::unique_ptr_type(new ::libmaus2::aio::InputStream); }
	size_t decodeReadAndRC(std::istream &, std::istream &, uint64_t const id, ::libmaus2::autoarray::AutoArray<char> & A) const
	{
		uint64_t const l = rlen[id];
		if ( A.size() < 2*l ) A = ::libmaus2::autoarray::AutoArray<char>(2*l,false);
		uint8_t const * p = bps + boff[id];
		for ( uint64_t i = 0; i < l; ++i ) A[i] = "ACGT"[(p[i>>2] >> (6-2*(i&3))) & 3];
		for ( uint64_t i = 0; i < l; ++i ) A[l+i] = ::libmaus2::fastx::invertUnmapped(A[l-1-i]);
		return l;
	}
	uint64_t size() const { return nreads; }
};
}
namespace align {
struct Path
{
	std::vector< std::pair<uint16_t,uint16_t> > path;
	int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
	Path() : tlen(0), diffs(0), abpos(0), bbpos(0), aepos(0), bepos(0) {}
};
struct OverlapDataInterface;
// One overlap record as DALIGNER stores it (the Overlap-struct twin of the handler and the window-parallel variant use
// it; the harness hands the used overload OverlapDataInterface objects)
struct Overlap
{
	Path path;
	uint32_t flags;
	int32_t aread, bread;
	uint64_t tag;      // (harness only: index of the record in the caller's array)
	Overlap() : flags(0), aread(0), bread(0), tag(0) {}
	bool isInverse() const { return flags & 1; }
	double getErrorRate() const { return (path.aepos > path.abpos) ? static_cast<double>(path.diffs) / (path.aepos-path.abpos) : 0.0; }
	uint64_t getNumErrors() const { return path.diffs; }
	static bool getPrimaryFlag(uint64_t const f) { return !(f & 0x40000000u); }
	bool isPrimary() const { return true; }
	// expansion of the trace points to an edit script: one global alignment per tspace block of A against the B span its
	// trace point names, block scripts appended (oracle/o_handle.hpp::computeTrace)
	void computeTrace(uint8_t const * aptr, uint8_t const * bptr, int64_t const tspace, ::libmaus2::lcs::AlignmentTraceContainer & ATC, ::libmaus2::lcs::Aligner & aligner) const
	{
		computeTracePoints(path.path.data(),path.path.size(),path.abpos,path.aepos,path.bbpos,tspace,aptr,bptr,ATC,aligner);
	}
	static void computeTracePoints(std::pair<uint16_t,uint16_t> const * tp, uint64_t const ntp, int64_t const abpos, int64_t const aepos, int64_t const bbpos, int64_t const tspace,
		uint8_t const * aptr, uint8_t const * bptr, ::libmaus2::lcs::AlignmentTraceContainer & ATC, ::libmaus2::lcs::Aligner & aligner)
	{
		std::vector< ::libmaus2::lcs::BaseConstants::step_type > S;
		int64_t a_i = (abpos/tspace)*tspace, b_i = bbpos;
		for ( uint64_t i = 0; i < ntp; ++i )
		{
			int64_t const a_i_1 = std::min<int64_t>(a_i+tspace,aepos);
			int64_t const b_i_1 = b_i + tp[i].second;
			int64_t const as = std::max<int64_t>(a_i,abpos);
			aligner.align(aptr+as,a_i_1-as,bptr+b_i,b_i_1-b_i);
			::libmaus2::lcs::AlignmentTraceContainer const & T = aligner.getTraceContainer();
			S.insert(S.end(),T.ta,T.te);
			b_i = b_i_1; a_i = a_i_1;
		}
		ATC.assign(S.begin(),S.end());
	}
	uint64_t getBBlockOffset(uint64_t const) const { return 0; }
	Overlap getSwapped(int64_t, uint8_t const *, uint64_t, uint8_t const *, uint64_t, ::libmaus2::lcs::Aligner &) const { return *this; }
};
inline std::ostream & operator<<(std::ostream & out, Overlap const & O)
{
	return out << "Overlap(aread=" << O.aread << ",bread=" << O.bread << ",flags=" << O.flags << ",[" << O.path.abpos << "," << O.path.aepos << ")x[" << O.path.bbpos << "," << O.path.bepos << "),diffs=" << O.path.diffs << ")";
}

// View of one overlap record of a pile (libmaus2 parses it out of the raw LAS bytes; here the harness fills the fields from
// the C ABI's dacc_overlap and points tp at the record's trace values)
struct OverlapDataInterface
{
	int32_t f_aread, f_bread; uint32_t f_flags; int32_t f_abpos, f_aepos, f_bbpos, f_bepos, f_diffs, f_tlen;
	void const * trace; int trace_bytes; uint64_t trace_off;
	OverlapDataInterface() : f_aread(0), f_bread(0), f_flags(0), f_abpos(0), f_aepos(0), f_bbpos(0), f_bepos(0), f_diffs(0), f_tlen(0), trace(0), trace_bytes(1), trace_off(0) {}
	int64_t aread() const { return f_aread; }
	int64_t bread() const { return f_bread; }
	int64_t abpos() const { return f_abpos; }
	int64_t aepos() const { return f_aepos; }
	int64_t bbpos() const { return f_bbpos; }
	int64_t bepos() const { return f_bepos; }
	int64_t diffs() const { return f_diffs; }
	int64_t tlen() const { return f_tlen; }
	uint64_t flags() const { return f_flags; }
	bool isInverse() const { return f_flags & 1; }
	// diffs / length of the A interval (src/daccord.cpp:2166-2178 uses the same quotient for its selection score)
	double getErrorRate() const { return static_cast<double>(f_diffs) / static_cast<double>(f_aepos-f_abpos); }
	uint64_t traceValue(uint64_t const i) const
	{
		return trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(trace)[trace_off+i] : reinterpret_cast<uint8_t const *>(trace)[trace_off+i];
	}
	void computeTrace(::libmaus2::autoarray::AutoArray<std::pair<uint16_t,uint16_t> > & Atrace, int64_t const tspace, uint8_t const * aptr, uint8_t const * bptr,
		::libmaus2::lcs::AlignmentTraceContainer & ATC, ::libmaus2::lcs::Aligner & aligner) const
	{
		uint64_t const ntp = f_tlen/2;
		Atrace.ensureSize(ntp);
		for ( uint64_t i = 0; i < ntp; ++i ) Atrace[i] = std::pair<uint16_t,uint16_t>(traceValue(2*i),traceValue(2*i+1));
		Overlap::computeTracePoints(Atrace.begin(),ntp,f_abpos,f_aepos,f_bbpos,tspace,aptr,bptr,ATC,aligner);
	}
	uint64_t getBBlockOffset(uint64_t const) const { return 0; }
	void getOverlap(Overlap & O) const
	{
		O.aread = f_aread; O.bread = f_bread; O.flags = f_flags; O.path.abpos = f_abpos; O.path.aepos = f_aepos; O.path.bbpos = f_bbpos; O.path.bepos = f_bepos;
		O.path.diffs = f_diffs; O.path.tlen = f_tlen; O.path.path.resize(f_tlen/2);
		for ( int64_t i = 0; i < f_tlen/2; ++i ) O.path.path[i] = std::pair<uint16_t,uint16_t>(traceValue(2*i),traceValue(2*i+1));
	}
};
inline std::ostream & operator<<(std::ostream & out, OverlapDataInterface const & O)
{
	return out << "OverlapData(aread=" << O.aread() << ",bread=" << O.bread() << ",flags=" << O.flags() << ",[" << O.abpos() << "," << O.aepos() << ")x[" << O.bbpos() << "," << O.bepos() << "),diffs=" << O.diffs() << ")";
}
// (window-parallel variant of the reference only)
struct BinIndexDecoder { BinIndexDecoder() {} BinIndexDecoder(std::string const &) {} };
struct LasRangeDecoder
{
	typedef ::libmaus2::util::unique_ptr<LasRangeDecoder>::type unique_ptr_type;
	LasRangeDecoder(std::string const &, BinIndexDecoder const &) {}
	void setup(uint64_t, uint64_t, uint64_t, uint64_t) {}
	bool getNext(Overlap &) { return false; }
	bool peekNext(Overlap &) { return false; }
};
}
}

}
#endif
