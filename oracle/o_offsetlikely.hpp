/*
 * ORACLE (test infrastructure, not product code).
 *
 * Restatement of the indel position model:
 *   DotProduct            src/DotProduct.hpp:29-137
 *   OffsetLikely::setup   src/OffsetLikely.hpp:59-99
 *   computeOffsetLikely   src/ComputeOffsetLikely.hpp:26-134
 *   KmerLimit             src/DebruijnGraph.hpp:28-75
 *
 * Third-party arithmetic absent from /root/reference (libmaus2 >= 2.0.352, configure.ac:163):
 *   math::Convolution::convolutionFFTRef  -> exact direct linear convolution in double,
 *                                            out[n] = sum_i x[i]*y[n-i], i ascending
 *   math::Binom::binomVector(p,n,512)     -> binomial pmf evaluated in __float128
 *                                            (113-bit) and rounded once to double
 *   math::Binom::binomRowUpperLimit(p,n,q)-> smallest c with CDF(c) >= q, in __float128
 * These are value-level "parity unpinned" (SURVEY.md 8c); thresholds 1e-7/1e-5/1e-3 can flip
 * on last-ulp differences against the real FFT path.
 *
 * NOTE on a reference defect: DotProduct::computeShifted() (DotProduct.hpp:54-60) is never
 * called anywhere in reference v0.0.14, yet getKmerPositionWeight reads DP.VS[...]
 * (DebruijnGraph.hpp:3852-3855) -- undefined behaviour on an empty vector.  The evident
 * intent (fixed-point copy of DPnormSquare, VS[i] = uint64(2^32 * V[i])) is what we implement.
 */
#ifndef ORACLE_OFFSETLIKELY_HPP
#define ORACLE_OFFSETLIKELY_HPP
#include <vector>
#include <cstdint>
#include <cmath>
#include <utility>
#include <algorithm>
#include <limits>
#include "o_align.hpp"   // variant()

namespace oracle {

struct DotProduct
{
	uint64_t firstsign;
	std::vector<double> V;
	std::vector<uint64_t> VS;

	DotProduct() : firstsign(0) {}
	DotProduct(uint64_t const f, std::vector<double> const & v) : firstsign(f), V(v) {}

	static double getMult() { return 4294967296.0; }

	// DotProduct.hpp:54-60
	void computeShifted()
	{
		VS.resize(V.size());
		double const mult = getMult();
		for ( uint64_t i = 0; i < V.size(); ++i )
			VS[i] = static_cast<uint64_t>(mult * V[i]);
	}
	uint64_t size() const { return firstsign + V.size(); }
	// DotProduct.hpp:72-83
	double operator[](uint64_t const i) const
	{
		if ( i < firstsign ) return 0.0;
		uint64_t const j = i-firstsign;
		return j < V.size() ? V[j] : 0.0;
	}
	// DotProduct.hpp:85-94
	void normaliseValue(uint64_t const i, double const div)
	{
		if ( i >= firstsign )
		{
			uint64_t const j = i-firstsign;
			if ( j < V.size() )
				V[j] /= div;
		}
	}
	// DotProduct.hpp:97-116
	double dotproduct(double const * O, uint64_t const Os) const
	{
		double s = 0;
		for ( uint64_t i = 0; i < V.size(); ++i )
		{
			uint64_t const j = firstsign + i;
			if ( j < Os )
				s += V[i] * O[j];
			else
				break;
		}
		return s;
	}
	// DotProduct.hpp:124-133
	void normalise()
	{
		double s = 0.0;
		for ( uint64_t i = 0; i < V.size(); ++i )
			s += V[i]*V[i];
		double const c = std::sqrt(1.0/s);
		for ( uint64_t i = 0; i < V.size(); ++i )
			V[i] *= c;
	}
};

struct OffsetLikely
{
	std::vector<DotProduct> DP;
	std::vector<double> dsum;
	std::vector<DotProduct> DPnorm;
	std::vector< std::pair<uint64_t,uint64_t> > Vsupport;
	std::vector<DotProduct> DPnormSquare;

	// OffsetLikely.hpp:37-45
	uint64_t getSupportLow(int64_t const i) const
	{
		return (i < static_cast<int64_t>(Vsupport.size())) ? Vsupport[i].first : DPnorm.size();
	}
	uint64_t getSupportHigh(int64_t const i) const
	{
		return (i < static_cast<int64_t>(Vsupport.size())) ? Vsupport[i].second : DPnorm.size();
	}
	uint64_t size() const { return DP.size(); }

	// OffsetLikely.hpp:59-99
	void setup()
	{
		dsum.resize(0);
		uint64_t maxsize = 0;
		for ( uint64_t i = 0; i < size(); ++i )
			maxsize = std::max(maxsize,DP[i].size());
		for ( uint64_t i = 0; i < maxsize; ++i )
		{
			double sum = 0.0;
			for ( uint64_t j = 0; j < size(); ++j )
				sum += DP[j][i];
			dsum.push_back(sum);
		}
		DPnorm = DP;
		for ( uint64_t i = 0; i < DPnorm.size(); ++i )
			for ( uint64_t j = 0; j < maxsize; ++j )
				DPnorm[i].normaliseValue(j,dsum[j]);
		uint64_t j = 0, k = 0;
		for ( uint64_t i = 0; i < maxsize; ++i )
		{
			while ( j < DPnorm.size() && i >= DPnorm[j].firstsign + DPnorm[j].V.size() )
				++j;
			while ( k < DPnorm.size() && DPnorm[k].firstsign <= i )
				++k;
			Vsupport.push_back(std::pair<uint64_t,uint64_t>(j,k));
		}
		DPnormSquare = DP;
		for ( uint64_t i = 0; i < DPnormSquare.size(); ++i )
		{
			DPnormSquare[i].normalise();
			DPnormSquare[i].computeShifted(); // intended by DebruijnGraph.hpp:3852 (see header note)
		}
	}
};

// exact direct convolution (stands in for libmaus2 convolutionFFTRef)
inline std::vector<double> convolve(std::vector<double> const & x, std::vector<double> const & y)
{
	if ( x.empty() || y.empty() ) return std::vector<double>();
	std::vector<double> r(x.size()+y.size()-1);
	for ( uint64_t n = 0; n < r.size(); ++n )
	{
		uint64_t const ilow = (n >= y.size()-1) ? (n-(y.size()-1)) : 0;
		uint64_t const ihigh = std::min<uint64_t>(n,x.size()-1);
		if ( variant().conv == 1 )
		{
			long double s = 0.0L;
			for ( uint64_t i = ilow; i <= ihigh; ++i ) s += static_cast<long double>(x[i])*static_cast<long double>(y[n-i]);
			r[n] = static_cast<double>(s);
		}
		else if ( variant().conv == 2 )
		{
			double s = 0.0;
			for ( uint64_t i = ihigh+1; i-- > ilow; ) s += x[i]*y[n-i];
			r[n] = s;
		}
		else
		{
			double s = 0.0;
			for ( uint64_t i = ilow; i <= ihigh; ++i )
				s += x[i]*y[n-i];
			r[n] = s;
		}
	}
	return r;
}

// binomial pmf C(n,d) p^d (1-p)^(n-d), d = 0..n (stands in for Binom::binomVector(p,n,512))
inline std::vector<double> binomVector(double const p, uint64_t const n)
{
	std::vector<double> V(n+1);
	__float128 const pp = p;
	__float128 const qq = static_cast<__float128>(1) - pp;
	for ( uint64_t d = 0; d <= n; ++d )
	{
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
		__float128 const v = (c * pw) * qw;
		V[d] = static_cast<double>(v);
	}
	return V;
}

// smallest c in [0,n] with P(X<=c) >= lim, X ~ Bin(n,p) (stands in for Binom::binomRowUpperLimit)
inline uint64_t binomRowUpperLimit(double const p, uint64_t const n, double const lim)
{
	__float128 const pp = p;
	__float128 const qq = static_cast<__float128>(1) - pp;
	__float128 const l = lim;
	__float128 pmf = 1;
	for ( uint64_t i = 0; i < n; ++i ) pmf = pmf * qq;
	__float128 cum = 0;
	for ( uint64_t c = 0; c <= n; ++c )
	{
		cum = cum + pmf;
		if ( cum >= l )
			return c;
		// pmf_{c+1} = pmf_c * (n-c)/(c+1) * p/q
		pmf = pmf * static_cast<__float128>(n-c);
		pmf = pmf / static_cast<__float128>(c+1);
		pmf = pmf * pp;
		pmf = pmf / qq;
	}
	return n;
}

// ComputeOffsetLikely.hpp:26-134
inline OffsetLikely computeOffsetLikely(uint64_t const maxl, double const p_i, double const p_d)
{
	OffsetLikely VD;
	double const q_i = 1.0 - p_i;
	double f_i = q_i;
	std::vector<double> P_I;
	while ( f_i >= 1e-7 )
	{
		P_I.push_back(f_i);
		f_i *= p_i;
	}
	std::vector<double> C_I(1,1.0);
	for ( uint64_t l = 0; l <= maxl; ++l )
	{
		C_I = convolve(C_I,P_I);
		std::vector<double> V_D = binomVector(p_d,l);
		std::vector<double> V_I(V_D.size()-1+C_I.size());
		std::copy(C_I.begin(),C_I.end(),V_I.begin()+(V_D.size()-1));
		std::reverse(V_D.begin(),V_D.end());
		std::vector<double> const F_I = convolve(V_D,V_I);
		bool signfound = false;
		int64_t firstsign = std::numeric_limits<int64_t>::min();
		std::vector<double> VP;
		for ( uint64_t j = 0; j < F_I.size(); ++j )
			if ( F_I[j] >= 1e-5 )
			{
				if ( ! signfound )
				{
					signfound = true;
					firstsign = static_cast<int64_t>(j)-static_cast<int64_t>(l);
				}
				uint64_t const offset = static_cast<int64_t>(j)-static_cast<int64_t>(l)-firstsign;
				while ( !(offset < VP.size()) )
					VP.push_back(0);
				VP[offset] = F_I[j];
			}
		VD.DP.push_back(DotProduct(firstsign,VP));
	}
	VD.setup();
	return VD;
}

// DebruijnGraph.hpp:28-75
struct KmerLimit
{
	double p_k;
	std::vector<uint64_t> Vlim;
	KmerLimit() : p_k(0) {}
	KmerLimit(double const rp_k, uint64_t const preload) : p_k(rp_k)
	{
		for ( uint64_t i = 0; i < preload; ++i )
			getLimit(i);
	}
	double getLimit(uint64_t const i)
	{
		if ( p_k )
		{
			while ( !(i < Vlim.size()) )
			{
				int64_t v = static_cast<int64_t>(binomRowUpperLimit(p_k,Vlim.size(),0.99)) + variant().klim_delta;
				if ( v < 0 ) v = 0;
				Vlim.push_back(static_cast<uint64_t>(v));
			}
			return Vlim[i];
		}
		else
			return 0;
	}
};

}
#endif
