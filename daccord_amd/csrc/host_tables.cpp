/*
 * Host-side model tables of the product (built once per error profile, uploaded, never
 * recomputed on the device): the indel position model OffsetLikely and the KmerLimit table.
 *
 * Follows src/ComputeOffsetLikely.hpp:26-134, src/OffsetLikely.hpp:59-99,
 * src/DotProduct.hpp:54-60,124-133 and src/DebruijnGraph.hpp:28-75 (KmerLimit), with the
 * libmaus2 arithmetic that is not in the reference tree replaced by: exact direct convolution
 * in double (summation index ascending), binomial terms in __float128 rounded once to double.
 * The tables are laid out dense and zero padded for O(1) device lookups (dev_types.hpp).
 * This is product code and deliberately independent of oracle/ (which holds its own
 * restatement used only to check this one).
 */
#include <vector>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include "host_tables.hpp"

namespace dacc {

typedef __float128 q128;

static void conv(std::vector<double> const & x, std::vector<double> const & y, std::vector<double> & r)
{
	size_t const nx = x.size(), ny = y.size();
	r.assign(nx+ny-1,0.0);
	for ( size_t n = 0; n < r.size(); ++n )
	{
		size_t const lo = (n+1 > ny) ? (n+1-ny) : 0;
		size_t const hi = std::min(n,nx-1);
		double acc = 0.0;
		for ( size_t i = lo; i <= hi; ++i )
			acc += x[i]*y[n-i];
		r[n] = acc;
	}
}

static double binomTerm(double const p, uint64_t const n, uint64_t const d)
{
	q128 const pp = p, qq = static_cast<q128>(1)-pp;
	q128 c = 1;
	for ( uint64_t i = 1; i <= d; ++i ) { c = c * static_cast<q128>(n-d+i); c = c / static_cast<q128>(i); }
	q128 pw = 1; for ( uint64_t i = 0; i < d; ++i ) pw = pw*pp;
	q128 qw = 1; for ( uint64_t i = 0; i < n-d; ++i ) qw = qw*qq;
	return static_cast<double>((c*pw)*qw);
}

static uint32_t binomUpper(double const p, uint64_t const n, double const lim)
{
	q128 const pp = p, qq = static_cast<q128>(1)-pp, l = lim;
	q128 pmf = 1;
	for ( uint64_t i = 0; i < n; ++i ) pmf = pmf*qq;
	q128 cum = 0;
	for ( uint64_t c = 0; c <= n; ++c )
	{
		cum = cum + pmf;
		if ( cum >= l ) return c;
		pmf = pmf * static_cast<q128>(n-c);
		pmf = pmf / static_cast<q128>(c+1);
		pmf = pmf * pp;
		pmf = pmf / qq;
	}
	return n;
}

void buildHostTables(HostTables & H, uint32_t const w, double const p_i, double const p_d, double const est_cor,
	uint32_t const klow, uint32_t const khigh, uint32_t const kln)
{
	// rows: distribution of the read offset for reference offsets 0..w
	uint32_t const nrows = w+1;
	std::vector<double> PI;
	for ( double f = 1.0-p_i; f >= 1e-7; f *= p_i ) PI.push_back(f);
	std::vector< std::vector<double> > V(nrows);
	std::vector<uint32_t> first(nrows,0);
	std::vector<double> CI(1,1.0), tmp, FI;
	for ( uint32_t l = 0; l < nrows; ++l )
	{
		conv(CI,PI,tmp); CI.swap(tmp);
		std::vector<double> VD(l+1);
		for ( uint32_t d = 0; d <= l; ++d ) VD[l-d] = binomTerm(p_d,l,d);   // reversed deletion vector
		std::vector<double> VI(l+CI.size(),0.0);
		std::copy(CI.begin(),CI.end(),VI.begin()+l);
		conv(VD,VI,FI);
		bool found = false; int64_t fs = 0;
		for ( size_t j = 0; j < FI.size(); ++j )
			if ( FI[j] >= 1e-5 )
			{
				int64_t const rel = static_cast<int64_t>(j)-static_cast<int64_t>(l);
				if ( !found ) { found = true; fs = rel; }
				size_t const o = rel-fs;
				if ( V[l].size() <= o ) V[l].resize(o+1,0.0);
				V[l][o] = FI[j];
			}
		first[l] = fs;
	}
	uint32_t nsup = 0;
	for ( uint32_t i = 0; i < nrows; ++i ) nsup = std::max<uint32_t>(nsup,first[i]+V[i].size());
	H.nrows = nrows; H.nsup = nsup; H.kln = kln; H.nk = khigh-klow+1;
	// column sums over rows (row index ascending) -> DPnorm
	std::vector<double> dsum(nsup,0.0);
	for ( uint32_t pos = 0; pos < nsup; ++pos )
	{
		double s = 0.0;
		for ( uint32_t i = 0; i < nrows; ++i )
			s += (pos >= first[i] && pos-first[i] < V[i].size()) ? V[i][pos-first[i]] : 0.0;
		dsum[pos] = s;
	}
	H.dpnorm.assign(static_cast<size_t>(nrows)*nsup,0.0);
	H.dpsq.assign(static_cast<size_t>(nrows)*nsup,0.0);
	H.dpsq_vs.assign(static_cast<size_t>(nrows)*nsup,0);
	H.dpsq_first.resize(nrows); H.dpsq_size.resize(nrows);
	for ( uint32_t i = 0; i < nrows; ++i )
	{
		double ss = 0.0;
		for ( size_t j = 0; j < V[i].size(); ++j ) ss += V[i][j]*V[i][j];
		double const c = std::sqrt(1.0/ss);
		for ( size_t j = 0; j < V[i].size(); ++j )
		{
			uint32_t const pos = first[i]+j;
			H.dpnorm[static_cast<size_t>(i)*nsup+pos] = V[i][j] / dsum[pos];
			double const sq = V[i][j]*c;
			H.dpsq[static_cast<size_t>(i)*nsup+pos] = sq;
			H.dpsq_vs[static_cast<size_t>(i)*nsup+pos] = static_cast<uint64_t>(4294967296.0*sq);
		}
		H.dpsq_first[i] = first[i]; H.dpsq_size[i] = V[i].size();
	}
	// transposed copy [pos][row]: lanes = consecutive rows read consecutive words (fast_window.hpp)
	H.dpsq_vst.assign(static_cast<size_t>(nrows)*nsup,0);
	for ( uint32_t i = 0; i < nrows; ++i )
		for ( uint32_t pos = 0; pos < nsup; ++pos )
			H.dpsq_vst[static_cast<size_t>(pos)*nrows+i] = H.dpsq_vs[static_cast<size_t>(i)*nsup+pos];
	H.tab32.assign(static_cast<size_t>(nrows+1)*(nsup+1),0);
	for ( uint32_t pos = 0; pos < nsup; ++pos )
		for ( uint32_t i = 0; i < nrows; ++i )
			H.tab32[static_cast<size_t>(pos)*(nrows+1)+i] = static_cast<uint32_t>(H.dpsq_vst[static_cast<size_t>(pos)*nrows+i]);
	// Vsupport: rows whose support covers read position pos (two monotone pointers, OffsetLikely.hpp:83-92)
	H.suplo.resize(nsup); H.suphi.resize(nsup);
	uint32_t j = 0, k = 0;
	for ( uint32_t pos = 0; pos < nsup; ++pos )
	{
		while ( j < nrows && pos >= first[j]+V[j].size() ) ++j;
		while ( k < nrows && first[k] <= pos ) ++k;
		H.suplo[pos] = j; H.suphi[pos] = k;
	}
	// KmerLimit(pow(est_cor,k)), daccord.cpp:1985
	H.klim.assign(static_cast<size_t>(H.nk)*kln,0);
	for ( uint32_t kk = klow; kk <= khigh; ++kk )
	{
		double const pk = ::std::pow(est_cor,static_cast<double>(kk));
		for ( uint32_t n = 0; n < kln; ++n )
			H.klim[static_cast<size_t>(kk-klow)*kln+n] = pk ? binomUpper(pk,n,0.99) : 0;
	}
	H.firsts = first;
	H.rowsizes.resize(nrows);
	for ( uint32_t i = 0; i < nrows; ++i ) H.rowsizes[i] = V[i].size();
}

// canonical serialisation (same format as oracle_tables) for bit-for-bit table parity tests
void serialiseHostTables(HostTables const & H, std::vector<uint64_t> & B, uint32_t const klimit_n)
{
	auto putd = [&B](double d){ uint64_t u; std::memcpy(&u,&d,8); B.push_back(u); };
	B.push_back(H.nrows); B.push_back(H.nsup);
	for ( uint32_t i = 0; i < H.nrows; ++i )
	{
		uint32_t const fs = H.firsts[i], sz = H.rowsizes[i];
		B.push_back(fs); B.push_back(sz);
		for ( uint32_t t = 0; t < sz; ++t ) putd(H.dpnorm[static_cast<size_t>(i)*H.nsup+fs+t]);
		B.push_back(fs); B.push_back(sz);
		for ( uint32_t t = 0; t < sz; ++t ) putd(H.dpsq[static_cast<size_t>(i)*H.nsup+fs+t]);
		for ( uint32_t t = 0; t < sz; ++t ) B.push_back(H.dpsq_vs[static_cast<size_t>(i)*H.nsup+fs+t]);
	}
	for ( uint32_t pos = 0; pos < H.nsup; ++pos ) { B.push_back(H.suplo[pos]); B.push_back(H.suphi[pos]); }
	for ( uint32_t kk = 0; kk < H.nk; ++kk )
		for ( uint32_t n = 0; n < klimit_n; ++n ) B.push_back(n < H.kln ? H.klim[static_cast<size_t>(kk)*H.kln+n] : 0);
}

}
