/*
 * ORACLE SUPPORT (test infrastructure).  NOT a reference file: it takes the place of the reference's
 * src/DebruijnGraphContainer.hpp:23-114 in the "k16" build of oracle/_ref ONLY, because that file instantiates
 * DebruijnGraph<k> for k in [3,12] and throws "k-mer size k is not compiled in" beyond, while
 * BASELINE.json's configurations run k = 14 (and a sweep up to 16).  The graph template itself (DebruijnGraph.hpp) is
 * the reference's and well defined up to k = 16 (32 bit k-mer words); only the factory below is ours.  The default build
 * of oracle/_ref uses the reference's own container and therefore stops at k = 12, like the reference.
 * A DebruijnGraph<k> holds a direct-addressed node cache of 4^k int32 (DebruijnGraph.hpp:858, :2363): 1 GiB at k = 14,
 * 16 GiB at k = 16, per context.
 */
#if ! defined(DEBRUIJNGRAPHCONTAINER_HPP)
#define DEBRUIJNGRAPHCONTAINER_HPP

#include <DebruijnGraph.hpp>

struct DebruijnGraphContainer
{
	typedef DebruijnGraphContainer this_type;
	typedef libmaus2::util::unique_ptr<this_type>::type unique_ptr_type;

	libmaus2::autoarray::AutoArray < DebruijnGraphInterface::unique_ptr_type > ADG;

	template<unsigned int k>
	static DebruijnGraphInterface * make(uint64_t const want, double const est_cor, std::map < uint64_t, KmerLimit::shared_ptr_type > const & MKL)
	{
		if ( want == k ) return new DebruijnGraph<k>(est_cor,*(MKL.find(k)->second));
		if constexpr ( k < 16 ) return make<k+1>(want,est_cor,MKL);
		libmaus2::exception::LibMausException lme;
		lme.getStream() << "k-mer size " << want << " is not compiled in" << std::endl;
		lme.finish();
		throw lme;
	}

	DebruijnGraphContainer(double const est_cor, uint64_t const kmersizelow, uint64_t const kmersizehigh, std::map < uint64_t, KmerLimit::shared_ptr_type > const & MKL)
	: ADG(kmersizehigh-kmersizelow+1)
	{
		for ( uint64_t k = kmersizelow; k <= kmersizehigh; ++k )
		{
			DebruijnGraphInterface::unique_ptr_type tptr(make<3>(k,est_cor,MKL));
			ADG[k-kmersizelow] = UNIQUE_PTR_MOVE(tptr);
		}
	}
};
#endif
