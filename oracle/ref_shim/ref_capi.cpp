/*
 * ORACLE SUPPORT (test infrastructure, not product code): C entry points that drive the REFERENCE'S OWN per-pile handler --
 * /root/reference/src/HandleContext.hpp (operator() at :1699-2901) with DebruijnGraph.hpp, OffsetLikely.hpp, DotProduct.hpp,
 * ComputeOffsetLikely.hpp, ... compiled where they lie, unmodified -- on the C ABI's structs (include/daccord_hip.h), so that
 * tests/test_oracle_vs_ref.py can compare oracle/'s restatement with the reference source on the same inputs.
 * libmaus2 is not available; oracle/ref_shim/libmaus2/shim.hpp stands in for the symbols the headers use (see its header for
 * what that pins and what it does not).  Built by oracle/ref_shim/build.sh into oracle/_ref/ (git-ignored); nothing
 * under daccord_amd/ links, loads or calls it, and the GPU box never needs /root/reference (the built .so travels).
 *
 * What this file does is what src/daccord.cpp does around the handler, cited line by line:
 *   :1913       OffsetLikely const DP = computeOffsetLikely(windowsize,p_i,p_d)
 *   :1981-1988  KmerLimit(pow(est_cor,k),100) per k
 *   :1989-2023  one HandleContext per thread (inner numthreads = 1)
 *   :2107-2112  the A-read loop, schedule(dynamic,1)
 *   :2402       context(outstr,logstr,ita,ite)
 * plus ONE line the reference lacks: DotProduct::computeShifted() (DotProduct.hpp:54-60) is defined but never called in
 * v0.0.14, yet getKmerPositionWeight reads VS[] (DebruijnGraph.hpp:3852-3855, :3894) -- undefined behaviour as shipped.  The
 * evident intent (VS[i] = uint64(2^32 V[i])) is applied here, after computeOffsetLikely, to every DPnormSquare row.
 */
#include <libmaus2/shim.hpp>
#include <HandleContext.hpp>
// The estimator functions of the reference, handleIndelEstimate<k> and handleIndelEstimateDeep<k> (src/daccord.cpp:271-995), live in
// the driver's translation unit next to main(); build.sh cuts exactly those lines out of /root/reference/src/daccord.cpp into a
// temporary file of the (git-ignored) output directory at build time and deletes it after the compile -- the repository never holds them.
#if defined(DACC_REF_ESTIMATE_EXCERPT)
#include DACC_REF_ESTIMATE_EXCERPT
#endif
#if defined(DACC_REF_CMP_EXCERPT) && defined(DACC_REF_PFG_EXCERPT)
#include DACC_REF_CMP_EXCERPT      // struct OverlapPosComparator, OverlapErrorDescComparator (daccord.cpp:997-1011)
#include DACC_REF_PFG_EXCERPT      // struct PairFirstGreaterComp (daccord.cpp:1406-1412)
#endif
#include "../../include/daccord_hip.h"
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
struct RefCtx
{
	dacc_params par;
	bool haveprofile;
	OffsetLikely OL;
	std::map < uint64_t, KmerLimit::shared_ptr_type > MKL;
	double est_cor;
	libmaus2::dazzler::db::DatabaseFile DB;
	std::vector<dacc_fragment> frags;
	std::string bases;
	std::string log;
	std::string err;
	int tb_order;
	RefCtx() : haveprofile(false), est_cor(0), tb_order(0) {}
};

// ">{aread+1}/{well}/{first}_{first+len} A=[{first},{last}]" + sequence lines of 80 columns (HandleContext.hpp:2710-2724)
static bool parseFasta(std::string const & txt, std::vector<dacc_fragment> & F, std::string & bases, std::string & err)
{
	std::istringstream in(txt);
	std::string line;
	while ( std::getline(in,line) )
	{
		if ( line.empty() ) continue;
		if ( line[0] == '>' )
		{
			long a = 0, well = 0, first = 0, end = 0, f2 = 0, last = 0;
			if ( std::sscanf(line.c_str(),">%ld/%ld/%ld_%ld A=[%ld,%ld]",&a,&well,&first,&end,&f2,&last) != 6 ) { err = "unparsable FASTA header: " + line; return false; }
			dacc_fragment f; f.aread = static_cast<int32_t>(a-1); f.first = static_cast<uint32_t>(first); f.last = static_cast<uint32_t>(last); f.len = 0; f.seq_off = bases.size();
			F.push_back(f);
		}
		else
		{
			if ( F.empty() ) { err = "sequence before header"; return false; }
			bases += line; F.back().len += line.size();
		}
	}
	return true;
}
}

extern "C" {

void * ref_create(dacc_params const * p)
{
	if ( !p || p->klow < 3 || p->klow > p->khigh || !p->w || !p->a ) return 0;
	RefCtx * c = new RefCtx;
	c->par = *p;
	char const * tb = std::getenv("ORACLE_TB_BLOCK");      // the same exposure switch as the oracle's (both aligners of the handler)
	c->tb_order = tb ? std::atoi(tb) : 0;
	return c;
}
void ref_destroy(void * v) { delete static_cast<RefCtx *>(v); }
char const * ref_error(void * v) { return static_cast<RefCtx *>(v)->err.c_str(); }
char const * ref_log(void * v) { return static_cast<RefCtx *>(v)->log.c_str(); }

int ref_set_error_profile(void * v, double p_i, double p_d, double est_cor)
{
	RefCtx * c = static_cast<RefCtx *>(v);
	try
	{
		c->OL = computeOffsetLikely(c->par.w,p_i,p_d);                                         // daccord.cpp:1913
		for ( uint64_t i = 0; i < c->OL.DPnormSquare.size(); ++i ) c->OL.DPnormSquare[i].computeShifted();   // the one added line (header)
		c->MKL.clear();
		for ( uint64_t k = c->par.klow; k <= c->par.khigh; ++k )                               // daccord.cpp:1981-1988
		{
			KmerLimit::shared_ptr_type tptr(new KmerLimit(::std::pow(est_cor,k),100));
			c->MKL[k] = tptr;
		}
	}
	catch ( std::exception const & ex ) { c->err = ex.what(); return -1; }
	c->est_cor = est_cor;
	c->haveprofile = true;
	return 0;
}

int ref_load_db(void * v, uint8_t const * bps, uint64_t, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads)
{
	RefCtx * c = static_cast<RefCtx *>(v);
	c->DB.bps = bps; c->DB.boff = boff; c->DB.rlen = rlen; c->DB.nreads = nreads;
	return 0;
}

// canonical serialisation of the model tables, the format of oracle_tables / dacc_debug_tables
int ref_tables(void * v, uint64_t * out, uint64_t cap, uint64_t * n, uint64_t klimit_n)
{
	RefCtx * c = static_cast<RefCtx *>(v);
	std::vector<uint64_t> B;
	auto put64 = [&](uint64_t x) { B.push_back(x); };
	auto putd = [&](double d) { uint64_t u; std::memcpy(&u,&d,8); B.push_back(u); };
	OffsetLikely const & OL = c->OL;
	put64(OL.DP.size()); put64(OL.Vsupport.size());
	for ( uint64_t i = 0; i < OL.DP.size(); ++i )
	{
		put64(OL.DPnorm[i].firstsign); put64(OL.DPnorm[i].V.size());
		for ( uint64_t j = 0; j < OL.DPnorm[i].V.size(); ++j ) putd(OL.DPnorm[i].V[j]);
		put64(OL.DPnormSquare[i].firstsign); put64(OL.DPnormSquare[i].V.size());
		for ( uint64_t j = 0; j < OL.DPnormSquare[i].V.size(); ++j ) putd(OL.DPnormSquare[i].V[j]);
		for ( uint64_t j = 0; j < OL.DPnormSquare[i].VS.size(); ++j ) put64(OL.DPnormSquare[i].VS[j]);
	}
	for ( uint64_t i = 0; i < OL.Vsupport.size(); ++i ) { put64(OL.Vsupport[i].first); put64(OL.Vsupport[i].second); }
	for ( uint64_t k = c->par.klow; k <= c->par.khigh; ++k )
	{
		KmerLimit & KL = *(c->MKL.find(k)->second);
		for ( uint64_t i = 0; i < klimit_n; ++i ) put64(static_cast<uint64_t>(KL.getLimit(i)));
	}
	*n = B.size();
	uint64_t const m = std::min<uint64_t>(cap,B.size());
	if ( m ) std::memcpy(out,B.data(),8*m);
	return 0;
}

// the A-read loop of daccord.cpp:2107-2112 around HandleContext::operator() (:2402); output re-ordered by pile, the well
// counter of the FASTA name is renumbered by the caller (tests compare fragments, not the counter)
int ref_run_piles(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t, void const * trace, uint64_t, int trace_bytes, int nthreads, int verbose)
{
	RefCtx * c = static_cast<RefCtx *>(v);
	if ( ! c->haveprofile ) return -4;
	if ( nthreads < 1 ) nthreads = 1;
	std::vector<std::string> POUT(npiles), PLOG(npiles);
	std::string failure;
	try
	{
		typedef libmaus2::dazzler::align::OverlapDataInterface ODI;
		// daccord.cpp:1378-1398: free lists shared by the contexts
		libmaus2::parallel::LockedGrowingFreeList<trace_type,TraceAllocator,TraceTypeInfo> traceFreeList;
		libmaus2::parallel::LockedGrowingFreeList<ReadData,ReadDataAllocator,ReadDataTypeInfo> readDataFreeList;
		ReadDecoderAllocator RDA(&c->DB);
		libmaus2::parallel::LockedGrowingFreeList<ReadDecoder,ReadDecoderAllocator,ReadDecoderTypeInfo> readDecoderFreeList(RDA);
		libmaus2::parallel::LockedGrowingFreeList<ReadDecoder,ReadDecoderAllocator,ReadDecoderTypeInfo> readDecoderFreeList2(RDA);
		libmaus2::parallel::SynchronousCounter<uint64_t> wellcounter(0);
		std::vector<HandleContext::unique_ptr_type> AHC(nthreads);
		for ( int i = 0; i < nthreads; ++i )
		{
			// daccord.cpp:1995-2020
			HandleContext::unique_ptr_type tptr(new HandleContext(
				c->par.maxalign,wellcounter,c->par.w,c->par.a,readDataFreeList,readDecoderFreeList,readDecoderFreeList2,traceFreeList,
				c->par.tspace,c->OL,c->par.producefull != 0,c->est_cor,c->par.klow,c->par.khigh,verbose,c->par.minwindowcov,c->par.eminrate,c->par.minlen,
				c->par.minfilterfreq,c->par.maxfilterfreq,c->MKL,1));
			// exposure switch only (default 0 = the shim's documented traceback priority)
			if ( c->tb_order )
				for ( uint64_t t = 0; t < tptr->Pthreadcontext.size(); ++t )
					if ( libmaus2::lcs::NP * np = dynamic_cast<libmaus2::lcs::NP *>(tptr->Pthreadcontext[t]->PNP.get()) ) np->order = c->tb_order;
			AHC[i] = std::move(tptr);
		}
		#ifdef _OPENMP
		#pragma omp parallel for num_threads(nthreads) schedule(dynamic,1)
		#endif
		for ( int64_t i = 0; i < static_cast<int64_t>(npiles); ++i )
		{
			#ifdef _OPENMP
			int const tid = omp_get_thread_num();
			#else
			int const tid = 0;
			#endif
			try
			{
				// (one spare element: the handler's closing log line reads ita[0] even for an empty pile, HandleContext.hpp:2900 --
				// the reference passes a non-empty array there, daccord.cpp:2402)
				std::vector<ODI> V(piles[i].novl+1);
				for ( uint64_t z = 0; z < piles[i].novl; ++z )
				{
					dacc_overlap const & o = ovl[piles[i].first_ovl+z];
					ODI & d = V[z];
					d.f_aread = o.aread; d.f_bread = o.bread; d.f_flags = o.flags; d.f_abpos = o.abpos; d.f_aepos = o.aepos; d.f_bbpos = o.bbpos; d.f_bepos = o.bepos;
					d.f_diffs = o.diffs; d.f_tlen = o.tlen; d.trace = trace; d.trace_bytes = trace_bytes; d.trace_off = o.trace_off;
				}
				std::ostringstream outstr, logstr;
				ODI const * ita = V.data();
				(*AHC[tid])(outstr,logstr,ita,ita+piles[i].novl);       // daccord.cpp:2402
				POUT[i] = outstr.str();
				if ( verbose ) PLOG[i] = logstr.str();
			}
			catch ( std::exception const & ex )
			{
				// daccord.cpp:2464-2478 logs the exception of one read and goes on; the harness reports it
				#ifdef _OPENMP
				#pragma omp critical
				#endif
				failure = std::string("read ") + std::to_string(piles[i].aread) + ": " + ex.what();
			}
		}
	}
	catch ( std::exception const & ex ) { c->err = ex.what(); return -1; }
	if ( failure.size() ) { c->err = failure; return -2; }
	c->frags.clear(); c->bases.clear(); c->log.clear();
	for ( uint64_t i = 0; i < npiles; ++i )
	{
		if ( !parseFasta(POUT[i],c->frags,c->bases,c->err) ) return -3;
		c->log += PLOG[i];
	}
	return 0;
}

int ref_collect(void * v, dacc_fragment const ** frags, uint64_t * nfrags, char const ** bases, uint64_t * nbases)
{
	RefCtx * c = static_cast<RefCtx *>(v);
	*frags = c->frags.data(); *nfrags = c->frags.size(); *bases = c->bases.data(); *nbases = c->bases.size();
	return 0;
}

// highest k the container compiled into this library instantiates (the reference's: 12; the k16 variant: 16)
int ref_max_k()
{
	#if defined(DACC_REF_K16)
	return 16;
	#else
	return 12;
	#endif
}

}

#if defined(DACC_REF_ESTIMATE_EXCERPT)
namespace {
typedef libmaus2::dazzler::align::Overlap RefOverlap;
static void fillOverlaps(std::vector<RefOverlap> & V, dacc_pile const & pile, dacc_overlap const * ovl, void const * trace, int const trace_bytes)
{
	V.assign(pile.novl+1,RefOverlap());
	for ( uint64_t z = 0; z < pile.novl; ++z )
	{
		dacc_overlap const & o = ovl[pile.first_ovl+z];
		RefOverlap & d = V[z];
		d.aread = o.aread; d.bread = o.bread; d.flags = o.flags; d.path.abpos = o.abpos; d.path.aepos = o.aepos; d.path.bbpos = o.bbpos; d.path.bepos = o.bepos;
		d.path.diffs = o.diffs; d.path.tlen = o.tlen; d.path.path.resize(o.tlen/2);
		for ( int32_t i = 0; i < o.tlen/2; ++i )
		{
			uint64_t const a = trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(trace)[o.trace_off+2*i] : reinterpret_cast<uint8_t const *>(trace)[o.trace_off+2*i];
			uint64_t const b = trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(trace)[o.trace_off+2*i+1] : reinterpret_cast<uint8_t const *>(trace)[o.trace_off+2*i+1];
			d.path.path[i] = std::pair<uint16_t,uint16_t>(a,b);
		}
	}
}
}
#endif

extern "C" {

// src/daccord.cpp:1653-1878 around handleIndelEstimate<8> (:271-631): the sampling loop over the given (already selected, :1705-1755)
// piles with the estimator's window 40 / advance 5 (:1665-1666), and the rates of :1867-1878.  deep != 0: handleIndelEstimateDeep<8>
// (:633-995, --deepprofileonly), the window error rates sorted ascending into out (at most cap; returns their number in *ndeep).
int ref_estimate_profile(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, void const * trace, int trace_bytes,
	uint64_t maxalign, uint64_t * counts, uint64_t * usable, uint64_t * unusable, double * prof, int deep, uint32_t * out, uint64_t cap, uint64_t * ndeep)
{
#if defined(DACC_REF_ESTIMATE_EXCERPT)
	RefCtx * c = static_cast<RefCtx *>(v);
	try
	{
		libmaus2::parallel::LockedGrowingFreeList<trace_type,TraceAllocator,TraceTypeInfo> traceFreeList;
		libmaus2::parallel::LockedGrowingFreeList<ReadData,ReadDataAllocator,ReadDataTypeInfo> readDataFreeList;
		ReadDecoderAllocator RDA(&c->DB);
		libmaus2::parallel::LockedGrowingFreeList<ReadDecoder,ReadDecoderAllocator,ReadDecoderTypeInfo> readDecoderFreeList(RDA);
		libmaus2::parallel::LockedGrowingFreeList<ReadDecoder,ReadDecoderAllocator,ReadDecoderTypeInfo> readDecoderFreeList2(RDA);
		libmaus2::lcs::Aligner::unique_ptr_type Pal(DebruijnGraphBase::getAligner());       // daccord.cpp:1690
		libmaus2::lcs::AlignmentStatistics GAS; uint64_t us = 0, un = 0;
		libmaus2::sorting::SerialisingSortingBufferedOutputFileArray< libmaus2::util::U<uint32_t> >::sorter_type usorter;
		std::ostringstream sink;
		for ( uint64_t i = 0; i < npiles; ++i )
		{
			std::vector<RefOverlap> RO; fillOverlaps(RO,piles[i],ovl,trace,trace_bytes);
			libmaus2::lcs::AlignmentStatistics LGAS; uint64_t lu = 0, lun = 0;
			DecodedReadContainer RDC(readDataFreeList,readDecoderFreeList);        // daccord.cpp:1775-1776: two containers, always
			DecodedReadContainer RDC2(readDataFreeList,readDecoderFreeList2);
			if ( deep )
				handleIndelEstimateDeep<8>(usorter,sink,maxalign,RO.data(),RO.data()+piles[i].novl,40,5,RDC,RDC2,traceFreeList,*Pal,c->par.tspace,LGAS,lu,lun);
			else
				handleIndelEstimate<8>(sink,maxalign,RO.data(),RO.data()+piles[i].novl,40,5,RDC,RDC2,traceFreeList,*Pal,c->par.tspace,LGAS,lu,lun);
			GAS += LGAS; us += lu; un += lun;
		}
		counts[0] = GAS.matches; counts[1] = GAS.mismatches; counts[2] = GAS.insertions; counts[3] = GAS.deletions;
		*usable = us; *unusable = un;
		if ( deep )
		{
			std::vector<uint32_t> D; for ( uint64_t i = 0; i < usorter.V.size(); ++i ) D.push_back(usorter.V[i].u);
			std::sort(D.begin(),D.end());
			for ( uint64_t i = 0; i < D.size() && i < cap; ++i ) out[i] = D[i];
			*ndeep = D.size();
		}
		if ( !(GAS.matches + GAS.mismatches + GAS.deletions) ) return -1;
		// the rates the tables are built from, daccord.cpp:1867-1878 (len, numerr, est_cor, p_i, p_d, ...), compiled from its lines
		#if defined(DACC_REF_RATES_EXCERPT)
		#include DACC_REF_RATES_EXCERPT
		(void)est_i_frac; (void)est_d_frac; (void)est_s_frac; (void)p_s;
		prof[0] = p_i; prof[1] = p_d; prof[2] = est_cor;
		return 0;
		#else
		return -9;
		#endif
	}
	catch ( std::exception const & ex ) { c->err = ex.what(); return -2; }
#else
	(void)v; (void)piles; (void)npiles; (void)ovl; (void)trace; (void)trace_bytes; (void)maxalign; (void)counts; (void)usable; (void)unusable; (void)prof; (void)deep; (void)out; (void)cap; (void)ndeep;
	return -9;
#endif
}

}
extern "C" {

// The estimator's pile selection, src/daccord.cpp:1696-1758: the records of one pile in file order through the reference's own
// selection loop (:1712-1742, compiled from its lines: keep the lmaxinput lowest scores, a record that replaces another takes its
// slot) and its sort by abpos (:1758).  out receives the selected records in the resulting order.
int ref_pile_select_lowest(dacc_overlap const * in, uint64_t n, uint64_t lmaxinput, dacc_overlap * out, uint64_t * nout)
{
#if defined(DACC_REF_SEL_EXCERPT)
	typedef libmaus2::dazzler::align::Overlap Overlap_t;
	struct Feed
	{
		dacc_overlap const * in; uint64_t n, i;
		bool getNextOverlap(Overlap_t & O)
		{
			if ( i == n ) return false;
			dacc_overlap const & o = in[i];
			O.aread = o.aread; O.bread = o.bread; O.flags = o.flags; O.path.abpos = o.abpos; O.path.aepos = o.aepos; O.path.bbpos = o.bbpos; O.path.bepos = o.bepos;
			O.path.diffs = o.diffs; O.path.tlen = o.tlen; O.tag = i++;
			return true;
		}
	} feed; feed.in = in; feed.n = n; feed.i = 0;
	Feed * pdec = &feed;
	if ( !lmaxinput ) { *nout = 0; return 0; }
	libmaus2::util::FiniteSizeHeap< std::pair<uint64_t,uint64_t>, PairFirstGreaterComp > RH(lmaxinput);      // :1414-1418, :1696-1701
	libmaus2::autoarray::AutoArray < Overlap_t > RO(lmaxinput);                                              // :1404, :1698-1703
	Overlap_t OVL;
	uint64_t f = 0;
	#include DACC_REF_SEL_EXCERPT
	std::sort(RO.begin(),RO.begin()+f,OverlapPosComparator());                                               // :1758
	for ( uint64_t i = 0; i < f; ++i ) out[i] = in[RO[i].tag];
	*nout = f;
	return 0;
#else
	(void)in; (void)n; (void)lmaxinput; (void)out; (void)nout;
	return -9;
#endif
}

}
