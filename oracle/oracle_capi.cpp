/*
 * ORACLE (test infrastructure, not product code): C entry points so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the CPU restatement
 * through ctypes with the very same structs as the product's C ABI (include/daccord_hip.h).
 * Nothing under daccord_amd/ links, loads or calls this library.
 *
 * Built with -ffp-contract=off: the reference is plain scalar C++ and the path compares
 * FP64 weights exactly.
 */
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <map>
#include <cmath>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "o_handle.hpp"
#include "o_eprof.hpp"

using namespace oracle;

struct OracleCtx
{
	dacc_params par;
	bool haveprofile;
	OffsetLikely OL;
	std::map<uint64_t,KmerLimit> MKL;
	double est_cor;
	ReadStore db;
	std::vector<dacc_fragment> frags;
	std::string bases;
	std::vector<dacc_window_result> windows;
	OracleCtx() : haveprofile(false), est_cor(0) {}
};

static Params makeParams(dacc_params const & p)
{
	Params q;
	q.maxalign = p.maxalign; q.windowsize = p.w; q.advancesize = p.a; q.tspace = p.tspace;
	q.producefull = p.producefull; q.minwindowcov = p.minwindowcov; q.eminrate = p.eminrate; q.minlen = p.minlen;
	q.minfilterfreq = p.minfilterfreq; q.maxfilterfreq = p.maxfilterfreq; q.klow = p.klow; q.khigh = p.khigh;
	return q;
}

extern "C" {

void * oracle_create(dacc_params const * p)
{
	if ( !p || p->klow < 3 || p->khigh > 16 || p->klow > p->khigh || !p->w || !p->a ) return 0;
	OracleCtx * c = new OracleCtx;
	c->par = *p;
	return c;
}
void oracle_destroy(void * v) { delete static_cast<OracleCtx *>(v); }

// src/daccord.cpp:1913 (computeOffsetLikely(windowsize,p_i,p_d)) and :1981-1988 (KmerLimit(pow(est_cor,k),100))
int oracle_set_error_profile(void * v, double p_i, double p_d, double est_cor)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	c->OL = computeOffsetLikely(c->par.w,p_i,p_d);
	c->MKL.clear();
	for ( uint64_t k = c->par.klow; k <= c->par.khigh; ++k )
		c->MKL[k] = KmerLimit(::std::pow(est_cor,static_cast<double>(k)),100);
	c->est_cor = est_cor;
	c->haveprofile = true;
	return 0;
}

int oracle_load_db(void * v, uint8_t const * bps, uint64_t, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	c->db.bps = bps; c->db.boff = boff; c->db.rlen = rlen; c->db.nreads = nreads;
	return 0;
}

// the A-read loop of src/daccord.cpp:2107-2112 (schedule(dynamic,1)), output re-ordered by pile
int oracle_run_piles(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t,
	void const * trace, uint64_t, int trace_bytes, int nthreads, int want_windows)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	if ( ! c->haveprofile ) return -4;
	if ( nthreads < 1 ) nthreads = 1;
	std::vector< std::vector<Fragment> > PF(npiles);
	std::vector< std::vector<dacc_window_result> > PW(npiles);
	Params const par = makeParams(c->par);
	#ifdef _OPENMP
	#pragma omp parallel num_threads(nthreads)
	#endif
	{
		HandleContext HC(par,c->OL,c->est_cor,c->MKL);
		ReadStore RS = c->db;
		#ifdef _OPENMP
		#pragma omp for schedule(dynamic,1)
		#endif
		for ( int64_t i = 0; i < static_cast<int64_t>(npiles); ++i )
		{
			HC.windowlog = want_windows ? &PW[i] : 0;
			HC.pileindex = i;
			dacc_overlap const * ita = ovl + piles[i].first_ovl;
			HC(PF[i],RS,ita,ita+piles[i].novl,trace,trace_bytes);
			RS.clear();
		}
	}
	c->frags.clear(); c->bases.clear(); c->windows.clear();
	for ( uint64_t i = 0; i < npiles; ++i )
	{
		for ( uint64_t j = 0; j < PF[i].size(); ++j )
		{
			dacc_fragment f;
			f.aread = PF[i][j].aread; f.first = PF[i][j].first; f.last = PF[i][j].last;
			f.len = PF[i][j].seq.size(); f.seq_off = c->bases.size();
			c->bases += PF[i][j].seq;
			c->frags.push_back(f);
		}
		c->windows.insert(c->windows.end(),PW[i].begin(),PW[i].end());
	}
	return 0;
}

int oracle_collect(void * v, dacc_fragment const ** frags, uint64_t * nfrags, char const ** bases, uint64_t * nbases)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	*frags = c->frags.data(); *nfrags = c->frags.size(); *bases = c->bases.data(); *nbases = c->bases.size();
	return 0;
}

int oracle_windows(void * v, dacc_window_result * out, uint64_t cap, uint64_t * n)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	*n = c->windows.size();
	uint64_t const m = std::min<uint64_t>(cap,c->windows.size());
	if ( m ) std::memcpy(out,c->windows.data(),m*sizeof(dacc_window_result));
	return 0;
}

// canonical serialisation of the model tables (compared bit for bit with dacc_debug_tables)
static void put64(std::vector<uint64_t> & B, uint64_t v) { B.push_back(v); }
static void putd(std::vector<uint64_t> & B, double d) { uint64_t u; std::memcpy(&u,&d,8); B.push_back(u); }
int oracle_tables(void * v, uint64_t * out, uint64_t cap, uint64_t * n, uint64_t klimit_n)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	std::vector<uint64_t> B;
	OffsetLikely const & OL = c->OL;
	put64(B,OL.DP.size());
	put64(B,OL.Vsupport.size());
	for ( uint64_t i = 0; i < OL.DP.size(); ++i )
	{
		put64(B,OL.DPnorm[i].firstsign); put64(B,OL.DPnorm[i].V.size());
		for ( uint64_t j = 0; j < OL.DPnorm[i].V.size(); ++j ) putd(B,OL.DPnorm[i].V[j]);
		put64(B,OL.DPnormSquare[i].firstsign); put64(B,OL.DPnormSquare[i].V.size());
		for ( uint64_t j = 0; j < OL.DPnormSquare[i].V.size(); ++j ) putd(B,OL.DPnormSquare[i].V[j]);
		for ( uint64_t j = 0; j < OL.DPnormSquare[i].VS.size(); ++j ) put64(B,OL.DPnormSquare[i].VS[j]);
	}
	for ( uint64_t i = 0; i < OL.Vsupport.size(); ++i ) { put64(B,OL.Vsupport[i].first); put64(B,OL.Vsupport[i].second); }
	for ( uint64_t k = c->par.klow; k <= c->par.khigh; ++k )
	{
		KmerLimit KL = c->MKL[k];
		for ( uint64_t i = 0; i < klimit_n; ++i ) put64(B,static_cast<uint64_t>(KL.getLimit(i)));
	}
	*n = B.size();
	uint64_t const m = std::min<uint64_t>(cap,B.size());
	if ( m ) std::memcpy(out,B.data(),8*m);
	return 0;
}

// ---- unit-test helpers ----
uint64_t oracle_align(uint8_t const * a, uint64_t m, uint8_t const * b, uint64_t n, uint8_t * trace_out, uint64_t * tracelen)
{
	Aligner A;
	uint64_t const d = A.align(a,m,b,n);
	*tracelen = A.trace.size();
	if ( trace_out ) std::memcpy(trace_out,A.trace.data(),A.trace.size());
	return d;
}
uint64_t oracle_edit_distance(uint8_t const * a, uint64_t m, uint8_t const * b, uint64_t n)
{
	std::vector<uint32_t> tmp;
	return editDistance(a,m,b,n,tmp);
}
uint64_t oracle_windows_count(uint64_t l, uint64_t a, uint64_t w) { return Windows::computeN(l,a,w); }

// one window through the graph engine: strings -> consensus (the DebruijnGraphInterface call
// sequence of HandleContext.hpp:2194-2344 for a single k), for window-level parity tests
int oracle_window_consensus(void * v, uint8_t const * strings, uint32_t const * lens, uint32_t nstr, int32_t elength,
	char * cons_out, uint32_t * conslen, uint64_t * minrate_out, int32_t * ff_out)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	std::vector<StringRef> MA;
	uint64_t off = 0;
	for ( uint32_t i = 0; i < nstr; ++i ) { MA.push_back(StringRef(strings+off,lens[i])); off += lens[i]; }
	uint64_t minrate = c->par.eminrate;
	*conslen = 0;
	bool found = false;
	for ( uint64_t k = c->par.klow; k <= c->par.khigh; ++k )
	{
		DebruijnGraph DG(k,c->est_cor,c->MKL[k]);
		for ( int64_t ff = c->par.maxfilterfreq; ff >= c->par.minfilterfreq; --ff )
		{
			DG.setup(MA.data(),nstr);
			DG.filterFreq(std::max<int64_t>(ff,1),nstr);
			DG.computeFeasibleKmerPositions(c->OL,1e-3);
			if ( ff == 0 )
			{
				DG.getLevelSuccessors(2); DG.setupNodes(); DG.setupAddHeap(nstr); DG.computeFeasibleKmerPositions(c->OL,1e-3);
			}
			uint64_t mintry = 0; bool lconsok = false;
			do
			{
				if ( DG.traverse(elength-4,elength+4,MA.data(),nstr,16) )
				{
					std::pair<uint64_t,uint64_t> const MR = DG.checkCandidatesU(MA.data(),nstr);
					if ( MR.second < minrate )
					{
						lconsok = true; minrate = MR.second;
						std::pair<uint8_t const *,uint8_t const *> const U = DG.getCandidate(MR.first);
						*conslen = U.second-U.first; std::memcpy(cons_out,U.first,*conslen); *ff_out = ff;
						found = true;
					}
					else if ( found ) lconsok = true;
					break;
				}
				else if ( ++mintry >= 3 ) break;
			} while ( DG.addNextFromHeap() );
			if ( lconsok ) break;
		}
	}
	*minrate_out = minrate;
	return found ? 1 : 0;
}

}

// ---- P1: pile load + top-D select + sort by abpos (src/daccord.cpp:2120-2288) ----
// Records arrive in .las order.  The reference streams the pile's byte range in 64 KiB blocks,
// keeps <= maxinput records in a min-heap on score = uint64(ldexp(diffs/(aepos-abpos),30))
// (evicting the LOWEST score when full, :2169-2177), then copies the survivors: those of the
// final block first, in descending entry order (:2199-2214), then the earlier blocks in
// ascending (block,entry) order (:2216-2262), and std::sorts the pointers by abpos (:2284-2288).
// Which block a record split across a 64 KiB boundary is attributed to is libmaus2
// OverlapParser behaviour (not in /root/reference): we attribute it to the block in which its
// last byte arrives.  PINNED: tests/test_oracle_vs_ref.py runs this function against :2026-2105 + :2120-2288 compiled from the
// reference's lines (oracle/ref_shim/ref_select.cpp) over piles of up to 9 input blocks; only the parser's attribution stays assumed.
namespace {
struct OverlapEntry
{
	uint64_t score, blockid, entryid, idx;
	bool operator<(OverlapEntry const & O) const { return score < O.score; }
};
struct OverlapEntryBlockComparator
{
	bool operator()(OverlapEntry const & A, OverlapEntry const & B) const
	{
		if ( A.blockid != B.blockid ) return A.blockid < B.blockid; else return A.entryid < B.entryid;
	}
};
struct AbposComparator
{
	bool operator()(dacc_overlap const & A, dacc_overlap const & B) const { return A.abpos < B.abpos; }
};
}
extern "C" int oracle_pile_select(dacc_overlap const * in, uint64_t n, int trace_bytes, uint64_t maxinput, dacc_overlap * out, uint64_t * nout)
{
	if ( ! maxinput ) { *nout = 0; return 0; }
	oracle::FiniteSizeHeap<OverlapEntry> RHO(maxinput);
	uint64_t const blocksize = 64*1024;
	uint64_t bytepos = 0, curblock = 0, entry = 0, lastblock = 0;
	for ( uint64_t i = 0; i < n; ++i )
	{
		uint64_t const s = 40 + static_cast<uint64_t>(in[i].tlen)*trace_bytes;
		uint64_t const endb = (bytepos+s-1)/blocksize;
		if ( endb != curblock ) { curblock = endb; entry = 0; }
		bytepos += s;
		uint64_t const score = static_cast<uint64_t>(ldexp(static_cast<double>(in[i].diffs)/static_cast<double>(in[i].aepos-in[i].abpos),30));
		if ( RHO.f == maxinput )
		{
			if ( score > RHO.top().score ) RHO.popvoid();
		}
		if ( RHO.f < maxinput )
		{
			OverlapEntry E; E.score = score; E.blockid = curblock; E.entryid = entry; E.idx = i;
			RHO.push(E);
		}
		++entry;
		lastblock = curblock;
	}
	uint64_t const nblocks = n ? lastblock+1 : 0;
	std::sort(RHO.H.begin(),RHO.H.begin()+RHO.f,OverlapEntryBlockComparator());
	uint64_t o = 0;
	while ( RHO.f && RHO.H[RHO.f-1].blockid == nblocks-1 )
	{
		out[o++] = in[RHO.H[RHO.f-1].idx];
		RHO.f--;
	}
	for ( uint64_t i = 0; i < RHO.f; ++i )
		out[o++] = in[RHO.H[i].idx];
	std::sort(out,out+o,AbposComparator());
	*nout = o;
	return 0;

}

extern "C" {

// src/daccord.cpp:1705-1737, 1755: the estimator's pile selection (lowest scores kept, slots reused, std::sort by abpos)
struct PairFirstGreaterComp { bool operator()(std::pair<uint64_t,uint64_t> const & A, std::pair<uint64_t,uint64_t> const & B) const { return A.first > B.first; } };
int oracle_pile_select_lowest(dacc_overlap const * in, uint64_t n, uint64_t lmaxinput, dacc_overlap * RO, uint64_t * nout)
{
	FiniteSizeHeap< std::pair<uint64_t,uint64_t>, PairFirstGreaterComp > RH(lmaxinput ? lmaxinput : 1);
	uint64_t f = 0;
	for ( uint64_t i = 0; i < n && lmaxinput; ++i )
	{
		uint64_t const score = static_cast<uint64_t>(ldexp((static_cast<double>(in[i].diffs) / static_cast<double>(in[i].aepos - in[i].abpos)),30));
		if ( RH.f == lmaxinput )
		{
			if ( score > RH.top().first ) continue;
			uint64_t const p = RH.top().second;
			RH.popvoid();
			RO[p] = in[i];
			RH.push(std::pair<uint64_t,uint64_t>(score,p));
		}
		else { uint64_t const p = f++; RH.push(std::pair<uint64_t,uint64_t>(score,p)); RO[p] = in[i]; }
	}
	std::sort(RO,RO+f,[](dacc_overlap const & A, dacc_overlap const & B){ return A.abpos < B.abpos; });
	*nout = f;
	return 0;
}

// src/daccord.cpp:1653-1878: the sampling loop over piles and the derived rates
int oracle_estimate_profile(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, void const * trace, int trace_bytes,
	uint64_t maxalign, int twodb, uint64_t * counts, uint64_t * usable, uint64_t * unusable, double * prof)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	AlignmentStatistics GAS; uint64_t us = 0, un = 0;
	for ( uint64_t i = 0; i < npiles; ++i )
	{
		AlignmentStatistics LGAS; uint64_t lu = 0, lun = 0;
		c->db.clear();
		handleIndelEstimate8(maxalign,ovl+piles[i].first_ovl,ovl+piles[i].first_ovl+piles[i].novl,40,5,c->db,twodb != 0,trace,trace_bytes,c->par.tspace,LGAS,lu,lun);
		GAS += LGAS; us += lu; un += lun;
	}
	counts[0] = GAS.matches; counts[1] = GAS.mismatches; counts[2] = GAS.insertions; counts[3] = GAS.deletions;
	*usable = us; *unusable = un;
	uint64_t const len = GAS.matches + GAS.mismatches + GAS.deletions;
	uint64_t const numerr = GAS.mismatches + GAS.deletions + GAS.insertions;
	if ( !len ) return -1;
	double const est_erate = static_cast<double>(numerr) / len;
	prof[0] = static_cast<double>(GAS.insertions) / len; prof[1] = static_cast<double>(GAS.deletions) / len; prof[2] = 1.0 - est_erate;
	return 0;
}

// src/daccord.cpp:1442-1650 (--deepprofileonly): handleIndelEstimateDeep<8> over the given (already selected) piles; the window
// error rates as sorted 32 bit values (what the merger of :1626-1648 reads back), at most cap of them; returns their number
uint64_t oracle_deep_profile(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, void const * trace, int trace_bytes,
	uint64_t maxalign, int twodb, uint32_t * out, uint64_t cap)
{
	OracleCtx * c = static_cast<OracleCtx *>(v);
	std::vector<uint32_t> D;
	for ( uint64_t i = 0; i < npiles; ++i )
	{
		AlignmentStatistics LGAS; uint64_t lu = 0, lun = 0;
		c->db.clear();
		handleIndelEstimate8(maxalign,ovl+piles[i].first_ovl,ovl+piles[i].first_ovl+piles[i].novl,40,5,c->db,twodb != 0,trace,trace_bytes,c->par.tspace,LGAS,lu,lun,&D);
	}
	std::sort(D.begin(),D.end());
	for ( uint64_t i = 0; i < D.size() && i < cap; ++i ) out[i] = D[i];
	return D.size();
}

}
