/*
 * ORACLE (test infrastructure, not product code): CPU restatement of the error profile estimation,
 * src/daccord.cpp:271-631 (handleIndelEstimate<8>) and :1653-1878 (driver, rates).  PARITY: handleIndelEstimate<8> and
 * handleIndelEstimateDeep<8> are pinned to the reference's own functions since round 4 (compiled into oracle/_ref from the lines
 * of daccord.cpp where they lie; tests/test_oracle_vs_ref.py) -- that comparison corrected this file: the A window ALWAYS joins
 * a window's strings.  The sampling loop around them (:1653-1755) and the libmaus2 primitives stay unpinned: libmaus2's aligner,
 * AlignmentStatistics and KmerRepeatDetector are not in the reference tree; their semantics are recalled
 * (KmerRepeatDetector(q).detect: true iff some q-mer occurs twice in the string; AlignmentStatistics: counts of MATCH /
 * MISMATCH / INS (second string only) / DEL (first string only) steps).
 */
#ifndef ORACLE_EPROF_HPP
#define ORACLE_EPROF_HPP
#include <map>
#include <set>
#include "o_handle.hpp"

namespace oracle {

struct AlignmentStatistics
{
	uint64_t matches, mismatches, insertions, deletions;
	AlignmentStatistics() : matches(0), mismatches(0), insertions(0), deletions(0) {}
	AlignmentStatistics & operator+=(AlignmentStatistics const & O) { matches += O.matches; mismatches += O.mismatches; insertions += O.insertions; deletions += O.deletions; return *this; }
	double getErrorRate() const { uint64_t const t = matches+mismatches+insertions+deletions; return t ? static_cast<double>(mismatches+insertions+deletions)/t : 0.0; }
};

static inline bool kmerRepeatDetect(uint8_t const * p, uint64_t const n, unsigned int const q)
{
	std::set<std::string> S;
	for ( uint64_t i = 0; i+q <= n; ++i )
		if ( !S.insert(std::string(p+i,p+i+q)).second ) return true;
	return false;
}

// src/daccord.cpp:271-631
static inline double handleIndelEstimate8(uint64_t const maxalign, dacc_overlap const * ita, dacc_overlap const * ite,
	uint64_t const windowsize, uint64_t const advancesize, ReadStore & RC, bool const twodb, void const * trace, int const trace_bytes,
	int64_t const tspace, AlignmentStatistics & RGAS, uint64_t & usable, uint64_t & unusable, std::vector<uint32_t> * deep = 0)
{
	unsigned int const k = 8;
	uint64_t const nintv = ite-ita;
	double maxerate = 0.0, minerate = 1.0;
	for ( uint64_t i = 0; i < nintv; ++i )
	{
		double const erate = getErrorRate(ita[i]);
		if ( erate > maxerate ) maxerate = erate;
		if ( erate < minerate ) minerate = erate;
	}
	double const ediv = (maxerate > minerate) ? (maxerate - minerate) : 1.0;
	std::map<uint64_t, std::vector<uint8_t> > Mtraces;
	Aligner NP;
	uint64_t maxaepos = 0;
	if ( nintv )
	{
		uint8_t const * ua = reinterpret_cast<uint8_t const *>(RC.getForwardRead(ita[0].aread));
		for ( uint64_t z = 0; z < nintv; ++z )
		{
			if ( ita[z].aepos > static_cast<int64_t>(maxaepos) ) maxaepos = ita[z].aepos;
			uint8_t const * ub = reinterpret_cast<uint8_t const *>((ita[z].flags&1) ? RC.getReverseComplementRead(ita[z].bread) : RC.getForwardRead(ita[z].bread));
			computeTrace(ita[z],trace,trace_bytes,tspace,ua,ub,Mtraces[z],NP);
		}
	}
	typedef std::pair<uint64_t,uint64_t> upair;
	FiniteSizeHeap<upair> E(1024);
	struct AE_t { uint8_t const * ua; uint8_t const * ub; uint8_t const * ta; uint8_t const * te; };
	std::map<uint64_t,AE_t> activeset;
	std::vector<StringRef> MA;
	KmerLimit KL(0.85,0);
	DebruijnGraph DG(k,0,KL);
	uint64_t z = 0;
	uint64_t const ylimit = (maxaepos + advancesize >= windowsize) ? ((maxaepos + advancesize - windowsize) / advancesize) : 0;
	double esum = 0; uint64_t ecnt = 0;
	for ( uint64_t y = 0; y < ylimit; ++y )
	{
		uint64_t const astart = y * advancesize, aend = astart + windowsize;
		while ( z < nintv && static_cast<int64_t>(astart) >= ita[z].abpos )
		{
			if ( ita[z].aepos >= static_cast<int64_t>(astart) )
			{
				uint64_t const aoff = astart - ita[z].abpos;
				std::vector<uint8_t> const & T = Mtraces.find(z)->second;
				uint8_t const * ta = T.data(); uint8_t const * te = T.data()+T.size();
				std::pair<uint64_t,uint64_t> const adv = advanceA(ta,te,aoff);
				uint8_t const * ua = reinterpret_cast<uint8_t const *>(RC.getForwardRead(ita[z].aread)) + ita[z].abpos + aoff;
				uint64_t const uboff = ita[z].bbpos + getStringLengthUsed(ta,ta+adv.second).second;
				uint8_t const * ub = reinterpret_cast<uint8_t const *>((ita[z].flags&1) ? RC.getReverseComplementRead(ita[z].bread) : RC.getForwardRead(ita[z].bread)) + uboff;
				ta += adv.second;
				uint64_t const escore = static_cast<uint64_t>(((getErrorRate(ita[z]) - minerate) / ediv) * std::numeric_limits<uint32_t>::max());
				uint64_t const eindex = (escore<<32) | z;
				AE_t A; A.ua = ua; A.ub = ub; A.ta = ta; A.te = te;
				activeset[eindex] = A;
				E.pushBump(upair(ita[z].aepos,eindex));
			}
			z += 1;
		}
		while ( (!E.empty()) && E.top().first <= aend ) { upair const UP = E.pop(); activeset.erase(UP.second); }
		MA.clear();
		for ( std::map<uint64_t,AE_t>::iterator s_ita = activeset.begin(); s_ita != activeset.end(); ++s_ita )
		{
			AE_t & AE = s_ita->second;
			std::pair<uint64_t,uint64_t> const adv = advanceA(AE.ta,AE.te,windowsize);
			std::pair<uint64_t,uint64_t> const sl = getStringLengthUsed(AE.ta,AE.ta+adv.second);
			// `(! MAo) && (&RC != &RC2)` (daccord.cpp:522): RC and RC2 are two local containers of the caller (daccord.cpp:1775-1776),
			// so the test is ALWAYS true in v0.0.14 and the A window always joins the strings -- also with one database.  (Rounds 1-3
			// read it as "two databases only"; oracle/_ref, the reference's own estimator, showed the difference in round 4.)
			(void)twodb;
			if ( MA.empty() ) MA.push_back(StringRef(AE.ua,windowsize));
			if ( MA.size() < maxalign ) MA.push_back(StringRef(AE.ub,sl.second));
			std::pair<uint64_t,uint64_t> const advadv = advanceA(AE.ta,AE.te,advancesize);
			std::pair<uint64_t,uint64_t> const sladv = getStringLengthUsed(AE.ta,AE.ta+advadv.second);
			AE.ua += advancesize; AE.ta += advadv.second; AE.ub += sladv.second;
		}
		uint64_t const MAo = MA.size();
		if ( MAo >= 3 )
		{
			bool ghasrep = false;
			for ( uint64_t i = 0; i < MAo; ++i ) { bool const hasrep = kmerRepeatDetect(MA[i].first,MA[i].second,k-1); ghasrep = ghasrep || hasrep; }
			if ( ghasrep ) unusable += 1;
			else
			{
				usable += 1;
				DG.setup(MA.data(),MAo);
				DG.filterFreq(2,MAo);
				bool const consok = DG.traverseTrivial();
				if ( consok )
				{
					std::string const consensus = DG.getConsensus();
					AlignmentStatistics GAS;
					for ( uint64_t i = 0; i < MAo; ++i )
					{
						NP.align(reinterpret_cast<uint8_t const *>(consensus.c_str()),consensus.size(),MA[i].first,MA[i].second);
						for ( size_t q = 0; q < NP.trace.size(); ++q )
							switch ( NP.trace[q] )
							{
								case STEP_MATCH: GAS.matches++; break; case STEP_MISMATCH: GAS.mismatches++; break;
								case STEP_INS: GAS.insertions++; break; default: GAS.deletions++; break;
							}
					}
					RGAS += GAS;
					esum += GAS.getErrorRate(); ecnt += 1;
					// handleIndelEstimateDeep (src/daccord.cpp:634-995) is this function with one more output (:963-968)
					if ( deep )
					{
						double const e = GAS.getErrorRate();
						uint64_t const v = static_cast<uint64_t>(static_cast<double>(std::numeric_limits<uint32_t>::max()) * e + 0.5);
						deep->push_back(static_cast<uint32_t>(std::min(v,static_cast<uint64_t>(std::numeric_limits<uint32_t>::max()))));
					}
				}
			}
		}
	}
	return ecnt ? (esum / ecnt) : 0.0;
}

}
#endif
