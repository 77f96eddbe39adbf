/*
 * ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the per-pile handler HandleContext::operator()
 * (src/HandleContext.hpp:1699-2901, the overload daccord.cpp:2402 calls), of the window
 * schedule Windows (HandleContext.hpp:382-447), of PileElement ordering (:232-240) and of
 * libmaus2's OverlapDataInterface::computeTrace / getErrorRate as recalled (SURVEY.md 8c:
 * trace points -> one global alignment per tspace block of A, concatenated).
 *
 * PARITY: pinned to the reference's own source since round 4 -- the unmodified reference headers are compiled against a libmaus2
 * stand-in (oracle/ref_shim/ -> oracle/_ref/) and this restatement equals them on model tables and FASTA over the small data set, k
 * ranges and options, slices of every BASELINE configuration (k up to 16) and 270+ random parameter sets (tests/test_oracle_vs_ref.py,
 * profiles/r04_oracle_vs_ref_*.log).  What stays UNPINNED are the libmaus2 primitives themselves (heap sift order, aligner traceback,
 * convolution / binomial arithmetic): our documented definitions on both sides, their exposure measured in DESIGN.md section 6.
 */
#ifndef ORACLE_HANDLE_HPP
#define ORACLE_HANDLE_HPP
#include <map>
#include <string>
#include <cfloat>
#include <cstring>
#include "../include/daccord_hip.h"
#include "o_debruijn.hpp"

namespace oracle {

// DecodedReadContainer (src/DecodedReadContainer.hpp:160-199): forward read as ASCII and its
// reverse complement, decoded from the 2-bit .bps store
struct ReadStore
{
	uint8_t const * bps; uint64_t const * boff; uint32_t const * rlen; uint64_t nreads;
	std::map<int64_t,std::string> fwd, rc;

	ReadStore() : bps(0), boff(0), rlen(0), nreads(0) {}
	uint64_t getReadLength(int64_t const id) const { return rlen[id]; }
	char const * getForwardRead(int64_t const id)
	{
		std::map<int64_t,std::string>::iterator it = fwd.find(id);
		if ( it == fwd.end() )
		{
			std::string s(rlen[id],'A');
			uint8_t const * p = bps + boff[id];
			for ( uint64_t i = 0; i < rlen[id]; ++i )
				s[i] = "ACGT"[(p[i>>2] >> (6-2*(i&3))) & 3];
			it = fwd.insert(std::make_pair(id,s)).first;
		}
		return it->second.c_str();
	}
	char const * getReverseComplementRead(int64_t const id)
	{
		std::map<int64_t,std::string>::iterator it = rc.find(id);
		if ( it == rc.end() )
		{
			char const * f = getForwardRead(id);
			uint64_t const l = rlen[id];
			std::string s(l,'A');
			for ( uint64_t i = 0; i < l; ++i )
			{
				char const c = f[l-1-i];
				s[i] = (c=='A')?'T':(c=='C')?'G':(c=='G')?'C':'A';
			}
			it = rc.insert(std::make_pair(id,s)).first;
		}
		return it->second.c_str();
	}
	void clear() { fwd.clear(); rc.clear(); }
};

struct Params
{
	uint64_t maxalign, windowsize, advancesize; int64_t tspace;
	bool producefull; uint64_t minwindowcov, eminrate, minlen; int64_t minfilterfreq, maxfilterfreq;
	uint64_t klow, khigh;
};

struct Fragment { int32_t aread; uint64_t first, last; std::string seq; };

// HandleContext.hpp:382-447
struct Windows
{
	uint64_t l, a, w, n;
	static uint64_t computeN(uint64_t const l, uint64_t const a, uint64_t const w)
	{
		uint64_t const npre = (l+a >= w) ? ((l+a-w)/a) : 0;
		if ( npre )
		{
			if ( (npre-1)*a+w == l ) return npre;
			else return npre+1;
		}
		else return ( l >= w ) ? 1 : 0;
	}
	Windows(uint64_t rl, uint64_t ra, uint64_t rw) : l(rl), a(ra), w(rw), n(computeN(rl,ra,rw)) {}
	uint64_t size() const { return n; }
	std::pair<uint64_t,uint64_t> operator[](uint64_t const i) const
	{
		if ( i*a+w <= l ) return std::pair<uint64_t,uint64_t>(i*a,i*a+w);
		else return std::pair<uint64_t,uint64_t>(l-w,l);
	}
	uint64_t offset(uint64_t const i) const
	{
		if ( i+1 < size() ) return operator[](i+1).first - operator[](i).first;
		else return 0;
	}
};

// HandleContext.hpp:219-248
struct PileElement
{
	int64_t apos, apre; char sym;
	PileElement() {}
	PileElement(int64_t a, int64_t p, char s) : apos(a), apre(p), sym(s) {}
	bool operator<(PileElement const & P) const
	{
		if ( apos != P.apos ) return apos < P.apos;
		else if ( apre != P.apre ) return apre < P.apre;
		else return sym < P.sym;
	}
};

// ActiveElement.hpp:26-49
struct ActiveElement
{
	uint8_t const * ua; uint8_t const * ub; uint8_t const * ta; uint8_t const * te; uint64_t uboff; double erate;
	ActiveElement() {}
	ActiveElement(uint8_t const * a, uint8_t const * b, uint8_t const * rta, uint8_t const * rte, uint64_t o, double e)
	: ua(a), ub(b), ta(rta), te(rte), uboff(o), erate(e) {}
};

static inline uint64_t traceValue(void const * trace, int const trace_bytes, uint64_t const i)
{
	return trace_bytes == 1 ? static_cast<uint8_t const *>(trace)[i] : static_cast<uint16_t const *>(trace)[i];
}

// libmaus2 OverlapDataInterface::getErrorRate (recalled; cf. the explicit formula daccord.cpp:2167)
static inline double getErrorRate(dacc_overlap const & o)
{
	return static_cast<double>(o.diffs) / static_cast<double>(o.aepos-o.abpos);
}

// libmaus2 OverlapDataInterface::computeTrace (recalled): align every tspace block of A against
// the B span its trace point gives, append the block scripts
static inline void computeTrace(dacc_overlap const & o, void const * trace, int const trace_bytes, int64_t const tspace,
	uint8_t const * aptr, uint8_t const * bptr, std::vector<uint8_t> & out, Aligner & NP)
{
	out.clear();
	int64_t a_i = (o.abpos/tspace)*tspace;
	int64_t b_i = o.bbpos;
	for ( int64_t i = 0; i < o.tlen/2; ++i )
	{
		int64_t const a_i_1 = std::min<int64_t>(a_i+tspace,o.aepos);
		int64_t const b_i_1 = b_i + traceValue(trace,trace_bytes,o.trace_off+2*i+1);
		int64_t const as = std::max<int64_t>(a_i,o.abpos);
		NP.align(aptr+as,a_i_1-as,bptr+b_i,b_i_1-b_i,variant().tb_block);
		out.insert(out.end(),NP.trace.begin(),NP.trace.end());
		b_i = b_i_1;
		a_i = a_i_1;
	}
}

struct HandleContext
{
	Params const par;
	OffsetLikely const & offsetLikely;
	std::vector<DebruijnGraph *> ADG;   // DebruijnGraphContainer.hpp:23-114: one graph per k in [klow,khigh]
	Aligner NP;
	std::vector<PileElement> PV, NPV;
	std::vector<StringRef> MA;
	std::vector< std::vector<uint8_t> > Mtraces;
	std::vector<dacc_window_result> * windowlog; // optional per-window dump (parity tests)
	int32_t pileindex;

	HandleContext(Params const & p, OffsetLikely const & OL, double const est_cor, std::map<uint64_t,KmerLimit> const & MKL)
	: par(p), offsetLikely(OL), windowlog(0), pileindex(0)
	{
		for ( uint64_t k = p.klow; k <= p.khigh; ++k )
			ADG.push_back(new DebruijnGraph(k,est_cor,MKL.find(k)->second));
	}
	~HandleContext() { for ( uint64_t i = 0; i < ADG.size(); ++i ) delete ADG[i]; }

	// HandleContext.hpp:1699-2901
	void operator()(std::vector<Fragment> & out, ReadStore & RC, dacc_overlap const * ita, dacc_overlap const * ite,
		void const * trace, int const trace_bytes)
	{
		uint64_t const windowsize = par.windowsize;
		uint64_t const nintv = ite-ita;
		if ( ! nintv ) return;
		FiniteSizeHeap< std::pair<uint64_t,uint64_t> > E(1024);
		std::map<uint64_t,ActiveElement> activeset;

		// :1769-1776
		uint64_t maxaepos = 0;
		for ( uint64_t z = 0; z < nintv; ++z )
			if ( ita[z].aepos > static_cast<int64_t>(maxaepos) ) maxaepos = ita[z].aepos;
		// :1780-1790
		double maxerate = 0.0, minerate = 1.0;
		for ( uint64_t i = 0; i < nintv; ++i )
		{
			double const erate = getErrorRate(ita[i]);
			if ( erate > maxerate ) maxerate = erate;
			if ( erate < minerate ) minerate = erate;
		}
		double const ediv = (maxerate > minerate) ? (maxerate-minerate) : 1.0;
		int64_t const aid = ita->aread;
		if ( Mtraces.size() < nintv ) Mtraces.resize(nintv);
		uint64_t PVo = 0;
		Windows const W(maxaepos,par.advancesize,windowsize);
		typedef std::pair<uint64_t,uint64_t> upair;

		uint64_t z = 0;
		for ( uint64_t y = 0; y < W.size(); ++y )
		{
			uint64_t const astart = W[y].first;
			uint64_t const aend = W[y].second;

			// add new active intervals :1904-1966
			while ( z < nintv && static_cast<int64_t>(astart) >= ita[z].abpos )
			{
				if ( ita[z].aepos >= static_cast<int64_t>(astart) )
				{
					bool const inv = ita[z].flags & 1;
					uint8_t const * ra = reinterpret_cast<uint8_t const *>(RC.getForwardRead(ita[z].aread));
					uint8_t const * rb = reinterpret_cast<uint8_t const *>(inv ? RC.getReverseComplementRead(ita[z].bread) : RC.getForwardRead(ita[z].bread));
					computeTrace(ita[z],trace,trace_bytes,par.tspace,ra,rb,Mtraces[z],NP);
					uint64_t const aoff = astart-ita[z].abpos;
					uint8_t const * ua = ra + astart;
					uint8_t const * ta = Mtraces[z].data();
					uint8_t const * te = ta + Mtraces[z].size();
					std::pair<uint64_t,uint64_t> const adv = advanceA(ta,te,aoff);
					assert ( adv.first == aoff );
					uint64_t const uboff = ita[z].bbpos + getStringLengthUsed(ta,ta+adv.second).second;
					ta += adv.second;
					uint8_t const * ub = rb + uboff;
					uint64_t const escore = static_cast<uint64_t>(((getErrorRate(ita[z])-minerate)/ediv) * std::numeric_limits<uint32_t>::max());
					uint64_t const eindex = (escore<<32)|z;
					activeset[eindex] = ActiveElement(ua,ub,ta,te,uboff,getErrorRate(ita[z]));
					E.pushBump(upair(ita[z].aepos,eindex));
				}
				z += 1;
			}
			// cleanup :1968-1977
			while ( !E.empty() && E.top().first < aend )
			{
				upair const UP = E.pop();
				uint64_t const zz = UP.second & 0xFFFFFFFFull;
				std::vector<uint8_t>().swap(Mtraces[zz]);
				activeset.erase(UP.second);
			}

			uint64_t MAo = 0;
			MA.clear();
			uint8_t const * w_ua = activeset.size() ? activeset.begin()->second.ua : 0;

			// :1984-2049
			for ( std::map<uint64_t,ActiveElement>::iterator s_ita = activeset.begin(); s_ita != activeset.end(); ++s_ita )
			{
				ActiveElement & AE = s_ita->second;
				std::pair<uint64_t,uint64_t> const adv = advanceA(AE.ta,AE.te,windowsize);
				assert ( adv.first == windowsize );
				uint64_t const bwindowsize = getStringLengthUsed(AE.ta,AE.ta+adv.second).second;
				std::pair<uint64_t,uint64_t> const advadv = advanceA(AE.ta,AE.te,W.offset(y));
				uint64_t const badvancesize = getStringLengthUsed(AE.ta,AE.ta+advadv.second).second;
				AE.ta += advadv.second;
				if ( ! MAo ) { MA.push_back(StringRef(AE.ua,windowsize)); ++MAo; }
				if ( MAo < par.maxalign ) { MA.push_back(StringRef(AE.ub,bwindowsize)); ++MAo; }
				AE.ua += W.offset(y);
				AE.ub += badvancesize;
				AE.uboff += badvancesize;
			}

			// length estimate :2051-2100
			int64_t maxvprodindex = -1;
			if ( MAo )
			{
				int64_t minSupLen = static_cast<int64_t>(MA[0].second)-1;
				int64_t maxSupLen = minSupLen;
				for ( uint64_t j = 1; j < MAo; ++j )
				{
					int64_t const lastpos = static_cast<int64_t>(MA[j].second)-1;
					minSupLen = std::min(minSupLen,lastpos);
					maxSupLen = std::max(maxSupLen,lastpos);
				}
				if ( minSupLen < 0 ) minSupLen = 0;
				if ( maxSupLen < 0 ) maxSupLen = 0;
				uint64_t const supStart = offsetLikely.getSupportLow(minSupLen);
				uint64_t const supEnd = offsetLikely.getSupportHigh(maxSupLen);
				double maxval = std::numeric_limits<double>::min();
				for ( uint64_t i = supStart; i < supEnd; ++i )
				{
					DotProduct const & DP = offsetLikely.DPnorm[i];
					double vprod = 1.0;
					for ( uint64_t j = 0; j < MAo; ++j )
					{
						uint64_t const len = MA[j].second;
						if ( len ) vprod *= DP[len-1];
					}
					if ( vprod > maxval ) { maxval = vprod; maxvprodindex = i; }
				}
			}
			// fallback :2103-2155
			if ( maxvprodindex == -1 )
			{
				int64_t maxoff = -1;
				double maxoffv = std::numeric_limits<double>::min();
				std::vector<uint64_t> Vdist;
				for ( uint64_t i = 0; i < MAo; ++i ) Vdist.push_back(MA[i].second);
				std::sort(Vdist.begin(),Vdist.end());
				std::vector< std::pair<uint64_t,uint64_t> > VPdist;
				{
					uint64_t low = 0;
					while ( low < Vdist.size() )
					{
						uint64_t high = low+1;
						while ( high < Vdist.size() && Vdist[high] == Vdist[low] ) ++high;
						VPdist.push_back(std::pair<uint64_t,uint64_t>(Vdist[low],high-low));
						low = high;
					}
				}
				std::vector<double> VVVV;
				for ( uint64_t i = 0; i < VPdist.size(); ++i )
				{
					while ( !(VPdist[i].first < VVVV.size()) ) VVVV.push_back(0);
					VVVV[VPdist[i].first] = VPdist[i].second-1;
				}
				for ( uint64_t i = 0; i < offsetLikely.DPnormSquare.size(); ++i )
				{
					double const v = offsetLikely.DPnormSquare[i].dotproduct(VVVV.data(),VVVV.size());
					if ( v > maxoffv ) { maxoff = i; maxoffv = v; }
				}
				if ( maxoff != -1 && maxoffv >= 1e-3 )
					maxvprodindex = maxoff;
			}

			dacc_window_result wr; std::memset(&wr,0,sizeof(wr));
			wr.pile = pileindex; wr.y = y; wr.mao = MAo; wr.elength = maxvprodindex+1;

			if ( MAo >= par.minwindowcov )
			{
				int64_t const elength = maxvprodindex+1;
				bool pathfailed = true;
				int64_t filterfreq = -1;
				uint64_t minindex = 0;
				uint64_t minrate = par.eminrate;
				DebruijnGraph * minDG = 0;
				int64_t minff = -1;

				// :2194-2344
				for ( uint64_t adgi = 0; adgi < ADG.size(); ++adgi )
				{
					DebruijnGraph & DG = *ADG[adgi];
					filterfreq = par.maxfilterfreq;
					for ( ; filterfreq >= par.minfilterfreq; --filterfreq )
					{
						DG.setup(MA.data(),MAo);
						DG.filterFreq(std::max(filterfreq,static_cast<int64_t>(1)),MAo);
						DG.computeFeasibleKmerPositions(offsetLikely,1e-3);
						if ( filterfreq == 0 )
						{
							DG.getLevelSuccessors(2);
							DG.setupNodes();
							DG.setupAddHeap(MAo);
							DG.computeFeasibleKmerPositions(offsetLikely,1e-3);
						}
						uint64_t mintry = 0;
						uint64_t const maxtries = 3;
						bool lconsok = false;
						do
						{
							bool const consok = DG.traverse(elength-4,elength+4,MA.data(),MAo,16);
							if ( consok )
							{
								std::pair<uint64_t,uint64_t> const MR = DG.checkCandidatesU(MA.data(),MAo);
								if ( MR.second < minrate )
								{
									lconsok = true;
									minrate = MR.second; minindex = MR.first; minDG = &DG; minff = filterfreq;
								}
								else if ( minDG )
									lconsok = true;
								break;
							}
							else
							{
								if ( ++mintry >= maxtries ) break;
							}
						} while ( DG.addNextFromHeap() );
						if ( lconsok ) { pathfailed = false; break; }
					}
				}

				if ( ! pathfailed )
				{
					std::pair<uint8_t const *,uint8_t const *> const consensus = minDG->getCandidate(minindex);
					uint8_t const * cdata = consensus.first;
					uint64_t const clen = consensus.second-consensus.first;
					wr.status = 1; wr.k = minDG->getKmerSize(); wr.filterfreq = minff; wr.conslen = clen; wr.minrate = minrate;
					std::memcpy(wr.cons,cdata,std::min<uint64_t>(clen,sizeof(wr.cons)-1));

					// :2429-2493
					NP.align(w_ua,windowsize,cdata,clen,variant().tb_cons);
					uint64_t apos = astart;
					uint8_t const * ta = NP.trace.data();
					uint8_t const * te = ta + NP.trace.size();
					while ( ta != te )
					{
						uint64_t numins = 0;
						while ( ta != te && *ta == STEP_INS ) { ++numins; ++ta; }
						for ( uint64_t i = 0; i < numins; ++i )
							PV.push_back(PileElement(apos,(-static_cast<int64_t>(numins))+static_cast<int64_t>(i),*(cdata++)));
						if ( ta != te )
						{
							switch ( *(ta++) )
							{
								case STEP_MATCH: case STEP_MISMATCH:
									PV.push_back(PileElement(apos++,0,*(cdata++))); break;
								case STEP_DEL:
									PV.push_back(PileElement(apos++,0,'D')); break;
								default: break;
							}
						}
					}
					PVo = PV.size();
					assert ( apos == aend );
				}
				else
					wr.status = 2;
			}
			if ( windowlog ) windowlog->push_back(wr);
		}

		while ( !E.empty() )
		{
			upair const UP = E.pop();
			activeset.erase(UP.second);
			std::vector<uint8_t>().swap(Mtraces[UP.second & 0xFFFFFFFFull]);
		}

		// :2541
		std::sort(PV.begin(),PV.begin()+PVo);

		// -f :2543-2580
		if ( par.producefull )
		{
			uint8_t const * ua = reinterpret_cast<uint8_t const *>(RC.getForwardRead(ita->aread));
			uint64_t next = 0, low = 0;
			NPV.clear();
			while ( low < PVo )
			{
				uint64_t high = low+1;
				while ( high < PVo && PV[low].apos == PV[high].apos ) ++high;
				for ( ; static_cast<int64_t>(next) < PV[low].apos; ++next )
					NPV.push_back(PileElement(next,0,::tolower(ua[next])));
				for ( uint64_t i = low; i < high; ++i ) NPV.push_back(PV[i]);
				next = PV[low].apos+1;
				low = high;
			}
			uint64_t const rl = RC.getReadLength(ita->aread);
			for ( ; next < rl; ++next )
				NPV.push_back(PileElement(next,0,::tolower(ua[next])));
			PV.swap(NPV);
			PVo = PV.size();
		}

		// :2582-2612
		std::vector< std::pair<uint64_t,uint64_t> > PVI;
		uint64_t il = 0;
		while ( il < PVo )
		{
			uint64_t ih = il+1;
			while ( ih != PVo && (PV[ih].apos-PV[ih-1].apos) <= 1 ) ++ih;
			uint64_t const first = PV[il].apos;
			uint64_t const lastp = PV[ih-1].apos;
			if ( lastp-first >= 100 )
				PVI.push_back(std::pair<uint64_t,uint64_t>(il,ih));
			il = ih;
		}

		// :2614-2724
		for ( uint64_t zi = 0; zi < PVI.size(); ++zi )
		{
			std::pair<uint64_t,uint64_t> const P = PVI[zi];
			uint64_t const first = PV[P.first].apos;
			uint64_t const lastp = PV[P.second-1].apos;
			std::string CO;
			int64_t l = P.second;
			int64_t depth = -1;
			while ( l > static_cast<int64_t>(P.first) )
			{
				int64_t const h = --l;
				while ( l >= 0 && PV[l].apos == PV[h].apos && PV[l].apre == PV[h].apre ) --l;
				l += 1;
				uint64_t const ld = (h-l)+1;
				if ( PV[l].apre == 0 ) depth = ld;
				std::pair<uint64_t,uint64_t> C[] = {
					std::pair<uint64_t,uint64_t>(0,'A'), std::pair<uint64_t,uint64_t>(0,'C'), std::pair<uint64_t,uint64_t>(0,'G'),
					std::pair<uint64_t,uint64_t>(0,'T'), std::pair<uint64_t,uint64_t>(0,'D'), std::pair<uint64_t,uint64_t>(0,'a'),
					std::pair<uint64_t,uint64_t>(0,'c'), std::pair<uint64_t,uint64_t>(0,'g'), std::pair<uint64_t,uint64_t>(0,'t'),
					std::pair<uint64_t,uint64_t>(0,0) };
				for ( int64_t i = l; i <= h; ++i )
					switch ( PV[i].sym )
					{
						case 'A': C[0].first++; break; case 'C': C[1].first++; break; case 'G': C[2].first++; break;
						case 'T': C[3].first++; break; case 'D': C[4].first++; break; case 'a': C[5].first++; break;
						case 'c': C[6].first++; break; case 'g': C[7].first++; break; case 't': C[8].first++; break;
						default: assert(0); break;
					}
				for ( int64_t i = ld; i < depth; ++i ) C[4].first++;
				std::sort(&C[0],&C[sizeof(C)/sizeof(C[0])],std::greater< std::pair<uint64_t,uint64_t> >());
				if ( C[0].first && C[0].second != 'D' )
					CO.push_back(C[0].second);
			}
			std::reverse(CO.begin(),CO.end());
			if ( par.producefull || CO.size() >= par.minlen )
			{
				Fragment F; F.aread = aid; F.first = first; F.last = lastp; F.seq = CO;
				out.push_back(F);
			}
		}
		PV.clear();
	}
};

}
#endif
