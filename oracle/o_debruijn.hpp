/*
 * ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the reference's per-window local de Bruijn graph engine,
 * DebruijnGraph<k> (src/DebruijnGraph.hpp:671-5483), with k a run-time value (3..16) and
 * the 4^k direct-addressed nodecache (DebruijnGraph.hpp:858, 2363) replaced by an
 * open-addressing table for k > 12 (semantically neutral: it is only kmer -> node id).
 * Each function cites the reference lines it follows.  Orderings copied from the reference:
 *   Stretch::operator<        DebruijnGraph.hpp:145-163      ReversePath::operator< :317-323
 *   EdgeActivationElement     :414-422                       SeqPos::operator<      :535-541
 *   ScoreInterval::operator<  :514-517                       Links (freq<<8|sym) desc  Links.hpp:35-57
 * libmaus2 primitives (absent from /root/reference) are replaced as documented in
 * o_heap.hpp / o_align.hpp / o_offsetlikely.hpp; RMQ and wavelet-tree queries
 * (DebruijnGraph.hpp:3499-3534) act on a permutation, so plain scans are exact equivalents.
 *
 * PARITY: pinned to the reference's own source since round 4 -- the unmodified reference headers are compiled against a libmaus2
 * stand-in (oracle/ref_shim/ -> oracle/_ref/) and this restatement equals them on model tables and FASTA over the small data set, k
 * ranges and options, slices of every BASELINE configuration (k up to 16) and 270+ random parameter sets (tests/test_oracle_vs_ref.py,
 * profiles/r04_oracle_vs_ref_*.log).  What stays UNPINNED are the libmaus2 primitives themselves (heap sift order, aligner traceback,
 * convolution / binomial arithmetic): our documented definitions on both sides, their exposure measured in DESIGN.md section 6.
 */
#ifndef ORACLE_DEBRUIJN_HPP
#define ORACLE_DEBRUIJN_HPP
#include <vector>
#include <cstdint>
#include <algorithm>
#include <numeric>
#include <limits>
#include <functional>
#include <utility>
#include <string>
#include <cassert>
#include "o_heap.hpp"
#include "o_align.hpp"
#include "o_offsetlikely.hpp"

namespace oracle {

typedef std::pair<uint8_t const *, uint64_t> StringRef; // (pointer to ASCII ACGT, length)

// libmaus2::fastx::mapChar / remapChar (DebruijnGraph.hpp:2086, 1433): A,C,G,T <-> 0,1,2,3
static inline uint64_t mapChar(uint8_t const c)
{
	switch ( c ) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 0; }
}
static inline uint8_t remapChar(uint64_t const v) { return "ACGT"[v&3]; }

// Node.hpp:21-58
struct Node
{
	uint64_t v, spo, freq, numsucc, numsuccactive, feaspos, cfeaspos, numfeaspos, numcfeaspos,
		pfostart, pfosize, cpfostart, cpfosize, plow, phigh, cplow, cphigh;
};

// Links.hpp:23-73
struct Links
{
	uint64_t A[4];
	uint64_t p;
	Links() : p(0) {}
	void reset() { p = 0; }
	void push(uint64_t const sym, uint64_t const freq) { if ( freq ) A[p++] = (freq<<8)|sym; }
	void setSize(uint64_t const rp) { p = rp; }
	void sort() { if ( p > 1 ) std::sort(&A[0],&A[p],std::greater<uint64_t>()); }
	uint64_t size() const { return p; }
	uint64_t getFreq(uint64_t const i) const { return A[i]>>8; }
	uint64_t getSym(uint64_t const i) const { return A[i]&0xFF; }
};

// DebruijnGraph.hpp:90-205
struct Stretch
{
	uint64_t first, ext, last, len, stretchO, feasposO, feasposL, cfeasposO, cfeasposL;
	Stretch() {}
	Stretch(uint64_t f, uint64_t e, uint64_t l, uint64_t n, uint64_t o)
	: first(f), ext(e), last(l), len(n), stretchO(o), feasposO(0), feasposL(0), cfeasposO(0), cfeasposL(0) {}
	bool isLoop() const { return first == last; }
	bool operator<(Stretch const & O) const
	{
		if ( first != O.first ) return first < O.first;
		else if ( ext != O.ext ) return ext < O.ext;
		else if ( len != O.len ) return len > O.len;
		else return last < O.last;
	}
	bool operator==(Stretch const & O) const
	{
		return first == O.first && ext == O.ext && last == O.last && len == O.len;
	}
};

// DebruijnGraph.hpp:207-259
struct Path
{
	uint64_t len, off, pos;
	double weight;
	uint64_t baselen;
	Path() : len(0), off(0), pos(0), weight(0.0), baselen(0) {}
	Path(uint64_t l, uint64_t o, uint64_t p, double w, uint64_t b) : len(l), off(o), pos(p), weight(w), baselen(b) {}
};
struct PathWeightComparator { bool operator()(Path const & A, Path const & B) const { return A.weight < B.weight; } };

// DebruijnGraph.hpp:261-324 (note the narrow field types)
struct ReversePath
{
	uint32_t linkoff;
	uint32_t front;
	double weight;
	uint16_t pos, len, baselen;
	ReversePath() : linkoff(0), front(0), weight(0.0), pos(0), len(0), baselen(0) {}
	ReversePath(uint64_t rlen, uint64_t rlinkoff, int64_t rpos, double rweight, uint64_t rfront, uint64_t rbaselen)
	: linkoff(rlinkoff), front(rfront), weight(rweight), pos(rpos), len(rlen), baselen(rbaselen) {}
	bool operator<(ReversePath const & P) const
	{
		if ( front != P.front ) return front < P.front;
		else return baselen < P.baselen;
	}
};
struct ReversePathFrontComparator { bool operator()(ReversePath const & A, ReversePath const & B) const { return A.front < B.front; } };
struct ReversePathBaseLenComparator { bool operator()(ReversePath const & A, ReversePath const & B) const { return A.baselen < B.baselen; } };
struct ReversePathWeightHeapComparator { bool operator()(ReversePath const & A, ReversePath const & B) const { return A.weight < B.weight; } };
struct ReversePathWeightQueueHeapComparator { bool operator()(ReversePath const & A, ReversePath const & B) const { return A.weight > B.weight; } };

// DebruijnGraph.hpp:396-423
struct EdgeActivationElement
{
	uint64_t freq, nodeid, edgeid;
	EdgeActivationElement() {}
	EdgeActivationElement(uint64_t f, uint64_t n, uint64_t e) : freq(f), nodeid(n), edgeid(e) {}
	bool operator<(EdgeActivationElement const & E) const
	{
		if ( E.freq != freq ) return freq > E.freq;
		else if ( nodeid != E.nodeid ) return nodeid < E.nodeid;
		else return edgeid < E.edgeid;
	}
};

// DebruijnGraph.hpp:434-483
struct ConsensusCandidate
{
	uint64_t o, l;
	double weight, error;
	ConsensusCandidate() {}
	ConsensusCandidate(uint64_t ro, uint64_t rl, double w, double e) : o(ro), l(rl), weight(w), error(e) {}
};
struct ConsensusCandidateDumpHeapComparator { bool operator()(ConsensusCandidate const & A, ConsensusCandidate const & B) const { return A.weight < B.weight; } };
struct ConsensusCandidateHeapComparator { bool operator()(ConsensusCandidate const & A, ConsensusCandidate const & B) const { return A.weight > B.weight; } };
struct ConsensusCandidateErrorComparator { bool operator()(ConsensusCandidate const & A, ConsensusCandidate const & B) const { return A.error < B.error; } };

// DebruijnGraph.hpp:496-518
struct ScoreInterval
{
	uint64_t left, right, current;
	double weight;
	Path P;
	ScoreInterval() {}
	ScoreInterval(uint64_t l, uint64_t r, uint64_t c, double w, Path const & p) : left(l), right(r), current(c), weight(w), P(p) {}
	bool operator<(ScoreInterval const & S) const { return weight > S.weight; }
};

// DebruijnGraph.hpp:526-557
struct SeqPos
{
	uint32_t seq, pos;
	SeqPos() {}
	SeqPos(uint32_t s, uint32_t p) : seq(s), pos(p) {}
	bool operator<(SeqPos const & O) const { if ( pos != O.pos ) return pos < O.pos; else return seq < O.seq; }
};
struct PosFreq { uint32_t pos, freq; PosFreq() {} PosFreq(uint32_t p, uint32_t f) : pos(p), freq(f) {} };
// DebruijnGraph.hpp:565-599
struct LevelAddElement { uint64_t from, to, v, off; LevelAddElement() {} LevelAddElement(uint64_t f, uint64_t t, uint64_t rv, uint64_t o) : from(f), to(t), v(rv), off(o) {} };
struct NodeAddElement
{
	uint64_t v, pos;
	NodeAddElement() {}
	NodeAddElement(uint64_t rv, uint64_t p) : v(rv), pos(p) {}
	bool operator<(NodeAddElement const & O) const { if ( v != O.v ) return v < O.v; else return pos < O.pos; }
};
struct StretchesFirstComparator { bool operator()(Stretch const & A, Stretch const & B) const { return A.first < B.first; } };

// DebruijnGraph.hpp:875-889
struct StretchFeasObject
{
	uint64_t p; double w, wf, wl;
	StretchFeasObject() {}
	StretchFeasObject(uint64_t rp, double rw, double rwf, double rwl) : p(rp), w(rw), wf(rwf), wl(rwl) {}
};

template<typename T>
static inline void vpush(std::vector<T> & V, uint64_t & o, T const & v)
{
	if ( o >= V.size() ) V.resize(std::max<uint64_t>(16,2*V.size()));
	V[o++] = v;
}

struct DebruijnGraph
{
	unsigned int const kmersize;
	double const p;               // est_cor (daccord.cpp:2008), 0 disables the KmerLimit rule
	KmerLimit KL;
	uint64_t const m;

	std::vector<uint64_t> prenodes; uint64_t numprenodes;
	std::vector<uint64_t> last; uint64_t lastn;
	std::vector<uint32_t> seqlen; uint64_t seqlenn;
	uint64_t maxk;
	std::vector<SeqPos> SP, RSP;
	std::vector<PosFreq> PF, RPF;
	std::vector<Node> nodes; uint64_t numnodes;
	FiniteSizeHeap<EdgeActivationElement> EAH;
	std::vector<uint64_t> stretchLinks; uint64_t stretchLinksO;
	uint64_t stretcho; std::vector<Stretch> stretches;
	std::vector<uint8_t> Acons; uint64_t conso;
	std::vector<uint64_t> splitA;
	std::vector<uint64_t> AP; uint64_t APo;
	std::vector<uint64_t> APR; uint64_t APRo;
	std::vector<ConsensusCandidate> ACC; uint64_t ACCo;
	std::vector< std::pair<uint64_t,double> > Afeaspos; uint64_t Afeasposo;
	std::vector< std::pair<uint64_t,double> > Acfeaspos; uint64_t Acfeasposo;
	std::vector<LevelAddElement> LS;
	std::vector< std::pair<uint64_t,double> > Atmpp;
	std::vector<NodeAddElement> ANE;
	std::vector< FiniteSizeHeap<Path,PathWeightComparator> > APQ;
	FiniteSizeHeap<ScoreInterval> SIQ;
	std::vector<StretchFeasObject> Astretchfeas, Acstretchfeas;
	std::vector< std::vector<double> > Afeasbuck;
	std::vector< std::pair<uint64_t,uint64_t> > reverseStretchLinks; uint64_t reverseStretchLinksO;
	std::vector<ReversePath> ARP; uint64_t ARPo;
	std::vector< std::pair<double,uint64_t> > ARWT;
	std::vector<uint64_t> ARW, ARWR;
	uint64_t maxkmerpos, maxstretchlength;
	std::vector< FiniteSizeHeap<ReversePath,ReversePathWeightHeapComparator> > ARPH;
	FiniteSizeHeap<ReversePath,ReversePathWeightQueueHeapComparator> RPST;
	FiniteSizeHeap<ConsensusCandidate,ConsensusCandidateDumpHeapComparator> CDH;
	FiniteSizeHeap<ConsensusCandidate,ConsensusCandidateHeapComparator> CH;
	uint64_t maxsupto;
	Aligner SNP;
	std::vector<uint32_t> edtmp;
	std::vector< std::pair<uint64_t,uint64_t> > maxFirst, maxLast;

	// node cache: direct array for k <= 12 (as the reference), open addressing above
	bool const directcache;
	std::vector<int32_t> nodecache;
	std::vector<uint64_t> hkeys; std::vector<int32_t> hvals; uint64_t hmask;

	static uint64_t lowbits(unsigned int const b) { return b >= 64 ? ~0ull : ((1ull<<b)-1); }

	// DebruijnGraph.hpp:2358-2384
	DebruijnGraph(unsigned int const k, double const rp, KmerLimit const & rKL)
	: kmersize(k), p(rp), KL(rKL), m(lowbits(2*k)), numprenodes(0), lastn(0), seqlenn(0), maxk(0), numnodes(0),
	  EAH(1024), stretchLinksO(0), stretcho(0), conso(0), APo(0), APRo(0), ACCo(0), Afeasposo(0), Acfeasposo(0),
	  SIQ(1024), reverseStretchLinksO(0), ARPo(0), maxkmerpos(0), maxstretchlength(0), RPST(1024), CDH(16), CH(16), maxsupto(0),
	  directcache(k <= 12), hmask(0)
	{
		if ( directcache ) nodecache.assign(1ull<<(2*k),-1);
		else { hkeys.assign(1024,~0ull); hvals.assign(1024,-1); hmask = 1023; }
		// the restatement takes &X[0] of these arrays; keep them non-empty
		stretches.resize(16); ARP.resize(16); Astretchfeas.resize(16); Acstretchfeas.resize(16);
		PF.resize(16); RPF.resize(16); Acons.resize(64); AP.resize(16); APR.resize(16);
		reverseStretchLinks.resize(16);
	}

	uint64_t getKmerSize() const { return kmersize; }

	static uint64_t hashk(uint64_t v) { v *= 0x9E3779B97F4A7C15ull; return v >> 20; }

	int64_t getNodeId(uint64_t const v) const
	{
		if ( directcache ) return nodecache[v];
		uint64_t h = hashk(v) & hmask;
		while ( hkeys[h] != ~0ull )
		{
			if ( hkeys[h] == v ) return hvals[h];
			h = (h+1)&hmask;
		}
		return -1;
	}
	Node const * getNode(uint64_t const v) const { int64_t const j = getNodeId(v); return j < 0 ? 0 : &nodes[j]; }
	Node * getNode(uint64_t const v) { int64_t const j = getNodeId(v); return j < 0 ? 0 : &nodes[j]; }

	// DebruijnGraph.hpp:1163-1178
	void clearNodeCache()
	{
		if ( directcache )
			for ( uint64_t i = 0; i < numnodes; ++i ) nodecache[nodes[i].v] = -1;
		else
		{
			std::fill(hkeys.begin(),hkeys.end(),~0ull);
		}
	}
	void setupNodeCache()
	{
		if ( directcache )
			for ( uint64_t i = 0; i < numnodes; ++i ) nodecache[nodes[i].v] = i;
		else
		{
			uint64_t sz = 1024;
			while ( sz < 4*numnodes ) sz <<= 1;
			hkeys.assign(sz,~0ull); hvals.assign(sz,-1); hmask = sz-1;
			for ( uint64_t i = 0; i < numnodes; ++i )
			{
				uint64_t h = hashk(nodes[i].v) & hmask;
				while ( hkeys[h] != ~0ull ) h = (h+1)&hmask;
				hkeys[h] = nodes[i].v; hvals[h] = i;
			}
		}
	}

	// DebruijnGraph.hpp:1209-1235
	static uint64_t combine(uint64_t const v, uint64_t const seq, uint64_t const pos) { return (v<<32)|(seq)|(pos<<16); }
	static uint64_t kmerMask(uint64_t const v) { return v>>32; }
	static uint64_t seqMask(uint64_t const v) { return v & 0xFFFFull; }
	static uint64_t posMask(uint64_t const v) { return (v>>16)&0xFFFFull; }

	uint64_t count(uint64_t const v) const { Node const * n = getNode(v); return n ? n->freq : 0; }

	// DebruijnGraph.hpp:2413-2426
	void getSuccessors(uint64_t const v, Links & L) const
	{
		L.reset();
		uint64_t const masked = (v<<2)&m;
		for ( uint64_t i = 0; i < 4; ++i )
			L.push(i,count(masked|i));
		L.sort();
	}
	// DebruijnGraph.hpp:2434-2457
	void getActiveSuccessors(uint64_t const v, Links & L) const
	{
		L.reset();
		Node const * node = getNode(v);
		if ( node ) { getSuccessors(v,L); L.setSize(node->numsuccactive); }
		else L.setSize(0);
	}
	uint64_t getUniqueActiveSuccessor(uint64_t const v) const { Links L; getActiveSuccessors(v,L); return ((v<<2)&m)|L.getSym(0); }
	uint64_t getNumActiveSuccessors(uint64_t const v) const { Links L; getActiveSuccessors(v,L); return L.size(); }
	// DebruijnGraph.hpp:2484-2500
	bool isEdgeActive(uint64_t const from, uint64_t const to) const
	{
		Links L; getActiveSuccessors(from,L);
		uint64_t const masked = (from<<2)&m;
		for ( uint64_t i = 0; i < L.size(); ++i )
			if ( (masked|L.getSym(i)) == to ) return true;
		return false;
	}
	// DebruijnGraph.hpp:2530-2581
	void getPredecessors(uint64_t const v, Links & L) const
	{
		L.reset();
		uint64_t const masked = (v>>2)&m;
		unsigned int const shift = 2*(kmersize-1);
		for ( uint64_t i = 0; i < 4; ++i )
			L.push(i,count(masked|(i<<shift)));
		L.sort();
	}
	void getActivePredecessors(uint64_t const v, Links & L) const
	{
		L.reset();
		Node const * node = getNode(v);
		if ( node )
		{
			uint64_t const masked = (v>>2)&m;
			unsigned int const shift = 2*(kmersize-1);
			getPredecessors(v,L);
			uint64_t o = 0;
			for ( uint64_t i = 0; i < L.size(); ++i )
			{
				uint64_t const prev = masked | (L.getSym(i)<<shift);
				if ( isEdgeActive(prev,v) )
					L.A[o++] = L.A[i];
			}
			L.setSize(o);
		}
	}
	uint64_t getNumActivePredecessors(uint64_t const v) const { Links L; getActivePredecessors(v,L); return L.size(); }

	// DebruijnGraph.hpp:2018-2304; the LSD radix passes (:2197-2290) produce the fully
	// ascending 64-bit order asserted at :2294-2295, so a plain sort is equivalent
	void setupPreNodes(StringRef const * I, uint64_t const o)
	{
		numprenodes = 0; lastn = 0; seqlenn = 0; maxk = 0;
		if ( kmersize )
		{
			for ( uint64_t j = 0; j < o; ++j )
			{
				if ( I[j].second >= kmersize )
				{
					uint64_t const numk = I[j].second-kmersize+1;
					uint8_t const * u = I[j].first;
					uint64_t v = 0;
					for ( unsigned int i = 0; i < kmersize-1; ++i ) { v <<= 2; v |= mapChar(*(u++)); }
					for ( uint64_t i = 0; i < numk; ++i )
					{
						v <<= 2; v &= m; v |= mapChar(*(u++));
						vpush(prenodes,numprenodes,combine(v,j,i));
					}
					vpush(last,lastn,combine(v,j,numk-1));
					maxk = std::max(maxk,numk);
				}
				vpush(seqlen,seqlenn,static_cast<uint32_t>(I[j].second));
			}
			std::sort(prenodes.begin(),prenodes.begin()+numprenodes);
			std::sort(last.begin(),last.begin()+lastn);
		}
	}

	// DebruijnGraph.hpp:1899-1916
	void setupFeasBuckets(uint64_t const mpos)
	{
		if ( Afeasbuck.size() < mpos+1 ) Afeasbuck.resize(mpos+1);
	}

	// DebruijnGraph.hpp:1918-2014
	void setupNodes()
	{
		clearNodeCache();
		numnodes = 0;
		uint64_t l = 0, spo = 0, rspo = 0, pfo = 0, rpfo = 0;
		maxkmerpos = 0;
		while ( l < numprenodes )
		{
			uint64_t h = l, li = l;
			uint64_t lp = posMask(prenodes[l]);
			uint64_t const pfostart = pfo, rpfostart = rpfo;
			while ( h < numprenodes && kmerMask(prenodes[h]) == kmerMask(prenodes[l]) )
			{
				uint64_t const seq = seqMask(prenodes[h]);
				uint64_t const pos = posMask(prenodes[h]);
				if ( pos != lp )
				{
					vpush(PF,pfo,PosFreq(lp,h-li));
					lp = pos; li = h;
				}
				vpush(SP,spo,SeqPos(seq,pos));
				assert ( pos+kmersize <= seqlen[seq] );
				vpush(RSP,rspo,SeqPos(seq,seqlen[seq]-pos-kmersize));
				++h;
			}
			vpush(PF,pfo,PosFreq(lp,h-li));
			uint64_t const freq = h-l;
			std::sort(RSP.begin()+rspo-freq,RSP.begin()+rspo);
			uint64_t cl = rspo-freq;
			while ( cl < rspo )
			{
				uint64_t ch = cl+1;
				while ( ch < rspo && RSP[ch].pos == RSP[cl].pos ) ++ch;
				vpush(RPF,rpfo,PosFreq(RSP[cl].pos,ch-cl));
				cl = ch;
			}
			Node node;
			node.v = kmerMask(prenodes[l]); node.spo = spo-freq; node.freq = freq;
			node.numsucc = 0; node.numsuccactive = 0; node.feaspos = 0; node.cfeaspos = 0; node.numfeaspos = 0; node.numcfeaspos = 0;
			node.pfostart = pfostart; node.pfosize = pfo-pfostart; node.cpfostart = rpfostart; node.cpfosize = rpfo-rpfostart;
			node.plow = PF[pfostart].pos; node.phigh = PF[pfo-1].pos; node.cplow = RPF[rpfostart].pos; node.cphigh = RPF[rpfo-1].pos;
			vpush(nodes,numnodes,node);
			maxkmerpos = std::max(maxkmerpos,std::max(node.cphigh,node.phigh));
			l = h;
		}
		setupFeasBuckets(maxkmerpos);
		setupNodeCache();
	}

	// DebruijnGraph.hpp:1770-1814
	void setNodesActive(bool const check, uint64_t const lim)
	{
		Links L;
		for ( uint64_t z = 0; z < numnodes; ++z )
		{
			Node & node = nodes[z];
			getSuccessors(node.v,L);
			if ( L.size() )
			{
				node.numsucc = L.size();
				node.numsuccactive = 1;
				while ( node.numsuccactive < L.size() &&
					( (L.getFreq(node.numsuccactive) >= L.getFreq(0)/2) || (check && (L.getFreq(node.numsuccactive) >= lim)) ) )
					++node.numsuccactive;
			}
			else { node.numsucc = 0; node.numsuccactive = 0; }
		}
	}
	// DebruijnGraph.hpp:1818-1859
	void setupAddHeap(uint64_t const no)
	{
		for ( uint64_t z = 0; z < numnodes; ++z ) { nodes[z].numsucc = 0; nodes[z].numsuccactive = 0; }
		if ( p ) setNodesActive(true,static_cast<uint64_t>(KL.getLimit(no)));
		else setNodesActive(false,0);
		EAH.clear();
		Links L;
		for ( uint64_t z = 0; z < numnodes; ++z )
		{
			Node const & node = nodes[z];
			getSuccessors(node.v,L);
			for ( uint64_t i = node.numsuccactive; i < L.size(); ++i )
				EAH.pushBump(EdgeActivationElement(L.getFreq(i),z,i));
		}
	}
	// DebruijnGraph.hpp:1861-1897
	bool addNextFromHeap()
	{
		if ( EAH.empty() ) return false;
		uint64_t const topfreq = EAH.top().freq;
		while ( !EAH.empty() && EAH.top().freq == topfreq )
		{
			EdgeActivationElement const EAE = EAH.pop();
			nodes[EAE.nodeid].numsuccactive += 1;
		}
		return true;
	}

	// DebruijnGraph.hpp:2307-2331
	void setup(StringRef const * I, uint64_t const o)
	{
		stretcho = 0;
		clearNodeCache();
		numnodes = 0;
		EAH.clear();
		numprenodes = 0; lastn = 0; seqlenn = 0; maxk = 0;
		setupPreNodes(I,o);
		setupNodes();
		setupAddHeap(o);
	}

	// DebruijnGraph.hpp:1181-1197
	void filterFreq(uint64_t const f, uint64_t const no)
	{
		clearNodeCache();
		uint64_t o = 0;
		for ( uint64_t i = 0; i < numnodes; ++i )
			if ( nodes[i].freq >= f )
				nodes[o++] = nodes[i];
		numnodes = o;
		setupNodeCache();
		setupAddHeap(no);
	}

	// DebruijnGraph.hpp:3826-3864 (VS fixed point, see o_offsetlikely.hpp header note)
	double getKmerPositionWeight(uint64_t const kmer, uint64_t const pp, OffsetLikely const & OL) const
	{
		if ( pp >= OL.size() ) return 0;
		Node const * node = getNode(kmer);
		if ( ! node ) return 0.0;
		DotProduct const & DP = OL.DPnormSquare[pp];
		PosFreq const * q = &PF[0] + node->pfostart;
		PosFreq const * qe = q + node->pfosize;
		while ( q != qe && q->pos < DP.firstsign ) ++q;
		uint64_t const e = DP.firstsign + DP.V.size();
		uint64_t uprr = 0;
		for ( ; q != qe && q->pos < e; ++q )
			uprr += q->freq * DP.VS[q->pos-DP.firstsign];
		return static_cast<double>(uprr) / DotProduct::getMult();
	}
	// DebruijnGraph.hpp:3866-3904
	double getKmerReversePositionWeight(uint64_t const kmer, uint64_t const pp, OffsetLikely const & OL) const
	{
		if ( pp >= OL.size() ) return 0;
		Node const * node = getNode(kmer);
		if ( ! node ) return 0.0;
		DotProduct const & DP = OL.DPnormSquare[pp];
		PosFreq const * q = &RPF[0] + node->cpfostart;
		PosFreq const * qe = q + node->cpfosize;
		while ( q != qe && q->pos < DP.firstsign ) ++q;
		uint64_t const e = DP.firstsign + DP.V.size();
		uint64_t uprr = 0;
		for ( ; q != qe && q->pos < e; ++q )
			uprr += q->freq * DP.VS[q->pos-DP.firstsign];
		return static_cast<double>(uprr) / DotProduct::getMult();
	}

	// DebruijnGraph.hpp:3117-3174
	void computeFeasibleKmerPositions(OffsetLikely const & OL, double const thres)
	{
		Afeasposo = 0; Acfeasposo = 0; maxsupto = 0;
		for ( uint64_t i = 0; i < numnodes; ++i )
		{
			Node & node = nodes[i];
			node.feaspos = Afeasposo; node.cfeaspos = Acfeasposo;
			uint64_t const kmer = node.v;
			uint64_t const pfrom = OL.getSupportLow(node.plow);
			uint64_t const pto = OL.getSupportHigh(node.phigh);
			maxsupto = std::max(maxsupto,pto);
			for ( uint64_t q = pfrom; q < pto; ++q )
			{
				double const weight = getKmerPositionWeight(kmer,q,OL);
				if ( weight >= thres ) vpush(Afeaspos,Afeasposo,std::pair<uint64_t,double>(q,weight));
			}
			uint64_t const cpfrom = OL.getSupportLow(node.cplow);
			uint64_t const cpto = OL.getSupportHigh(node.cphigh);
			maxsupto = std::max(maxsupto,cpto);
			for ( uint64_t q = cpfrom; q < cpto; ++q )
			{
				double const weight = getKmerReversePositionWeight(kmer,q,OL);
				if ( weight >= thres ) vpush(Acfeaspos,Acfeasposo,std::pair<uint64_t,double>(q,weight));
			}
			node.numfeaspos = Afeasposo-node.feaspos;
			node.numcfeaspos = Acfeasposo-node.cfeaspos;
		}
	}

	// DebruijnGraph.hpp:1016-1071
	uint64_t getLevelSuccessors(uint64_t const v, unsigned int const s, std::vector<LevelAddElement> & A, uint64_t o) const
	{
		uint64_t const low = (v<<(2*s))&m;
		uint64_t const high = low | lowbits(2*s);
		uint64_t il = 0;
		while ( il < numnodes && nodes[il].v < low ) ++il;      // lower_bound
		uint64_t ih = il;
		while ( ih < numnodes && nodes[ih].v <= high ) ++ih;    // upper_bound
		for ( uint64_t np = il; np < ih; ++np )
		{
			uint64_t const nv = nodes[np].v;
			for ( unsigned int i = 1; i < s; ++i )
			{
				uint64_t const vhigh = (v<<(2*i))&m;
				uint64_t const vlow = nv >> ((s-i)*2);
				uint64_t const cv = vlow|vhigh;
				if ( ! getNode(cv) )
					vpush(A,o,LevelAddElement(v,nv,cv,i));
			}
		}
		return o;
	}
	// DebruijnGraph.hpp:1073-1161
	void getLevelSuccessors(unsigned int const s)
	{
		uint64_t o = 0;
		for ( uint64_t i = 0; i < numnodes; ++i )
			o = getLevelSuccessors(nodes[i].v,s,LS,o);
		uint64_t aneo = 0;
		for ( uint64_t i = 0; i < o; ++i )
		{
			LevelAddElement const & L = LS[i];
			Node const & from = *getNode(L.from);
			Node const & to = *getNode(L.to);
			uint64_t tmpo = 0;
			for ( uint64_t j = 0; j < from.numfeaspos; ++j )
				vpush(Atmpp,tmpo,std::pair<uint64_t,double>(Afeaspos[from.feaspos+j].first+s,Afeaspos[from.feaspos+j].second));
			for ( uint64_t j = 0; j < to.numfeaspos; ++j )
				vpush(Atmpp,tmpo,Afeaspos[to.feaspos+j]);
			std::sort(Atmpp.begin(),Atmpp.begin()+tmpo);
			uint64_t l = 0, mp = 0;
			double mweight = std::numeric_limits<double>::min();
			while ( l < tmpo )
			{
				uint64_t h = l+1;
				while ( h < tmpo && Atmpp[l].first == Atmpp[h].first ) ++h;
				if ( h-l > 1 )
				{
					uint64_t const pp = Atmpp[l].first;
					uint64_t const q = pp - s + L.off;
					double const weight = Atmpp[l].second + Atmpp[h-1].second;
					if ( weight > mweight ) { mweight = weight; mp = q; }
				}
				l = h;
			}
			if ( mweight != std::numeric_limits<double>::min() )
				vpush(ANE,aneo,NodeAddElement(L.v,mp));
		}
		std::sort(ANE.begin(),ANE.begin()+aneo);
		for ( uint64_t i = 0; i < aneo; ++i )
		{
			int64_t seqid = -1;
			for ( uint64_t j = 0; j < seqlenn && seqid < 0; ++j )
				if ( ANE[i].pos+kmersize <= seqlen[j] )
					seqid = j;
			if ( seqid != -1 )
				vpush(prenodes,numprenodes,combine(ANE[i].v,seqid,ANE[i].pos));
		}
		std::sort(prenodes.begin(),prenodes.begin()+numprenodes);
	}

	// DebruijnGraph.hpp:1256-1277
	uint64_t maxForPos(uint64_t const q) const
	{
		uint64_t maxv = 0, maxc = 0;
		for ( uint64_t i = 0; i < numnodes; ++i )
		{
			Node const & node = nodes[i];
			uint64_t c = 0;
			for ( uint64_t j = 0; j < node.freq; ++j )
				if ( SP[node.spo+j].pos == q ) ++c;
			if ( c > maxc ) { maxc = c; maxv = node.v; }
		}
		return maxv;
	}
	// DebruijnGraph.hpp:1329-1358
	uint64_t maxLastWord() const
	{
		uint64_t maxv = 0, maxc = 0, l = 0;
		while ( l < lastn )
		{
			uint64_t c = 1, h = l+1;
			while ( h < lastn && (last[h]>>32) == (last[l]>>32) ) { ++c; ++h; }
			if ( c > maxc ) { maxv = (last[l]>>32); maxc = c; }
			l = h;
		}
		return maxv;
	}
	// DebruijnGraph.hpp:3794-3824 with the stretch part of prepareTraverse(false,false,first,last,0,0,false) (:3541-3566);
	// the feasibility / enumeration part of that call does not touch the stretches
	bool traverseTrivial()
	{
		conso = 0;
		uint64_t const first = maxForPos(0);
		uint64_t const lastw = maxLastWord();
		computeStretches(false);
		splitStretches(first);
		splitStretches(lastw);
		stretchesUnique();
		for ( uint64_t i = 0; i < stretcho; ++i )
			if ( stretches[i].first == first && stretches[i].last == lastw )
			{
				consPushWord(first,Acons,conso);
				for ( uint64_t j = 1; j < stretches[i].len; ++j )
					vpush(Acons,conso,remapChar(stretchLinks[stretches[i].stretchO+j]&3));
				return true;
			}
		return false;
	}
	std::string getConsensus() const { return std::string(Acons.begin(),Acons.begin()+conso); }

	// DebruijnGraph.hpp:1280-1304
	uint64_t maxForPosList(uint64_t const q, std::vector< std::pair<uint64_t,uint64_t> > & PL) const
	{
		uint64_t PLo = 0;
		for ( uint64_t i = 0; i < numnodes; ++i )
		{
			Node const & node = nodes[i];
			uint64_t c = 0;
			for ( uint64_t j = 0; j < node.freq; ++j )
				if ( SP[node.spo+j].pos == q ) ++c;
			if ( c ) vpush(PL,PLo,std::pair<uint64_t,uint64_t>(c,node.v));
		}
		std::sort(PL.begin(),PL.begin()+PLo,std::greater< std::pair<uint64_t,uint64_t> >());
		return PLo;
	}
	// DebruijnGraph.hpp:1360-1391
	uint64_t maxLastList(std::vector< std::pair<uint64_t,uint64_t> > & PL) const
	{
		uint64_t PLo = 0, l = 0;
		while ( l < lastn )
		{
			uint64_t h = l+1;
			while ( h < lastn && (last[h]>>32) == (last[l]>>32) ) ++h;
			vpush(PL,PLo,std::pair<uint64_t,uint64_t>(h-l,last[l]>>32));
			l = h;
		}
		std::sort(PL.begin(),PL.begin()+PLo,std::greater< std::pair<uint64_t,uint64_t> >());
		return PLo;
	}

	// DebruijnGraph.hpp:2592-2619
	void copyStretch(uint64_t const low, uint64_t const high)
	{
		uint64_t const first = stretchLinks[low], firstext = stretchLinks[low+1], lastw = stretchLinks[high-1];
		uint64_t const len = high-low;
		uint64_t const start = stretchLinksO;
		for ( uint64_t i = low; i < high; ++i )
		{
			uint64_t const link = stretchLinks[i];
			vpush(stretchLinks,stretchLinksO,link);
		}
		vpush(stretches,stretcho,Stretch(first,firstext,lastw,len,start));
	}

	// DebruijnGraph.hpp:2772-2841
	void splitStretches(uint64_t const v)
	{
		uint64_t splito = 0;
		uint64_t const loopend = stretcho;
		for ( uint64_t z = 0; z < loopend; ++z )
		{
			Stretch const stretch = stretches[z];
			uint64_t const start = stretch.stretchO, len = stretch.len;
			int64_t splitindex = -1;
			for ( uint64_t i = 1; i+1 < len; ++i )
				if ( stretchLinks[start+i] == v ) { splitindex = i; break; }
			if ( splitindex != -1 )
			{
				copyStretch(start,start+splitindex+1);
				copyStretch(start+splitindex,start+len);
				vpush(splitA,splito,z);
			}
		}
		uint64_t l = 0, idx = 0, o = 0;
		for ( ; idx < splito; ++l )
		{
			if ( l == splitA[idx] ) ++idx;
			else stretches[o++] = stretches[l];
		}
		while ( l < stretcho ) stretches[o++] = stretches[l++];
		stretcho = o;
	}

	// DebruijnGraph.hpp:2844-2986
	void computeStretches(bool const checkpredecessors)
	{
		std::vector<uint8_t> stretchBV(numnodes,0);
		stretchLinksO = 0; stretcho = 0; maxstretchlength = 0;
		for ( uint64_t z = 0; z < numnodes; ++z )
		{
			Node const & node = nodes[z];
			uint64_t const refk = node.v;
			uint64_t const numpred = getNumActivePredecessors(refk);
			uint64_t const numsucc = node.numsuccactive;
			if ( numsucc && (numpred != 1 || numsucc > 1) )
			{
				Links L;
				getActiveSuccessors(refk,L);
				for ( uint64_t i = 0; i < numsucc; ++i )
				{
					uint64_t const start = stretchLinksO;
					uint64_t const first = refk;
					uint64_t const firstext = ((refk<<2)&m)|L.getSym(i);
					uint64_t extk = firstext;
					vpush(stretchLinks,stretchLinksO,refk);
					stretchBV[getNodeId(refk)] = 1;
					vpush(stretchLinks,stretchLinksO,extk);
					stretchBV[getNodeId(extk)] = 1;
					uint64_t len = 2;
					bool loop = (refk == extk);
					while ( !loop && getNumActiveSuccessors(extk) == 1 && ( !checkpredecessors || getNumActivePredecessors(extk) == 1 ) )
					{
						extk = getUniqueActiveSuccessor(extk);
						vpush(stretchLinks,stretchLinksO,extk);
						len += 1;
						uint64_t const extid = getNodeId(extk);
						if ( stretchBV[extid] ) loop = true;
						else stretchBV[extid] = 1;
					}
					uint64_t const lastw = extk;
					for ( uint64_t j = start; j < start+len; ++j )
						stretchBV[getNodeId(stretchLinks[j])] = 0;
					if ( loop && first != lastw )
					{
						uint64_t j = 0;
						while ( stretchLinks[start+j] != lastw ) ++j;
						j += 1;
						uint64_t const retract = len-j;
						len -= retract;
						stretchLinksO -= retract;
					}
					maxstretchlength = std::max(maxstretchlength,len);
					vpush(stretches,stretcho,Stretch(first,firstext,lastw,len,start));
				}
			}
		}
	}

	// DebruijnGraph.hpp:3087-3114
	void stretchesUnique()
	{
		std::sort(stretches.begin(),stretches.begin()+stretcho);
		stretcho = std::unique(stretches.begin(),stretches.begin()+stretcho) - stretches.begin();
		uint64_t l = 0, o = 0;
		while ( l < stretcho )
		{
			uint64_t h = l+1;
			while ( h < stretcho && stretches[h].first == stretches[l].first && stretches[h].ext == stretches[l].ext ) ++h;
			stretches[o++] = stretches[l];
			l = h;
		}
		stretcho = o;
	}

	// DebruijnGraph.hpp:3176-3330.  The bucket/bit-vector machinery visits bucket indices in
	// ascending order and sums the pushed weights in push order; we keep exactly that.
	void computeFeasibleStretchPositions()
	{
		uint64_t Astretchfeaso = 0, Acstretchfeaso = 0;
		for ( uint64_t i = 0; i < stretcho; ++i )
		{
			Stretch & stretch = stretches[i];
			uint64_t const len = stretch.len;
			{
				stretch.feasposO = Astretchfeaso; stretch.feasposL = 0;
				for ( uint64_t j = 0; j < len; ++j )
				{
					Node const * node = getNode(stretchLinks[stretch.stretchO+j]);
					uint64_t const poff = len-j-1;
					for ( uint64_t q = 0; q < node->numfeaspos; ++q )
					{
						std::pair<uint64_t,double> const & FP = Afeaspos[node->feaspos+q];
						Afeasbuck[FP.first+poff].push_back(FP.second);
					}
				}
				for ( uint64_t zz = 0; zz < Afeasbuck.size(); ++zz )
					if ( ! Afeasbuck[zz].empty() )
					{
						std::vector<double> const & A = Afeasbuck[zz];
						if ( A.size() == len && zz >= len-1 )
						{
							double weight = 0.0;
							for ( uint64_t q = 0; q < len; ++q ) weight += A[q];
							vpush(Astretchfeas,Astretchfeaso,StretchFeasObject(zz-(len-1),weight,A[0],A[len-1]));
							stretch.feasposL += 1;
						}
						Afeasbuck[zz].clear();
					}
			}
			{
				stretch.cfeasposO = Acstretchfeaso; stretch.cfeasposL = 0;
				for ( uint64_t jj = 0; jj < len; ++jj )
				{
					uint64_t const j = len-jj-1;
					Node const * node = getNode(stretchLinks[stretch.stretchO+j]);
					uint64_t const poff = len-jj-1;
					for ( uint64_t q = 0; q < node->numcfeaspos; ++q )
					{
						std::pair<uint64_t,double> const & FP = Acfeaspos[node->cfeaspos+q];
						Afeasbuck[FP.first+poff].push_back(FP.second);
					}
				}
				for ( uint64_t zz = 0; zz < Afeasbuck.size(); ++zz )
					if ( ! Afeasbuck[zz].empty() )
					{
						std::vector<double> const & A = Afeasbuck[zz];
						if ( A.size() == len && zz >= len-1 )
						{
							double weight = 0.0;
							for ( uint64_t q = 0; q < len; ++q ) weight += A[q];
							vpush(Acstretchfeas,Acstretchfeaso,StretchFeasObject(zz-(len-1),weight,A[0],A[len-1]));
							stretch.cfeasposL += 1;
						}
						Afeasbuck[zz].clear();
					}
			}
		}
	}

	// DebruijnGraph.hpp:3388-3440
	double getReverseStretchLinkWeight(Stretch const & A, Stretch const & B)
	{
		uint64_t const shift = B.len-1;
		StretchFeasObject const * Pa = &Acstretchfeas[0] + B.cfeasposO;
		StretchFeasObject const * Pe = Pa + B.cfeasposL;
		for ( ; Pa != Pe; ++Pa )
			Afeasbuck[Pa->p+shift].push_back(Pa->w);
		Pa = &Acstretchfeas[0] + A.cfeasposO;
		Pe = Pa + A.cfeasposL;
		for ( ; Pa != Pe; ++Pa )
			Afeasbuck[Pa->p].push_back(Pa->w - Pa->wf);
		double weight = 0.0;
		for ( uint64_t zz = 0; zz < Afeasbuck.size(); ++zz )
			if ( ! Afeasbuck[zz].empty() )
			{
				if ( Afeasbuck[zz].size() == 2 )
				{
					double const lweight = Afeasbuck[zz][0] + Afeasbuck[zz][1];
					weight = std::max(weight,lweight);
				}
				Afeasbuck[zz].clear();
			}
		return weight;
	}

	// DebruijnGraph.hpp:3442-3480
	void computeStretchLinks()
	{
		reverseStretchLinksO = 0;
		if ( Acstretchfeas.empty() ) Acstretchfeas.resize(1);
		for ( uint64_t i = 0; i < stretcho; ++i )
		{
			Stretch const & stretch = stretches[i];
			Stretch ref; ref.first = stretch.last;
			std::pair<Stretch *,Stretch *> ER = std::equal_range(&stretches[0],&stretches[0]+stretcho,ref,StretchesFirstComparator());
			for ( Stretch * q = ER.first; q != ER.second; ++q )
			{
				double const rweight = getReverseStretchLinkWeight(stretch,*q);
				uint64_t const linkid = q-&stretches[0];
				if ( rweight >= 1e-1 )
					vpush(reverseStretchLinks,reverseStretchLinksO,std::pair<uint64_t,uint64_t>(linkid,i));
			}
		}
		std::sort(reverseStretchLinks.begin(),reverseStretchLinks.begin()+reverseStretchLinksO);
	}

	// DebruijnGraph.hpp:3906-3934
	StretchFeasObject const * getCachedStretchPositionWeight(uint64_t const stretchid, uint64_t const q) const
	{
		if ( Astretchfeas.empty() ) return 0;
		StretchFeasObject const * a = &Astretchfeas[0] + stretches[stretchid].feasposO;
		StretchFeasObject const * e = a + stretches[stretchid].feasposL;
		while ( a != e && a->p < q ) ++a;
		return ( a != e && a->p == q ) ? a : 0;
	}
	StretchFeasObject const * getCachedStretchReversePositionWeight(uint64_t const stretchid, uint64_t const q) const
	{
		if ( Acstretchfeas.empty() ) return 0;
		StretchFeasObject const * a = &Acstretchfeas[0] + stretches[stretchid].cfeasposO;
		StretchFeasObject const * e = a + stretches[stretchid].cfeasposL;
		while ( a != e && a->p < q ) ++a;
		return ( a != e && a->p == q ) ? a : 0;
	}

	// DebruijnGraph.hpp:3936-3971
	Path copyPath(Path const & P)
	{
		uint64_t const o = APo;
		for ( uint64_t i = 0; i < P.len; ++i ) { uint64_t const v = AP[P.off+i]; vpush(AP,APo,v); }
		return Path(P.len,o,P.pos,P.weight,P.baselen);
	}
	ReversePath copyReversePath(ReversePath const & P)
	{
		uint64_t const o = APRo;
		for ( uint64_t i = 0; i < P.len; ++i ) { uint64_t const v = APR[P.linkoff+i]; vpush(APR,APRo,v); }
		return ReversePath(P.len,o,P.pos,P.weight,P.front,P.baselen);
	}
	// DebruijnGraph.hpp:3989-4056
	Path extendPath(Path const P, uint64_t const stretchid)
	{
		Path NP = copyPath(P);
		StretchFeasObject const * SFO = getCachedStretchPositionWeight(stretchid,P.pos);
		vpush(AP,APo,stretchid);
		NP.len += 1;
		if ( NP.len == 1 )
		{
			NP.baselen = stretches[stretchid].len+kmersize-1;
			NP.weight = SFO ? SFO->w : 0;
		}
		else
		{
			NP.baselen += stretches[stretchid].len-1;
			if ( SFO ) NP.weight += SFO->w - SFO->wf;
		}
		NP.pos += (stretches[stretchid].len-1);
		return NP;
	}
	// DebruijnGraph.hpp:4058-4105
	ReversePath extendReversePath(ReversePath const P, uint64_t const stretchid)
	{
		ReversePath NP = copyReversePath(P);
		StretchFeasObject const * SFO = getCachedStretchReversePositionWeight(stretchid,P.pos);
		vpush(APR,APRo,stretchid);
		NP.len += 1;
		if ( NP.len == 1 )
		{
			NP.baselen = stretches[stretchid].len+kmersize-1;
			NP.weight = SFO ? SFO->w : 0.0;
		}
		else
		{
			NP.baselen += stretches[stretchid].len-1;
			if ( SFO ) NP.weight += SFO->w - SFO->wf;
		}
		NP.pos += stretches[stretchid].len-1;
		NP.front = stretches[stretchid].first;
		return NP;
	}
	// DebruijnGraph.hpp:4130-4159
	bool checkReversePathFeasiblePosition(ReversePath const & RP) const
	{
		if ( RP.len )
		{
			uint64_t const laststretchid = APR[RP.linkoff+RP.len-1];
			Stretch const & laststretch = stretches[laststretchid];
			uint64_t const checkpos = RP.pos-(laststretch.len-1);
			for ( uint64_t i = 0; i < laststretch.cfeasposL; ++i )
			{
				StretchFeasObject const & SFO = Acstretchfeas[laststretch.cfeasposO+i];
				if ( SFO.p == checkpos && SFO.w >= 0.5 ) return true;
			}
			return false;
		}
		else return true;
	}

	// DebruijnGraph.hpp:3482-3497
	double getPairScore(Path const & P, ReversePath const & RP) const
	{
		uint64_t const laststretchidP = AP[P.off+P.len-1];
		Stretch const & laststretchP = stretches[laststretchidP];
		uint64_t const laststretchPpos = P.pos-(laststretchP.len-1);
		StretchFeasObject const * SFO = getCachedStretchPositionWeight(laststretchidP,laststretchPpos);
		if ( SFO ) return P.weight + RP.weight - SFO->wl;
		else return P.weight + RP.weight;
	}
	// DebruijnGraph.hpp:3499-3510: ARWR_RMQ(left,right-1) = index of the minimum of ARWR = maximum of ARW
	ScoreInterval getPrimaryScoreInterval(uint64_t const left, uint64_t const right, Path const & P) const
	{
		uint64_t mi = left;
		for ( uint64_t i = left+1; i < right; ++i )
			if ( ARWR[i] < ARWR[mi] ) mi = i;
		return ScoreInterval(left,right,mi,getPairScore(P,ARP[mi]),P);
	}
	// DebruijnGraph.hpp:3513-3534: rpv(left,right,v-1) = largest value <= v-1 in ARW[left,right),
	// select(u,0) = its position (ARW is a permutation)
	bool nextScoreInterval(ScoreInterval & S)
	{
		uint64_t const v = ARW[S.current];
		if ( v )
		{
			bool found = false; uint64_t bu = 0, bi = 0;
			for ( uint64_t i = S.left; i < S.right; ++i )
				if ( ARW[i] <= v-1 && ( !found || ARW[i] > bu ) ) { found = true; bu = ARW[i]; bi = i; }
			if ( ! found ) return false;
			S.current = bi;
			S.weight = getPairScore(S.P,ARP[S.current]);
			return true;
		}
		else return false;
	}

	// DebruijnGraph.hpp:3541-3787
	void prepareTraverse(bool const checkpredecessors, uint64_t const first, uint64_t const lastw, int64_t const lmax)
	{
		computeStretches(checkpredecessors);
		splitStretches(first);
		splitStretches(lastw);
		stretchesUnique();
		setupFeasBuckets(std::max(maxkmerpos,maxsupto)+maxstretchlength);
		computeFeasibleStretchPositions();
		computeStretchLinks();
		APRo = 0; ARPo = 0;
		for ( uint64_t i = 0; i < ARPH.size(); ++i ) ARPH[i].clear();
		if ( getNode(lastw) )
			RPST.push(ReversePath(0,0,0,0.0,lastw,kmersize));
		while ( !RPST.empty() )
		{
			ReversePath const RP = RPST.top();
			RPST.popvoid();
			uint64_t const srcbaselen = RP.baselen;
			while ( !(srcbaselen < ARPH.size()) )
				ARPH.push_back(FiniteSizeHeap<ReversePath,ReversePathWeightHeapComparator>(12));
			if ( ARPH[srcbaselen].full() )
			{
				if ( RP.weight <= ARPH[srcbaselen].top().weight ) continue;
				else ARPH[srcbaselen].popvoid();
			}
			ARPH[srcbaselen].push(RP);
			vpush(ARP,ARPo,RP);
			if ( RP.len == 0 )
			{
				for ( uint64_t i = 0; i < stretcho; ++i )
					if ( stretches[i].last == lastw )
					{
						ReversePath const RPE = extendReversePath(RP,i);
						if ( checkReversePathFeasiblePosition(RPE) ) RPST.pushBump(RPE);
					}
			}
			else if ( RP.baselen < static_cast<uint64_t>((lmax+1)/2) )
			{
				uint64_t const laststretchid = APR[RP.linkoff+RP.len-1];
				uint64_t el = 0;
				while ( el < reverseStretchLinksO && reverseStretchLinks[el].first < laststretchid ) ++el;
				uint64_t eh = el;
				while ( eh < reverseStretchLinksO && reverseStretchLinks[eh].first == laststretchid ) ++eh;
				for ( uint64_t q = el; q < eh; ++q )
				{
					ReversePath const RPE = extendReversePath(RP,reverseStretchLinks[q].second);
					if ( checkReversePathFeasiblePosition(RPE) ) RPST.pushBump(RPE);
				}
			}
		}
		std::sort(ARP.begin(),ARP.begin()+ARPo);
		ARWT.resize(ARPo);
		for ( uint64_t i = 0; i < ARPo; ++i ) ARWT[i] = std::pair<double,uint64_t>(ARP[i].weight,i);
		std::sort(ARWT.begin(),ARWT.begin()+ARPo);
		ARW.resize(ARPo); ARWR.resize(ARPo);
		for ( uint64_t i = 0; i < ARPo; ++i )
		{
			ARW[ARWT[i].second] = i;
			ARWR[ARWT[i].second] = ARPo-i-1;
		}
	}

	// DebruijnGraph.hpp:1441-1448, 4267-4300
	void consPushWord(uint64_t const w, std::vector<uint8_t> & cons, uint64_t & o) const
	{
		unsigned int shift = 2*(kmersize-1);
		for ( unsigned int i = 0; i < kmersize; ++i, shift -= 2 )
			vpush(cons,o,remapChar((w>>shift)&3));
	}
	uint64_t decodePathPair(Path const & P, ReversePath const & RP, std::vector<uint8_t> & A, uint64_t o) const
	{
		uint64_t const firststretch = AP[P.off];
		consPushWord(stretches[firststretch].first,A,o);
		for ( uint64_t i = 0; i < P.len; ++i )
		{
			Stretch const & stretch = stretches[AP[P.off+i]];
			for ( uint64_t j = 1; j < stretch.len; ++j )
				vpush(A,o,remapChar(stretchLinks[stretch.stretchO+j]&3));
		}
		for ( uint64_t ii = 0; ii < RP.len; ++ii )
		{
			uint64_t const i = RP.len-ii-1;
			Stretch const & stretch = stretches[APR[RP.linkoff+i]];
			for ( uint64_t j = 1; j < stretch.len; ++j )
				vpush(A,o,remapChar(stretchLinks[stretch.stretchO+j]&3));
		}
		return o;
	}

	// DebruijnGraph.hpp:5355-5363
	double getSimpleCandidateError(StringRef const * I, uint64_t const o, uint8_t const * ca, uint8_t const * ce)
	{
		uint64_t s = 0;
		for ( uint64_t j = 0; j < o; ++j )
			s += editDistance(ca,ce-ca,I[j].first,I[j].second,edtmp);
		return s;
	}

	void apqPush(Path const & P)
	{
		while ( !(P.baselen < APQ.size()) )
			APQ.push_back(FiniteSizeHeap<Path,PathWeightComparator>(12));
		if ( APQ[P.baselen].full() )
		{
			if ( P.weight > APQ[P.baselen].top().weight )
			{
				APQ[P.baselen].popvoid();
				APQ[P.baselen].push(P);
			}
		}
		else APQ[P.baselen].push(P);
	}

	// DebruijnGraph.hpp:4496-5170 (stretch based branch, :4769-5097)
	bool traverse(int64_t const lmin, int64_t const lmax, StringRef const * MA, uint64_t const MAo, uint64_t const maxfullpath)
	{
		conso = 0; APo = 0; ACCo = 0;
		CDH.clear();
		uint64_t const maxFirstO = maxForPosList(0,maxFirst);
		uint64_t const maxLastO = maxLastList(maxLast);
		uint64_t const firstthres = maxFirstO ? (maxFirst[0].first*3)/4 : 0;
		uint64_t const lastthres = maxLastO ? (maxLast[0].first*3)/4 : 0;

		for ( uint64_t maxFirstI = 0; maxFirstI < maxFirstO && maxFirst[maxFirstI].first >= firstthres; ++maxFirstI )
			for ( uint64_t maxLastI = 0; maxLastI < maxLastO && maxLast[maxLastI].first >= lastthres; ++maxLastI )
			{
				uint64_t const first = maxFirst[maxFirstI].second;
				uint64_t const lastw = maxLast[maxLastI].second;
				prepareTraverse(true,first,lastw,lmax);
				APo = 0;
				for ( uint64_t i = 0; i < stretcho; ++i )
					if ( stretches[i].first == first )
					{
						Path const P = extendPath(Path(),i);
						apqPush(P);
					}
				SIQ.clear();
				for ( uint64_t zz = 0; zz < APQ.size(); ++zz )
					while ( !APQ[zz].empty() )
					{
						Path const P = APQ[zz].top();
						APQ[zz].popvoid();
						int64_t const candlen = P.pos+kmersize;
						ReversePath RP;
						RP.front = stretches[AP[P.off+P.len-1]].last;
						std::pair<ReversePath const *,ReversePath const *> RPP =
							std::equal_range(&ARP[0],&ARP[0]+ARPo,RP,ReversePathFrontComparator());
						RP.baselen = std::max(lmin+static_cast<int64_t>(kmersize)-candlen,static_cast<int64_t>(0));
						ReversePath const * subRPP = std::lower_bound(RPP.first,RPP.second,RP,ReversePathBaseLenComparator());
						RP.baselen = std::max(lmax+static_cast<int64_t>(kmersize)-candlen,static_cast<int64_t>(0));
						ReversePath const * supRPP = std::upper_bound(subRPP,RPP.second,RP,ReversePathBaseLenComparator());
						if ( subRPP != supRPP )
							SIQ.pushBump(getPrimaryScoreInterval(subRPP-&ARP[0],supRPP-&ARP[0],P));
						uint64_t const Plaststretchid = AP[P.off+P.len-1];
						Stretch ref; ref.first = stretches[Plaststretchid].last;
						std::pair<Stretch *,Stretch *> ER = std::equal_range(&stretches[0],&stretches[0]+stretcho,ref,StretchesFirstComparator());
						if ( P.baselen < kmersize || ( static_cast<int64_t>(P.baselen-kmersize) < ((lmax+1)/2) ) )
						{
							for ( Stretch * q = ER.first; q != ER.second; ++q )
							{
								uint64_t const addstretchid = q-&stretches[0];
								StretchFeasObject const * SFO = getCachedStretchPositionWeight(addstretchid,P.pos);
								double const eweight = SFO ? SFO->w : 0.0;
								double const eweightthres = 0.1;
								if ( eweight > eweightthres )
								{
									Path const EP = extendPath(P,addstretchid);
									if ( EP.weight > eweightthres && static_cast<int64_t>(EP.pos+kmersize) <= lmax )
										apqPush(EP);
								}
							}
						}
					}
				uint64_t prevo = 0, prevlen = std::numeric_limits<uint64_t>::max();
				for ( uint64_t numfullpath = 0; !SIQ.empty() && numfullpath < maxfullpath; ++numfullpath )
				{
					ScoreInterval const SI = SIQ.top();
					SIQ.popvoid();
					ScoreInterval SIC = SI;
					if ( nextScoreInterval(SIC) )
						SIQ.pushBump(SIC);
					Path const & P = SI.P;
					ReversePath const & RP = ARP[SI.current];
					double const weight = SI.weight;
					if ( CDH.full() )
					{
						if ( weight <= CDH.top().weight ) continue;
						else CDH.popvoid();
					}
					uint64_t const consstart = conso;
					conso = decodePathPair(P,RP,Acons,conso);
					uint64_t const conslen = conso-consstart;
					if ( conslen == prevlen && std::equal(Acons.begin()+prevo,Acons.begin()+prevo+prevlen,Acons.begin()+consstart) )
						continue;
					prevo = consstart; prevlen = conslen;
					CDH.push(ConsensusCandidate(consstart,conslen,weight,0.0));
				}
			}
		CH.clear();
		while ( !CDH.empty() ) CH.pushBump(CDH.pop());
		while ( !CH.empty() )
		{
			ConsensusCandidate CC = CH.pop();
			CC.error = getSimpleCandidateError(MA,MAo,&Acons[0]+CC.o,&Acons[0]+CC.o+CC.l);
			vpush(ACC,ACCo,CC);
		}
		std::sort(ACC.begin(),ACC.begin()+ACCo,ConsensusCandidateErrorComparator());
		return ACCo != 0;
	}

	uint64_t getNumCandidates() const { return ACCo; }
	std::pair<uint8_t const *,uint8_t const *> getCandidate(uint64_t const i) const
	{
		uint8_t const * a = &Acons[0] + ACC[i].o;
		return std::pair<uint8_t const *,uint8_t const *>(a,a+ACC[i].l);
	}
	// DebruijnGraph.hpp:5408-5447: sum of the aligner's edit distances (optimal => unique)
	uint64_t getCandidateErrorU(StringRef const * I, uint64_t const o, uint64_t const id)
	{
		std::pair<uint8_t const *,uint8_t const *> const U = getCandidate(id);
		uint64_t s = 0;
		for ( uint64_t j = 0; j < o; ++j )
			s += editDistance(U.first,U.second-U.first,I[j].first,I[j].second,edtmp);
		return s;
	}
	// DebruijnGraph.hpp:5476-5482
	std::pair<uint64_t,uint64_t> checkCandidatesU(StringRef const * I, uint64_t const o)
	{
		if ( getNumCandidates() )
			return std::pair<uint64_t,uint64_t>(0,getCandidateErrorU(I,o,0));
		else
			return std::pair<uint64_t,uint64_t>(0,static_cast<uint64_t>(std::numeric_limits<double>::max()));
	}
};

}
#endif
