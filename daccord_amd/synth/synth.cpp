/*
 * Synthetic pile generator (bench/test input, SURVEY.md 8d "Configs restated as concrete
 * inputs"): random genome -> fixed-length noisy reads from both strands -> all true read
 * pairs sharing >= min_overlap genome bases as DALIGNER-style overlap records with trace
 * points (tspace-aligned A blocks: (diffs, B length)) taken from the TRUE edit script.
 * Error profile default: 15 % split 80 % ins / 13.33 % del / 6.67 % sub, the profile hinted
 * at src/daccord.cpp:1905-1910.  Output uses the structs of include/daccord_hip.h and the
 * Dazzler .bps 2-bit layout, i.e. exactly what dacc_load_db / dacc_submit_piles take.
 *
 * Overlap records come out in .las order (aread, bread); selecting/sorting a pile the way
 * daccord.cpp:2166-2288 does is the pile loader's job (host.cpp).
 */
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../../include/daccord_hip.h"

namespace {

struct Rng
{
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
	uint64_t next()
	{
		uint64_t z = (s += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	double uni() { return (next() >> 11) * (1.0/9007199254740992.0); }
	uint64_t below(uint64_t n) { return next() % n; }
};

struct SynthRead
{
	uint64_t gs, ge;        // genome span [gs,ge) covered by non-inserted bases
	int strand;             // 0 forward, 1 reverse
	std::vector<uint8_t> F; // genome-forward version of the read, 2-bit codes
	std::vector<uint32_t> gpos; // genome coordinate (relative to gs) of each F base (insertions: next genome base)
	std::vector<uint8_t> isins;
};

struct Synth
{
	std::vector<uint8_t> genome;
	std::vector<SynthRead> reads;
	std::vector<uint8_t> bps; std::vector<uint64_t> boff; std::vector<uint32_t> rlen;
	std::vector<dacc_overlap> ovl; std::vector<uint8_t> trace; std::vector<dacc_pile> piles;
	std::vector<int64_t> truth; // per read: gs, ge, strand
};

}

extern "C" {

typedef struct synth_params
{
	uint64_t genome_len; uint32_t nreads; uint32_t read_len;
	double p_ins, p_del, p_sub;
	uint32_t min_overlap; int32_t tspace; uint64_t seed; int32_t nthreads; int32_t reserved;
	uint32_t afirst, alast;      // overlaps / piles only for the A reads in [afirst,alast) (alast == 0: all); every read is generated either way
} synth_params;

void * synth_generate(synth_params const * P)
{
	Synth * S = new Synth;
	Rng grng(P->seed*7919+1);
	S->genome.resize(P->genome_len);
	for ( uint64_t i = 0; i < P->genome_len; ++i ) S->genome[i] = grng.next() & 3;
	uint32_t const L = P->read_len;
	S->reads.resize(P->nreads);
	uint64_t const maxspan = static_cast<uint64_t>(L) + 64; // a read never spans more genome than this
	// reads
	for ( uint32_t r = 0; r < P->nreads; ++r )
	{
		Rng rng(P->seed*1000003ull + 17ull*r + 5);
		SynthRead & R = S->reads[r];
		R.strand = rng.next() & 1;
		uint64_t const gs = rng.below(P->genome_len > maxspan ? P->genome_len-maxspan : 1);
		R.gs = gs;
		uint64_t g = gs;
		R.F.reserve(L); R.gpos.reserve(L); R.isins.reserve(L);
		bool first = true;
		while ( R.F.size() < L && g < P->genome_len )
		{
			if ( !first )
				while ( R.F.size() < L && rng.uni() < P->p_ins )
				{
					R.F.push_back(rng.next()&3); R.gpos.push_back(g-gs); R.isins.push_back(1);
				}
			if ( R.F.size() >= L ) break;
			double const u = rng.uni();
			if ( !first && u < P->p_del ) { ++g; continue; }
			uint8_t b = S->genome[g];
			if ( !first && u < P->p_del + P->p_sub ) b = (b + 1 + rng.below(3)) & 3;
			R.F.push_back(b); R.gpos.push_back(g-gs); R.isins.push_back(0);
			++g; first = false;
		}
		// make the last base a non-insertion so that [gs,ge) is exact
		while ( !R.F.empty() && R.isins.back() ) { R.F.pop_back(); R.gpos.pop_back(); R.isins.pop_back(); }
		R.ge = gs + R.gpos.back() + 1;
	}
	// 2-bit store of the reads as stored (reverse strand reads are reverse complemented)
	S->boff.resize(P->nreads); S->rlen.resize(P->nreads);
	uint64_t off = 0;
	for ( uint32_t r = 0; r < P->nreads; ++r )
	{
		S->boff[r] = off; S->rlen[r] = S->reads[r].F.size();
		off += (S->rlen[r]+3)/4;
	}
	S->bps.assign(off+8,0);
	for ( uint32_t r = 0; r < P->nreads; ++r )
	{
		SynthRead const & R = S->reads[r];
		uint64_t const n = R.F.size();
		for ( uint64_t i = 0; i < n; ++i )
		{
			uint8_t const b = R.strand ? (3-R.F[n-1-i]) : R.F[i];
			S->bps[S->boff[r]+(i>>2)] |= b << (6-2*(i&3));
		}
		S->truth.push_back(R.gs); S->truth.push_back(R.ge); S->truth.push_back(R.strand);
	}
	// candidate pairs via genome order
	std::vector<uint32_t> order(P->nreads);
	for ( uint32_t r = 0; r < P->nreads; ++r ) order[r] = r;
	std::sort(order.begin(),order.end(),[&](uint32_t a, uint32_t b){ return S->reads[a].gs < S->reads[b].gs; });
	std::vector<uint32_t> rank(P->nreads);
	for ( uint32_t i = 0; i < P->nreads; ++i ) rank[order[i]] = i;

	std::vector< std::vector<dacc_overlap> > PO(P->nreads);
	// trace values: one byte each up to tspace 125, two bytes (little endian) beyond, as DALIGNER writes them
	bool const wide = P->tspace > 125; uint32_t const tvmax = wide ? 65535u : 255u;
	std::vector< std::vector<uint16_t> > PT(P->nreads);
	int const nth = P->nthreads > 0 ? P->nthreads : 1;
	int64_t const ts = P->tspace;
	int64_t const a_lo = P->alast ? static_cast<int64_t>(P->afirst) : 0, a_hi = P->alast ? std::min<int64_t>(P->alast,P->nreads) : static_cast<int64_t>(P->nreads);
	#ifdef _OPENMP
	#pragma omp parallel for schedule(dynamic,8) num_threads(nth)
	#endif
	for ( int64_t a = a_lo; a < a_hi; ++a )
	{
		SynthRead const & A = S->reads[a];
		std::vector<uint32_t> partners;
		for ( int64_t i = static_cast<int64_t>(rank[a])-1; i >= 0; --i )
		{
			SynthRead const & B = S->reads[order[i]];
			if ( B.gs + maxspan + 8 < A.gs ) break;
			partners.push_back(order[i]);
		}
		for ( uint64_t i = rank[a]+1; i < P->nreads; ++i )
		{
			SynthRead const & B = S->reads[order[i]];
			if ( B.gs >= A.ge ) break;
			partners.push_back(order[i]);
		}
		std::sort(partners.begin(),partners.end());
		std::vector<uint8_t> ops; // bit0: consumes A, bit1: consumes B, bit2: diff
		for ( uint64_t pi = 0; pi < partners.size(); ++pi )
		{
			uint32_t const b = partners[pi];
			SynthRead const & B = S->reads[b];
			uint64_t const g0 = std::max(A.gs,B.gs), g1 = std::min(A.ge,B.ge);
			if ( g1 <= g0 || g1-g0 < P->min_overlap ) continue;
			// first / last genome column where both reads have an aligned base
			uint64_t ia = std::lower_bound(A.gpos.begin(),A.gpos.end(),static_cast<uint32_t>(g0-A.gs)) - A.gpos.begin();
			uint64_t ib = std::lower_bound(B.gpos.begin(),B.gpos.end(),static_cast<uint32_t>(g0-B.gs)) - B.gpos.begin();
			uint64_t const na = A.F.size(), nb = B.F.size();
			bool ok = false;
			while ( ia < na && ib < nb )
			{
				if ( A.isins[ia] ) { ++ia; continue; }
				if ( B.isins[ib] ) { ++ib; continue; }
				uint64_t const ga = A.gs+A.gpos[ia], gb = B.gs+B.gpos[ib];
				if ( ga == gb ) { ok = true; break; }
				if ( ga < gb ) ++ia; else ++ib;
			}
			if ( !ok ) continue;
			uint64_t ja = std::lower_bound(A.gpos.begin(),A.gpos.end(),static_cast<uint32_t>(g1-A.gs)) - A.gpos.begin();
			uint64_t jb = std::lower_bound(B.gpos.begin(),B.gpos.end(),static_cast<uint32_t>(g1-B.gs)) - B.gpos.begin();
			// step back to the last common aligned column
			int64_t ea = static_cast<int64_t>(ja)-1, eb = static_cast<int64_t>(jb)-1;
			ok = false;
			while ( ea > static_cast<int64_t>(ia) && eb > static_cast<int64_t>(ib) )
			{
				if ( A.isins[ea] ) { --ea; continue; }
				if ( B.isins[eb] ) { --eb; continue; }
				uint64_t const ga = A.gs+A.gpos[ea], gb = B.gs+B.gpos[eb];
				if ( ga == gb ) { ok = true; break; }
				if ( ga > gb ) --ea; else --eb;
			}
			if ( !ok ) continue;
			uint64_t const glast = A.gs+A.gpos[ea];
			// true path in genome-forward orientation
			ops.clear();
			uint64_t xa = ia, xb = ib;
			while ( true )
			{
				if ( A.isins[xa] ) { ops.push_back(1|4); ++xa; continue; }
				if ( B.isins[xb] ) { ops.push_back(2|4); ++xb; continue; }
				uint64_t const ga = A.gs+A.gpos[xa], gb = B.gs+B.gpos[xb];
				if ( ga == gb )
				{
					ops.push_back(3 | ((A.F[xa] != B.F[xb]) ? 4 : 0));
					++xa; ++xb;
					if ( ga == glast ) break;
				}
				else if ( ga < gb ) { ops.push_back(1|4); ++xa; }
				else { ops.push_back(2|4); ++xb; }
			}
			int64_t abpos = ia, aepos = xa, bbpos = ib, bepos = xb;
			if ( A.strand )
			{
				std::reverse(ops.begin(),ops.end());
				int64_t const t0 = na-aepos, t1 = na-abpos; abpos = t0; aepos = t1;
				int64_t const u0 = nb-bepos, u1 = nb-bbpos; bbpos = u0; bepos = u1;
			}
			if ( aepos-abpos < static_cast<int64_t>(P->min_overlap) ) continue;
			dacc_overlap O; std::memset(&O,0,sizeof(O));
			O.aread = a; O.bread = b; O.flags = (A.strand != B.strand) ? 1 : 0;
			O.abpos = abpos; O.aepos = aepos; O.bbpos = bbpos; O.bepos = bepos;
			O.trace_off = PT[a].size();
			// trace points: per tspace-aligned A block (diffs, B length)
			int64_t apos = abpos; int64_t blockend = std::min<int64_t>((abpos/ts)*ts+ts,aepos);
			uint32_t diffs = 0, blen = 0, totaldiffs = 0; bool bad = false;
			for ( uint64_t q = 0; q < ops.size(); ++q )
			{
				uint8_t const op = ops[q];
				if ( (op & 1) && apos == blockend )
				{
					// next A base opens a new block: close the current one
					if ( diffs > tvmax || blen > tvmax ) bad = true;
					PT[a].push_back(diffs); PT[a].push_back(blen);
					totaldiffs += diffs; diffs = 0; blen = 0;
					blockend = std::min<int64_t>(blockend+ts,aepos);
				}
				if ( op & 4 ) ++diffs;
				if ( op & 2 ) ++blen;
				if ( op & 1 ) ++apos;
			}
			if ( diffs > tvmax || blen > tvmax ) bad = true;
			PT[a].push_back(diffs); PT[a].push_back(blen); totaldiffs += diffs;
			if ( bad ) { PT[a].resize(O.trace_off); continue; }
			O.diffs = totaldiffs;
			O.tlen = PT[a].size()-O.trace_off;
			PO[a].push_back(O);
		}
	}
	for ( uint32_t a = static_cast<uint32_t>(a_lo); a < static_cast<uint32_t>(a_hi); ++a )
	{
		dacc_pile pile; pile.aread = a; pile.novl = PO[a].size(); pile.first_ovl = S->ovl.size();
		uint64_t const tbase = wide ? S->trace.size()/2 : S->trace.size();      // offsets count trace values
		for ( uint64_t i = 0; i < PO[a].size(); ++i )
		{
			dacc_overlap O = PO[a][i];
			O.trace_off += tbase;
			S->ovl.push_back(O);
		}
		for ( uint64_t i = 0; i < PT[a].size(); ++i )
		{
			S->trace.push_back(static_cast<uint8_t>(PT[a][i] & 0xFF));
			if ( wide ) S->trace.push_back(static_cast<uint8_t>(PT[a][i] >> 8));
		}
		S->piles.push_back(pile);
	}
	S->trace.resize(S->trace.size()+16,0);
	return S;
}

void synth_destroy(void * v) { delete static_cast<Synth *>(v); }

void synth_get(void * v,
	uint8_t const ** bps, uint64_t * bps_bytes, uint64_t const ** boff, uint32_t const ** rlen, uint64_t * nreads,
	dacc_overlap const ** ovl, uint64_t * novl, uint8_t const ** trace, uint64_t * ntrace,
	dacc_pile const ** piles, uint64_t * npiles, uint8_t const ** genome, int64_t const ** truth)
{
	Synth * S = static_cast<Synth *>(v);
	*bps = S->bps.data(); *bps_bytes = S->bps.size(); *boff = S->boff.data(); *rlen = S->rlen.data(); *nreads = S->rlen.size();
	*ovl = S->ovl.data(); *novl = S->ovl.size(); *trace = S->trace.data(); *ntrace = S->trace.size()-16;      /* bytes */
	*piles = S->piles.data(); *npiles = S->piles.size(); *genome = S->genome.data(); *truth = S->truth.data();
}

}
