/*
 * Per-wavefront scratch arena of the window kernel: one contiguous HBM slab per resident
 * wavefront, carved into the arrays below (all wave-uniform pointers).  Sizes come from
 * ArenaCaps, which the host derives from the batch (max active depth etc.); exceeding a
 * capacity never corrupts memory: the window is reported as WS_OVERFLOW and the library
 * returns DACC_ENOTSUP for the batch.
 */
#ifndef DACC_ARENA_HPP
#define DACC_ARENA_HPP
#include "wave.hpp"
#include "dev_types.hpp"

namespace dacc {

struct HeapWI { double w; int32_t idx; int32_t pad; };                    // (weight, pool index)
struct HeapSI { double w; uint32_t left, right, current; int32_t path; }; // ScoreInterval
struct HeapCC { double w; uint32_t o, l; };                               // ConsensusCandidate (weight,o,l)

struct Arena
{
	// window strings
	uint8_t * str; uint16_t * slen; uint64_t * peq; uint64_t * mst; uint64_t * akeys; uint32_t * koff;
	// k-mer instances
	uint64_t * pre; uint64_t * lastk; uint32_t * nstart0;
	// nodes
	uint32_t * nv; uint32_t * nps; uint16_t * nfreq; uint16_t * plow; uint16_t * phigh; uint16_t * cplow; uint16_t * cphigh; uint16_t * cnt0;
	uint8_t * nsucc; uint8_t * nsuccact; uint8_t * npred; uint32_t * succ; int32_t * succid;
	uint32_t * feasoff; uint32_t * cfeasoff; uint16_t * nfeas; uint16_t * ncfeas;
	uint8_t * fp_p; double * fp_w; uint8_t * cfp_p; double * cfp_w;
	uint64_t * mfirst; uint64_t * mlast;
	// stretches
	int32_t * sfirst; int32_t * sext; int32_t * slast; uint32_t * sslen; uint32_t * slink;
	uint32_t * sfo; uint32_t * sfl; uint32_t * scfo; uint32_t * scfl;
	int32_t * tfirst; int32_t * text; int32_t * tlast; uint32_t * tslen; uint32_t * tlink; // pre-sort copies
	uint64_t * skey; uint32_t * sidx; uint32_t * scnt;
	int32_t * links;
	uint16_t * sf_p; double * sf_w; double * sf_wf; double * sf_wl;
	uint16_t * csf_p; double * csf_w; double * csf_wf; double * csf_wl;
	uint64_t * rlkey;
	// reverse path pool + ARP
	int32_t * rp_parent; int32_t * rp_stretch; uint32_t * rp_front; double * rp_weight; uint16_t * rp_pos; uint16_t * rp_len; uint16_t * rp_baselen;
	int32_t * arp; uint32_t * arw; uint32_t * arwr; uint64_t * arwt;
	HeapWI * rpst; HeapWI * arph; uint8_t * arph_n;
	// forward path pool
	int32_t * p_parent; int32_t * p_stretch; uint32_t * p_len; uint32_t * p_pos; double * p_weight; uint32_t * p_baselen;
	HeapWI * apq; uint8_t * apq_n;
	HeapSI * siq;
	// candidates
	uint8_t * cons; HeapCC * cdh; HeapCC * ch; HeapCC * acc; double * accerr; uint32_t * canderr;
	// gap filling (filterfreq 0)
	uint64_t * ls; uint64_t * ane; double * tmpw; uint32_t * tmpp;
	// consensus -> A alignment
	uint64_t * alpv; uint64_t * almv; uint16_t * albot; uint8_t * alops;
};

HDEV uint64_t arena_align(uint64_t o) { return (o + 15) & ~static_cast<uint64_t>(15); }

// Host emulation only (tests/emul): a guard gap behind every field.  The harness fills the arena with a pattern and checks the gaps after
// every window, so a write past a field's capacity is caught where it happens instead of landing in the next field unnoticed.  On the
// device the gap is zero bytes and the layout is what it always was.
#if defined(DACC_EMUL)
  #include <vector>
  enum : uint64_t { ARENA_GUARD = 64 };
  inline std::vector<uint64_t> * & arenaGuardSink() { static std::vector<uint64_t> * p = 0; return p; }
  #define DACC_CARVE_GUARD { if ( arenaGuardSink() ) arenaGuardSink()->push_back(o); o += ARENA_GUARD; }
#else
  #define DACC_CARVE_GUARD
#endif
#define DACC_CARVE(field,type,count) A.field = reinterpret_cast<type *>(base + o); o = arena_align(o + sizeof(type)*static_cast<uint64_t>(count)); DACC_CARVE_GUARD

// carve the arena; returns total bytes (call with base = 0 to size it)
HDEV uint64_t arena_carve(Arena & A, uint8_t * base, ArenaCaps const & C, uint32_t const w = 0)
{
	uint64_t o = 0;
	uint32_t const keycap = next_pow2(C.maxs < 2 ? 2 : C.maxs);
	DACC_CARVE(str,uint8_t,static_cast<uint64_t>(C.maxs)*C.lstr)
	DACC_CARVE(slen,uint16_t,C.maxs)
	DACC_CARVE(peq,uint64_t,static_cast<uint64_t>(C.maxs)*4*(C.lstr>>6))
	DACC_CARVE(mst,uint64_t,64*2*(C.lstr>>6))    // per lane column state (Pv, Mv) of the block-wise Myers for strings beyond LSTR
	DACC_CARVE(akeys,uint64_t,C.precap)       // also used for the (possibly > maxs) active list
	DACC_CARVE(koff,uint32_t,C.maxs+1)
	DACC_CARVE(pre,uint64_t,C.precap)
	DACC_CARVE(lastk,uint64_t,keycap)
	DACC_CARVE(nstart0,uint32_t,C.precap+1)
	DACC_CARVE(nv,uint32_t,C.nodecap)
	DACC_CARVE(nps,uint32_t,C.nodecap)
	DACC_CARVE(nfreq,uint16_t,C.nodecap)
	DACC_CARVE(plow,uint16_t,C.nodecap)
	DACC_CARVE(phigh,uint16_t,C.nodecap)
	DACC_CARVE(cplow,uint16_t,C.nodecap)
	DACC_CARVE(cphigh,uint16_t,C.nodecap)
	DACC_CARVE(cnt0,uint16_t,C.nodecap)
	DACC_CARVE(nsucc,uint8_t,C.nodecap)
	DACC_CARVE(nsuccact,uint8_t,C.nodecap)
	DACC_CARVE(npred,uint8_t,C.nodecap)
	DACC_CARVE(succ,uint32_t,C.nodecap*4)
	DACC_CARVE(succid,int32_t,C.nodecap*4)
	DACC_CARVE(feasoff,uint32_t,C.nodecap+1)
	DACC_CARVE(cfeasoff,uint32_t,C.nodecap+1)
	DACC_CARVE(nfeas,uint16_t,C.nodecap)
	DACC_CARVE(ncfeas,uint16_t,C.nodecap)
	DACC_CARVE(fp_p,uint8_t,C.fcap)
	DACC_CARVE(fp_w,double,C.fcap)
	DACC_CARVE(cfp_p,uint8_t,C.fcap)
	DACC_CARVE(cfp_w,double,C.fcap)
	DACC_CARVE(mfirst,uint64_t,keycap)
	DACC_CARVE(mlast,uint64_t,keycap)
	DACC_CARVE(sfirst,int32_t,C.strcap)
	DACC_CARVE(sext,int32_t,C.strcap)
	DACC_CARVE(slast,int32_t,C.strcap)
	DACC_CARVE(sslen,uint32_t,C.strcap)
	DACC_CARVE(slink,uint32_t,C.strcap)
	DACC_CARVE(sfo,uint32_t,C.strcap)
	DACC_CARVE(sfl,uint32_t,C.strcap)
	DACC_CARVE(scfo,uint32_t,C.strcap)
	DACC_CARVE(scfl,uint32_t,C.strcap)
	DACC_CARVE(tfirst,int32_t,C.strcap)
	DACC_CARVE(text,int32_t,C.strcap)
	DACC_CARVE(tlast,int32_t,C.strcap)
	DACC_CARVE(tslen,uint32_t,C.strcap)
	DACC_CARVE(tlink,uint32_t,C.strcap)
	DACC_CARVE(skey,uint64_t,C.strcap)
	DACC_CARVE(sidx,uint32_t,C.strcap)
	DACC_CARVE(scnt,uint32_t,C.nodecap+1)
	DACC_CARVE(links,int32_t,C.linkcap)
	DACC_CARVE(sf_p,uint16_t,C.sfcap)
	DACC_CARVE(sf_w,double,C.sfcap)
	DACC_CARVE(sf_wf,double,C.sfcap)
	DACC_CARVE(sf_wl,double,C.sfcap)
	DACC_CARVE(csf_p,uint16_t,C.sfcap)
	DACC_CARVE(csf_w,double,C.sfcap)
	DACC_CARVE(csf_wf,double,C.sfcap)
	DACC_CARVE(csf_wl,double,C.sfcap)
	DACC_CARVE(rlkey,uint64_t,C.rlcap)
	DACC_CARVE(rp_parent,int32_t,C.poolcap)
	DACC_CARVE(rp_stretch,int32_t,C.poolcap)
	DACC_CARVE(rp_front,uint32_t,C.poolcap)
	DACC_CARVE(rp_weight,double,C.poolcap)
	DACC_CARVE(rp_pos,uint16_t,C.poolcap)
	DACC_CARVE(rp_len,uint16_t,C.poolcap)
	DACC_CARVE(rp_baselen,uint16_t,C.poolcap)
	DACC_CARVE(arp,int32_t,C.poolcap)
	DACC_CARVE(arw,uint32_t,C.poolcap)
	DACC_CARVE(arwr,uint32_t,C.poolcap)
	DACC_CARVE(arwt,uint64_t,C.poolcap)
	DACC_CARVE(rpst,HeapWI,C.poolcap)
	DACC_CARVE(arph,HeapWI,C.blcap*12)
	DACC_CARVE(arph_n,uint8_t,C.blcap)
	DACC_CARVE(p_parent,int32_t,C.poolcap)
	DACC_CARVE(p_stretch,int32_t,C.poolcap)
	DACC_CARVE(p_len,uint32_t,C.poolcap)
	DACC_CARVE(p_pos,uint32_t,C.poolcap)
	DACC_CARVE(p_weight,double,C.poolcap)
	DACC_CARVE(p_baselen,uint32_t,C.poolcap)
	DACC_CARVE(apq,HeapWI,C.blcap*12)
	DACC_CARVE(apq_n,uint8_t,C.blcap)
	DACC_CARVE(siq,HeapSI,C.poolcap)
	DACC_CARVE(cons,uint8_t,C.conscap)
	DACC_CARVE(cdh,HeapCC,16)
	DACC_CARVE(ch,HeapCC,16)
	DACC_CARVE(acc,HeapCC,16)
	DACC_CARVE(accerr,double,16)
	DACC_CARVE(canderr,uint32_t,16*C.maxs)
	DACC_CARVE(ls,uint64_t,C.nodecap*4)
	DACC_CARVE(ane,uint64_t,C.nodecap*4)
	DACC_CARVE(tmpw,double,256)
	DACC_CARVE(tmpp,uint32_t,256)
	// consensus -> A alignment: one column of vertical deltas per consensus symbol (two words per column for w in 65..128)
	uint32_t const alcons = DACC_MAXCONS_OF(w), alw = DACC_WIDE_W(w) ? 2u : 1u;
	DACC_CARVE(alpv,uint64_t,alw*(alcons+1))
	DACC_CARVE(almv,uint64_t,alw*(alcons+1))
	DACC_CARVE(albot,uint16_t,alcons+1)
	DACC_CARVE(alops,uint8_t,2*alcons+2*64*alw+8)
	return o;
}

}
#endif
