/*
 * LDS-resident fast path of the per-window de Bruijn consensus (one wavefront = one window).
 *
 * Same algorithm and exactly the same results as the generic engine (dbg_window.hpp, a close
 * restatement of the reference's control flow); what changes is where the state lives and how
 * the work is organised for CDNA4:
 *
 *  - all per-window working state sits in the workgroup's LDS slice with 8/16-bit fields; the
 *    sort buffer of the build phase is overlaid by the traversal structures;
 *  - node successors need no table: the <= 4 successor k-mers of a node are adjacent in the sorted
 *    node-key array, one lower bound per node is kept;
 *  - weights are exact 64-bit fixed point: every weight the reference forms is an integer multiple of
 *    2^-32 (w = sum of VS entries / 2^32, DotProduct.hpp:54-60) and all its partial sums stay below
 *    2^21, so the reference's FP64 sums/differences are exact and equal our integer sums times 2^-32.
 *    Thresholds become integer compares (>= 1e-3 <=> >= 4294968, > 0.1 / >= 0.1 <=> >= 429496730,
 *    >= 0.5 <=> >= 2^31), heap orders are unchanged;
 *  - k-mer feasibility is never materialised: stretch feasibility is evaluated with lanes = candidate
 *    start positions straight from a 32-bit copy of the fixed-point table held in LDS, results kept as
 *    a 64-bit position mask per stretch plus a compact weight list; stretch links are mask
 *    intersections evaluated on demand;
 *  - the reference recomputes stretches, feasibility and both path enumerations for every
 *    (first k-mer, last k-mer) candidate pair (traverse, DebruijnGraph.hpp:4774-5097; hundreds of
 *    pairs per window at k=14).  Here stretches and feasibility are computed once per activation
 *    state; the split pieces of every candidate first / last k-mer are added to the pool once;
 *    the reverse enumeration is cached per last k-mer and the forward enumeration per first k-mer,
 *    and reused for a pair whenever the other k-mer's split provably cannot touch it (no scan
 *    target of the enumeration equals a node that identifies the split stretch or its pieces);
 *    otherwise the pair is enumerated on its exact stretch set.
 *
 *  - stretches are looked up by first / last node through small index arrays, the weights of a feasible (stretch,
 *    position) entry carry the weights of its end nodes, candidates are kept as stretch sequences and decoded once.
 *
 * The same code is instantiated for five capacity tiers (FastTier<1..3>: 3, 2 and 1 wavefronts per CU; <4>: deep piles,
 * 3 per CU; <5>: strings of up to 128 bases, 1 per CU; the layout
 * FastLds<CT> is a compile time constant, so every LDS access has an immediate offset).  processWindowFast returns
 * FW_NEXT when a window overflows a tier (flags say what overflowed: 1 instances, 2 nodes, 8 candidates, 16 walk,
 * 32 stretches, 64 links, 128 weights, 512 pools (0x4000 reverse cache, 0x8000 path ids, 0x10000 forward pool,
 * 0x20000 popped paths, 0x40000 score intervals), 1024 introsort depth, 2048 base length, 4096 candidate length /
 * sequence, 8192 gap filling) and FW_GENERIC for shapes the tier does not support (w > 63, a string longer than its
 * string stride: 64 bases in tiers 1-4, 128 in tier 5); those go to tier 5 / the generic engine (dbg_window.hpp).
 */
#ifndef DACC_FAST_WINDOW_HPP
#define DACC_FAST_WINDOW_HPP
#include "wave.hpp"
#include "dev_types.hpp"
#include "arena.hpp"
#include "window_main.hpp"

namespace dacc {

enum { WS_RETRY = 4 };
enum { FNOPAR = 0xFF };
enum { FSUPCAP = 128, FSUPCAPW = 192 };      // (W: the wide tier -- the table of w = 127 covers 165 read offsets)
enum { FSEQCAP = 48 };      // max stretches of one candidate path     // max width (read offsets) of the model table copy in LDS

// run time description of a capacity tier (host planning, launch parameters)
struct FastCaps
{
	uint32_t maxs, precap, ncap, scap, lcap, wcap, rccap, fcap, siqcap, blcap;
	uint32_t tabcap;             // 32-bit words the table overlay of this tier can hold
	uint32_t nrows, nsup;        // dimensions of the fixed-point table copy held in LDS
	uint32_t ldsbytes;
	uint32_t gbytes;             // gw tiers: bytes of global scratch per workgroup (0: none)
};

struct FSI { uint64_t w; uint16_t left, right, current, path; };   // ScoreInterval (left/right/current: sorted reverse entries, path: forward pop index)
struct FCC { uint64_t w; uint32_t o, l; };                                       // ConsensusCandidate

// compile time capacities of the two tiers: every LDS offset below is an instruction immediate
template<int TIER> struct FastTier;
// gw: the weights of the feasible (stretch, position) pairs and the model table live in global memory (a scratch slab per
// workgroup / the padded 32 bit table), the build-phase arrays are overlaid by the enumeration pools (spilled to the slab
// while the enumerations run): layout FastLds<CT,true>.  Round 3 measured that a second wavefront per SIMD hides the LDS
// round trips of the first almost completely (profiles/r03a_occupancy_experiment.md), so LDS bytes per window decide the
// throughput: tier 1 is 26.3 KB = 6 wavefronts per CU with (almost) the capacities it had at 53.8 KB = 3 per CU.
// (round 6) reverse pool of tiers 0 and 1 in chunks of TWO paths: a lane takes pool entries a chunk at a time, and with one lane per last
// k-mer candidate (9 on average, up to 24) most chunks of four were half empty -- the reverse pool had become what sent tier 0's windows on
// (82 % of its hand-overs, each after the whole build phase and the feasibility; emulation, 8 piles of config 2: 261 of 5251 windows -> 14)
#if !defined(DACC_RCH01)
#define DACC_RCH01 2
#endif
template<> struct FastTier<1> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 1024, rch = DACC_RCH01, fch = 4, fnw = 2, fnc = 40, idmax = 250, rpstcap = 256, lstr = 64, maxs = 32, precap = 1024, ncap = 776, scap = 112, lcap = 960, wcap = 1024, rccap = 144, fcap = 192, siqcap = 56, blcap = 96, seqcap = 32, psiq = 8, consrow = 64, lscrids = 2048, wide = 0 }; };
// tier 0 (size classes): the windows a pre-pass (classifyWindow, k_classify) finds small -- few strings, at most T0INST k-mer
// instances -- in 20 KB = 8 wavefronts per CU (two on every SIMD).  What overflows it joins the other windows in tier 1.
template<> struct FastTier<0> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 768, rch = DACC_RCH01, fch = 4, fnw = 2, fnc = 32, idmax = 250, rpstcap = 256, lstr = 64, maxs = 28, precap = 576, ncap = 524, scap = 88, lcap = 640, wcap = 768, rccap = 96, fcap = 128, siqcap = 56, blcap = 96, seqcap = 32, psiq = 8, consrow = 64, lscrids = 2048, wide = 0 }; };
// tier 7 (round 6, the MIDDLE size class of shallow batches): 22.9 KB = 7 wavefronts per CU.  Four fifths of the windows tier 1 (6 per CU)
// used to run have at most 28 strings, 704 k-mer instances, 640 nodes and 800 links (emulation, 16 piles of config 2: 4244 of 5295); the
// pre-pass sends them here, what overflows joins tier 1's list like tier 0's hand-overs join this one.
template<> struct FastTier<7> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 896, rch = DACC_RCH01, fch = 4, fnw = 2, fnc = 32, idmax = 250, rpstcap = 256, lstr = 64, maxs = 28, precap = 704, ncap = 640, scap = 104, lcap = 800, wcap = 896, rccap = 112, fcap = 148, siqcap = 56, blcap = 96, seqcap = 32, psiq = 8, consrow = 64, lscrids = 2048, wide = 0 }; };
enum : uint32_t { T0INST_DEFAULT = 576, T7INST_DEFAULT = 704 };      // a window with more k-mer instances (upper bound of the pre-pass) starts in tier 1; run-time
                                               // argument of the pre-pass (DACC_T0INST overrides it for sweeps)
// tier 2: gw layout as well since round 3 (43 KB: 3 wavefronts per CU; the legacy layout was 80.5 KB: 2 per CU)
// (round 4: 76 KB = 2 wavefronts per CU instead of 46.5 KB = 3, with tier 3's node capacity: at 54x more than half of what the deep tier
// hands on has more than 1024 nodes at filter frequency 1 and used to go through this tier only to be handed on again to tier 3, which
// runs one wavefront per CU)
// (round 6: 53 744 B = 42 LDS granules = THREE wavefronts per CU instead of 63.9 KB = two, with 1536 nodes and 2032 links instead of 2048 / 2304:
// of the 65 903 windows the deep tier hands on in 1000 piles of the 54x shape 3045 went on to tier 3 before and 3329 do now, and the tier's
// time falls from 96 to 66 ms -- 27.4 -> 30.1 Mbase/s, profiles/r06s; the node tables are 20 of its bytes per node)
#if !defined(DACC_T2_NCAP)
#define DACC_T2_NCAP 1536
#endif
#if !defined(DACC_T2_LCAP)
#define DACC_T2_LCAP 2032
#endif
template<> struct FastTier<2> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 3072, rch = 4, fch = 4, fnw = 4, fnc = 64, idmax = 250, rpstcap = 256, lstr = 64, maxs = 96, precap = 2048, ncap = DACC_T2_NCAP, scap = 232, lcap = DACC_T2_LCAP, wcap = 3072, rccap = 256, fcap = 192, siqcap = 96, blcap = 96, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };
// tier 6 (second slot of SHALLOW batches since round 3, gw layout, 36 KB = 4 wavefronts per CU): what tier 1 hands on at 20x
// are windows with more than its 608 nodes (82 % of the hand-overs) or fuller pools, not more strings or instances, so this
// tier keeps tier 1's string / instance capacities and spends its LDS on nodes, stretches and pools.  Deep batches keep
// tier 2 (64 strings, 2048 instances) behind the deep tier.
// (round 6: the tier sat at 34 720 B, 6 KB below what four wavefronts per CU allow (40 960 B): the bytes went to what sends its windows on to
// tier 3 -- ONE wavefront per CU -- on the ONT mix at small k (emulation, 6 piles of config 5 at k = 10 / 12: reverse pool 73 / 130, weights 53 / 1,
// forward pool 45 / 7 of about 200 hand-overs): reverse pool 192 -> 256 paths in chunks of two, 248 stretches, 2048 weight records)
template<> struct FastTier<6> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 2048, rch = 2, fch = 8, fnw = 3, fnc = 40, idmax = 250, rpstcap = 256, lstr = 128, maxs = 40, precap = 1024, ncap = 1024, scap = 248, lcap = 1280, wcap = 2048, rccap = 256, fcap = 256, siqcap = 96, blcap = 96, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

// tier 3 (one wavefront per CU): 16 bit path ids, so that an enumeration may hold more than 250 paths; sized for deep piles
// too (BASELINE config 4, 54x: up to 96 strings with 4096 k-mer instances, 72 first / last k-mer candidates)
// gw layout and 16 bit STRETCH ids since round 3: what the legacy tier 3 handed to the generic engine at 54x were windows with
// more than 250 stretches (13 of 19 per 60 000 windows) or more than 2112 feasible weights (5 of 19), and each of them cost the
// generic engine seconds (685 such windows were 92 % of a 2000-pile 54x batch, profiles/r03c_bench_54x_2000piles.log)
template<> struct FastTier<3> { typedef uint16_t id_t; typedef uint16_t sid_t; enum : uint32_t { smax = 1000, gw = 1, wcapg = 8192, rch = 8, fch = 16, fnw = 4, fnc = 72, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 96, precap = 4096, ncap = 2048, scap = 1024, lcap = 4096, wcap = 8192, rccap = 2048, fcap = 512, siqcap = 256, blcap = 128, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

// tier 4 (three wavefronts per CU, takes the place of tier 1 in batches of deep piles): many strings and k-mer instances,
// small graph.  At 54x (BASELINE config 4) 96 % of the windows find their consensus at filter frequency 2, where the graph
// has about a hundred nodes, while the 55 strings of a window carry 1500 k-mer instances.
template<> struct FastTier<4> { typedef uint8_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 608, rch = 4, fch = 4, fnw = 2, fnc = 40, idmax = 250, rpstcap = 256, lstr = 64, maxs = 96, precap = 2048, ncap = 256, scap = 48, lcap = 256, wcap = 608, rccap = 128, fcap = 96, siqcap = 56, blcap = 96, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

// tier 5 (one wavefront per CU, only for the windows the pre-scan found): B strings of up to 128 bases (string stride 128,
// two words per pattern mask); everything else as tier 3 with 64 strings.  Window strings of more than 64 bases are rare
// at the default window (a few per ten million windows of config 2) but each of them costs the generic engine seconds.
template<> struct FastTier<5> { typedef uint16_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 0, wcapg = 0, rch = 8, fch = 16, fnw = 4, fnc = 72, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 64, precap = 4096, ncap = 1792, scap = 250, lcap = 2048, wcap = 2040, rccap = 512, fcap = 512, siqcap = 256, blcap = 128, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

// tier 8 (round 6, two wavefronts per CU): WIDE windows, -w 64 ... 127 (the reference takes any -w, daccord.cpp:1282-1305; rounds 4-5 ran them in the generic
// engine, two orders of magnitude slower per window).  What a wide window needs beyond tier 6's two-word pattern masks: feasible start positions
// of a stretch up to nrows = w+1 <= 128 (two words per position mask: maskF / maskFh), candidates and a consensus of up to 128 symbols, the
// two-word consensus -> A alignment and the wide window record (dev_types.hpp: WRECW).  A batch with w > 63 runs this tier alone in front of the
// generic engine (BatchPlan::wide); sized for 20x piles at w = 64 ... 96 (20 strings of 83 k-mers at w = 96, k = 14).
template<> struct FastTier<8> { typedef uint16_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 6144, rch = 4, fch = 8, fnw = 4, fnc = 64, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 40, precap = 3072, ncap = 2304, scap = 248, lcap = 3072, wcap = 6144, rccap = 640, fcap = 512, siqcap = 96, blcap = 128, seqcap = 96, psiq = 10, consrow = 128, lscrids = 2560, wide = 1 }; };

// tier 9 (round 6, one wavefront per CU): what overflows tier 8 -- its reverse pool (512 paths), forward pool, score intervals -- before the generic
// engine, which recomputes both enumerations for every (first, last) k-mer pair and takes seconds for such a window (30 of 37 000 windows at
// w = 64 were 6.2 of a step's 6.7 s, profiles/r06v).  Tier 3's pools with the wide tier's masks, alignment and record.
template<> struct FastTier<9> { typedef uint16_t id_t; typedef uint16_t sid_t; enum : uint32_t { smax = 500, gw = 1, wcapg = 8192, rch = 16, fch = 16, fnw = 4, fnc = 72, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 64, precap = 4096, ncap = 3072, scap = 512, lcap = 4096, wcap = 8192, rccap = 2048, fcap = 1024, siqcap = 256, blcap = 128, seqcap = 96, psiq = 10, consrow = 128, lscrids = 2560, wide = 1 }; };

// tier 10 (round 6, two wavefronts per CU): the DENSE-graph tier of shallow batches, between tier 6 (four per CU, 8 bit path ids: 256 reverse
// paths) and tier 3 (ONE per CU).  On the ONT error mix at k = 10 tier 6 hands 1.4 % of the windows on -- reverse pools of more than 256 paths,
// forward pools, more than 1024 k-mer instances -- and tier 3 needs 17 % of the step for them (profiles/r06k/bench_ont_k10.log; VERDICT r05 task 4:
// "more than 1 % in tier 3: add a dense-graph tier").  Tier 8's pools and 16 bit ids without its wide parts.
template<> struct FastTier<10> { typedef uint16_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 6144, rch = 4, fch = 8, fnw = 4, fnc = 64, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 64, precap = 2560, ncap = 2048, scap = 248, lcap = 3072, wcap = 6144, rccap = 896, fcap = 512, siqcap = 128, blcap = 128, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

// tier 11 (round 6, two wavefronts per CU): tier 10's place in batches of DEEP piles, between tier 2 (three per CU, 2048 k-mer instances) and tier 3 (one per CU).
// At 54x what tier 2 hands on are windows of more than 2048 k-mer instances (75 strings and more: 186 of 208 hand-overs in 12 piles of config 4), a few node
// tables and pools; tier 3 needs 6.8 % of the step for them (profiles/r06zz/bench_54x_2000piles.log).  Tier 3's string / instance capacities with tier 10's pools.
template<> struct FastTier<11> { typedef uint16_t id_t; typedef uint8_t sid_t; enum : uint32_t { smax = 250, gw = 1, wcapg = 4096, rch = 4, fch = 8, fnw = 4, fnc = 72, idmax = 4000, rpstcap = 768, lstr = 128, maxs = 96, precap = 4096, ncap = 1536, scap = 248, lcap = 2048, wcap = 4096, rccap = 640, fcap = 384, siqcap = 128, blcap = 128, seqcap = 48, psiq = 10, consrow = 96, lscrids = 2560, wide = 0 }; };

HDEV constexpr uint32_t fcpow2(uint32_t v) { uint32_t p = 1; while ( p < v ) p <<= 1; return p; }
HDEV constexpr uint32_t fcmax(uint32_t a, uint32_t b) { return a > b ? a : b; }

/*
 * LDS layout of one wavefront.  FLD(name,type,count,start) declares an array at byte offset `start` (a constant
 * expression), its end offset e_name (8 byte aligned) and an accessor returning an LDS (address space 3) pointer, so
 * that every access compiles to ds_read/ds_write with a constant offset.
 */
#define FLD(name,type,count,start) \
	static constexpr uint32_t o_##name = (start); \
	static constexpr uint32_t e_##name = (o_##name + static_cast<uint32_t>(sizeof(type))*static_cast<uint32_t>(count) + 7u) & ~7u; \
	HDEV LDSQ type * name() const { return reinterpret_cast<LDSQ type *>(base + o_##name); }

template<typename CT, bool GW = (CT::gw != 0)> struct FastLds;

// legacy layout: everything in LDS (tiers 2-5)
template<typename CT>
struct FastLds<CT,false>
{
	LDSQ uint8_t * base;
	static constexpr uint32_t keycap = fcpow2(CT::maxs < 2 ? 2 : CT::maxs);
	static_assert((CT::precap & (CT::precap-1)) == 0,"precap must be a power of two: the bitonic sorts pad to one");
	static_assert(sizeof(typename CT::id_t) > 1 || (CT::fcap <= 256 && CT::rccap <= 256),"pool slots are recorded as id_t (pout)");
	static_assert(CT::blcap <= 128 && (CT::blcap & 7) == 0,"base length buckets: two 64 bit occupancy words, cleared 8 at a time");
	// ---- live during the whole window ----
	static_assert(CT::lstr == 64 || CT::lstr == 128,"string stride: one or two 64 bit words per pattern mask");
	static constexpr uint32_t pw = CT::lstr/64;      // words per pattern mask
	typedef uint8_t sid_t;      // stretch ids are 8 bits in the legacy layout
	static_assert(sizeof(typename CT::sid_t) == 1,"16 bit stretch ids need the gw layout");
	FLD(str,uint8_t,CT::maxs*CT::lstr,0)
	FLD(slen,uint8_t,CT::maxs,e_str)
	FLD(peq,uint64_t,CT::maxs*4*pw,e_slen)      // pattern masks of string j: words 4*pw*j + pw*symbol + word
	FLD(ipos,uint8_t,CT::precap,e_peq)
	FLD(irpos,uint8_t,CT::precap,e_ipos)
	FLD(nv,uint32_t,CT::ncap,e_irpos)
	FLD(nps,uint16_t,CT::ncap+1,e_nv)
	FLD(nfreq,uint8_t,CT::ncap,e_nps)
	FLD(succ0,uint16_t,CT::ncap,e_nfreq)
	FLD(sinfo,uint16_t,CT::ncap,e_succ0)
	FLD(npred,uint8_t,CT::ncap,e_sinfo)
	FLD(nrange,uint32_t,CT::ncap,e_npred)   // feasible position range of a node: pfrom | pto<<8 | cpfrom<<16 | cpto<<24
	FLD(mfirst,uint64_t,keycap,e_nrange)
	FLD(mlast,uint64_t,keycap,e_mfirst)
	FLD(suplo8,uint8_t,FSUPCAP,e_mlast)
	FLD(suphi8,uint8_t,FSUPCAP,e_suplo8)
	// small scratch of the serial code (dynamically indexed, so it must not live in registers)
	FLD(vrem,uint8_t,8,e_suphi8)
	FLD(vadd,uint8_t,8,e_vrem)
	FLD(vapos,uint8_t,8,e_vadd)
	FLD(vfn,uint16_t,8,e_vapos)
	FLD(vln,uint16_t,8,e_vfn)
	FLD(sstack,uint16_t,3*24,e_vln)
	FLD(chain,uint8_t,64,e_sstack)
	FLD(midpar,uint16_t,8,e_chain)          // middle pieces of twice-split stretches: parent stretch, node range [midA,midB]
	FLD(midA,uint8_t,8,e_midpar)
	FLD(midB,uint8_t,8,e_midA)
	static constexpr uint32_t ubase = e_midB;
	// ---- overlay A: build phase ----
	FLD(pre,uint64_t,CT::precap,ubase)
	FLD(lastk,uint64_t,keycap,e_pre)
	static constexpr uint32_t uA = e_lastk;
	// ---- overlay B: traversal phase ----
	FLD(sfirst,uint16_t,CT::scap,ubase)
	FLD(slast,uint16_t,CT::scap,e_sfirst)
	FLD(sslen,uint16_t,CT::scap,e_slast)
	FLD(slink,uint16_t,CT::scap,e_sslen)
	FLD(maskF,uint64_t,CT::scap,e_slink)
	FLD(maskR,uint64_t,CT::scap,e_maskF)
	FLD(woffF,uint16_t,CT::scap,e_maskR)
	FLD(woffR,uint16_t,CT::scap,e_woffF)
	FLD(links,uint16_t,CT::lcap,e_woffR)
	// weights of the feasible (stretch, position) pairs: whole stretch (48 bit), its first node in walking direction and,
	// forward only, its last node (40 bit each), split into 32 bit low words and high parts
	FLD(wF_lo,uint32_t,CT::wcap,e_links)
	FLD(wF1_lo,uint32_t,CT::wcap,e_wF_lo)
	FLD(wFl_lo,uint32_t,CT::wcap,e_wF1_lo)
	FLD(wR_lo,uint32_t,CT::wcap,e_wFl_lo)
	FLD(wR1_lo,uint32_t,CT::wcap,e_wR_lo)
	FLD(wF_hi,uint16_t,CT::wcap,e_wR1_lo)
	FLD(wR_hi,uint16_t,CT::wcap,e_wF_hi)
	FLD(wF1_hi,uint8_t,CT::wcap,e_wR_hi)
	FLD(wFl_hi,uint8_t,CT::wcap,e_wF1_hi)
	FLD(wR1_hi,uint8_t,CT::wcap,e_wFl_hi)
	// lookup of stretches by first / last node: head index per node, base stretches ordered by (last node, id)
	FLD(lhead,uint8_t,CT::ncap,e_wR1_hi)
	FLD(lord,uint32_t,CT::scap,e_lhead)
	FLD(ppos,uint8_t,CT::scap,e_lord)      // pieces: number of base stretches sorting before them
	FLD(fkmer,uint32_t,CT::fnc,e_ppos)
	FLD(lkmer,uint32_t,CT::fnc,e_fkmer)
	FLD(fnode,uint16_t,CT::fnc,e_lkmer)
	FLD(lnode,uint16_t,CT::fnc,e_fnode)
	FLD(parF,uint8_t,CT::fnc,e_lnode)
	FLD(posF,uint8_t,CT::fnc,e_parF)
	FLD(parL,uint8_t,CT::fnc,e_posF)
	FLD(posL,uint8_t,CT::fnc,e_parL)
	FLD(pieF,uint8_t,CT::fnc,e_posL)      // first piece id of candidate i (second = +1), FNOPAR if none
	FLD(pieL,uint8_t,CT::fnc,e_pieF)
	static constexpr uint32_t consmax = MAXCONS;
	static_assert(CT::wide == 0,"the wide tier needs the gw layout");
	FLD(bestL,uint8_t,MAXCONS,e_pieL)      // best consensus so far (survives the tries)
	FLD(cdh,FCC,16,e_bestL)
	static_assert(uA <= o_cdh,"the model table copy (from o_cdh on) is loaded for gap filling while the instance array is live");
	FLD(cseq,uint8_t,18*CT::seqcap,e_cdh)      // stretch sequences of the kept candidates (16 slots) + current + previous
	// dead until the kept candidates are decoded: shared with the lane scratch of the enumerations (lscr: per lane a
	// heap of 32 path ids / of 12 path ids / of PSIQ score intervals; all of it for an enumeration on lane 0 alone)
	FLD(ch,FCC,16,e_cseq)
	FLD(acc,FCC,16,e_ch)
	FLD(accerr,uint32_t,16,e_acc)
	FLD(canderr,uint8_t,16*CT::maxs,e_accerr)
	FLD(consL,uint8_t,16*CT::consrow,e_canderr)   // decoded candidates
	static constexpr uint32_t lscrbytes = static_cast<uint32_t>(CT::lscrids)*static_cast<uint32_t>(sizeof(typename CT::id_t));
	FLD(lscr,uint8_t,lscrbytes,e_cseq)
	FLD(siq,FSI,CT::siqcap,e_cseq)          // score intervals of a pair combined on lane 0
	static_assert(sizeof(FSI)*CT::siqcap <= lscrbytes,"the serial score interval heap shares the lane scratch");
	static constexpr uint32_t pbase = fcmax(e_consL,e_lscr);
	// reverse pool: paths by pool id
	FLD(rc_w,uint64_t,CT::rccap,pbase)
	FLD(rc_parent,typename CT::id_t,CT::rccap,e_rc_w)
	FLD(rc_stretch,uint8_t,CT::rccap,e_rc_parent)
	FLD(rc_pos,uint8_t,CT::rccap,e_rc_stretch)
	FLD(rc_len,uint8_t,CT::rccap,e_rc_pos)
	FLD(rc_baselen,uint8_t,CT::rccap,e_rc_len)
	FLD(rc_acc,typename CT::id_t,CT::rccap,e_rc_baselen)     // accepted paths of an enumeration, in its own slots
	// sorted blocks of the reverse enumerations
	FLD(rc_ord,typename CT::id_t,CT::rccap,e_rc_acc)
	FLD(rc_arw,typename CT::id_t,CT::rccap,e_rc_ord)
	FLD(rc_sbl,uint8_t,CT::rccap,e_rc_arw)         // base length of the i-th entry in sorted order
	FLD(rc_front,uint32_t,CT::rccap,e_rc_sbl)      // front k-mer of the i-th entry in sorted order
	FLD(rbase,uint16_t,CT::fnc+1,e_rc_front)
	FLD(rn,uint16_t,CT::fnc+1,e_rbase)
	FLD(rmaxw,uint64_t,CT::fnc+1,e_rn)
	FLD(rtmask,uint64_t,CT::fnc+1,e_rmaxw)
	FLD(rfmask,uint64_t,CT::fnc+1,e_rtmask)    // one bit per front k-mer (value mod 64) of the accepted reverse paths
	// forward pool: paths by pool id
	FLD(f_w,uint64_t,CT::fcap,e_rfmask)
	FLD(f_parent,typename CT::id_t,CT::fcap,e_f_w)
	FLD(f_stretch,uint8_t,CT::fcap,e_f_parent)
	FLD(f_pos,uint8_t,CT::fcap,e_f_stretch)
	FLD(f_baselen,uint8_t,CT::fcap,e_f_pos)
	FLD(f_len,uint8_t,CT::fcap,e_f_baselen)
	// popped paths of a tree (pop order, in the tree's own slots): path, candidate length, k-mer of its last node, weight
	// minus the junction node
	FLD(fp_id,typename CT::id_t,CT::fcap,e_f_len)
	FLD(fp_cl,uint8_t,CT::fcap,e_fp_id)
	FLD(fp_front,uint32_t,CT::fcap,e_fp_cl)
	FLD(fp_adj,uint64_t,CT::fcap,e_fp_front)
	// per first k-mer candidate: chunk list of its tree, popped paths, junction k-mer bits, scan target bits, heaviest path
	FLD(fchb,uint8_t,8*CT::fnw*(CT::fnc+1),e_fp_adj)   // chunk ids of the forward trees: 8*fnw per first k-mer candidate (+ one exact tree)
	FLD(fnp,uint16_t,CT::fnc+1,e_fchb)
	FLD(ffm,uint64_t,CT::fnc+1,e_fnp)
	FLD(ftm,uint64_t,CT::fnc+1,e_ffm)
	FLD(fmx,uint64_t,CT::fnc+1,e_ftm)
	// recorded (path, entry) sequences of the pairs of a round
	FLD(pout,typename CT::id_t,32*32,e_fmx)
	FLD(poutn,uint8_t,64,e_pout)
	FLD(ctr,uint32_t,4,e_poutn)
	FLD(rchx,uint8_t,32,e_ctr)              // chunk ids of a reverse enumeration on lane 0 alone
	static constexpr uint32_t upool = e_rchx;
	// chunk ids of the reverse enumerations of all last k-mers (32 per lane, lanes < min(fnc,64)): over the per first k-mer tables, which are
	// written after those enumerations have been copied to their blocks
	FLD(rchb,uint8_t,(CT::fnc < 64 ? CT::fnc : 64u)*32,o_fchb)
	static_assert(e_rchb <= o_pout,"reverse chunk lists must fit the per first k-mer tables");
	// raw stretches (overlay of the caches)
	FLD(tfirst,uint16_t,CT::scap,pbase)
	FLD(tlast,uint16_t,CT::scap,e_tfirst)
	FLD(tslen,uint16_t,CT::scap,e_tlast)
	FLD(tlink,uint16_t,CT::scap,e_tslen)
	FLD(skey,uint64_t,fcpow2(CT::scap),e_tlink)
	static constexpr uint32_t uraw = e_skey;
	// final alignment (overlay of the caches)
	FLD(alpv,uint64_t,MAXCONS+1,pbase)
	FLD(almv,uint64_t,MAXCONS+1,e_alpv)
	FLD(albot,uint16_t,MAXCONS+1,e_almv)
	FLD(alops,uint8_t,2*MAXCONS+2*64+8,e_albot)
	static constexpr uint32_t ualn = e_alops;
	static constexpr uint32_t uend = fcmax(fcmax(uA,uraw),fcmax(upool,ualn));
	// scratch behind the build overlay (gap filling)
	FLD(gfbuf,uint64_t,(pbase-uA)/8,uA)
	static constexpr uint32_t gfcap = (pbase-uA)/8;
	// fixed-point model table [pos][row], row stride nrows+1.  It shares the bytes of the enumeration pools: it is
	// (re)loaded from HBM/L2 for gap filling and for the stretch feasibility of a traverse call, both of which are over
	// before the pools are used.
	// feasibility tasks (with the table, dead before the pools are used): first task, first and end start position of
	// every (stretch, direction)
	static constexpr uint32_t taskbytes = ((2*CT::scap+2)*2 + 2*(2*CT::scap) + 15u) & ~15u;
	FLD(toff,uint16_t,2*CT::scap+2,upool-taskbytes)
	FLD(ulo,uint8_t,2*CT::scap,e_toff)
	FLD(uhi,uint8_t,2*CT::scap,e_ulo)
	FLD(tab,uint32_t,(upool-taskbytes-o_cdh)/4,o_cdh)   // also over the candidate buffers, which are dead at that time
	static constexpr uint32_t tabcap = (upool-taskbytes-o_cdh)/4;
	HDEV static uint32_t bytes(uint32_t const, uint32_t const) { return (uend + 15u) & ~15u; }
	// scratch of the stretch construction, the candidate search and the reachability test: borrowed from the weight arrays,
	// which are written after them.  Walk slots (wtmp) of 64 node ids each in front, the walking table behind them.
	static_assert(6u*CT::ncap + 16u <= e_wR1_hi - o_wF_lo,"scratch tables of the stretch walk must fit the weight arrays");
	static_assert(CT::wcap*4 >= CT::ncap*2,"interior node table must fit the weight array it borrows");
	static_assert(2u*CT::ncap + CT::scap + 8u <= 4u*CT::wcap,"reachability scratch must fit the first weight array");
	HDEV LDSQ uint32_t * xcnt32() const { return reinterpret_cast<LDSQ uint32_t *>(base + o_wF_lo); }
	HDEV LDSQ uint16_t * xstep() const { return reinterpret_cast<LDSQ uint16_t *>(base + ((e_wR1_hi - 2u*CT::ncap) & ~7u)); }
	HDEV LDSQ uint16_t * xwtmp() const { return reinterpret_cast<LDSQ uint16_t *>(base + o_wF_lo); }
	static constexpr uint32_t xnslot = (((e_wR1_hi - 2u*CT::ncap) & ~7u) - o_wF_lo) / 128u;
	HDEV LDSQ uint16_t * xsid() const { return reinterpret_cast<LDSQ uint16_t *>(base + o_wR_lo); }
	HDEV LDSQ uint8_t * xspos() const { return reinterpret_cast<LDSQ uint8_t *>(base + o_wR1_lo); }
	HDEV LDSQ uint8_t * xreach() const { return reinterpret_cast<LDSQ uint8_t *>(base + o_wF_lo); }
};

/*
 * gw layout (tier 1 since round 3).  Three regions:
 *   P  live during the whole window: string lengths, support bounds, serial scratch, node keys, fhead, best consensus;
 *   S  results of the build phase that the first phases of a traversal (stretches, candidates, feasibility) and its last
 *      ones (candidate errors, next activation state) need, but not the enumerations and the pairs in between: the
 *      enumeration pools are laid over S, whose bytes are spilled to the workgroup's global slab before the
 *      enumerations and restored after the pairs (two bulk copies of 12 KB per traversal);
 *   X  overlay A (sorted instances, build phase) / overlay B (stretches, candidates, candidate heap, lane scratch).
 * The weights of the feasible (stretch, position) pairs (16 of the 54 KB of the legacy tier 1) live in the global slab,
 * the model table is read from its padded 32 bit copy in global memory (both stay in L1/L2).  The scratch tables of the
 * stretch construction sit in parts of overlay B that are written later (pattern masks: raw stretches; candidate heap,
 * sequences and lane scratch: predecessor counts, walking table, interior nodes, reachability, feasibility tasks).
 */
template<typename CT>
struct FastLds<CT,true>
{
	LDSQ uint8_t * base;
	static constexpr uint32_t keycap = fcpow2(CT::maxs < 2 ? 2 : CT::maxs);
	// (precap need not be a power of two in this layout: the sorts touch only the n keys there are; the padding of the overlap
	// sort at the start of a window is checked against it)
	static_assert(sizeof(typename CT::id_t) > 1 || (CT::fcap <= 256 && CT::rccap <= 256),"pool slots are recorded as id_t (pout)");
	static_assert(CT::blcap <= 128 && (CT::blcap & 7) == 0,"base length buckets: two 64 bit occupancy words, cleared 8 at a time");
	static_assert(CT::lstr == 64 || CT::lstr == 128,"the gw layout holds strings of up to 64 bases in one word per pattern mask, of up to 128 in two (round 6)");
	static constexpr uint32_t pw = CT::lstr/64;
	typedef typename CT::sid_t sid_t;      // stretch ids: 8 bits (at most 250 stretches) or 16 bits
	// ---- P ----
	FLD(slen,uint8_t,CT::maxs,0)
	static constexpr uint32_t supcap = CT::wide ? static_cast<uint32_t>(FSUPCAPW) : static_cast<uint32_t>(FSUPCAP);
	FLD(suplo8,uint8_t,supcap,e_slen)
	FLD(suphi8,uint8_t,supcap,e_suplo8)
	FLD(vrem,uint8_t,8,e_suphi8)
	FLD(vadd,uint8_t,8,e_vrem)
	FLD(vapos,uint8_t,8,e_vadd)
	FLD(vfn,uint16_t,8,e_vapos)
	FLD(vln,uint16_t,8,e_vfn)
	FLD(sstack,uint16_t,3*24,e_vln)
	FLD(chain,uint8_t,64,e_sstack)
	FLD(midpar,uint16_t,8,e_chain)          // middle pieces of twice-split stretches: parent stretch, node range [midA,midB]
	FLD(midA,uint8_t,8,e_midpar)
	FLD(midB,uint8_t,8,e_midA)
	FLD(nv,uint32_t,CT::ncap,e_midB)
	FLD(npred,sid_t,CT::ncap,e_nv)
	// longest consensus / candidate of this tier: MAXCONS, 128 in the wide tier (consensus of a window of up to 127 bases)
	static constexpr uint32_t consmax = CT::wide ? 128u : static_cast<uint32_t>(MAXCONS);
	static constexpr uint32_t alw = CT::wide ? 2u : 1u;      // 64 bit words per column of the consensus -> A alignment
	FLD(bestL,uint8_t,consmax,e_npred)
	static constexpr uint32_t sbase = (e_bestL + 15u) & ~15u;
	// ---- S (spilled while the pools are live) ----
	// (round 5: no string array -- the pattern masks ARE the strings: symbol c at position p of string j <=> bit p of peq[4*j+c]; they are
	// built by ballots while the bases are gathered and the k-mers are cut from them; the 64 bytes per string went into the node tables)
	FLD(peq,uint64_t,CT::maxs*4*pw,sbase)
	FLD(ipos,uint8_t,CT::precap,e_peq)
	FLD(irpos,uint8_t,CT::precap,e_ipos)
	FLD(nps,uint16_t,CT::ncap+1,e_irpos)
	FLD(nfreq,uint8_t,CT::ncap,e_nps)
	FLD(succ0,uint16_t,CT::ncap,e_nfreq)
	FLD(sinfo,uint16_t,CT::ncap,e_succ0)
	FLD(nrange,uint32_t,CT::ncap,e_sinfo)
	FLD(mfirst,uint64_t,keycap,e_nrange)
	FLD(mlast,uint64_t,keycap,e_mfirst)
	static constexpr uint32_t send = (e_mlast + 15u) & ~15u;
	static constexpr uint32_t sbytes = send - sbase;
	// ---- enumeration pools: overlay of S ----
	FLD(rc_w,uint64_t,CT::rccap,sbase)
	FLD(rc_parent,typename CT::id_t,CT::rccap,e_rc_w)
	FLD(rc_stretch,sid_t,CT::rccap,e_rc_parent)
	FLD(rc_pos,uint8_t,CT::rccap,e_rc_stretch)
	FLD(rc_len,uint8_t,CT::rccap,e_rc_pos)
	FLD(rc_baselen,uint8_t,CT::rccap,e_rc_len)
	FLD(rc_acc,typename CT::id_t,CT::rccap,e_rc_baselen)
	FLD(rc_ord,typename CT::id_t,CT::rccap,e_rc_acc)
	FLD(rc_arw,typename CT::id_t,CT::rccap,e_rc_ord)
	FLD(rc_sbl,uint8_t,CT::rccap,e_rc_arw)
	FLD(rc_front,uint32_t,CT::rccap,e_rc_sbl)
	FLD(rbase,uint16_t,CT::fnc+1,e_rc_front)
	FLD(rn,uint16_t,CT::fnc+1,e_rbase)
	FLD(rmaxw,uint64_t,CT::fnc+1,e_rn)
	FLD(rtmask,uint64_t,CT::fnc+1,e_rmaxw)
	FLD(rfmask,uint64_t,CT::fnc+1,e_rtmask)
	FLD(f_w,uint64_t,CT::fcap,e_rfmask)
	FLD(f_parent,typename CT::id_t,CT::fcap,e_f_w)
	FLD(f_stretch,sid_t,CT::fcap,e_f_parent)
	FLD(f_pos,uint8_t,CT::fcap,e_f_stretch)
	FLD(f_baselen,uint8_t,CT::fcap,e_f_pos)
	FLD(f_len,uint8_t,CT::fcap,e_f_baselen)
	FLD(fp_id,typename CT::id_t,CT::fcap,e_f_len)
	FLD(fp_cl,uint8_t,CT::fcap,e_fp_id)
	FLD(fp_front,uint32_t,CT::fcap,e_fp_cl)
	FLD(fp_adj,uint64_t,CT::fcap,e_fp_front)
	FLD(fchb,uint8_t,8*CT::fnw*(CT::fnc+1),e_fp_adj)
	FLD(fnp,uint16_t,CT::fnc+1,e_fchb)
	FLD(ffm,uint64_t,CT::fnc+1,e_fnp)
	FLD(ftm,uint64_t,CT::fnc+1,e_ffm)
	FLD(fmx,uint64_t,CT::fnc+1,e_ftm)
	FLD(pout,typename CT::id_t,32*32,e_fmx)
	FLD(poutn,uint8_t,64,e_pout)
	FLD(ctr,uint32_t,4,e_poutn)
	FLD(rchx,uint8_t,32,e_ctr)
	static constexpr uint32_t upool = e_rchx;
	// (pools larger than the build-phase arrays run on into a gap in front of the overlays: tier 3, whose reverse pool is sized
	// for the 72 last k-mer candidates of a 54x pile -- 256 chunks of 8 paths, the most an 8 bit chunk id can name)
	FLD(rchb,uint8_t,(CT::fnc < 64 ? CT::fnc : 64u)*32,o_fchb)
	static_assert(e_rchb <= o_pout,"reverse chunk lists must fit the per first k-mer tables");
	static constexpr uint32_t ubase = fcmax(send,(upool + 15u) & ~15u);
	// ---- overlay A: build phase ----
	FLD(pre,uint64_t,CT::precap,ubase)
	FLD(lastk,uint64_t,keycap,e_pre)
	static constexpr uint32_t uA = e_lastk;
	// ---- overlay B: traversal ----
	FLD(sfirst,uint16_t,CT::scap,ubase)
	FLD(slast,uint16_t,CT::scap,e_sfirst)
	FLD(sslen,uint16_t,CT::scap,e_slast)
	FLD(slink,uint16_t,CT::scap,e_sslen)
	FLD(maskF,uint64_t,CT::scap,e_slink)
	FLD(maskR,uint64_t,CT::scap,e_maskF)
	FLD(maskFh,uint64_t,(CT::wide ? CT::scap : 0u),e_maskR)      // wide tier: start positions 64 ... 127
	FLD(maskRh,uint64_t,(CT::wide ? CT::scap : 0u),e_maskFh)
	FLD(woffF,uint16_t,CT::scap,e_maskRh)
	FLD(woffR,uint16_t,CT::scap,e_woffF)
	FLD(links,uint16_t,CT::lcap,e_woffR)
	FLD(lhead,sid_t,CT::ncap,e_links)
	FLD(lord,uint32_t,CT::scap,e_lhead)
	FLD(ppos,sid_t,CT::scap,e_lord)
	FLD(fkmer,uint32_t,CT::fnc,e_ppos)
	FLD(lkmer,uint32_t,CT::fnc,e_fkmer)
	FLD(fnode,uint16_t,CT::fnc,e_lkmer)
	FLD(lnode,uint16_t,CT::fnc,e_fnode)
	FLD(parF,sid_t,CT::fnc,e_lnode)
	FLD(posF,uint8_t,CT::fnc,e_parF)
	FLD(parL,sid_t,CT::fnc,e_posF)
	FLD(posL,uint8_t,CT::fnc,e_parL)
	FLD(pieF,sid_t,CT::fnc,e_posL)
	FLD(pieL,sid_t,CT::fnc,e_pieF)
	static constexpr uint32_t xbase = (e_pieL + 15u) & ~15u;
	FLD(cdh,FCC,16,xbase)
	FLD(cseq,sid_t,18*CT::seqcap,e_cdh)
	FLD(ch,FCC,16,e_cseq)
	FLD(acc,FCC,16,e_ch)
	FLD(accerr,uint32_t,16,e_acc)
	FLD(canderr,uint8_t,16*CT::maxs,e_accerr)
	FLD(consL,uint8_t,16*CT::consrow,e_canderr)
	static constexpr uint32_t lscrbytes = static_cast<uint32_t>(CT::lscrids)*static_cast<uint32_t>(sizeof(typename CT::id_t));
	FLD(lscr,uint8_t,lscrbytes,e_cseq)
	FLD(siq,FSI,CT::siqcap,e_cseq)
	static_assert(sizeof(FSI)*CT::siqcap <= lscrbytes,"the serial score interval heap shares the lane scratch");
	// (the scratch tables of the stretch construction need 3 bytes per node: the region is at least that long)
	static constexpr uint32_t alnbytes = 2u*8u*alw*(consmax+1u) + 2u*(consmax+1u) + 8u + (2u*consmax + 2u*64u + 8u) + 16u;      // alpv, almv, albot, alops
	static constexpr uint32_t uB = fcmax(fcmax(fcmax(fcmax(e_consL,e_lscr),xbase + 3u*CT::ncap + 16u),xbase + (2u*CT::scap+2u)*2u + 8u + 8u*CT::scap + 16u),CT::wide ? xbase + alnbytes : 0u);
	static constexpr uint32_t xbytes = uB - xbase;
	// raw stretches: over the pattern masks and weight offsets, which the feasibility writes later
	FLD(tfirst,uint16_t,CT::scap,o_maskF)
	FLD(tlast,uint16_t,CT::scap,e_tfirst)
	FLD(tslen,uint16_t,CT::scap,e_tlast)
	FLD(tlink,uint16_t,CT::scap,e_tslen)
	FLD(skey,uint64_t,fcpow2(CT::scap),e_tlink)
	static_assert(e_skey <= o_links,"raw stretches must fit the mask / offset arrays they borrow");
	// scratch over [xbase,uB): candidate heap, sequences and lane scratch are written by the enumerations only
	FLD(alpv,uint64_t,alw*(consmax+1),xbase)
	FLD(almv,uint64_t,alw*(consmax+1),e_alpv)
	FLD(albot,uint16_t,consmax+1,e_almv)
	FLD(alops,uint8_t,2*consmax+2*64+8,e_albot)
	static_assert(e_alops <= uB,"final alignment scratch");
	FLD(toff,uint16_t,2*CT::scap+2,xbase)
	FLD(urec,uint32_t,2*CT::scap,e_toff)      // unit at position q of the processing order: lo | unit << 7
	static_assert(e_urec <= uB,"feasibility tasks");
	static_assert(3u*CT::ncap + 16u <= xbytes,"predecessor counts (a byte per node) and walking table must fit the scratch");
	static_assert(3u*CT::ncap + 16u <= xbytes && 2u*CT::ncap + CT::scap + 8u <= xbytes,"interior node table / reachability scratch");
	HDEV LDSQ uint32_t * xcnt32() const { return reinterpret_cast<LDSQ uint32_t *>(base + xbase); }
	HDEV LDSQ uint16_t * xstep() const { return reinterpret_cast<LDSQ uint16_t *>(base + xbase + ((CT::ncap + 3u) & ~3u) + 8u); }
	HDEV LDSQ uint16_t * xsid() const { return reinterpret_cast<LDSQ uint16_t *>(base + xbase); }
	HDEV LDSQ uint8_t * xspos() const { return reinterpret_cast<LDSQ uint8_t *>(base + xbase + 2u*CT::ncap + 8u); }
	HDEV LDSQ uint8_t * xreach() const { return reinterpret_cast<LDSQ uint8_t *>(base + xbase); }
	static constexpr uint32_t uend = fcmax(uA,uB);
	// scratch behind the build overlay (gap filling): small in this layout, a window that needs more goes to the next tier
	FLD(gfbuf,uint64_t,(uB > uA ? (uB-uA)/8 : 0u),uA)
	static constexpr uint32_t gfcap = uB > uA ? (uB-uA)/8 : 0u;
	static constexpr uint32_t tabcap = 0x7FFFFFFFu;      // no LDS copy of the model table
	HDEV static uint32_t bytes(uint32_t const, uint32_t const) { return (uend + 15u) & ~15u; }
	// the global slab of a workgroup: forward weights, reverse weights (16 bytes per entry), spill of S; the walk slots of
	// the stretch construction borrow the weight part
	// (round 4) behind the spill image: the sorted k-mer instances and last k-mers of the current k, saved before the first
	// traversal of a pass overwrites overlay A, so that the next filter frequency pass reloads them instead of sorting again
	static constexpr uint32_t g_wF = 0, g_wR = 16u*CT::wcapg, g_spill = 32u*CT::wcapg, g_inst = ((32u*CT::wcapg + sbytes + 255u) & ~255u),
		g_bytes = g_inst + (((CT::precap + keycap)*8u + 255u) & ~255u);
	static_assert((o_pre & 15u) == 0 && (CT::precap & 1u) == 0 && (keycap & 1u) == 0,"instance arrays are saved 16 bytes at a time");
	static constexpr uint32_t xnslot = (32u*CT::wcapg) / 128u;
};
#undef FLD

template<typename CT>
HDEV FastCaps fastCapsOf(uint32_t const nrows, uint32_t const nsup)
{
	FastCaps C;
	C.maxs = CT::maxs; C.precap = CT::precap; C.ncap = CT::ncap; C.scap = CT::scap; C.lcap = CT::lcap; C.wcap = CT::wcap; C.rccap = CT::rccap;
	C.fcap = CT::fcap; C.siqcap = CT::siqcap; C.blcap = CT::blcap;
	C.nrows = nrows; C.nsup = nsup;
	C.ldsbytes = FastLds<CT>::bytes(nrows,nsup);
	C.tabcap = FastLds<CT>::tabcap;
	if constexpr ( CT::gw != 0 ) C.gbytes = FastLds<CT>::g_bytes; else C.gbytes = 0;
	return C;
}

struct FastBatch
{
	WindowBatch W;
	FastCaps F;
	uint64_t const * dpsq_vst;  // [nsup][nrows] transposed fixed-point table (HBM); copied to LDS as 32-bit
	uint32_t * retry;           // [0] = count, [1..] = window indices for the next capacity tier
	uint32_t * gearly;          // same layout: windows no LDS tier can run (w > 63, a string > 64): straight to the generic engine
	uint8_t * gslab;            // gw tiers: global scratch, gstride bytes per workgroup (weights, spill of the build-phase arrays)
	uint64_t gstride;
	uint32_t const * tab32;     // gw tiers: [nsup+1][nrows+1] 32 bit copy of the fixed-point table, zero row / column at the end
	// round 4: hand-over without restart.  A window that overflows a tier's node table leaves its sorted k-mer instances and last k-mer
	// list in a slot of this buffer (handwords 64 bit words per slot: header {npre, nlast}, instances, last k-mers); the tier that picks
	// the window up loads them instead of generating and sorting them again.  hand == 0: every hand-over restarts from the strings.
	uint64_t * hand; uint32_t * handctr; uint32_t handcap, handwords;
#if defined(DACC_LEDGER)
	uint32_t ledger;            // instruction ledger build (scripts/ledger.py): bit p set = phase p of every window runs twice
#endif
};

// once per workgroup: the support bounds of the model table
template<typename CT>
DEV void fast_load_tables(FastLds<CT> const & L, uint32_t const nrows, uint32_t const nsup, DevTables const & T, uint64_t const * vst)
{
	int const lane = wv_lane();
	for ( uint32_t i = lane; i < nsup; i += WSZ ) { L.suplo8()[i] = T.suplo[i]; L.suphi8()[i] = T.suphi[i]; }
	wv_sync();
}

// Instruction ledger (round 6, VERDICT r05 task 1b).  PC sampling is refused on this pool and thread trace has no decoder, but the SQ
// counters are per kernel: a -DDACC_LEDGER build runs the phase whose bit is set in FastBatch::ledger TWICE (every phase listed is
// idempotent: it rewrites the same bytes from inputs it does not modify), so the difference of SQ_INSTS_VALU / _SALU / _LDS / ... and of
// SQ_WAVE_CYCLES between a run with the bit and a run without is that phase's own instruction count and wave time, on the product's code
// generation, at the product's occupancy, with bit-identical output.  What cannot run twice (the lane 0 replay of the pairs, which feeds
// the candidate heap) is the remainder.  The product build compiles LEDGER_REP to nothing.
#if defined(DACC_LEDGER)
#define LEDGER_REP(bit) for ( uint32_t lrep_ = 0, lrn_ = 1u + ((ledger >> (bit)) & 1u); lrep_ < lrn_; ++lrep_ )
#define LEDGER_REPX(E,bit) for ( uint32_t lrep_ = 0, lrn_ = 1u + (((E).ledger >> (bit)) & 1u); lrep_ < lrn_; ++lrep_ )
#else
#define LEDGER_REP(bit)
#define LEDGER_REPX(E,bit)
#endif
#define FW_THRES_FEAS 4294968ull        /* weight >= 1e-3 */
#define FW_THRES_01   429496730ull      /* weight > 0.1 and weight >= 0.1 (no integer lies between) */
#define FW_THRES_05   2147483648ull     /* weight >= 0.5 */

#if defined(DACC_EMUL)
// emulation only: per-window maxima of the variable-size structures (design aid, DACC_EMUL_STATS=<file>)
struct FastStats { uint32_t v[32]; void clear() { for ( int i = 0; i < 32; ++i ) v[i] = 0; } void mx(int i, uint32_t x) { if ( x > v[i] ) v[i] = x; } void add(int i, uint32_t x) { v[i] += x; } };
static FastStats g_fstats;
static inline void fstats_dump(int const tier)
{
	static FILE * sf = 0; static bool tried = false;
	if ( !tried ) { tried = true; char const * fn = getenv("DACC_EMUL_STATS"); if ( fn ) sf = fopen(fn,"w"); }
	if ( sf ) { fprintf(sf,"%d",tier); for ( int i = 0; i < 32; ++i ) fprintf(sf," %u",g_fstats.v[i]); fprintf(sf,"\n"); fflush(sf); }
}
static inline FILE * ftrav_file() { static FILE * f = 0; static bool tried = false; if ( !tried ) { tried = true; char const * fn = getenv("DACC_EMUL_TRAV"); if ( fn ) f = fopen(fn,"w"); } return f; }
#define FSTAT_MX(i,x) g_fstats.mx(i,x)
#define FSTAT_ADD(i,x) g_fstats.add(i,x)
// design aid (DACC_EMUL_FEAS=<file>): per traversal the feasibility tasks' iteration counts, grouped as the 64 lanes would run them
#include <vector>
#include <algorithm>
struct FeasIters { std::vector<uint32_t> it, ln; void clear() { it.clear(); ln.clear(); } };
static FeasIters g_feasit;
static inline void feasit_dump()
{
	static FILE * f = 0; static bool tried = false;
	if ( !tried ) { tried = true; char const * fn = getenv("DACC_EMUL_FEAS"); if ( fn ) f = fopen(fn,"w"); }
	if ( !f ) { g_feasit.clear(); return; }
	std::vector<uint32_t> const & it = g_feasit.it; std::vector<uint32_t> const & ln = g_feasit.ln;
	uint64_t tot = 0, lock = 0, locksorted = 0;
	for ( size_t i = 0; i < it.size(); ++i ) tot += it[i];
	for ( size_t i = 0; i < it.size(); i += 64 ) { uint32_t m = 0; for ( size_t q = i; q < it.size() && q < i+64; ++q ) m = std::max(m,it[q]); lock += m; }
	std::vector< std::pair<uint32_t,uint32_t> > P; for ( size_t i = 0; i < it.size(); ++i ) P.push_back(std::make_pair(ln[i],it[i]));
	std::stable_sort(P.begin(),P.end(),[](std::pair<uint32_t,uint32_t> const & a, std::pair<uint32_t,uint32_t> const & b){ return a.first > b.first; });
	for ( size_t i = 0; i < P.size(); i += 64 ) { uint32_t m = 0; for ( size_t q = i; q < P.size() && q < i+64; ++q ) m = std::max(m,P[q].second); locksorted += m; }
	uint64_t lockcls = 0;
	{
		std::vector< std::pair<uint32_t,uint32_t> > Q; for ( size_t i = 0; i < it.size(); ++i ) { uint32_t c = 0; while ( (2u<<c) <= ln[i] ) ++c; Q.push_back(std::make_pair(c,it[i])); }
		std::stable_sort(Q.begin(),Q.end(),[](std::pair<uint32_t,uint32_t> const & a, std::pair<uint32_t,uint32_t> const & b){ return a.first > b.first; });
		for ( size_t i = 0; i < Q.size(); i += 64 ) { uint32_t m = 0; for ( size_t q = i; q < Q.size() && q < i+64; ++q ) m = std::max(m,Q[q].second); lockcls += m; }
	}
	fprintf(f,"%zu %llu %llu %llu %llu\n",it.size(),(unsigned long long)tot,(unsigned long long)lock,(unsigned long long)locksorted,(unsigned long long)lockcls);
	g_feasit.clear();
}
static inline bool feasit_on() { static int on = -1; if ( on < 0 ) on = getenv("DACC_EMUL_FEAS") ? 1 : 0; return on == 1; }      // single threaded design aid only
#define FEAS_ITERS(t,len,n) { if ( feasit_on() ) { g_feasit.it.push_back(n); g_feasit.ln.push_back(len); } }
#define FEAS_DUMP() { if ( feasit_on() ) feasit_dump(); }
#else
#define FEAS_ITERS(t,len,n)
#define FEAS_DUMP()
#define FSTAT_MX(i,x)
#define FSTAT_ADD(i,x)
#endif

template<typename CT>
struct FastEngine
{
	typedef typename CT::id_t id_t;      // path ids inside one enumeration
	typedef typename CT::sid_t sid_t;    // stretch ids (pool ids of the stretches and pieces of a traversal)
	enum : uint32_t { SB = 8u*sizeof(sid_t), SMASK = (1u<<SB)-1u, SNONE = SMASK, SMAX = CT::smax };      // SNONE: no stretch (SNONE of the 8 bit tiers)
	FastLds<CT> L; DevTables T; DevParams P;
	uint32_t nrows, nsup;
	uint64_t const * vst;        // [nsup][nrows] fixed-point model table in HBM
	enum : bool { GW = (CT::gw != 0) };
	uint8_t * gslab;             // gw: this workgroup's global slab
	uint32_t const * gtab;       // gw: padded 32 bit table in global memory
	struct G4 { uint32_t x, y, z, w; };      // one 16 byte weight record
	int lane; uint32_t flags;
#if defined(DACC_LEDGER)
	uint32_t ledger;
#endif
	uint64_t * prof;
	uint32_t mao, k; uint64_t kmask;
	uint32_t npre, nlast, nn, nmfirst, nmlast;
	uint32_t n0, npool, nlinks, nwF, nwR;
	uint32_t nF, nL;

#if defined(DACC_EMUL)
	// emulation only: DACC_EMUL_OVER=1 reports where a tier overflowed
	DEV void over_(uint32_t b, int line) { flags |= b; static char const * ov = getenv("DACC_EMUL_OVER"); if ( ov ) fprintf(stderr,"[over] tier maxs=%d line %d bits 0x%x nn=%u n0=%u npool=%u nF=%u nL=%u nwF=%u nwR=%u rstop=%u\n",int(CT::maxs)*10000+int(CT::ncap),line,b,nn,n0,npool,nF,nL,nwF,nwR,rstop); }
	#define over(b) over_(b,__LINE__)
#else
	DEV void over(uint32_t b) { flags |= b; }
#endif
#if defined(DACC_PROFILE) && !defined(DACC_EMUL)
	DEV void pcount(int id, uint64_t v) { if ( prof ) atomicAdd(reinterpret_cast<unsigned long long *>(prof+id),static_cast<unsigned long long>(v)); }
	DEV uint64_t pclock() { return clock64(); }
#else
	DEV void pcount(int, uint64_t) {}
	DEV uint64_t pclock() { return 0; }
#endif

#if defined(DACC_PROFILE) && !defined(DACC_EMUL)
	// fine sites (round 5 hot-spot ledger): shader cycles between two probes, charged once per wavefront visit (the first active
	// lane reports; inside lane-divergent code the clock is the wavefront's).  s_memtime waits for the LDS operations in flight, so a
	// site is charged with the round trips it started.
	#define SITE_T0 uint64_t _ps = clock64();
	#define SITE_RESET { _ps = clock64(); }
	#define SITE(id) { uint64_t const _n = clock64(); if ( prof && lane == static_cast<int>(__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) ) { atomicAdd(reinterpret_cast<unsigned long long *>(prof+32+(id)),static_cast<unsigned long long>(_n-_ps)); atomicAdd(reinterpret_cast<unsigned long long *>(prof+80+(id)),1ull); } _ps = clock64(); }
	#define PROFX_T0 uint64_t _px = clock64();
	#define PROFX(id) { uint64_t const _n = clock64(); if ( lane == 0 && prof ) atomicAdd(reinterpret_cast<unsigned long long *>(prof+(id)),static_cast<unsigned long long>(_n-_px)); _px = _n; }
#else
	#define PROFX_T0
	#define PROFX(id)
	#define SITE_T0
	#define SITE_RESET
	#define SITE(id)
#endif
	DEV int32_t findNode(uint32_t const v) const
	{
		int32_t lo = 0, hi = static_cast<int32_t>(nn)-1;
		while ( lo <= hi )
		{
			int32_t const mid = (lo+hi)>>1;
			uint32_t const x = L.nv()[mid];
			if ( x == v ) return mid;
			if ( x < v ) lo = mid+1; else hi = mid-1;
		}
		return -1;
	}
	DEV uint32_t lowerNode(uint32_t const v) const
	{
		uint32_t lo = 0, hi = nn;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( L.nv()[mid] < v ) lo = mid+1; else hi = mid; }
		return lo;
	}
	DEV int32_t succNode(uint32_t const z, uint32_t const i) const
	{
		uint32_t const sym = (L.sinfo()[z]>>(2*i))&3;
		uint32_t const target = static_cast<uint32_t>((static_cast<uint64_t>(L.nv()[z])<<2) & kmask) | sym;
		for ( uint32_t q = L.succ0()[z]; q < nn; ++q )
		{
			uint32_t const x = L.nv()[q];
			if ( x == target ) return q;
			if ( x > target ) break;
		}
		return -1;
	}
	DEV uint32_t nsucc(uint32_t z) const { return (L.sinfo()[z]>>8)&7; }
	DEV uint32_t nsuccact(uint32_t z) const { return (L.sinfo()[z]>>11)&7; }

	// ================= build: instances, nodes, successors =================
	DEV void buildInstances()
	{
		PROFX_T0
		// k-mer instances of all strings at once, lane = instance.  String j (lane j%64 of round j/64) has nk k-mers, its first
		// output slot is the exclusive prefix sum off, and among the strings with k-mers it is number rk.  (Round 5: an instance finds
		// its string in two LDS round trips -- a byte per output slot marks where a string begins, so the number of marks up to
		// slot t, counted with a ballot per 64 slots, names the string of slot t in a compact list (string, first slot).  Rounds 1-4
		// compared t against the offsets of all strings, broadcast one by one: 64 x strings scalar steps per 64 instances, 3 % of a
		// window of config 2 and a quarter of one at 54x, profiles/r05c_sites_*.log.)
		enum { NCH = (CT::maxs+WSZ-1)/WSZ };
#if defined(DACC_GEN_MARKS_NONE)
		enum : bool { GENMARKS = false };      // (A/B builds: the broadcast loop of rounds 1-4)
#else
		enum : bool { GENMARKS = true };
#endif
		static_assert(4u*CT::maxs <= CT::precap && CT::maxs <= 256 && CT::precap <= (1u<<24),"compact string list: string | first slot << 8, in the bytes of irpos");
		LDSQ uint8_t * const marks = L.ipos();                                             // free until buildNodes writes the positions
		LDSQ uint32_t * const clist = reinterpret_cast<LDSQ uint32_t *>(L.irpos());
		uint32_t nkc[NCH], offc[NCH], rkc[NCH]; uint32_t base = 0, nl = 0;
		#pragma unroll
		for ( int c = 0; c < NCH; ++c )
		{
			uint32_t const j = c*WSZ + lane;
			uint32_t const len = j < mao ? L.slen()[j] : 0u;
			nkc[c] = len >= k ? (len-k+1) : 0u;
			uint32_t tot; uint32_t const pre = wv_scan_excl(nkc[c],tot);
			offc[c] = base + pre; base += tot;
			uint32_t t2; rkc[c] = nl + wv_scan_flag(nkc[c] != 0,t2); nl += t2;
		}
		npre = base;
		nlast = 0;
		if ( npre > CT::precap ) { over(1); npre = 0; return; }
		SITE_T0
		if constexpr ( GENMARKS )
		{
			for ( uint32_t i = 8*lane; i < npre; i += 8*WSZ ) *reinterpret_cast<LDSQ uint64_t *>(marks + i) = 0;      // (precap is a multiple of 8)
			wv_sync();
			#pragma unroll
			for ( int c = 0; c < NCH; ++c )
				if ( nkc[c] ) { marks[offc[c]] = 1; clist[rkc[c]] = (c*WSZ + lane) | (offc[c] << 8); }
			wv_sync();
		}
		uint32_t cum = 0;
		for ( uint32_t t0 = 0; t0 < npre; t0 += WSZ )
		{
			uint32_t const t = t0 + lane;
			uint32_t j = 0, oj = 0, lo = 0;
			if constexpr ( GENMARKS )
			{
				bool const mk = t < npre && marks[t] != 0;
				uint32_t tot; uint32_t const before = wv_scan_flag(mk,tot);
				lo = cum + before + (mk ? 1u : 0u);      // strings with k-mers that begin at or before slot t (slot 0 begins one)
				cum += tot;
				if ( t < npre ) { uint32_t const e = clist[lo-1]; j = e & 0xFFu; oj = e >> 8; }
			}
			else
			{
				// rounds 1-4: t against the offsets of all strings, broadcast one by one (kept for A/B builds; the first measurement of
				// the marks on the shallow tiers, profiles/r05d_ab_generation.log, was 2.6 % slower only because that build also capped
				// the registers with amdgpu_waves_per_eu -- without the cap the marks are 1.9 % faster there, profiles/r05n_*)
				#pragma unroll
				for ( int c = 0; c < NCH; ++c )
					for ( uint32_t jj = 0; jj < WSZ && c*WSZ + jj < mao; ++jj )
					{
						uint32_t const o = wv_bcast(offc[c],jj), n = wv_bcast(nkc[c],jj);
						if ( t >= o ) { j = c*WSZ + jj; oj = o; lo += (n != 0); }
					}
			}
			if ( t < npre )
			{
				uint32_t const i = t - oj;
				uint32_t v = 0;
				if constexpr ( GW )
				{
					// symbols i ... i+k-1 from the pattern masks: low / high bit planes, reversed (symbol i is the most significant) and interleaved
					enum { PW = FastLds<CT>::pw };
					LDSQ uint64_t const * PEQ = L.peq() + 4*PW*j;
					uint32_t const km = (1u << k) - 1u;
					uint32_t lo, hi;
					if constexpr ( PW == 1 )
					{
						uint64_t const e1 = PEQ[1], e2 = PEQ[2], e3 = PEQ[3];
						lo = static_cast<uint32_t>((e1|e3) >> i) & km; hi = static_cast<uint32_t>((e2|e3) >> i) & km;
					}
					else
					{
						// (two words per mask, strings of up to 128 bases: the 16 bits from position i on may straddle the words)
						uint64_t const l0 = PEQ[1*PW] | PEQ[3*PW], l1 = PEQ[1*PW+1] | PEQ[3*PW+1], h0 = PEQ[2*PW] | PEQ[3*PW], h1 = PEQ[2*PW+1] | PEQ[3*PW+1];
						uint64_t const sl = i < 64 ? ((l0 >> i) | (i ? (l1 << (64u-i)) : 0ull)) : (l1 >> (i-64u));
						uint64_t const sh = i < 64 ? ((h0 >> i) | (i ? (h1 << (64u-i)) : 0ull)) : (h1 >> (i-64u));
						lo = static_cast<uint32_t>(sl) & km; hi = static_cast<uint32_t>(sh) & km;
					}
					v = dacc_spread16(dacc_rev32(lo) >> (32u-k)) | (dacc_spread16(dacc_rev32(hi) >> (32u-k)) << 1);
				}
				else
				{
					LDSQ uint8_t const * sp = L.str() + j*CT::lstr + i;
					#pragma unroll
					for ( uint32_t q = 0; q < 16; ++q ) { uint32_t const c = sp[q]; v = q < k ? ((v<<2) | c) : v; }
				}
				uint64_t const word = (static_cast<uint64_t>(v)<<32) | (static_cast<uint64_t>(i)<<16) | j;
				L.pre()[t] = word;
				if ( i + k == L.slen()[j] ) L.lastk()[lo-1] = word;
			}
		}
		nlast = nl;
		uint32_t const lp2 = next_pow2(nlast < 2 ? 2 : nlast);
		for ( uint32_t i = nlast + lane; i < lp2; i += WSZ ) L.lastk()[i] = ~0ull;
		wv_sync();
		SITE(31)      // buildInstances: generation of the k-mer instances (lane = instance)
		wv_sort_keys<FastLds<CT>::keycap>(L.lastk(),nlast);
		SITE(32)      // buildInstances: sort of the last k-mers
		wv_sort_keys<CT::precap,(CT::precap == 2048 && CT::ncap == 256)>(L.pre(),npre);      // 2048 keys in registers: the deep tier (FastTier<4>) only
		SITE(33)      // buildInstances: sort of the instances (register bitonic network)
	}

	DEV void buildNodes(uint32_t const f)
	{
		uint32_t base = 0;
		// (round 6, measured and dropped: run lengths from ballots of the head flags instead of every head walking its run: 0.45 % slower --
		// the runs are short (1.3 instances per k-mer), the extra pass over the sorted array is not, profiles/r06n)
		for ( uint32_t c = 0; c < npre; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t keep = 0, e = i;
			if ( i < npre && (i == 0 || (L.pre()[i]>>32) != (L.pre()[i-1]>>32)) )
			{
				uint64_t const km = L.pre()[i]>>32;
				e = i+1;
				while ( e < npre && (L.pre()[e]>>32) == km ) ++e;
				keep = (e-i) >= f;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(keep,tot);
			if ( keep )
			{
				uint32_t const z = base+pre;
				if ( z < CT::ncap ) { L.nv()[z] = static_cast<uint32_t>(L.pre()[i]>>32); L.nps()[z] = i; L.nfreq()[z] = (e-i) > 255 ? 255 : (e-i); if ( (e-i) > 255 ) over(2); }
			}
			base += tot;
		}
		nn = base;
		if ( nn > CT::ncap ) { over(2); nn = 0; }
		for ( uint32_t i = lane; i < npre; i += WSZ )
		{
			uint32_t const pos = (L.pre()[i]>>16)&0xFFFF, seq = L.pre()[i]&0xFFFF;
			// positions behind the support of the model table (a string much longer than the error profile makes likely)
			// carry no weight (`pos < first+size`, getKmerPositionWeight :3826-3864): they are stored as nsup, the all-zero
			// row behind the table copy, and count as "beyond the support" in the range lookups below
			uint32_t const rpos = L.slen()[seq]-pos-k;
			L.ipos()[i] = pos < nsup ? pos : nsup; L.irpos()[i] = rpos < nsup ? rpos : nsup;
		}
		wv_sync();
		if ( lane == 0 ) L.nps()[nn] = npre;
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const s0 = L.nps()[z], f2 = L.nfreq()[z];
			uint32_t const lo = L.ipos()[s0], hi = L.ipos()[s0+f2-1];
			uint32_t rlo = 255, rhi = 0;
			for ( uint32_t q = 0; q < f2; ++q ) { uint32_t const r = L.irpos()[s0+q]; rlo = r < rlo ? r : rlo; rhi = r > rhi ? r : rhi; }
			uint32_t const pf = lo < nsup ? L.suplo8()[lo] : nrows, pt = hi < nsup ? L.suphi8()[hi] : nrows;
			uint32_t const cf = rlo < nsup ? L.suplo8()[rlo] : nrows, ct = rhi < nsup ? L.suphi8()[rhi] : nrows;
			L.nrange()[z] = pf | (pt<<8) | (cf<<16) | (ct<<24);
		}
		uint32_t const kp2 = next_pow2(CT::maxs < 2 ? 2 : CT::maxs);
		base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t c0 = 0;
			if ( z < nn ) { uint32_t const s = L.nps()[z]; for ( uint32_t q = 0; q < L.nfreq()[z] && L.ipos()[s+q] == 0; ++q ) ++c0; }
			uint32_t tot; uint32_t const pre = wv_scan_excl(c0 ? 1 : 0,tot);
			if ( c0 && base+pre < kp2 ) L.mfirst()[base+pre] = ~((static_cast<uint64_t>(c0)<<32) | L.nv()[z]);
			base += tot;
		}
		nmfirst = base;
		if ( nmfirst > kp2 ) { over(8); nmfirst = 0; }
		uint32_t const p2 = next_pow2(nmfirst < 2 ? 2 : nmfirst);
		for ( uint32_t i = nmfirst + lane; i < p2; i += WSZ ) L.mfirst()[i] = ~0ull;
		base = 0;
		for ( uint32_t c = 0; c < nlast; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t const head = (i < nlast) && (i == 0 || (L.lastk()[i]>>32) != (L.lastk()[i-1]>>32));
			uint32_t tot; uint32_t const pre = wv_scan_excl(head,tot);
			if ( head )
			{
				uint32_t e = i+1;
				while ( e < nlast && (L.lastk()[e]>>32) == (L.lastk()[i]>>32) ) ++e;
				L.mlast()[base+pre] = ~((static_cast<uint64_t>(e-i)<<32) | (L.lastk()[i]>>32));
			}
			base += tot;
		}
		nmlast = base;
		uint32_t const q2 = next_pow2(nmlast < 2 ? 2 : nmlast);
		for ( uint32_t i = nmlast + lane; i < q2; i += WSZ ) L.mlast()[i] = ~0ull;
		wv_sync();
		wv_sort_keys<FastLds<CT>::keycap>(L.mfirst(),nmfirst);
		wv_sort_keys<FastLds<CT>::keycap>(L.mlast(),nmlast);
	}

	DEV void buildSuccessors(uint32_t const no)
	{
		uint32_t const lim = T.klim[(k-P.klow)*T.kln + (no < static_cast<uint32_t>(T.kln) ? no : T.kln-1)];
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const masked = static_cast<uint32_t>((static_cast<uint64_t>(L.nv()[z])<<2) & kmask);
			uint32_t const s0 = lowerNode(masked);
			// up to four successors (freq<<8 | symbol), absent = 0; sorted descending by a fixed network (keys are distinct)
			uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, n = 0;
			for ( uint32_t q = s0; q < nn && L.nv()[q] <= (masked|3); ++q )
			{
				uint32_t const kv = (static_cast<uint32_t>(L.nfreq()[q])<<8) | (L.nv()[q]&3);
				if ( n == 0 ) k0 = kv; else if ( n == 1 ) k1 = kv; else if ( n == 2 ) k2 = kv; else k3 = kv;
				++n;
			}
			#define DACC_CX(a,b) { uint32_t const hi_ = a > b ? a : b, lo_ = a > b ? b : a; a = hi_; b = lo_; }
			DACC_CX(k0,k1) DACC_CX(k2,k3) DACC_CX(k0,k2) DACC_CX(k1,k3) DACC_CX(k1,k2)
			#undef DACC_CX
			uint32_t act = 0;
			if ( n )
			{
				uint32_t const half = (k0>>8)/2;
				bool const a1 = n > 1 && ( ((k1>>8) >= half) || (P.checklim && ((k1>>8) >= lim)) );
				bool const a2 = n > 2 && ( ((k2>>8) >= half) || (P.checklim && ((k2>>8) >= lim)) );
				bool const a3 = n > 3 && ( ((k3>>8) >= half) || (P.checklim && ((k3>>8) >= lim)) );
				act = 1 + (a1 ? (1 + (a2 ? (1 + (a3 ? 1 : 0)) : 0)) : 0);
			}
			uint32_t const order = (k0&3) | ((k1&3)<<2) | ((k2&3)<<4) | ((k3&3)<<6);
			L.succ0()[z] = s0;
			L.sinfo()[z] = order | (n<<8) | (act<<11);
		}
		wv_sync();
		sfresh = true;      // S has been rebuilt (instances / nodes / successors of this pass): the next traversal spills all of it
	}
	// after buildNodes: no first k-mer candidate, or none of the last k-mer candidates (count >= 3/4 of the best, :4774-4784)
	// is a node of the filtered graph
	DEV bool passIsDead() const
	{
		if ( nmfirst == 0 || nmlast == 0 ) return true;
		uint32_t const lastthres = ((static_cast<uint32_t>((~L.mlast()[0])>>32))*3)/4;
		uint32_t any = 0;
		for ( uint32_t i = lane; i < nmlast; i += WSZ )
		{
			uint64_t const e = ~L.mlast()[i];
			if ( static_cast<uint32_t>(e>>32) >= lastthres && findNode(static_cast<uint32_t>(e)) >= 0 ) any = 1;
		}
		return !wv_any(any);
	}
	DEV bool addNextFromHeap()
	{
		uint32_t best = 0;
		for ( uint32_t z = lane; z < nn; z += WSZ )
			if ( nsuccact(z) < nsucc(z) )
			{
				uint32_t const fq = L.nfreq()[succNode(z,nsuccact(z))];
				best = fq > best ? fq : best;
			}
		best = wv_max(best);
		if ( ! best ) return false;
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t a = nsuccact(z); uint32_t const n = nsucc(z);
			while ( a < n && L.nfreq()[succNode(z,a)] == best ) ++a;
			L.sinfo()[z] = (L.sinfo()[z] & 0x7FF) | (a<<11);
		}
		wv_sync();
		sdirty = true;
		return true;
	}

	// ================= gap filling at filter frequency 0 (getLevelSuccessors(2), :1016-1161) =================
	// feasible position list of node z is implicit: p in [pfrom,pto) with nodeU(z,p) >= 1e-3
	DEV void levelSuccessors2()
	{
		uint32_t const s = 2;
		LDSQ uint64_t * REC = L.gfbuf();          // records (from<<48 | to<<32 | cv)
		uint32_t const cap = FastLds<CT>::gfcap/2;    // second half holds the (cv,pos) candidates
		LDSQ uint64_t * ANE = L.gfbuf() + cap;
		// pass 1: missing intermediate k-mers between nodes two steps apart
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t cnt = 0; uint32_t lo = 0; uint32_t low = 0, vhigh = 0;
			if ( i < nn )
			{
				uint32_t const v = L.nv()[i];
				low = static_cast<uint32_t>((static_cast<uint64_t>(v)<<(2*s)) & kmask);
				vhigh = static_cast<uint32_t>((static_cast<uint64_t>(v)<<2) & kmask);
				lo = lowerNode(low);
				for ( uint32_t q = lo; q < nn && L.nv()[q] <= (low|0xF); ++q )
					if ( findNode((L.nv()[q]>>2)|vhigh) < 0 ) ++cnt;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( i < nn && cnt && base+pre+cnt <= cap )
			{
				uint32_t o = base+pre;
				for ( uint32_t q = lo; q < nn && L.nv()[q] <= (low|0xF); ++q )
				{
					uint32_t const cv = (L.nv()[q]>>2)|vhigh;
					if ( findNode(cv) < 0 ) REC[o++] = (static_cast<uint64_t>(i)<<48) | (static_cast<uint64_t>(q)<<32) | cv;
				}
			}
			base += tot;
		}
		uint32_t const nrec = base;
		if ( nrec > cap ) { over(8192); return; }
		wv_sync();
		// pass 2: best common feasible position of `from` (shifted by s) and `to`
		base = 0;
		for ( uint32_t c = 0; c < nrec; c += WSZ )
		{
			uint32_t const r = c + lane;
			uint32_t have = 0; uint64_t cand = 0;
			if ( r < nrec )
			{
				uint32_t const from = REC[r]>>48, to = (REC[r]>>32)&0xFFFF; uint32_t const cv = static_cast<uint32_t>(REC[r]);
				uint64_t mweight = 0; uint32_t mp = 0;
				// common position pp: from feasible at pp-s, to feasible at pp; ascending pp, strict > keeps the first maximum
				uint32_t const rto = L.nrange()[to], rfrom = L.nrange()[from];
				for ( uint32_t pp = rto & 0xFF; pp < ((rto>>8)&0xFF); ++pp )
				{
					if ( pp < s ) continue;
					uint32_t const pf = pp-s;
					if ( pf < (rfrom&0xFF) || pf >= ((rfrom>>8)&0xFF) ) continue;
					uint64_t const ua = nodeU(from,pf,false), ub = nodeU(to,pp,false);
					if ( ua < FW_THRES_FEAS || ub < FW_THRES_FEAS ) continue;
					uint64_t const weight = ua+ub;
					if ( !have || weight > mweight ) { have = 1; mweight = weight; mp = pp - s + 1; }
				}
				cand = (static_cast<uint64_t>(cv)<<8) | mp;
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(have,tot);
			if ( have ) ANE[base+pre] = cand;
			base += tot;
		}
		uint32_t const nane = base;
		uint32_t const p2 = next_pow2(nane < 2 ? 2 : nane);
		if ( p2 > cap ) { over(8192); return; }
		for ( uint32_t i = nane + lane; i < p2; i += WSZ ) ANE[i] = ~0ull;
		wv_sync();
		wv_bitonic_sort(ANE,p2);       // (v,pos) ascending, NodeAddElement::operator<
		// pass 3: attach each candidate to the first string that is long enough and append the instance
		base = 0;
		for ( uint32_t c = 0; c < nane; c += WSZ )
		{
			uint32_t const i = c + lane;
			uint32_t ok = 0; uint32_t seqid = 0, pos = 0; uint64_t cv = 0;
			if ( i < nane )
			{
				pos = ANE[i]&0xFF; cv = ANE[i]>>8;
				for ( uint32_t j = 0; j < mao; ++j ) if ( pos + k <= L.slen()[j] ) { seqid = j; ok = 1; break; }
			}
			uint32_t tot; uint32_t const pre = wv_scan_excl(ok,tot);
			if ( ok && npre+base+pre < CT::precap ) L.pre()[npre+base+pre] = (cv<<32) | (static_cast<uint64_t>(pos)<<16) | seqid;
			base += tot;
		}
		if ( npre + base > CT::precap ) { over(1); return; }
		npre += base;
		wv_sync();
		wv_sort_keys<CT::precap>(L.pre(),npre);
	}

	// ================= stretches, once per activation state =================
	// Active predecessor counts and the walking table of the stretches, both from the successor side: node u adds one to
	// each of its active successors (LDS atomics on a scratch word array) instead of every node searching its four
	// possible predecessors; stepT[z] = the node a walk continues with from z (exactly one active successor and one active
	// predecessor), 0xFFFF where a stretch ends.  Both tables borrow the weight arrays, which are written much later.
	DEV LDSQ uint16_t * stepTable() const { return L.xstep(); }
	DEV void computePredCounts()
	{
		// (round 5: four 8 bit counters per word -- a node has at most four predecessors --, so that the scratch is 3 bytes per node
		// instead of 6 and stops being what binds the tiers' last region when their node tables grow)
		LDSQ uint32_t * const cnt32 = L.xcnt32();
		LDSQ uint16_t * const stepT = stepTable();
		for ( uint32_t z = lane; 4u*z < nn; z += WSZ ) cnt32[z] = 0;
		wv_sync();
		for ( uint32_t u = lane; u < nn; u += WSZ )
		{
			uint32_t const na = nsuccact(u);
			uint32_t first = 0xFFFF;
			for ( uint32_t i = 0; i < na; ++i ) { int32_t const v = succNode(u,i); wv_atomic_add(cnt32+(static_cast<uint32_t>(v)>>2),1u << (8u*(static_cast<uint32_t>(v)&3u))); if ( i == 0 ) first = v; }
			stepT[u] = (na == 1) ? first : 0xFFFF;
		}
		wv_sync();
		for ( uint32_t z = lane; z < nn; z += WSZ )
		{
			uint32_t const c = (cnt32[z>>2] >> (8u*(z&3u))) & 0xFFu;
			L.npred()[z] = c;
			if ( c != 1 ) stepT[z] = 0xFFFF;
		}
		wv_sync();
	}
	// out (if given) receives the first `cap` nodes
	template<typename OUTP>
	DEV uint32_t walkStretch(uint32_t const z, uint32_t const i, OUTP out, uint32_t & lastnode, uint32_t const cap = 0xFFFFFFFFu)
	{
		LDSQ uint16_t const * const stepT = stepTable();
		int32_t cur = succNode(z,i);
		uint32_t len = 2;
		if ( out ) { out[0] = z; out[1] = cur; }
		bool loop = (cur == static_cast<int32_t>(z));
		while ( !loop )
		{
			uint32_t const nx = stepT[cur];
			if ( nx == 0xFFFF ) break;
			cur = nx;
			if ( out && len < cap ) out[len] = cur;
			++len;
			if ( cur == static_cast<int32_t>(z) ) loop = true;
			if ( len > nn+1 ) { over(16); break; }
		}
		lastnode = cur;
		return len;
	}
	// (first, ext, len desc, last) packed into 56 bits: Stretch::operator< (node ids ascend with k-mer value)
	// (8 bit stretch ids: four 14 bit fields = 56 bits next to the id; 16 bit ids: four 12 bit fields = 48 bits)
	enum : uint32_t { KF = (sizeof(sid_t) == 1 ? 14u : 12u), KFMAX = (1u<<KF)-1u };
	DEV static uint64_t sortKey(uint32_t first, uint32_t ext, uint32_t len, uint32_t last)
	{
		return (static_cast<uint64_t>(first)<<(3*KF)) | (static_cast<uint64_t>(ext)<<(2*KF)) | (static_cast<uint64_t>(KFMAX-len)<<KF) | last;
	}
	DEV uint64_t poolKey(uint32_t const s) const { uint32_t const len = L.sslen()[s]; return sortKey(L.sfirst()[s],L.links()[L.slink()[s]+1],len < KFMAX ? len : KFMAX,L.slast()[s]); }

	// raw stretches of the current activation state, sorted by Stretch::operator< (pool ids 0..n0-1)
	DEV void computeBaseStretches()
	{
		computePredCounts();
		uint32_t base = 0;
		for ( uint32_t c = 0; c < nn; c += WSZ )
		{
			uint32_t const z = c + lane;
			uint32_t cnt = 0;
			if ( z < nn ) { uint32_t const ns = nsuccact(z); if ( ns && (L.npred()[z] != 1 || ns > 1) ) cnt = ns; }
			uint32_t tot; uint32_t const pre = wv_scan_excl(cnt,tot);
			if ( base+pre+cnt <= CT::scap )
				for ( uint32_t i = 0; i < cnt; ++i ) { L.tfirst()[base+pre+i] = z; L.tlast()[base+pre+i] = i; }
			base += tot;
		}
		uint32_t const ns = base;
		if ( ns + 2 > CT::scap || ns > SMAX || nn >= KFMAX ) { over(32); n0 = 0; return; }
		wv_sync();
		// walk every stretch once: the nodes go to a scratch slot of WSLOT entries per stretch in the (not yet used) weight
		// arrays and are compacted into `links` below; a stretch without a slot or longer than it is walked a second time
		enum { WSLOT = 64 };
		// (round 6, measured and dropped: walking every stretch twice instead of sending its nodes through the slab: 0.4 % slower, profiles/r06f)
		uint32_t const nslot = FastLds<CT>::xnslot;     // legacy: the walking table sits behind the slots; gw: slots in the global slab
		for ( uint32_t q = lane; q < ns; q += WSZ )
		{
			uint32_t ln;
			uint32_t const sub = L.tlast()[q];
			if constexpr ( GW )
			{
				uint16_t * const wtmp = reinterpret_cast<uint16_t *>(gslab);
				L.tslen()[q] = walkStretch(L.tfirst()[q],sub,q < nslot ? wtmp + q*WSLOT : static_cast<uint16_t *>(0),ln,WSLOT);
			}
			else
			{
				LDSQ uint16_t * const wtmp = L.xwtmp();
				L.tslen()[q] = walkStretch(L.tfirst()[q],sub,q < nslot ? wtmp + q*WSLOT : static_cast<LDSQ uint16_t *>(0),ln,WSLOT);
			}
			L.skey()[q] = (static_cast<uint64_t>(sub)<<32) | ln;     // successor index (for a second walk) and last node
		}
		wv_sync();
		base = 0;
		for ( uint32_t c = 0; c < ns; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const len = q < ns ? L.tslen()[q] : 0;
			uint32_t tot; uint32_t const pre = wv_scan_excl(len,tot);
			if ( q < ns ) L.tlink()[q] = base+pre;
			base += tot;
		}
		nlinks = base;
		if ( nlinks > CT::lcap ) { over(64); n0 = 0; return; }
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ )
		{
			uint32_t const len = L.tslen()[q]; uint64_t const sk = L.skey()[q];
			LDSQ uint16_t * const dst = L.links() + L.tlink()[q];
			if ( q < nslot && len <= WSLOT )
			{
				if constexpr ( GW ) { uint16_t const * src = reinterpret_cast<uint16_t const *>(gslab) + q*WSLOT; for ( uint32_t t = 0; t < len; ++t ) dst[t] = src[t]; }
				else { LDSQ uint16_t const * src = L.xwtmp() + q*WSLOT; for ( uint32_t t = 0; t < len; ++t ) dst[t] = src[t]; }
			}
			else { uint32_t ln; walkStretch(L.tfirst()[q],static_cast<uint32_t>(sk>>32),dst,ln); }
			L.tlast()[q] = static_cast<uint32_t>(sk);
		}
		wv_sync();
		uint32_t const p2 = next_pow2(ns < 2 ? 2 : ns);
		for ( uint32_t q = lane; q < p2; q += WSZ )
			L.skey()[q] = q < ns ? ((sortKey(L.tfirst()[q],L.links()[L.tlink()[q]+1],L.tslen()[q] < KFMAX ? L.tslen()[q] : KFMAX,L.tlast()[q])<<SB) | q) : ~0ull;
		wv_sync();
		wv_sort_keys<fcpow2(CT::scap)>(L.skey(),ns);
		// distinct (first,ext) by construction; a duplicate would need stretchesUnique's tie handling -> generic engine
		uint32_t dup = 0;
		for ( uint32_t q = lane; q < ns; q += WSZ )
		{
			if ( q && (L.skey()[q]>>(SB+2*KF)) == (L.skey()[q-1]>>(SB+2*KF)) ) dup = 1;
			uint32_t const raw = L.skey()[q]&SMASK;
			L.sfirst()[q] = L.tfirst()[raw]; L.slast()[q] = L.tlast()[raw]; L.sslen()[q] = L.tslen()[raw]; L.slink()[q] = L.tlink()[raw];
		}
		if ( wv_any(dup) ) { over(32); n0 = 0; return; }
		n0 = ns;
		wv_sync();
		// lookup by first node (base order is sorted by it) and by last node (ordered by (last, id))
		for ( uint32_t z = lane; z < nn; z += WSZ ) { L.npred()[z] = SNONE; L.lhead()[z] = SNONE; }
		for ( uint32_t q = lane; q < p2; q += WSZ ) L.skey()[q] = q < ns ? ((static_cast<uint64_t>(L.slast()[q])<<SB) | q) : ~0ull;
		wv_sync();
		for ( uint32_t q = lane; q < ns; q += WSZ ) if ( q == 0 || L.sfirst()[q-1] != L.sfirst()[q] ) L.npred()[L.sfirst()[q]] = q;
		wv_sort_keys<fcpow2(CT::scap)>(L.skey(),ns);
		for ( uint32_t q = lane; q < ns; q += WSZ )
		{
			uint64_t const e = L.skey()[q];
			L.lord()[q] = static_cast<uint32_t>(e);
			if ( q == 0 || (L.skey()[q-1]>>SB) != (e>>SB) ) L.lhead()[e>>SB] = q;
		}
		wv_sync();
	}

	// pool stretch id = nodes [a,b] of parent stretch par
	DEV void makePiece(uint32_t const id, uint32_t const par, uint32_t const a, uint32_t const b)
	{
		LDSQ uint16_t const * Lk = L.links() + L.slink()[par];
		L.sfirst()[id] = Lk[a]; L.slast()[id] = Lk[b]; L.sslen()[id] = b-a+1; L.slink()[id] = L.slink()[par]+a;
	}
	// candidates + the pool stretch each one splits (splitStretches :2772-2841: first occurrence strictly inside)
	DEV void findCandidatesAndPieces()
	{
		uint32_t const firstthres = nmfirst ? ((static_cast<uint32_t>((~L.mfirst()[0])>>32))*3)/4 : 0;
		uint32_t const lastthres = nmlast ? ((static_cast<uint32_t>((~L.mlast()[0])>>32))*3)/4 : 0;
		uint32_t cf = 0, cl = 0;
		while ( cf < nmfirst && static_cast<uint32_t>((~L.mfirst()[cf])>>32) >= firstthres ) ++cf;
		while ( cl < nmlast && static_cast<uint32_t>((~L.mlast()[cl])>>32) >= lastthres ) ++cl;
		nF = cf; npool = n0;
		if ( nF > FNC ) { nL = cl; over(8); return; }
		for ( uint32_t i = lane; i < nF; i += WSZ ) { uint32_t const km = static_cast<uint32_t>(~L.mfirst()[i]); L.fkmer()[i] = km; int32_t const z = findNode(km); L.fnode()[i] = z < 0 ? 0xFFFF : z; L.parF()[i] = SNONE; L.pieF()[i] = SNONE; }
		// Last k-mer candidates that are not nodes of the (filtered) graph are left out: their reverse enumeration is empty
		// (prepareTraverse starts from the node of `last`), so none of their pairs can offer a candidate, and the pairs of
		// the others keep their relative order.  In deep piles most strings end in a k-mer of their own, the list of
		// candidates with a count of at least 3/4 of a small best count is then as long as the pile is deep.
		{
			uint32_t nl = 0;
			for ( uint32_t c = 0; c < cl; c += WSZ )
			{
				uint32_t const i = c + lane;
				uint32_t const km = i < cl ? static_cast<uint32_t>(~L.mlast()[i]) : 0u;
				int32_t const z = i < cl ? findNode(km) : -1;
				uint32_t tot; uint32_t const pre = wv_scan_flag(z >= 0,tot);
				uint32_t const o = nl + pre;
				if ( z >= 0 && o < FNC ) { L.lkmer()[o] = km; L.lnode()[o] = z; L.parL()[o] = SNONE; L.pieL()[o] = SNONE; }
				nl += tot;
			}
			nL = nl;
		}
		if ( nL > FNC ) { over(8); return; }
		wv_sync();
		// parents: lanes over base stretches.  An interior node has a unique active predecessor and successor, so it
		// lies strictly inside at most one stretch; anything else goes to the generic engine
		uint32_t multi = 0;
		{
			// node -> (stretch, index) for the interior nodes, in the not yet used weight arrays: one pass over the stretches
			// (lane per stretch, indices downwards so that the first occurrence of a node stays), then one lookup per candidate
			LDSQ uint16_t * const sid = L.xsid();
			LDSQ uint8_t * const spos = L.xspos();
			for ( uint32_t z = lane; z < nn; z += WSZ ) sid[z] = 0xFFFF;
			wv_sync();
			for ( uint32_t s = lane; s < n0; s += WSZ )
			{
				uint32_t const len = L.sslen()[s]; LDSQ uint16_t const * Lk = L.links() + L.slink()[s];
				for ( uint32_t i = len > 2 ? len-2 : 0; i >= 1; --i ) { uint32_t const z = Lk[i]; sid[z] = s; spos[z] = i; }
			}
			wv_sync();
			// safety net: a node strictly inside two stretches cannot be modelled by one parent per candidate
			for ( uint32_t s = lane; s < n0; s += WSZ )
			{
				uint32_t const len = L.sslen()[s]; LDSQ uint16_t const * Lk = L.links() + L.slink()[s];
				for ( uint32_t i = 1; i+1 < len; ++i ) if ( sid[Lk[i]] != s ) multi = 1;
			}
			for ( uint32_t c = lane; c < nF; c += WSZ ) { uint32_t const z = L.fnode()[c]; if ( z != 0xFFFF && sid[z] != 0xFFFF ) { L.parF()[c] = sid[z]; L.posF()[c] = spos[z]; } }
			for ( uint32_t c = lane; c < nL; c += WSZ ) { uint32_t const z = L.lnode()[c]; if ( z != 0xFFFF && sid[z] != 0xFFFF ) { L.parL()[c] = sid[z]; L.posL()[c] = spos[z]; } }
		}
		if ( wv_any(multi) ) { over(32); return; }
		wv_sync();
		{
			uint32_t cnt = 0;
			for ( uint32_t c = 0; c < nF; ++c ) cnt += (L.parF()[c] != SNONE);
			for ( uint32_t c = 0; c < nL; ++c ) cnt += (L.parL()[c] != SNONE);
			if ( n0 + 2*cnt + 1 > CT::scap || n0 + 2*cnt + 1 > SMAX ) { over(32); return; }
		}
		{
			// pieces of the candidates that split a stretch, one lane per candidate (first k-mers, then last k-mers): two
			// pool ids each, numbered in candidate order
			uint32_t id = n0;
			for ( uint32_t c0 = 0; c0 < nF + nL; c0 += WSZ )
			{
				uint32_t const c = c0 + lane;
				bool const isF = c < nF, act = c < nF + nL;
				uint32_t const ci = isF ? c : c - nF;
				uint32_t const par = act ? (isF ? L.parF()[ci] : L.parL()[ci]) : static_cast<uint32_t>(SNONE);
				bool const has = act && par != SNONE;
				uint32_t tot; uint32_t const pre = wv_scan_flag(has,tot);
				if ( has )
				{
					uint32_t const pid = id + 2*pre, pos = isF ? L.posF()[ci] : L.posL()[ci];
					if ( isF ) L.pieF()[ci] = pid; else L.pieL()[ci] = pid;
					makePiece(pid,par,0,pos); makePiece(pid+1,par,pos,L.sslen()[par]-1);
				}
				id += 2*tot;
			}
			npool = id;
		}
		wv_sync();
		// Middle pieces.  A first and a last k-mer candidate that lie strictly inside the SAME stretch at different nodes make a
		// pair whose stretch set holds three parts of that stretch (split at first, then at last: splitStretches :2772-2841 applied
		// twice); the part between the two nodes belongs to no candidate alone.  Every such pair is replayed on its exact stretch
		// set (classifyPair: 4), so the middle pieces a traversal will need are known here: they join the pool now and get their
		// feasibility with all other stretches, while the node tables are still in LDS.  (Until round 3 they were appended when
		// their pair came up, with a feasibility pass of their own that read the model table from HBM -- which the gw layout
		// cannot do, its node tables are spilled then, and whose registers cost the gw tiers their second wavefront per SIMD.)
		nmid = 0; midbase = npool;
		{
			uint64_t pm = 0;
			for ( uint32_t c = lane; c < nF; c += WSZ ) { uint32_t const par = L.parF()[c]; if ( par != SNONE ) pm |= 1ull << (par & 63u); }
			pm = wv_or64(pm);
			uint32_t hit = 0;
			for ( uint32_t c = lane; c < nL; c += WSZ ) { uint32_t const par = L.parL()[c]; if ( par != SNONE && ((pm >> (par & 63u)) & 1ull) ) hit = 1; }
			if ( wv_any(hit) )
			{
				uint32_t nm = 0;
				if ( lane == 0 )
				{
					for ( uint32_t f = 0; f < nF && nm <= MIDCAP; ++f )
					{
						uint32_t const pf = L.parF()[f];
						if ( pf == SNONE ) continue;
						for ( uint32_t l = 0; l < nL && nm <= MIDCAP; ++l )
						{
							if ( L.parL()[l] != pf ) continue;
							uint32_t const a = L.posF()[f], b = L.posL()[l];
							if ( a == b ) continue;
							uint32_t const lo = a < b ? a : b, hi = a < b ? b : a;
							uint32_t m = 0;
							while ( m < nm && !(L.midpar()[m] == pf && L.midA()[m] == lo && L.midB()[m] == hi) ) ++m;
							if ( m < nm ) continue;
							if ( nm < MIDCAP ) { L.midpar()[nm] = pf; L.midA()[nm] = lo; L.midB()[nm] = hi; }
							++nm;      // MIDCAP+1: more middle pieces than the list holds
						}
					}
				}
				wv_sync();
				nm = wv_bcast(nm,0);
				if ( nm > MIDCAP || npool + nm > CT::scap || npool + nm > SMAX ) { over(32); return; }
				for ( uint32_t m = lane; m < nm; m += WSZ ) makePiece(npool+m,L.midpar()[m],L.midA()[m],L.midB()[m]);      // (nm <= MIDCAP < 64: one round on the device; the 1-lane emulation needs the loop)
				nmid = nm; npool += nm;
				wv_sync();
			}
		}
		for ( uint32_t id = n0 + lane; id < npool; id += WSZ ) L.ppos()[id] = basePos(id);
		wv_sync();

	}

	// Can any (first, last) candidate pair of this activation state offer a candidate at all?  A candidate is a chain of
	// view stretches from the first k-mer to the last k-mer (forward path, junction, reverse path: traverse :4838-5097), i.e.
	// a walk of at least one active edge from the node of `first` to the node of `last`.  Base stretches cover every active
	// edge that can be reached from a branch node (computeStretches :2844-2986), pieces are parts of them, so a breadth first
	// search over the base stretches from the first k-mer candidates decides it: if it reaches no last k-mer candidate,
	// traverse() has nothing to enumerate and returns false (ACCo == 0) -- feasibility, both enumerations and the pairs are
	// skipped.  The condition is necessary only (feasibility and the length bounds are not looked at): a reachable pair
	// goes through the enumerations as before.  Scratch: the weight arrays, which are written after this.
	DEV bool pairReachable()
	{
		LDSQ uint8_t * const seedN = L.xreach();
		LDSQ uint8_t * const reachN = seedN + CT::ncap;
		LDSQ uint8_t * const vis = reachN + CT::ncap;
		if ( nF == 0 || nL == 0 ) return false;
		for ( uint32_t z = lane; z < nn; z += WSZ ) { seedN[z] = 0; reachN[z] = 0; }
		for ( uint32_t s = lane; s < n0; s += WSZ ) vis[s] = 0;
		wv_sync();
		// a first k-mer at a stretch end starts the stretches that begin there; one strictly inside a stretch leads to its end
		for ( uint32_t c = lane; c < nF; c += WSZ )
		{
			uint32_t const z = L.fnode()[c], par = L.parF()[c];
			if ( z == 0xFFFF ) continue;
			if ( par == SNONE ) seedN[z] = 1; else reachN[L.slast()[par]] = 1;
		}
		wv_sync();
		while ( true )
		{
			uint32_t changed = 0;
			for ( uint32_t s = lane; s < n0; s += WSZ )
				if ( !vis[s] )
				{
					uint32_t const f = L.sfirst()[s];
					if ( seedN[f] | reachN[f] ) { vis[s] = 1; reachN[L.slast()[s]] = 1; changed = 1; }
				}
			wv_sync();
			if ( !wv_any(changed) ) break;
		}
		uint32_t ok = 0;
		for ( uint32_t c = lane; c < nL; c += WSZ )
		{
			uint32_t const z = L.lnode()[c], par = L.parL()[c];
			if ( z == 0xFFFF ) continue;
			if ( par == SNONE ) ok |= reachN[z];
			else
			{
				ok |= vis[par];      // entered through its first node
				uint32_t const pl = L.posL()[c];
				for ( uint32_t f = 0; f < nF; ++f ) if ( L.parF()[f] == par && L.posF()[f] < pl ) ok = 1;     // a first k-mer before it inside the same stretch
			}
		}
		bool const res = wv_any(ok) != 0;
		wv_sync();
		return res;
	}

	// copy of the model table in LDS, row stride nrows+1: the extra row is zero so that positions beyond the table can be
	// clamped instead of branched on; one more all-zero position (nsup) for read positions behind the support
	DEV void loadTab()
	{
		if constexpr ( GW ) return;      // gw tiers read the padded 32 bit table from global memory (L1 / L2 resident) where they need it
		else {
		uint32_t const stride = nrows+1;
		uint32_t pos = static_cast<uint32_t>(lane) / stride, row = static_cast<uint32_t>(lane) - pos*stride;     // the one division of the copy
		uint32_t const dpos = WSZ / stride, drow = WSZ - dpos*stride;
		for ( uint32_t i = lane; i < stride*(nsup+1); i += WSZ )
		{
			L.tab()[i] = (row < nrows && pos < nsup) ? static_cast<uint32_t>(vst[pos*nrows+row]) : 0u;
			pos += dpos; row += drow; if ( row >= stride ) { row -= stride; ++pos; }
		}
		wv_sync();
		}
	}
	// entry [pos][pc] of the padded table (row stride nrows+1, zero row nrows, zero position nsup)
	DEV uint32_t tabR(uint32_t const idx) const
	{
		if constexpr ( GW ) return gtab[idx];
		else return L.tab()[idx];
	}
	// ---- stretch feasibility, one LANE per (stretch, direction, start position) ----
	// The wavefront-per-stretch form above walks the stretches one after the other with lanes = start positions and leaves
	// most lanes idle (few positions survive the node supports).  Here one lane per (stretch, direction) first intersects
	// the support ranges of its nodes; the surviving (stretch, direction, position) triples are then numbered and
	// evaluated 64 at a time, a stretch after the other in each direction, so that the feasible ones can be appended to
	// the weight lists with a ballot.  Same sums in the same order as computeStretchFeas, entries in ascending start
	// position.  The loads of a node (link -> node -> first instance -> table) run ahead of the table reads.
	DEV auto unitRecords() const
	{
		if constexpr ( GW ) return L.urec();
		else { static_assert(2*CT::scap < 512,"unit ids are packed into 9 bits of a 16 bit record"); return reinterpret_cast<LDSQ uint16_t *>(L.ulo()); }
	}
	DEV void computeStretchFeasLanes(uint32_t const sfrom, uint32_t const sto)
	{
		PROFX_T0
		uint32_t const ns = sto-sfrom, nu = 2*ns;          // units: forward stretches, then reverse stretches
		uint32_t const stride = nrows+1;
		// The tasks of a round run in lock step for as long as the longest of them walks its stretch, so the units are
		// processed in classes of decreasing stretch length (lock step iterations per traversal at config 2: 248 in pool
		// order, 97 sorted by length, about 105 in these classes; 65 is the sum of the iterations over 64).  Any order of
		// the units gives the same weights: a unit's entries stay consecutive and ascending in its own list, only the place
		// of the list in the weight arrays changes.  Position q of the order holds (lo | unit << 7) in the bytes of ulo/uhi.
		enum { NCHU = (2*CT::scap + WSZ - 1)/WSZ, NCLS = 12 };
		// (legacy layout: 16 bit records over the bytes of ulo / uhi, at most 511 units; gw layout: 32 bit records)
		auto const urec = unitRecords();
		uint32_t ulo_r[NCHU], uw_r[NCHU], ucls_r[NCHU], upos_r[NCHU];
		#pragma unroll
		for ( uint32_t cc = 0; cc < NCHU; ++cc )
		{
			uint32_t const u = cc*WSZ + lane;
			uint32_t w = 0, lo_ = 0, cls = NCLS;
			if ( u < nu )
			{
				bool const rev = u >= ns; uint32_t const s = sfrom + (rev ? u-ns : u);
				uint32_t const len = L.sslen()[s]; LDSQ uint16_t const * Lk = L.links() + L.slink()[s];
				int32_t lo = 0, hi = static_cast<int32_t>(nrows);
				uint32_t const sh = rev ? 16 : 0;
				for ( uint32_t j = 0; j < len; ++j )
				{
					uint32_t const g = L.nrange()[Lk[rev ? (len-1-j) : j]] >> sh;
					int32_t const jj = static_cast<int32_t>(j);
					int32_t const a = static_cast<int32_t>(g&0xFF)-jj, b = static_cast<int32_t>((g>>8)&0xFF)-jj;
					lo = a > lo ? a : lo; hi = b < hi ? b : hi;
				}
				if ( hi < lo ) hi = lo;
				w = static_cast<uint32_t>(hi-lo); lo_ = static_cast<uint32_t>(lo);
				cls = len >= 49 ? 0u : len >= 33 ? 1u : len >= 25 ? 2u : len >= 17 ? 3u : len >= 13 ? 4u : len >= 9 ? 5u : len >= 7 ? 6u : len >= 5 ? 7u : len == 4 ? 8u : len == 3 ? 9u : len == 2 ? 10u : 11u;
				if ( rev ) { L.maskR()[s] = 0; L.woffR()[s] = 0; } else { L.maskF()[s] = 0; L.woffF()[s] = 0; }
				if constexpr ( CT::wide != 0 ) { if ( rev ) L.maskRh()[s] = 0; else L.maskFh()[s] = 0; }
			}
			ulo_r[cc] = lo_; uw_r[cc] = w; ucls_r[cc] = cls; upos_r[cc] = 0;
		}
		{
			uint32_t run = 0;
			for ( uint32_t k = 0; k < NCLS; ++k )
			{
				#pragma unroll
				for ( uint32_t cc = 0; cc < NCHU; ++cc )
				{
					if ( cc*WSZ >= nu ) break;
					bool const mine = ucls_r[cc] == k;
					uint32_t tot; uint32_t const pre = wv_scan_flag(mine,tot);
					if ( mine ) upos_r[cc] = run + pre;
					run += tot;
				}
			}
		}
		#pragma unroll
		for ( uint32_t cc = 0; cc < NCHU; ++cc )
		{
			uint32_t const u = cc*WSZ + lane;
			if ( u < nu ) { urec[upos_r[cc]] = ulo_r[cc] | (u << 7); L.toff()[upos_r[cc]] = static_cast<uint16_t>(uw_r[cc]); }
		}
		wv_sync();
		uint32_t tbase = 0;
		for ( uint32_t c = 0; c < nu; c += WSZ )
		{
			uint32_t const q = c + lane;
			uint32_t const w = q < nu ? L.toff()[q] : 0u;
			uint32_t tot; uint32_t const pre = wv_scan_excl(w,tot);
			if ( q < nu ) L.toff()[q] = tbase + pre;
			tbase += tot;
		}
		PROFX(18)
		if ( tbase > 0xFFFF ) { over(128); return; }
		if ( lane == 0 ) L.toff()[nu] = tbase;
		wv_sync();
		uint32_t const ntask = tbase;
		uint64_t const ltmask = wv_lanemask_lt();
		for ( uint32_t c = 0; c < ntask; c += WSZ )
		{
			uint32_t const t = c + lane;
			bool const act = t < ntask;
			uint32_t u = 0;
			if ( act )
			{
				// unit of task t: last u with toff[u] <= t (empty units share their successor's offset)
				uint32_t lo = 0, hi = nu;
				while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( L.toff()[mid] <= t ) lo = mid; else hi = mid; }
				u = lo;
			}
			uint32_t const ur = act ? urec[u] : 0u;          // u = position in the processing order
			uint32_t const uu = ur >> 7;
			bool const rev = uu >= ns; uint32_t const s = sfrom + (rev ? uu-ns : uu);
			uint32_t const t0 = act ? L.toff()[u] : 0u;
			uint32_t const P = act ? ((ur & 127u) + (t-t0)) : 0u;
			bool ok = act;
			uint64_t sum = 0, f1 = 0, fl = 0;
			if ( act )
			{
				uint32_t const len = L.sslen()[s]; LDSQ uint16_t const * Lk = L.links() + L.slink()[s];
				LDSQ uint8_t const * IP = rev ? L.irpos() : L.ipos();
				#define DACC_LKN(J) static_cast<uint32_t>(Lk[rev ? (len-1-(J)) : (J)])
				// software pipeline over the nodes: a = node j (first instance known), b = node j+1 (index known), c = node j+2
				uint32_t z_b = len > 1 ? DACC_LKN(1) : 0u, z_c = len > 2 ? DACC_LKN(2) : 0u;
				uint32_t i0_a, f_a, ip_a, i0_b, f_b;
				{ uint32_t const z_a = DACC_LKN(0); i0_a = L.nps()[z_a]; f_a = L.nfreq()[z_a]; ip_a = IP[i0_a]; i0_b = L.nps()[z_b]; f_b = L.nfreq()[z_b]; }
				for ( uint32_t j = 0; j < len; ++j )
				{
					uint32_t const z_d = j+3 < len ? DACC_LKN(j+3) : 0u;
					uint32_t const i0_c = L.nps()[z_c], f_c = L.nfreq()[z_c];
					uint32_t const ip_b = IP[i0_b];
					uint32_t const pp = P+j;
					uint32_t const pc = pp < nrows ? pp : nrows;
					uint64_t U = tabR(ip_a*stride + pc);
					// (round 5) the other instances of the node four at a time: their position bytes, then their table words, are in
					// flight together -- one instance per step was two dependent round trips (LDS byte, table word in L1 / L2) each
					for ( uint32_t q = 1; q < f_a; q += 4 )
					{
						uint32_t const m = f_a - q;
						uint32_t const b0 = IP[i0_a+q], b1 = IP[i0_a+q+(m > 1 ? 1u : 0u)], b2 = IP[i0_a+q+(m > 2 ? 2u : 0u)], b3 = IP[i0_a+q+(m > 3 ? 3u : 0u)];
						uint32_t const t0 = tabR(b0*stride + pc), t1 = tabR(b1*stride + pc), t2 = tabR(b2*stride + pc), t3 = tabR(b3*stride + pc);
						U += t0; U += m > 1 ? t1 : 0u; U += m > 2 ? t2 : 0u; U += m > 3 ? t3 : 0u;
					}
					if ( U < FW_THRES_FEAS ) { ok = false; FEAS_ITERS(t,len,j+1) break; }
					sum += U;
					if ( j == 0 ) f1 = U;
					fl = U;
					i0_a = i0_b; f_a = f_b; ip_a = ip_b; i0_b = i0_c; f_b = f_c; z_c = z_d;
				}
				if ( ok ) { FEAS_ITERS(t,len,len) }
				#undef DACC_LKN
			}
			// append the feasible ones: forward tasks precede reverse tasks, the tasks of a stretch are consecutive
			uint64_t const okb = wv_ballot(ok);
			uint64_t const revb = wv_ballot(act && rev);
			uint64_t const mydir = rev ? revb : ~revb;
			uint32_t const base = rev ? nwR : nwF;
			if ( ok )
			{
				uint32_t const o = base + dacc_popc64(okb & mydir & ltmask);
				if ( o < CT::wcap )
				{
					if ( rev ) putR(o,sum,f1); else putF(o,sum,f1,fl);
				}
			}
			if ( act )
			{
				// the first lane of a stretch in this round adds the round's bits to its mask; the lane with the stretch's first
				// task sets the offset of its list
				uint32_t const firstlane = (t0 > c) ? (t0-c) : 0u;
				if ( static_cast<uint32_t>(lane) == firstlane )
				{
					uint32_t const tend = L.toff()[u+1];
					uint32_t const lastlane = (tend-c < WSZ) ? (tend-c) : static_cast<uint32_t>(WSZ);     // exclusive
					uint64_t const span = (lastlane-firstlane >= 64) ? ~0ull : ((1ull << (lastlane-firstlane))-1ull);
					if constexpr ( CT::wide != 0 )
					{
						// start positions up to 127: the round's bits (at most 64, from position P on) may straddle the two words
						uint64_t const seg = (okb >> firstlane) & span;
						uint64_t const blo = P < 64u ? (seg << P) : 0ull, bhi = P < 64u ? (P ? (seg >> (64u-P)) : 0ull) : (seg << (P-64u));
						if ( rev ) { L.maskR()[s] |= blo; L.maskRh()[s] |= bhi; } else { L.maskF()[s] |= blo; L.maskFh()[s] |= bhi; }
					}
					else
					{
					uint64_t const bits = ((okb >> firstlane) & span) << P;
					if ( rev ) L.maskR()[s] |= bits; else L.maskF()[s] |= bits;
					}
					if ( t == t0 ) { uint32_t const wo = base + dacc_popc64(okb & mydir & ltmask); if ( rev ) L.woffR()[s] = wo; else L.woffF()[s] = wo; }
				}
			}
			nwF += dacc_popc64(okb & ~revb); nwR += dacc_popc64(okb & revb);
			if ( nwF > CT::wcap || nwR > CT::wcap ) { over(128); FEAS_DUMP() return; }
		}
		FEAS_DUMP()
		wv_sync();
	}
	// weights of feasible (stretch, position) entry i.  A node weight is a sum of at most 255 table words (< 2^40), a
	// feasible stretch has at most nrows <= 64 nodes (< 2^46)
	// One record per feasible (stretch, position) entry.  Forward: whole stretch (w), its first node at the start position
	// (w1), its last node at the end position (wl); reverse: whole stretch and its last node (= first in reverse direction).
	// gw tiers: 16 byte records in the workgroup's global slab (one load per lookup); legacy tiers: split LDS arrays.
	struct WF { uint64_t w, w1, wl; };
	struct WR { uint64_t w, w1; };
	DEV WF recF(uint32_t const i) const
	{
		WF r;
		if constexpr ( GW )
		{
			G4 const v = reinterpret_cast<G4 const *>(gslab + FastLds<CT>::g_wF)[i];
			r.w = v.x | (static_cast<uint64_t>(v.w & 0xFFFFu)<<32); r.w1 = v.y | (static_cast<uint64_t>((v.w>>16)&0xFFu)<<32); r.wl = v.z | (static_cast<uint64_t>(v.w>>24)<<32);
		}
		else
		{
			r.w = L.wF_lo()[i] | (static_cast<uint64_t>(L.wF_hi()[i])<<32);
			r.w1 = L.wF1_lo()[i] | (static_cast<uint64_t>(L.wF1_hi()[i])<<32);
			r.wl = L.wFl_lo()[i] | (static_cast<uint64_t>(L.wFl_hi()[i])<<32);
		}
		return r;
	}
	DEV WR recR(uint32_t const i) const
	{
		WR r;
		if constexpr ( GW )
		{
			G4 const v = reinterpret_cast<G4 const *>(gslab + FastLds<CT>::g_wR)[i];
			r.w = v.x | (static_cast<uint64_t>(v.z & 0xFFFFu)<<32); r.w1 = v.y | (static_cast<uint64_t>((v.z>>16)&0xFFu)<<32);
		}
		else
		{
			r.w = L.wR_lo()[i] | (static_cast<uint64_t>(L.wR_hi()[i])<<32);
			r.w1 = L.wR1_lo()[i] | (static_cast<uint64_t>(L.wR1_hi()[i])<<32);
		}
		return r;
	}
	DEV void putF(uint32_t const o, uint64_t const sum, uint64_t const f1, uint64_t const fl) const
	{
		if constexpr ( GW )
		{
			G4 v; v.x = static_cast<uint32_t>(sum); v.y = static_cast<uint32_t>(f1); v.z = static_cast<uint32_t>(fl);
			v.w = (static_cast<uint32_t>(sum>>32)&0xFFFFu) | ((static_cast<uint32_t>(f1>>32)&0xFFu)<<16) | ((static_cast<uint32_t>(fl>>32)&0xFFu)<<24);
			reinterpret_cast<G4 *>(gslab + FastLds<CT>::g_wF)[o] = v;
		}
		else
		{
			L.wF_lo()[o] = static_cast<uint32_t>(sum); L.wF_hi()[o] = static_cast<uint16_t>(sum>>32);
			L.wF1_lo()[o] = static_cast<uint32_t>(f1); L.wF1_hi()[o] = static_cast<uint8_t>(f1>>32);
			L.wFl_lo()[o] = static_cast<uint32_t>(fl); L.wFl_hi()[o] = static_cast<uint8_t>(fl>>32);
		}
	}
	DEV void putR(uint32_t const o, uint64_t const rsum, uint64_t const r1) const
	{
		if constexpr ( GW )
		{
			G4 v; v.x = static_cast<uint32_t>(rsum); v.y = static_cast<uint32_t>(r1);
			v.z = (static_cast<uint32_t>(rsum>>32)&0xFFFFu) | ((static_cast<uint32_t>(r1>>32)&0xFFu)<<16); v.w = 0;
			reinterpret_cast<G4 *>(gslab + FastLds<CT>::g_wR)[o] = v;
		}
		else
		{
			L.wR_lo()[o] = static_cast<uint32_t>(rsum); L.wR_hi()[o] = static_cast<uint16_t>(rsum>>32);
			L.wR1_lo()[o] = static_cast<uint32_t>(r1); L.wR1_hi()[o] = static_cast<uint8_t>(r1>>32);
		}
	}
	DEV uint64_t wuF(uint32_t const i) const { return recF(i).w; }
	DEV uint64_t wuR(uint32_t const i) const { return recR(i).w; }
	// fixed-point weight of a single node at (reverse) position p
	DEV uint64_t nodeU(uint32_t const z, uint32_t const p, bool const rev) const
	{
		if ( p >= nrows ) return 0;
		uint32_t const i0 = L.nps()[z], f = L.nfreq()[z];
		LDSQ uint8_t const * IP = rev ? L.irpos() : L.ipos();
		uint64_t u = 0;
		for ( uint32_t q = 0; q < f; ++q ) u += tabR(static_cast<uint32_t>(IP[i0+q])*(nrows+1) + p);
		return u;
	}
	// (wide tier) is position p set in the two-word mask (m0: positions 0-63, m1: 64-127), and how many positions below it are
	DEV static bool mask2Test(uint64_t const m0, uint64_t const m1, uint32_t const p) { return p < 64u ? ((m0 >> p) & 1ull) != 0 : ((m1 >> (p-64u)) & 1ull) != 0; }
	DEV static uint32_t mask2Rank(uint64_t const m0, uint64_t const m1, uint32_t const p)
	{
		return p < 64u ? dacc_popc64(m0 & ((1ull<<p)-1ull)) : dacc_popc64(m0) + dacc_popc64(m1 & ((1ull<<(p-64u))-1ull));
	}
	DEV int32_t sfFind(uint32_t const s, uint32_t const p) const
	{
		if constexpr ( CT::wide != 0 )
		{
			if ( p >= 128 ) return -1;
			uint64_t const m0 = L.maskF()[s], m1 = L.maskFh()[s];
			if ( !mask2Test(m0,m1,p) ) return -1;
			return L.woffF()[s] + mask2Rank(m0,m1,p);
		}
		if ( p >= 64 ) return -1;
		uint64_t const m = L.maskF()[s];
		if ( !((m>>p)&1) ) return -1;
		return L.woffF()[s] + dacc_popc64(m & ((1ull<<p)-1));
	}
	DEV int32_t csfFind(uint32_t const s, uint32_t const p) const
	{
		if constexpr ( CT::wide != 0 )
		{
			if ( p >= 128 ) return -1;
			uint64_t const m0 = L.maskR()[s], m1 = L.maskRh()[s];
			if ( !mask2Test(m0,m1,p) ) return -1;
			return L.woffR()[s] + mask2Rank(m0,m1,p);
		}
		if ( p >= 64 ) return -1;
		uint64_t const m = L.maskR()[s];
		if ( !((m>>p)&1) ) return -1;
		return L.woffR()[s] + dacc_popc64(m & ((1ull<<p)-1));
	}
	// getReverseStretchLinkWeight(A=i,B=b) >= 0.1 (computeStretchLinks :3388-3480), evaluated on demand
	DEV bool linkOk(uint32_t const i, uint32_t const b) const
	{
		uint32_t const shift = L.sslen()[b]-1;
		if constexpr ( CT::wide != 0 )
		{
			if ( shift >= 128 ) return false;
			uint64_t const a0 = L.maskR()[i], a1 = L.maskRh()[i], b0 = L.maskR()[b], b1 = L.maskRh()[b];
			// (b0,b1) << shift over 128 bits
			uint64_t const s0 = shift < 64u ? (b0 << shift) : 0ull;
			uint64_t const s1 = shift < 64u ? ((b1 << shift) | (shift ? (b0 >> (64u-shift)) : 0ull)) : (b0 << (shift-64u));
			uint64_t c0 = a0 & s0, c1 = a1 & s1;
			uint64_t weight = 0;
			while ( c0 | c1 )
			{
				uint32_t pa;
				if ( c0 ) { pa = __builtin_ctzll(c0); c0 &= c0-1; } else { pa = 64u + static_cast<uint32_t>(__builtin_ctzll(c1)); c1 &= c1-1; }
				uint32_t const ia = L.woffR()[i] + mask2Rank(a0,a1,pa);
				uint32_t const pb = pa-shift;
				uint32_t const ib = L.woffR()[b] + mask2Rank(b0,b1,pb);
				WR const ra = recR(ia);
				uint64_t const lweight = wuR(ib) + (ra.w - ra.w1);
				weight = lweight > weight ? lweight : weight;
			}
			return weight >= FW_THRES_01;
		}
		if ( shift >= 64 ) return false;
		uint64_t const mA = L.maskR()[i], mB = L.maskR()[b];
		uint64_t common = mA & (mB<<shift);
		uint64_t weight = 0;
		while ( common )
		{
			uint32_t const pa = __builtin_ctzll(common); common &= common-1;
			uint32_t const ia = L.woffR()[i] + dacc_popc64(mA & ((1ull<<pa)-1));
			uint32_t const pb = pa-shift;
			uint32_t const ib = L.woffR()[b] + dacc_popc64(mB & ((1ull<<pb)-1));
			WR const ra = recR(ia);
			uint64_t const lweight = wuR(ib) + (ra.w - ra.w1);
			weight = lweight > weight ? lweight : weight;
		}
		return weight >= FW_THRES_01;
	}

	// ================= views: the stretch set of a pair in sorted order =================
	// view = base stretches (pool ids 0..n0-1, already in Stretch::operator< order) minus the split parents plus the
	// pieces.  It is never materialised: the enumerations look stretches up by first / last node (fhead = npred region,
	// lhead/lord) and merge the <= 4 inserted pieces by their position ppos among the base stretches.  A view lives in
	// registers (every lane has its own): e[i] = piece | position << 8 | first node << 16 | last node << 32.
	struct View { uint32_t nadd; uint32_t r0, r1; uint64_t e0, e1, e2, e3; };
	struct MIt { uint32_t i, a, target; };
	// number of base stretches whose key is smaller than the key of pool stretch s
	DEV uint32_t basePos(uint32_t const s) const
	{
		uint64_t const key = poolKey(s);
		uint32_t lo = 0, hi = n0;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( poolKey(mid) < key ) lo = mid+1; else hi = mid; }
		return lo;
	}
	DEV static void viewClear(View & V) { V.nadd = 0; V.r0 = 0xFFFF; V.r1 = 0xFFFF; V.e0 = V.e1 = V.e2 = V.e3 = 0; }
	DEV static void viewRemove(View & V, uint32_t const s) { if ( V.r0 == 0xFFFF ) V.r0 = s; else V.r1 = s; }
	// (masks instead of selects: a select chain over four registers is turned into an indexed access of a scratch array)
	DEV static uint64_t viewE(View const & V, uint32_t const i)
	{
		uint64_t const m0 = 0ull - static_cast<uint64_t>(i == 0), m1 = 0ull - static_cast<uint64_t>(i == 1), m2 = 0ull - static_cast<uint64_t>(i == 2), m3 = 0ull - static_cast<uint64_t>(i == 3);
		return (V.e0 & m0) | (V.e1 & m1) | (V.e2 & m2) | (V.e3 & m3);
	}
	DEV static uint32_t vAdd(uint64_t const e) { return static_cast<uint32_t>(e) & SMASK; }
	DEV static uint32_t vPos(uint64_t const e) { return static_cast<uint32_t>(e>>SB) & SMASK; }
	DEV static uint32_t vFn(uint64_t const e) { return static_cast<uint32_t>(e>>(2*SB)) & 0xFFFF; }
	DEV static uint32_t vLn(uint64_t const e) { return static_cast<uint32_t>(e>>(2*SB+16)) & 0xFFFF; }
	DEV void viewAdd(View & V, uint32_t const s) const
	{
		uint32_t const pos = L.ppos()[s];
		uint64_t const key = poolKey(s);
		uint64_t const ne = s | (static_cast<uint64_t>(pos)<<SB) | (static_cast<uint64_t>(L.sfirst()[s])<<(2*SB)) | (static_cast<uint64_t>(L.slast()[s])<<(2*SB+16));
		// keep the insertions sorted by (position, key)
		uint32_t i = V.nadd;
		#define DACC_VSHIFT(Q,PE,DST) if ( i == Q ) { uint64_t const pe = PE; if ( vPos(pe) > pos || (vPos(pe) == pos && poolKey(vAdd(pe)) > key) ) { DST = pe; i = Q-1; } }
		DACC_VSHIFT(3,V.e2,V.e3) DACC_VSHIFT(2,V.e1,V.e2) DACC_VSHIFT(1,V.e0,V.e1)
		#undef DACC_VSHIFT
		{
			uint64_t const m0 = 0ull - static_cast<uint64_t>(i == 0), m1 = 0ull - static_cast<uint64_t>(i == 1), m2 = 0ull - static_cast<uint64_t>(i == 2), m3 = 0ull - static_cast<uint64_t>(i == 3);
			V.e0 = (V.e0 & ~m0) | (ne & m0); V.e1 = (V.e1 & ~m1) | (ne & m1); V.e2 = (V.e2 & ~m2) | (ne & m2); V.e3 = (V.e3 & ~m3) | (ne & m3);
		}
		++V.nadd;
	}
	DEV static bool viewRemoved(View const & V, uint32_t const s) { return s == V.r0 || s == V.r1; }
	// view stretches whose first node is `node`, in view order
	DEV void byFirstBegin(View const &, MIt & it, uint32_t const node) const { it.i = fheadOf(node); it.a = 0; it.target = node; }
	DEV int32_t byFirstNext(View const & V, MIt & it) const
	{
		while ( true )
		{
			while ( it.a < V.nadd && vFn(viewE(V,it.a)) != it.target ) ++it.a;
			bool const haveb = it.i < n0 && L.sfirst()[it.i] == it.target;
			if ( it.a < V.nadd && ( !haveb || vPos(viewE(V,it.a)) <= it.i ) ) return vAdd(viewE(V,it.a++));
			if ( !haveb ) return -1;
			uint32_t const sx = it.i++;
			if ( !viewRemoved(V,sx) ) return sx;
		}
	}
	// view stretches whose last node is `node`, in view order
	DEV void byLastBegin(View const &, MIt & it, uint32_t const node) const { it.i = L.lhead()[node]; it.a = 0; it.target = node; }
	DEV int32_t byLastNext(View const & V, MIt & it) const
	{
		while ( true )
		{
			while ( it.a < V.nadd && vLn(viewE(V,it.a)) != it.target ) ++it.a;
			uint32_t const e = it.i < n0 ? L.lord()[it.i] : 0xFFFFFFFFu;
			bool const haveb = (e>>SB) == it.target;
			uint32_t const sx = e & SMASK;
			if ( it.a < V.nadd && ( !haveb || vPos(viewE(V,it.a)) <= sx ) ) return vAdd(viewE(V,it.a++));
			if ( !haveb ) return -1;
			++it.i;
			if ( !viewRemoved(V,sx) ) return sx;
		}
	}
	DEV uint32_t fheadOf(uint32_t const node) const { return L.npred()[node]; }   // npred is free once the stretches are walked

	// 16 byte records are moved as two 64 bit words (struct copies cannot bind LDS lvalues to generic references)
	template<typename TT> DEV static TT ldget(LDSQ TT const * p)
	{
		static_assert(sizeof(TT) == 16,"record size");
		LDSQ uint64_t const * q = reinterpret_cast<LDSQ uint64_t const *>(p);
		uint64_t u[2]; u[0] = q[0]; u[1] = q[1];
		TT t; __builtin_memcpy(&t,u,16); return t;
	}
	template<typename TT> DEV static void ldput(LDSQ TT * p, TT const & t)
	{
		uint64_t u[2]; __builtin_memcpy(u,&t,16);
		LDSQ uint64_t * q = reinterpret_cast<LDSQ uint64_t *>(p);
		q[0] = u[0]; q[1] = u[1];
	}
	template<typename TT> DEV static void ldswap(LDSQ TT * a, LDSQ TT * b)
	{
		LDSQ uint64_t * x = reinterpret_cast<LDSQ uint64_t *>(a); LDSQ uint64_t * y = reinterpret_cast<LDSQ uint64_t *>(b);
		uint64_t const x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
		x[0] = y0; x[1] = y1; y[0] = x0; y[1] = x1;
	}
	// ---- heaps (same sift algorithm as oracle/o_heap.hpp) ----
	template<bool MINHEAP> DEV static bool hless(uint64_t a, uint64_t b) { return MINHEAP ? (a < b) : (a > b); }
	// (id heaps: a version that keeps the moving id and its weight in registers measured 7 % slower in the forward trees --
	// the re-read of W[H[i]] is in flight together with the children's weights, it costs no round trip)
	template<bool MINHEAP>
	DEV void ipush(LDSQ id_t * H, uint32_t & f, id_t const id, LDSQ uint64_t const * W)
	{
		uint32_t i = f++; H[i] = id;
		while ( i )
		{
			uint32_t const p = (i-1)>>1;
			if ( hless<MINHEAP>(W[H[i]],W[H[p]]) ) { id_t const t = H[i]; H[i] = H[p]; H[p] = t; i = p; }
			else break;
		}
	}
	template<bool MINHEAP>
	DEV void ipop(LDSQ id_t * H, uint32_t & f, LDSQ uint64_t const * W)
	{
		H[0] = H[--f];
		uint32_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			uint32_t const m = hless<MINHEAP>(W[H[r-1]],W[H[r]]) ? (r-1) : r;
			if ( hless<MINHEAP>(W[H[i]],W[H[m]]) ) return;
			id_t const t = H[i]; H[i] = H[m]; H[m] = t; i = m;
		}
		uint32_t const l = 2*i+1;
		if ( l < f && !hless<MINHEAP>(W[H[i]],W[H[l]]) ) { id_t const t = H[i]; H[i] = H[l]; H[l] = t; }
	}
	// The record that moves stays in registers and the records it passes are shifted (same comparisons and the same final
	// arrangement as swapping level by level): one LDS round trip per level.
	template<typename TT, bool MINHEAP>
	DEV void spush(LDSQ TT * H, uint32_t & f, TT const & e)
	{
		uint32_t i = f++;
		while ( i )
		{
			uint32_t const p = (i-1)>>1;
			TT const ep = ldget(H+p);
			if ( hless<MINHEAP>(e.w,ep.w) ) { ldput(H+i,ep); i = p; }
			else break;
		}
		ldput(H+i,e);
	}
	template<typename TT, bool MINHEAP>
	DEV void spop(LDSQ TT * H, uint32_t & f)
	{
		--f; TT const e = ldget(H+f);
		uint32_t i = 0, r;
		while ( (r = 2*i+2) < f )
		{
			TT const a = ldget(H+r-1), b = ldget(H+r);
			bool const pa = hless<MINHEAP>(a.w,b.w);
			uint64_t const wm = pa ? a.w : b.w;
			if ( hless<MINHEAP>(e.w,wm) ) { ldput(H+i,e); return; }
			if ( pa ) { ldput(H+i,a); i = r-1; } else { ldput(H+i,b); i = r; }
		}
		uint32_t const l = 2*i+1;
		if ( l < f )
		{
			TT const el = ldget(H+l);
			if ( !hless<MINHEAP>(e.w,el.w) ) { ldput(H+i,el); i = l; }
		}
		ldput(H+i,e);
	}

	// ================= per-lane path enumerations =================
	// The reverse enumeration of every last k-mer candidate and the forward enumeration of every first k-mer candidate
	// are independent of each other, so they run on one lane each (lane = candidate), all candidates of a window at once;
	// the same routines run on lane 0 alone for the rare pair that needs its exact stretch set.  Paths live in shared
	// pools (rc_* / f_*, indexed by a pool-wide id); a lane takes pool entries in chunks of RCH / FCH through an LDS
	// counter and keeps its chunk list in registers (entry i of an enumeration -> clSlot).
	enum { RCH = CT::rch, RNW = 4, FCH = CT::fch, FNW = CT::fnw, FNC = CT::fnc };         // entries per chunk, 64 bit words of chunk ids (8 per word)
	// The chunk ids of an enumeration are a row of bytes in LDS (a register array indexed at run time would end up in
	// scratch memory): NW*8 chunks per enumeration.
	template<int NW> struct ChunkList { LDSQ uint8_t * ids; uint32_t n; };
	template<int NW> DEV static void clInit(ChunkList<NW> & C, LDSQ uint8_t * row) { C.ids = row; C.n = 0; }
	template<int PCH, int NW> DEV static uint32_t clSlot(ChunkList<NW> const & C, uint32_t const i) { return static_cast<uint32_t>(C.ids[i/PCH])*PCH + (i%PCH); }
	// slot of entry i, taking a new chunk from the pool when i starts one; ~0u: pool or chunk list exhausted
	template<int PCH, int NW> DEV uint32_t clEnsure(ChunkList<NW> & C, uint32_t const i, LDSQ uint32_t * ctr, uint32_t const nchunks)
	{
		uint32_t const ch = i/PCH;
		if ( ch >= C.n )
		{
			if ( ch >= 8*NW ) return ~0u;
			uint32_t const id = wv_atomic_add(ctr,1u);
			if ( id >= nchunks || id > 255 ) return ~0u;
			C.ids[ch] = id;
			C.n = ch+1;
			return id*PCH + (i%PCH);
		}
		return clSlot<PCH>(C,i);
	}

	// ---- reverse enumeration (prepareTraverse :3582-3765) of one last k-mer on view V ----
	struct REnum { ChunkList<RNW> C; uint32_t nrp, narp, lastk; };
	// extendReversePath :4058-4105 with the parent's fields and the feasible entry sfo = csfFind(s,ppos) in registers
	// (checkReversePathFeasiblePosition :4130-4159 looks the same entry up again: the check position is the parent's)
	DEV int32_t extendReversePath(REnum & R, uint32_t const parent, uint32_t const s, uint32_t const ppos, uint32_t const plen, uint64_t const pw, uint32_t const pbl, int32_t const sfo, uint64_t const wr, uint64_t const wr1)
	{
		uint32_t const slot = clEnsure<RCH>(R.C,R.nrp,L.ctr()+0,CT::rccap/RCH);
		if ( slot == ~0u ) { over(512|0x4000); return -1; }
		uint32_t const slen = L.sslen()[s];
		uint64_t weight = pw; uint32_t baselen = pbl;
		if ( plen == 0 ) { baselen = slen+k-1; weight = sfo >= 0 ? wr : 0; }
		else { baselen += slen-1; if ( sfo >= 0 ) weight += wr - wr1; }
		uint32_t const npos = ppos + slen-1;
		if ( baselen > 255 || npos > 255 || plen+1 > 255 ) { over(2048); return -1; }
		++R.nrp;
		L.rc_parent()[slot] = parent; L.rc_stretch()[slot] = s; L.rc_len()[slot] = plen+1; L.rc_pos()[slot] = npos;
		L.rc_w()[slot] = weight; L.rc_baselen()[slot] = baselen;
		return slot;
	}
	DEV uint32_t rpFront(uint32_t const id, uint32_t const lastk) const { return L.rc_len()[id] ? L.nv()[L.sfirst()[L.rc_stretch()[id]]] : lastk; }
	// accepted paths are listed in acceptance order in the enumeration's own slots (rc_acc); rpst = the lane's heap
	DEV void reverseEnumerateLane(REnum & R, LDSQ uint8_t * chunkrow, View const & V, uint32_t const lastkmer, int32_t const lastnode, int64_t const lmax, LDSQ id_t * rpst, uint32_t const rpstcap)
	{
		clInit(R.C,chunkrow); R.nrp = 0; R.narp = 0; R.lastk = lastkmer;
		uint32_t nrpst = 0;
		if ( lastnode >= 0 )
		{
			uint32_t const slot = clEnsure<RCH>(R.C,0,L.ctr()+0,CT::rccap/RCH);
			if ( slot == ~0u ) { over(512|0x4000); return; }
			R.nrp = 1;
			L.rc_parent()[slot] = 0; L.rc_stretch()[slot] = SNONE; L.rc_len()[slot] = 0; L.rc_pos()[slot] = 0; L.rc_w()[slot] = 0; L.rc_baselen()[slot] = k;
			rpst[nrpst++] = slot;
		}
		LDSQ uint64_t const * W = L.rc_w();
		SITE_T0
		while ( nrpst )
		{
			uint32_t const rp = rpst[0];
			ipop<false>(rpst,nrpst,W);
			uint32_t const bl = L.rc_baselen()[rp], ppos = L.rc_pos()[rp], plen = L.rc_len()[rp];
			uint64_t const pw = W[rp];
			SITE(17)      // reverse enumeration: pop of the heaviest pending path + its fields
			// ARPH[bl] (:3626-3665) keeps the 12 heaviest accepted paths of a base length and admits a path only if it is
			// heavier than the lightest of them: a path is dropped iff 12 accepted paths of its length weigh at least as much
			if ( R.narp >= 12 )
			{
				uint32_t cnt = 0;
				for ( uint32_t i = 0; i < R.narp; ++i )
				{
					uint32_t const a = L.rc_acc()[clSlot<RCH>(R.C,i)];
					cnt += (L.rc_baselen()[a] == bl && W[a] >= pw) ? 1 : 0;
				}
				SITE(18)      // reverse enumeration: the 12 heaviest accepted paths of this base length
				if ( cnt >= 12 ) continue;
			}
			L.rc_acc()[clSlot<RCH>(R.C,R.narp)] = rp; ++R.narp;
			if ( plen == 0 )
			{
				MIt it; byLastBegin(V,it,lastnode);
				for ( int32_t sx = byLastNext(V,it); sx >= 0; sx = byLastNext(V,it) )
				{
					int32_t const sfo = csfFind(sx,ppos);
					WR rr; rr.w = 0; rr.w1 = 0; if ( sfo >= 0 ) rr = recR(sfo);
					uint64_t const wr = rr.w;
					if ( !(sfo >= 0 && wr >= FW_THRES_05) ) continue;     // the new path would be dropped right away
					int32_t const rpe = extendReversePath(R,rp,sx,ppos,plen,pw,bl,sfo,wr,rr.w1);
					if ( rpe < 0 ) return;
					if ( nrpst >= rpstcap ) { over(512|0x8000); return; }
					ipush<false>(rpst,nrpst,static_cast<id_t>(rpe),W);
				}
				SITE(19)      // reverse enumeration: the root's extensions
			}
			else if ( static_cast<int64_t>(bl) < (lmax+1)/2 )
			{
				uint32_t const b = L.rc_stretch()[rp];
				uint32_t const bf = L.sfirst()[b];
				MIt it; byLastBegin(V,it,bf);
				for ( int32_t ax = byLastNext(V,it); ax >= 0; ax = byLastNext(V,it) )
				{
					uint32_t const a = ax;
					if ( linkOk(a,b) )
					{
						int32_t const sfo = csfFind(a,ppos);
						WR rr; rr.w = 0; rr.w1 = 0; if ( sfo >= 0 ) rr = recR(sfo);
						uint64_t const wr = rr.w;
						if ( !(sfo >= 0 && wr >= FW_THRES_05) ) continue;
						int32_t const rpe = extendReversePath(R,rp,a,ppos,plen,pw,bl,sfo,wr,rr.w1);
						if ( rpe < 0 ) return;
						if ( nrpst >= rpstcap ) { over(512|0x8000); return; }
						ipush<false>(rpst,nrpst,static_cast<id_t>(rpe),W);
					}
					SITE(20)      // reverse enumeration: one predecessor stretch of a popped path (iterator, link weight, record, push)
				}
			}
			SITE_RESET
		}
	}
	// ---- block of a finished reverse enumeration in sorted order at rc_ord/rc_arw/rc_sbl/rc_front[sbase..sbase+narp) ----
	DEV bool arpLess(uint32_t const a, uint32_t const b, uint32_t const lastk) const
	{
		uint32_t const fa = rpFront(a,lastk), fb = rpFront(b,lastk);
		if ( fa != fb ) return fa < fb;
		return L.rc_baselen()[a] < L.rc_baselen()[b];
	}
	DEV void arpULI(LDSQ id_t * last, uint32_t const lk) { id_t const val = *last; LDSQ id_t * next = last-1; while ( arpLess(val,*next,lk) ) { *last = *next; last = next; --next; } *last = val; }
	DEV void arpIns(LDSQ id_t * first, LDSQ id_t * last, uint32_t const lk)
	{
		if ( first == last ) return;
		for ( LDSQ id_t * i = first+1; i != last; ++i )
		{
			if ( arpLess(*i,*first,lk) ) { id_t const val = *i; for ( LDSQ id_t * q = i; q != first; --q ) *q = *(q-1); *first = val; }
			else arpULI(i,lk);
		}
	}
	// libstdc++ std::sort permutation (introsort + final insertion sort), see dbg_window.hpp arpSort; more than 16
	// elements need the (single) explicit stack L.sstack, so only one lane at a time may sort such a block
	DEV void arpSort(LDSQ id_t * first, LDSQ id_t * last, uint32_t const lk)
	{
		if ( first == last ) return;
		int32_t const n = last-first;
		if ( n > 16 )
		{
			int depth = 0; { int32_t t = n; while ( t > 1 ) { t >>= 1; ++depth; } depth *= 2; }
			LDSQ uint16_t * stF = L.sstack(); LDSQ uint16_t * stL = stF+24; LDSQ uint16_t * stD = stF+48; int sp = 0;   // offsets from first
			stF[0] = 0; stL[0] = n; stD[0] = depth; sp = 1;
			while ( sp )
			{
				--sp;
				LDSQ id_t * f = first+stF[sp]; LDSQ id_t * l = first+stL[sp]; int d = stD[sp];
				while ( l-f > 16 )
				{
					if ( d == 0 ) { over(1024); return; }
					--d;
					LDSQ id_t * mid = f + (l-f)/2; LDSQ id_t * a = f+1; LDSQ id_t * b = mid; LDSQ id_t * c = l-1;
					if ( arpLess(*a,*b,lk) )
					{
						if ( arpLess(*b,*c,lk) ) { id_t t = *f; *f = *b; *b = t; }
						else if ( arpLess(*a,*c,lk) ) { id_t t = *f; *f = *c; *c = t; }
						else { id_t t = *f; *f = *a; *a = t; }
					}
					else if ( arpLess(*a,*c,lk) ) { id_t t = *f; *f = *a; *a = t; }
					else if ( arpLess(*b,*c,lk) ) { id_t t = *f; *f = *c; *c = t; }
					else { id_t t = *f; *f = *b; *b = t; }
					LDSQ id_t * lo = f+1; LDSQ id_t * hi = l;
					while ( true )
					{
						while ( arpLess(*lo,*f,lk) ) ++lo;
						--hi;
						while ( arpLess(*f,*hi,lk) ) --hi;
						if ( !(lo < hi) ) break;
						id_t t = *lo; *lo = *hi; *hi = t;
						++lo;
					}
					if ( sp < 24 ) { stF[sp] = lo-first; stL[sp] = l-first; stD[sp] = d; ++sp; } else { over(1024); return; }
					l = lo;
				}
			}
			arpIns(first,first+16,lk);
			for ( LDSQ id_t * i = first+16; i != last; ++i ) arpULI(i,lk);
		}
		else arpIns(first,last,lk);
	}
	// The same sort on (id, key) pairs: key = front k-mer << 8 | base length, so that key order is arpLess order and a comparison is two
	// independent loads instead of two chains of four (path -> stretch -> first node -> k-mer).  Same comparisons with the same
	// outcomes in the same order, hence the same permutation.  I / K: ids and keys of the block, indexed from its start.
	DEV static void arpSwapK(LDSQ id_t * I, LDSQ uint64_t * K, int32_t const a, int32_t const b) { id_t const t = I[a]; I[a] = I[b]; I[b] = t; uint64_t const u = K[a]; K[a] = K[b]; K[b] = u; }
	DEV static void arpULIK(LDSQ id_t * I, LDSQ uint64_t * K, int32_t last) { id_t const vi = I[last]; uint64_t const vk = K[last]; int32_t next = last-1; while ( vk < K[next] ) { I[last] = I[next]; K[last] = K[next]; last = next; --next; } I[last] = vi; K[last] = vk; }
	DEV static void arpInsK(LDSQ id_t * I, LDSQ uint64_t * K, int32_t const first, int32_t const last)
	{
		if ( first == last ) return;
		for ( int32_t i = first+1; i != last; ++i )
		{
			if ( K[i] < K[first] ) { id_t const vi = I[i]; uint64_t const vk = K[i]; for ( int32_t q = i; q != first; --q ) { I[q] = I[q-1]; K[q] = K[q-1]; } I[first] = vi; K[first] = vk; }
			else arpULIK(I,K,i);
		}
	}
	DEV void arpSortK(LDSQ id_t * I, LDSQ uint64_t * K, int32_t const n)
	{
		if ( n == 0 ) return;
		if ( n > 16 )
		{
			int depth = 0; { int32_t t = n; while ( t > 1 ) { t >>= 1; ++depth; } depth *= 2; }
			LDSQ uint16_t * stF = L.sstack(); LDSQ uint16_t * stL = stF+24; LDSQ uint16_t * stD = stF+48; int sp = 0;
			stF[0] = 0; stL[0] = n; stD[0] = depth; sp = 1;
			while ( sp )
			{
				--sp;
				int32_t f = stF[sp], l = stL[sp]; int d = stD[sp];
				while ( l-f > 16 )
				{
					if ( d == 0 ) { over(1024); return; }
					--d;
					int32_t const mid = f + (l-f)/2, a = f+1, b = mid, c = l-1;
					if ( K[a] < K[b] )
					{
						if ( K[b] < K[c] ) arpSwapK(I,K,f,b);
						else if ( K[a] < K[c] ) arpSwapK(I,K,f,c);
						else arpSwapK(I,K,f,a);
					}
					else if ( K[a] < K[c] ) arpSwapK(I,K,f,a);
					else if ( K[b] < K[c] ) arpSwapK(I,K,f,c);
					else arpSwapK(I,K,f,b);
					int32_t lo = f+1, hi = l;
					while ( true )
					{
						while ( K[lo] < K[f] ) ++lo;
						--hi;
						while ( K[f] < K[hi] ) --hi;
						if ( !(lo < hi) ) break;
						arpSwapK(I,K,lo,hi);
						++lo;
					}
					if ( sp < 24 ) { stF[sp] = lo; stL[sp] = l; stD[sp] = d; ++sp; } else { over(1024); return; }
					l = lo;
				}
			}
			arpInsK(I,K,0,16);
			for ( int32_t i = 16; i != n; ++i ) arpULIK(I,K,i);
		}
		else arpInsK(I,K,0,n);
	}
	// copy the accepted list to its block (acceptance order)
	DEV void reverseBlockCopy(REnum const & R, uint32_t const sbase)
	{
		for ( uint32_t i = 0; i < R.narp; ++i ) L.rc_ord()[sbase+i] = L.rc_acc()[clSlot<RCH>(R.C,i)];
	}
	// sort the block and derive what the score intervals need; slot li of the per-candidate tables gets the summary
	// WTMP: the forward pools are not in use yet (the cached blocks are finished before the first batch of trees): the block's weights
	// are laid out in block order in the forward weight array first, so that the quadratic rank loop reads one value per step, four
	// steps at a time, instead of two dependent loads per step (round 5: site 21 of the ledger, 3 % of a window of config 2)
	template<bool WTMP = false>
	DEV void reverseBlockFinish(REnum const & R, uint32_t const sbase, uint32_t const li, int32_t const lastnode, int64_t const lmax)
	{
		uint32_t const narp = R.narp;
		LDSQ uint64_t const * W = L.rc_w();
		constexpr bool TMP = WTMP && CT::rccap <= CT::fcap;
		LDSQ uint64_t * const WT = L.f_w() + sbase;
		uint64_t rmaxw = 0, rfm = 0;
		for ( uint32_t i = 0; i < narp; ++i ) { uint64_t const w = W[L.rc_ord()[sbase+i]]; rmaxw = w > rmaxw ? w : rmaxw; }
		if constexpr ( TMP )
		{
			// keys of the block in acceptance order, then the sort on (id, key); WT holds the keys until the ranks need it for the weights
			for ( uint32_t i = 0; i < narp; ++i ) { uint32_t const rp = L.rc_ord()[sbase+i]; WT[i] = (static_cast<uint64_t>(rpFront(rp,R.lastk)) << 8) | L.rc_baselen()[rp]; }
			arpSortK(L.rc_ord()+sbase,WT,static_cast<int32_t>(narp));
			for ( uint32_t i = 0; i < narp; ++i ) { uint64_t const key = WT[i]; L.rc_front()[sbase+i] = static_cast<uint32_t>(key >> 8); L.rc_sbl()[sbase+i] = static_cast<uint8_t>(key & 0xFF); }
			for ( uint32_t i = 0; i < narp; ++i ) WT[i] = W[L.rc_ord()[sbase+i]];
		}
		else arpSort(L.rc_ord()+sbase,L.rc_ord()+sbase+narp,R.lastk);
		// rank of every entry by (weight, sorted position); one bit per scan target (node id mod 64) of the enumeration
		uint64_t tm = lastnode >= 0 ? (1ull << (lastnode & 63)) : 0ull;
		for ( uint32_t i = 0; i < narp; ++i )
		{
			uint32_t const rp = L.rc_ord()[sbase+i];
			uint64_t const wi = W[rp];
			uint32_t r = 0;
			if constexpr ( TMP )
			{
				uint32_t j = 0;
				for ( ; j+4 <= narp; j += 4 )
				{
					uint64_t const w0 = WT[j], w1 = WT[j+1], w2 = WT[j+2], w3 = WT[j+3];
					r += (w0 < wi || (w0 == wi && j < i)) ? 1u : 0u; r += (w1 < wi || (w1 == wi && j+1 < i)) ? 1u : 0u;
					r += (w2 < wi || (w2 == wi && j+2 < i)) ? 1u : 0u; r += (w3 < wi || (w3 == wi && j+3 < i)) ? 1u : 0u;
				}
				for ( ; j < narp; ++j ) { uint64_t const wj = WT[j]; if ( wj < wi || (wj == wi && j < i) ) ++r; }
			}
			else
			for ( uint32_t j = 0; j < narp; ++j )
			{
				uint64_t const wj = W[L.rc_ord()[sbase+j]];
				if ( wj < wi || (wj == wi && j < i) ) ++r;
			}
			L.rc_arw()[sbase+i] = r;
			uint32_t fr;
			if constexpr ( TMP ) fr = L.rc_front()[sbase+i];      // (written from the sort keys above)
			else { fr = rpFront(rp,R.lastk); L.rc_front()[sbase+i] = fr; L.rc_sbl()[sbase+i] = L.rc_baselen()[rp]; }
			rfm |= 1ull << (fr & 63);
			if ( L.rc_len()[rp] && static_cast<int64_t>(L.rc_baselen()[rp]) < (lmax+1)/2 ) tm |= 1ull << (L.sfirst()[L.rc_stretch()[rp]] & 63);
		}
		L.rbase()[li] = sbase; L.rn()[li] = narp; L.rmaxw()[li] = rmaxw; L.rtmask()[li] = tm; L.rfmask()[li] = rfm;
	}
	// can the cached reverse block (computed on the view of `last` alone) be used when stretch `par` is split at node `fn`?
	// the scans of the enumeration look for stretches whose last node equals a target; the split changes the answer
	// only for targets par.last (parent vs second piece) and fn (first piece)
	DEV bool reverseUnaffected(uint32_t const sbase, uint32_t const nacc2, int32_t const lastnode, uint32_t const par, uint32_t const fn, int64_t const lmax) const
	{
		uint32_t const plast = L.slast()[par];
		if ( lastnode >= 0 && (static_cast<uint32_t>(lastnode) == plast || static_cast<uint32_t>(lastnode) == fn) ) return false;
		for ( uint32_t i = 0; i < nacc2; ++i )
		{
			uint32_t const rp = L.rc_ord()[sbase+i];
			if ( L.rc_len()[rp] && static_cast<int64_t>(L.rc_baselen()[rp]) < (lmax+1)/2 )
			{
				uint32_t const target = L.sfirst()[L.rc_stretch()[rp]];
				if ( target == plast || target == fn ) return false;
			}
		}
		return true;
	}

	// ---- forward enumeration (path tree of traverse :4838-5041) of one first k-mer on view V ----
	// APQ[baselen] (:4851-4862, 5003-5017) are bounded heaps of 12 that are filled completely before they are drained
	// (an extension is strictly longer than its parent and the buckets are drained in increasing base length), so a
	// bucket's heap is rebuilt right before it is drained by pushing its paths in creation order: one 12 entry heap per
	// enumeration instead of one per base length, same heap arrays, same pop order among equal weights.
	// (m0, m1: base lengths with pending paths, a bit per length; wide tiers: m2 for 128 ... 191 -- a forward path of a window of 100 bases and more
	// ends beyond 127 when its last stretch is long, profiles/r06w: 364 of 120 000 windows at w = 104 went to the generic engine on this)
	template<bool W, int DUMMY = 0> struct FEnumHi { }; template<int DUMMY> struct FEnumHi<true,DUMMY> { uint64_t m2; };
	struct FEnum : FEnumHi<(CT::wide != 0)> { ChunkList<FNW> C; uint32_t np, nfpop; uint64_t fmaxw, ffm, m0, m1; };
	enum : uint32_t { FBLMAX = CT::wide ? 191u : 127u };
	DEV static void fblSet(FEnum & F, uint32_t const nbl)
	{
		// (wide tiers only; the other tiers keep their two-word statements in place: moved into a function, the compiler turns their
		// branch into selects -- a different instruction stream than the measured one.  Unconditional updates here: a three-way choice of
		// the word makes the compiler address F in scratch memory)
		if constexpr ( CT::wide != 0 )
		{
			uint64_t const b = 1ull << (nbl & 63u); uint32_t const wd = nbl >> 6;
			F.m0 |= wd == 0 ? b : 0ull; F.m1 |= wd == 1 ? b : 0ull; F.m2 |= wd == 2 ? b : 0ull;
		}
	}
	DEV static bool fblAny(FEnum const & F) { if constexpr ( CT::wide != 0 ) return (F.m0 | F.m1 | F.m2) != 0; else return (F.m0 | F.m1) != 0; }
	// lowest pending base length, removed from the masks
	DEV static uint32_t fblPop(FEnum & F)
	{
		if constexpr ( CT::wide == 0 ) return 0;
		else
		{
			uint32_t const wd = F.m0 ? 0u : (F.m1 ? 1u : 2u);
			uint64_t const m = wd == 0 ? F.m0 : (wd == 1 ? F.m1 : F.m2);
			uint64_t const c = m & (m-1);
			F.m0 = wd == 0 ? c : F.m0; F.m1 = wd == 1 ? c : F.m1; F.m2 = wd == 2 ? c : F.m2;
			return 64u*wd + static_cast<uint32_t>(__builtin_ctzll(m));
		}
	}
	DEV int32_t extendPath(FEnum & F, uint32_t const parent, uint32_t const s, uint32_t const ppos, uint32_t const plen, uint64_t const pw, uint32_t const pbl,
		int32_t const sfo, uint64_t const wf, uint64_t const wf1, uint32_t & npos, uint32_t & nbl, uint64_t & nw)
	{
		uint32_t const slot = clEnsure<FCH>(F.C,F.np,L.ctr()+1,CT::fcap/FCH);
		if ( slot == ~0u ) { over(512|0x10000); return -1; }
		uint32_t const slen = L.sslen()[s];
		uint64_t weight = pw; uint32_t baselen = pbl;
		if ( plen == 0 ) { baselen = slen+k-1; weight = sfo >= 0 ? wf : 0; }
		else { baselen += slen-1; if ( sfo >= 0 ) weight += wf - wf1; }
		npos = ppos + (slen-1); nbl = baselen; nw = weight;
		if ( baselen > FBLMAX || npos > 255 || plen+1 > 255 ) { over(2048); return -1; }
		++F.np;
		L.f_parent()[slot] = parent; L.f_stretch()[slot] = s; L.f_len()[slot] = plen+1; L.f_pos()[slot] = npos;
		L.f_w()[slot] = weight; L.f_baselen()[slot] = baselen;
		return slot;
	}
	DEV void forwardEnumerateLane(FEnum & F, LDSQ uint8_t * chunkrow, View const & V, int32_t const firstnode, int64_t const lmax, LDSQ id_t * hp)
	{
		clInit(F.C,chunkrow); F.np = 0; F.nfpop = 0; F.fmaxw = 0; F.ffm = 0; F.m0 = 0; F.m1 = 0; if constexpr ( CT::wide != 0 ) F.m2 = 0;
		if ( firstnode < 0 ) return;
		PROFX_T0      // profiling builds: lane 0's tree, split into filling the bucket heap (19) and draining it (20)
		SITE_T0
		{
			MIt it; byFirstBegin(V,it,firstnode);
			for ( int32_t sx = byFirstNext(V,it); sx >= 0; sx = byFirstNext(V,it) )
			{
				int32_t const sfo = sfFind(sx,0);
				uint64_t const wf = sfo >= 0 ? wuF(sfo) : 0;
				uint32_t npos, nbl; uint64_t nw;
				int32_t const id = extendPath(F,0,sx,0,0,0,0,sfo,wf,0,npos,nbl,nw);
				if ( id < 0 ) return;
				if constexpr ( CT::wide == 0 ) { if ( nbl < 64 ) F.m0 |= 1ull << nbl; else F.m1 |= 1ull << (nbl-64); } else fblSet(F,nbl);
			}
		}
		LDSQ uint64_t const * W = L.f_w();
		SITE(11)      // forward tree: the root's extensions
		// (round 6, measured and dropped: starting a bucket's scan behind the leading entries whose base lengths are already drained -- three
		// more registers and a compare per group of four made the window kernels 0.6 % slower, profiles/r06d)
		while ( fblAny(F) )
		{
			uint32_t zz;
			if constexpr ( CT::wide == 0 )
			{
				zz = F.m0 ? static_cast<uint32_t>(__builtin_ctzll(F.m0)) : 64u + static_cast<uint32_t>(__builtin_ctzll(F.m1));
				if ( zz < 64 ) F.m0 &= F.m0-1; else F.m1 &= F.m1-1;
			}
			else zz = fblPop(F);
			uint32_t hn = 0;
			for ( uint32_t i0 = 0; i0 < F.np; i0 += 64 )
			{
				// members of the bucket among entries i0..i0+63 (loads only, so that they overlap), then the pushes in order
				uint64_t mem = 0;
				uint32_t const ie = (F.np-i0 < 64) ? (F.np-i0) : 64u;
				// four consecutive entries lie in one chunk (FCH is a multiple of four): one chunk id and one 32 bit load of
				// their base lengths
				static_assert(FCH % 4 == 0,"entries per chunk");
				for ( uint32_t i = 0; i < ie; i += 4 )
				{
					uint32_t const slot = clSlot<FCH>(F.C,i0+i);
					uint32_t const b4 = *reinterpret_cast<LDSQ uint32_t const *>(L.f_baselen() + slot);
					#pragma unroll
					for ( uint32_t u = 0; u < 4; ++u ) if ( i+u < ie ) mem |= static_cast<uint64_t>(((b4 >> (8*u)) & 0xFF) == zz) << (i+u);
				}
				SITE(12)      // forward tree: members of the bucket among 64 entries (base length loads)
				while ( mem )
				{
					uint32_t const i = __builtin_ctzll(mem); mem &= mem-1;
					uint32_t const e = clSlot<FCH>(F.C,i0+i);
					if ( hn == 12 )
					{
						if ( W[e] > W[hp[0]] ) { ipop<true>(hp,hn,W); ipush<true>(hp,hn,static_cast<id_t>(e),W); }
					}
					else ipush<true>(hp,hn,static_cast<id_t>(e),W);
				}
				SITE(13)      // forward tree: pushes of the bucket's members into the heap of 12
			}
			PROFX(19)
			while ( hn )
			{
				uint32_t const path = hp[0];
				ipop<true>(hp,hn,W);
				uint32_t const ps = L.f_stretch()[path], ppos = L.f_pos()[path], plen = L.f_len()[path], pbl = zz;
				uint64_t const pw = W[path];
				uint32_t const lastn = L.slast()[ps];
				SITE(14)      // forward tree: pop of the heaviest path of the bucket + its fields
				{
					// what the score intervals need of this path: junction k-mer, candidate length, weight without the junction node
					int32_t const psfo = sfFind(ps,ppos - (L.sslen()[ps]-1));
					uint32_t const pfront = L.nv()[lastn];
					uint32_t const o = clSlot<FCH>(F.C,F.nfpop);
					L.fp_id()[o] = path; L.fp_front()[o] = pfront; L.fp_cl()[o] = ppos; F.ffm |= 1ull << (pfront & 63);
					L.fp_adj()[o] = psfo >= 0 ? (pw - recF(psfo).wl) : pw;
					if ( pw > F.fmaxw ) F.fmaxw = pw;
					++F.nfpop;
				}
				SITE(15)      // forward tree: the popped path's record for the score intervals (weight record from the slab)
				if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
				{
					MIt it; byFirstBegin(V,it,lastn);
					for ( int32_t sx = byFirstNext(V,it); sx >= 0; sx = byFirstNext(V,it) )
					{
						uint32_t const s = sx;
						int32_t const sfo = sfFind(s,ppos);
						WF fr; fr.w = 0; fr.w1 = 0; fr.wl = 0; if ( sfo >= 0 ) fr = recF(sfo);
						uint64_t const eweight = fr.w;
						if ( eweight >= FW_THRES_01 )
						{
							uint32_t npos, nbl; uint64_t nw;
							int32_t const ep = extendPath(F,path,s,ppos,plen,pw,pbl,sfo,eweight,fr.w1,npos,nbl,nw);
							if ( ep < 0 ) return;
							if ( nw >= FW_THRES_01 && static_cast<int64_t>(npos) + k <= lmax )
							{
								if constexpr ( CT::wide == 0 ) { if ( nbl < 64 ) F.m0 |= 1ull << nbl; else F.m1 |= 1ull << (nbl-64); } else fblSet(F,nbl);
							}
							else --F.np;
						}
						SITE(16)      // forward tree: one successor stretch of a popped path (view iterator, weight record, new path)
					}
				}
				SITE_RESET
			}
			SITE_RESET      // (lanes that left the drain loop early wait here for the others: not the scan's time)
			PROFX(20)
		}
	}
	// summary of a finished forward tree in slot fi of the per-candidate tables; one bit per scan target
	DEV void forwardTreeFinish(FEnum const & F, uint32_t const fi, int32_t const firstnode, int64_t const lmax)
	{
		uint64_t m = firstnode >= 0 ? (1ull << (firstnode & 63)) : 0ull;
		for ( uint32_t i = 0; i < F.nfpop; ++i )
		{
			uint32_t const path = L.fp_id()[clSlot<FCH>(F.C,i)];
			uint32_t const pbl = L.f_baselen()[path];
			if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) ) m |= 1ull << (L.slast()[L.f_stretch()[path]] & 63);
		}
		L.fnp()[fi] = F.nfpop; L.ffm()[fi] = F.ffm; L.ftm()[fi] = m; L.fmx()[fi] = F.fmaxw;
	}
	DEV void forwardTreeLoad(ChunkList<FNW> & C, uint32_t const fi) const { C.ids = L.fchb() + 8*FNW*fi; C.n = 8*FNW; }
	// scans of the forward enumeration look for stretches whose first node equals a target; splitting `par` at `ln`
	// changes the answer only for targets par.first (parent vs first piece) and ln (second piece)
	DEV bool forwardUnaffected(ChunkList<FNW> const & C, uint32_t const nfpop, int32_t const firstnode, uint32_t const par, uint32_t const ln, int64_t const lmax) const
	{
		uint32_t const pfirst = L.sfirst()[par];
		if ( static_cast<uint32_t>(firstnode) == pfirst || static_cast<uint32_t>(firstnode) == ln ) return false;
		for ( uint32_t i = 0; i < nfpop; ++i )
		{
			uint32_t const path = L.fp_id()[clSlot<FCH>(C,i)];
			uint32_t const pbl = L.f_baselen()[path];
			if ( pbl < k || ( static_cast<int64_t>(pbl-k) < ((lmax+1)/2) ) )
			{
				uint32_t const target = L.slast()[L.f_stretch()[path]];
				if ( target == pfirst || target == ln ) return false;
			}
		}
		return true;
	}

	// ================= combining a forward tree with a reverse block (score intervals) =================
	// capacities of this tier that rounds 1-4 had as globals (tier 0 trades them for nodes, round 5): stretches of a candidate, bytes of a decoded candidate's row
	enum : uint32_t { FSEQCAP = CT::seqcap, CONSROW = CT::consrow };
	static_assert(CONSROW <= FastLds<CT>::consmax && (CONSROW & 7u) == 0,"rows of the decoded candidates: read as 64 bit words");
	uint32_t cfree;   // free candidate sequence slots
	// A candidate is kept as its sequence of view stretches (forward chain, then reverse chain).  Within one view two
	// candidates spell the same string iff their sequences are equal (nodes are distinct k-mers and every edge lies on
	// exactly one stretch of the view), so the duplicate test of traverse (:5098-5110) compares sequences; strings are
	// decoded once, for the final candidates (decodePathPair :4267-4300).
	DEV uint32_t buildSeq(uint32_t const path, uint32_t const rp, LDSQ sid_t * dst, uint32_t & conslen)
	{
		uint32_t const nf = L.f_len()[path], nr = L.rc_len()[rp];
		if ( nf + nr > FSEQCAP ) { over(4096); return ~0u; }
		conslen = static_cast<uint32_t>(L.f_pos()[path]) + k + L.rc_pos()[rp];
		if ( conslen > CONSROW ) { over(4096); return ~0u; }
		// the two parent chains are walked in one loop, so that their dependent loads are in flight together (rc_len drops
		// by one per step and is zero at the root: the reverse chain has nr steps)
		uint32_t qf = path, qr = rp;
		uint32_t const ns = nf > nr ? nf : nr;
		for ( uint32_t t = 0; t < ns; ++t )
		{
			if ( t < nf ) { dst[nf-1-t] = L.f_stretch()[qf]; qf = L.f_parent()[qf]; }
			if ( t < nr ) { dst[nf+t] = L.rc_stretch()[qr]; qr = L.rc_parent()[qr]; }
		}
		return nf+nr;
	}
	DEV uint32_t decodeSeq(LDSQ sid_t const * seq, uint32_t const n, LDSQ uint8_t * dst) const
	{
		uint32_t o = 0;
		uint32_t const firstv = L.nv()[L.sfirst()[seq[0]]];
		for ( uint32_t i = 0; i < k; ++i ) dst[o++] = (firstv >> (2*(k-1-i))) & 3;
		for ( uint32_t ii = 0; ii < n; ++ii )
		{
			uint32_t const s = seq[ii]; LDSQ uint16_t const * Lk = L.links() + L.slink()[s];
			uint32_t const len = L.sslen()[s];
			for ( uint32_t j = 1; j < len; ++j ) dst[o++] = L.nv()[Lk[j]] & 3;
		}
		return o;
	}
	// matching interval [sub,sup) of popped path pi (junction k-mer `front`, candidate length candlen) in the sorted
	// reverse block and its heaviest entry (:4864-4950)
	DEV bool scoreInterval(uint32_t const sbase, uint32_t const nacc2, uint32_t const front, int64_t const candlen, int64_t const lmin, int64_t const lmax,
		uint32_t & sub, uint32_t & sup, uint32_t & mi) const
	{
		uint32_t lo = 0, hi = nacc2;
		while ( lo < hi ) { uint32_t const mid = (lo+hi)>>1; if ( L.rc_front()[sbase+mid] < front ) lo = mid+1; else hi = mid; }
		uint32_t e = lo;
		while ( e < nacc2 && L.rc_front()[sbase+e] == front ) ++e;
		if ( e == lo ) return false;
		int64_t bllo = lmin + static_cast<int64_t>(k) - candlen; if ( bllo < 0 ) bllo = 0;
		int64_t blhi = lmax + static_cast<int64_t>(k) - candlen; if ( blhi < 0 ) blhi = 0;
		uint32_t const bllo16 = static_cast<uint16_t>(bllo), blhi16 = static_cast<uint16_t>(blhi);
		sub = lo;
		while ( sub < e && L.rc_sbl()[sbase+sub] < bllo16 ) ++sub;
		sup = sub;
		while ( sup < e && !(blhi16 < L.rc_sbl()[sbase+sup]) ) ++sup;
		if ( sub == sup ) return false;
		mi = sub; uint32_t mr = L.rc_arw()[sbase+sub];
		for ( uint32_t i = sub+1; i < sup; ++i ) { uint32_t const r = L.rc_arw()[sbase+i]; if ( r > mr ) { mi = i; mr = r; } }
		return true;
	}
	// next lighter entry of a score interval (nextScoreInterval :3513-3534)
	DEV bool scoreNext(uint32_t const sbase, uint32_t const left, uint32_t const right, uint32_t const current, uint32_t & bi) const
	{
		uint32_t const v = L.rc_arw()[sbase+current];
		if ( !v ) return false;
		bool found = false; uint32_t bu = 0;
		for ( uint32_t i = left; i < right; ++i )
		{
			uint32_t const r = L.rc_arw()[sbase+i];
			if ( r <= v-1 && (!found || r > bu) ) { found = true; bu = r; bi = i; }
		}
		return found;
	}
	// one candidate offered to the candidate heap CDH (:5049-5092, including the shrink quirk); false: stop (error)
	//
	// Round 5: a candidate enters the heap as its pair of pool ids (forward path | reverse path << 16, flag UNMAT) and its stretch
	// sequence is written to a slot only if the candidate is still in the heap when its pools are about to be reused (materializeKept,
	// one lane per candidate, at the end of a batch of forward trees; materializeSerial after a pair on its exact stretch set).  The
	// duplicate test (:5077-5088: same string as the previous candidate kept of this pair <=> same stretch sequence) needs the two
	// sequences only when their stretch counts AND consensus lengths agree, both of which follow from the ids with four loads.
	// Rounds 1-4 walked the two parent chains of EVERY offered candidate on lane 0 (33 walks of 2 300 cycles per window of config 2,
	// 5.6 % of its time: site 6 of profiles/r05b_sites_cfg2_256piles_first_ledger.log) and copied the sequence twice.
	enum : uint32_t { UNMAT = 0x80000000u };
	uint32_t pvpath, pvrp, pvcl, pvnf;      // lane 0: the previous candidate kept of the current pair (pool ids, consensus length, stretches of its forward part); its stretch count is `pn`
	// Do (forward entry pa with na stretches, reverse entry ra) and (pb with nb, rb) spell the same stretch sequence, given equal
	// totals and na != nb?  The pool entries of an enumeration are the nodes of a tree (one entry per (parent, stretch)), so two
	// entries of one enumeration are equal sequences iff they are the same entry.  With na < nb = na + d the sequences are equal iff
	// pa is the d-th ancestor of pb, rb is the d-th ancestor of ra, and the d stretches in between agree: d steps along two parent
	// chains (d = distance of the two junctions, mostly 1) instead of both full chains.  Half of the offered candidates of config 2
	// are such duplicates -- the same chain met at a neighbouring junction k-mer.
	DEV bool seqEqual(uint32_t pa, uint32_t ra, uint32_t na, uint32_t pb, uint32_t rb, uint32_t nb)
	{
		if ( na > nb ) { uint32_t t = pa; pa = pb; pb = t; t = ra; ra = rb; rb = t; t = na; na = nb; nb = t; }
		uint32_t const d = nb - na;
		if ( static_cast<uint32_t>(L.rc_len()[ra]) < d ) return false;      // (equal totals: ra has d stretches more than rb)
		LDSQ sid_t * mid = L.cseq() + 16*FSEQCAP;      // the first d stretches of ra in candidate order
		uint32_t qr = ra;
		for ( uint32_t t = 0; t < d; ++t ) { mid[t] = L.rc_stretch()[qr]; qr = L.rc_parent()[qr]; }
		if ( qr != rb ) return false;
		uint32_t qf = pb;
		for ( uint32_t t = 0; t < d; ++t )
		{
			if ( L.f_stretch()[qf] != mid[d-1-t] ) return false;
			qf = L.f_parent()[qf];
		}
		return qf == pa;
	}
	DEV bool offerCandidate(uint64_t const weight, uint32_t const path, uint32_t const rp, uint32_t & pn)
	{
		FSTAT_ADD(18,1);
		SITE_T0
		if ( ncdh == 16 ) { uint32_t const to = L.cdh()[0].o, tl = L.cdh()[0].l; if ( !(tl & UNMAT) ) cfree |= 1u << to; spop<FCC,true>(L.cdh(),ncdh); SITE(5) }   // weight > top here
		uint32_t const nf = L.f_len()[path], nr = L.rc_len()[rp];
		uint32_t const conslen = static_cast<uint32_t>(L.f_pos()[path]) + k + L.rc_pos()[rp];
		if ( nf + nr > FSEQCAP || conslen > CONSROW ) { over(4096); return false; }      // (the checks of buildSeq, in its order)
		uint32_t const n = nf+nr;
		SITE(6)      // offerCandidate: stretch count and consensus length of the candidate
		// Within a pair both candidates come from one forward tree and one reverse enumeration, whose pool entries are the nodes of
		// a tree each: two entries of the same enumeration are different stretch sequences.  So equal strings need equal stretch counts
		// and consensus lengths, a different forward entry AND a different reverse entry, and forward parts of different lengths (the
		// same chain cut at another junction); half of the offers of config 2 pass the first two tests, hardly any all of them.
		if ( n == pn && conslen == pvcl && path != pvpath && rp != pvrp && nf != pvnf )
		{
			bool const same = seqEqual(path,rp,nf,pvpath,pvrp,pvnf);
			SITE(7)      // offerCandidate: comparison with the previous kept candidate (the stretches between the two junctions)
			if ( same ) return true;
		}
		pn = n; pvpath = path; pvrp = rp; pvcl = conslen; pvnf = nf;
		FCC cc; cc.w = weight; cc.o = path | (rp << 16); cc.l = n | (conslen<<8) | UNMAT;
		FSTAT_ADD(19,1);
		spush<FCC,true>(L.cdh(),ncdh,cc);
		SITE(8)      // offerCandidate: push
		return true;
	}
	// all lanes: the kept candidates that are still pool ids get a sequence slot each and their sequences, one lane per candidate
	DEV void materializeKept()
	{
		wv_sync();
		uint32_t const n = wv_bcast(ncdh,0);
		if ( !n ) return;
		uint32_t const cf = wv_bcast(cfree,0);
		uint32_t taken = 0, done = 0;
		for ( uint32_t c0 = 0; c0 < n; c0 += WSZ )
		{
			uint32_t const c = c0 + lane;
			FCC e; e.w = 0; e.o = 0; e.l = 0;
			if ( c < n ) e = ldget(L.cdh()+c);
			bool const um = c < n && (e.l & UNMAT) != 0;
			uint32_t tot; uint32_t const rank = done + wv_scan_flag(um,tot);
			done += tot;
			if ( um )
			{
				uint32_t m = cf; for ( uint32_t i = 0; i < rank; ++i ) m &= m-1;      // the rank-th free slot
				uint32_t const slot = __builtin_ctz(m);
				uint32_t cl;
				buildSeq(e.o & 0xFFFFu,e.o >> 16,L.cseq() + FSEQCAP*slot,cl);
				e.o = slot; e.l &= ~static_cast<uint32_t>(UNMAT); ldput(L.cdh()+c,e);
				taken |= 1u << slot;
			}
		}
		taken = wv_or(taken);
		if ( lane == 0 ) cfree &= ~taken;
		wv_sync();
	}
	// lane 0: the same for the candidates that name pool entries from slot f0 (forward) / r0 (reverse) on -- the enumerations of a
	// pair on its exact stretch set, whose pool space the next such pair reuses
	DEV void materializeSerial(uint32_t const f0, uint32_t const r0)
	{
		for ( uint32_t c = 0; c < ncdh; ++c )
		{
			uint32_t const l = L.cdh()[c].l;
			if ( !(l & UNMAT) ) continue;
			uint32_t const o = L.cdh()[c].o, pa = o & 0xFFFFu, ra = o >> 16;
			if ( pa < f0 && ra < r0 ) continue;
			uint32_t const slot = __builtin_ctz(cfree); cfree &= cfree-1;
			uint32_t cl;
			buildSeq(pa,ra,L.cseq() + FSEQCAP*slot,cl);
			L.cdh()[c].o = slot; L.cdh()[c].l = l & ~static_cast<uint32_t>(UNMAT);
		}
	}
	// serial form (lane 0): score intervals in the shared heap L.siq, candidates offered as they are popped
	// skip: that many pops have been offered already (recorded sequence of the lane form), pn: sequence length of the last
	// candidate kept from them
	DEV void combinePair(ChunkList<FNW> const & FC, uint32_t const nfpop, uint32_t const sbase, uint32_t const nacc2, uint64_t const rfm, int64_t const lmin, int64_t const lmax, uint32_t const maxfullpath,
		uint32_t const skip = 0, uint32_t pn = ~0u)
	{
		nsiq = 0;
		if ( !skip ) { pvpath = pvrp = pvnf = 0; pvcl = ~0u; }      // a pair of its own: no previous candidate
		for ( uint32_t pi = 0; pi < nfpop; ++pi )
		{
			uint32_t const o = clSlot<FCH>(FC,pi);
			uint32_t const front = L.fp_front()[o];
			if ( !((rfm >> (front & 63)) & 1) ) continue;   // no reverse path starts at this k-mer
			uint32_t sub, sup, mi;
			if ( scoreInterval(sbase,nacc2,front,static_cast<int64_t>(L.fp_cl()[o]) + k,lmin,lmax,sub,sup,mi) )
			{
				if ( nsiq >= CT::siqcap ) { over(512|0x40000); return; }
				FSI si; si.left = sub; si.right = sup; si.current = mi; si.path = pi; si.w = L.fp_adj()[o] + L.rc_w()[L.rc_ord()[sbase+mi]];
				spush<FSI,false>(L.siq(),nsiq,si);
			}
		}
		for ( uint32_t numfullpath = 0; nsiq && numfullpath < maxfullpath; ++numfullpath )
		{
			FSI const si = ldget(L.siq());
			// the score intervals leave the heap in non increasing weight order and everything still inside is not
			// heavier, so once the candidate heap is full and its top cannot be beaten, nothing of this pair can enter
			if ( numfullpath >= skip && ncdh == 16 && si.w <= L.cdh()[0].w ) break;
			spop<FSI,false>(L.siq(),nsiq);
			uint32_t bi;
			if ( scoreNext(sbase,si.left,si.right,si.current,bi) )
			{
				FSI sic = si; sic.current = bi; sic.w = L.fp_adj()[clSlot<FCH>(FC,si.path)] + L.rc_w()[L.rc_ord()[sbase+bi]];
				if ( nsiq >= CT::siqcap ) { over(512|0x40000); return; }
				spush<FSI,false>(L.siq(),nsiq,sic);
			}
			if ( numfullpath < skip ) continue;
			if ( !offerCandidate(si.w,L.fp_id()[clSlot<FCH>(FC,si.path)],L.rc_ord()[sbase+si.current],pn) ) return;
		}
	}
	// lane form: the same pops without the candidate heap; the (forward pool slot, reverse pool id) sequence goes to `out`
	// (at most 16 pairs of ids) and is offered to the candidate heap later, in pair order (replayPair).  The lane's heap holds the intervals
	// as (path, current, left, right); weights are recomputed from the tables.  Returns the count or 0xFF if the heap
	// is too small (the pair is then combined serially).
	struct PSI { id_t path, current, left, right; };
	// (round 4: 10 intervals per lane and 8 recorded pops per pair instead of 20 / 16, so that the same lane scratch and the same
	// record area serve 64 pairs per round instead of 32: a window has 96 pairs on average, its lanes are the pairs; 0.24 % of the
	// pairs have more than 10 matching intervals and are combined serially, a pair with more than 8 pops continues serially)
	enum { PSIQ = CT::psiq, POUTE = 8 };
	static constexpr id_t PSENT = static_cast<id_t>(~static_cast<id_t>(0));      // forward pop index no path has (8 bit ids: fewer than 255 pops per tree)
	static_assert(sizeof(id_t) > 1 || 8u*FNW*FCH < 255u,"the end mark of a lane's interval heap must not be a forward pop index");
	bool lscrdirty;      // lane 0: the lane scratch of the current round has been overwritten (an exact pair's enumerations, a serial combine)
	DEV uint64_t psiW(PSI const & e, ChunkList<FNW> const & FC, uint32_t const sbase) const { return L.fp_adj()[clSlot<FCH>(FC,e.path)] + L.rc_w()[L.rc_ord()[sbase+e.current]]; }
	DEV uint32_t combineLane(ChunkList<FNW> const & FC, uint32_t const nfpop, uint32_t const sbase, uint32_t const nacc2, uint64_t const rfm, int64_t const lmin, int64_t const lmax,
		uint32_t const maxfullpath, LDSQ PSI * H, LDSQ id_t * out, bool const prune, uint64_t const T0)
	{
		uint32_t n = 0;
		SITE_T0
		for ( uint32_t pi = 0; pi < nfpop; ++pi )
		{
			uint32_t const o = clSlot<FCH>(FC,pi);
			uint32_t const front = L.fp_front()[o];
			if ( !((rfm >> (front & 63)) & 1) ) continue;
			uint32_t sub, sup, mi;
			if ( scoreInterval(sbase,nacc2,front,static_cast<int64_t>(L.fp_cl()[o]) + k,lmin,lmax,sub,sup,mi) )
			{
				if ( n >= PSIQ || sup > 255 && sizeof(id_t) == 1 ) return 0xFF;
				PSI e; e.path = pi; e.current = mi; e.left = sub; e.right = sup;
				// FiniteSizeHeap push (max heap on the weight)
				uint32_t i = n++; H[i] = e;
				uint64_t const we = psiW(e,FC,sbase);
				while ( i )
				{
					uint32_t const p = (i-1)>>1;
					PSI const ep = H[p];
					if ( we > psiW(ep,FC,sbase) ) { H[i] = ep; H[p] = e; i = p; }
					else break;
				}
			}
		}
		FSTAT_ADD(25,1); FSTAT_ADD(26,n > 8 ? 1u : 0u); FSTAT_ADD(27,n > 10 ? 1u : 0u); FSTAT_ADD(28,n > 12 ? 1u : 0u); FSTAT_ADD(29,n > 16 ? 1u : 0u); FSTAT_MX(30,n);
		SITE(1)      // combineLane: the intervals of the pair (one scoreInterval per forward pop) and their heap
		return combineLanePops(FC,sbase,n,maxfullpath,H,out,prune,T0);
	}
	// the pops of the lane form on a heap of n intervals (second half of combineLane)
	DEV uint32_t combineLanePops(ChunkList<FNW> const & FC, uint32_t const sbase, uint32_t n, uint32_t const maxfullpath, LDSQ PSI * H, LDSQ id_t * out, bool const prune, uint64_t const T0)
	{
		SITE_T0
		uint32_t cnt = 0;
		for ( uint32_t numfullpath = 0; n && numfullpath < maxfullpath; ++numfullpath )
		{
			PSI const top = H[0];
			// the candidate heap was full with lightest weight T0 when the round began: as long as that still holds when
			// the pair is offered, nothing from here on can enter (replayRound checks it and continues serially otherwise)
			// (round 5: a cut pair leaves its heap behind -- the entry behind the last one marked -- and is continued from it, continueLane)
			if ( prune && psiW(top,FC,sbase) <= T0 ) { if ( n < PSIQ ) H[n].path = PSENT; return cnt | 0x20; }
			// the record of a pair holds POUTE pops: the rest of this pair's sequence is produced serially at its place in the pair order
			if ( cnt == POUTE ) { if ( n < PSIQ ) H[n].path = PSENT; return cnt | 0x10; }
			// pop
			{
				--n; PSI const last = H[n]; H[0] = last;
				uint32_t i = 0, r;
				while ( (r = 2*i+2) < n )
				{
					uint32_t const m = psiW(H[r-1],FC,sbase) > psiW(H[r],FC,sbase) ? (r-1) : r;
					PSI const em = H[m], ei = H[i];
					if ( psiW(ei,FC,sbase) > psiW(em,FC,sbase) ) break;
					H[i] = em; H[m] = ei; i = m;
				}
				if ( r >= n )
				{
					uint32_t const l = 2*i+1;
					if ( l < n ) { PSI const el = H[l], ei = H[i]; if ( !(psiW(ei,FC,sbase) > psiW(el,FC,sbase)) ) { H[i] = el; H[l] = ei; } }
				}
			}
			uint32_t bi;
			if ( scoreNext(sbase,top.left,top.right,top.current,bi) )
			{
				PSI e = top; e.current = bi;
				uint32_t i = n++; H[i] = e;
				uint64_t const we = psiW(e,FC,sbase);
				while ( i )
				{
					uint32_t const p = (i-1)>>1;
					PSI const ep = H[p];
					if ( we > psiW(ep,FC,sbase) ) { H[i] = ep; H[p] = e; i = p; }
					else break;
				}
			}
			out[2*cnt] = static_cast<id_t>(clSlot<FCH>(FC,top.path)); out[2*cnt+1] = L.rc_ord()[sbase+top.current]; ++cnt;
			SITE(2)      // combineLane: one pop (sift down, next lighter entry, push, record)
		}
		return cnt;
	}
	// Round 5: the interval construction of a round of pairs as a flat task list.  The lane form above walks a pair's forward pops and runs
	// scoreInterval for every pop whose junction k-mer (mod 64) the reverse block knows -- 64 lanes with different trees, different match
	// patterns and data dependent scan lengths, so the wavefront pays, iteration by iteration, for its slowest lane (site 1 of the ledger:
	// 6 % of a window).  Here a pair only LISTS its matching pops (matchPops: two loads per pop), the (pair, pop) matches of the whole
	// round become tasks dealt 64 at a time to the lanes (intervalTask: every lane runs one scoreInterval), and the pair then pushes its
	// valid intervals in pop order (heapOfList) -- the same pushes in the same order as before, hence the same heap.  A pair with more
	// than PSIQ matches, or a round with more tasks than the task list holds, takes the lane form as before.
	DEV uint32_t matchPops(ChunkList<FNW> const & FC, uint32_t const nfpop, uint64_t const rfm, LDSQ PSI * H) const
	{
		uint32_t nm = 0;
		for ( uint32_t pi = 0; pi < nfpop; ++pi )
		{
			uint32_t const front = L.fp_front()[clSlot<FCH>(FC,pi)];
			if ( !((rfm >> (front & 63)) & 1) ) continue;
			if ( nm < PSIQ ) H[nm].path = static_cast<id_t>(pi);
			++nm;
		}
		return nm;
	}
	// one (pair, matching pop) task: the pop's interval in the pair's reverse block, written over the list entry; false: the interval does
	// not fit an 8 bit id (the pair is combined serially, as in the lane form)
	DEV bool intervalTask(ChunkList<FNW> const & FC, uint32_t const sbase, uint32_t const nacc2, int64_t const lmin, int64_t const lmax, LDSQ PSI * e) const
	{
		uint32_t const pi = e->path;
		uint32_t const o = clSlot<FCH>(FC,pi);
		uint32_t sub, sup, mi;
		if ( scoreInterval(sbase,nacc2,L.fp_front()[o],static_cast<int64_t>(L.fp_cl()[o]) + k,lmin,lmax,sub,sup,mi) )
		{
			if ( sup > 255 && sizeof(id_t) == 1 ) return false;
			PSI v; v.path = static_cast<id_t>(pi); v.current = static_cast<id_t>(mi); v.left = static_cast<id_t>(sub); v.right = static_cast<id_t>(sup);
			*e = v;
		}
		else e->path = PSENT;
		return true;
	}
	// the valid entries of a pair's list, pushed in list (= pop) order: FiniteSizeHeap pushes as in combineLane; returns the heap's size
	DEV uint32_t heapOfList(ChunkList<FNW> const & FC, uint32_t const sbase, LDSQ PSI * H, uint32_t const nm) const
	{
		uint32_t n = 0;
		for ( uint32_t j = 0; j < nm; ++j )
		{
			PSI const e = H[j];
			if ( e.path == PSENT ) continue;
			uint32_t i = n++; H[i] = e;
			uint64_t const we = psiW(e,FC,sbase);
			while ( i )
			{
				uint32_t const p = (i-1)>>1;
				PSI const ep = H[p];
				if ( we > psiW(ep,FC,sbase) ) { H[i] = ep; H[p] = e; i = p; }
				else break;
			}
		}
		return n;
	}
	// offers the recorded sequence of a pair to the candidate heap, exactly as combinePair would have
	// returns true if the sequence was used up (false: ended at an entry that cannot enter the full heap, or error)
	DEV bool replayPair(ChunkList<FNW> const & FC, uint32_t const sbase, LDSQ id_t const * out, uint32_t const cnt, uint32_t & pn)
	{
		pn = ~0u; pvpath = pvrp = pvnf = 0; pvcl = ~0u;      // (dead between pairs: the compiler must not keep them alive across the enumerations)
		FSTAT_ADD(16,1); FSTAT_ADD(17,cnt);
		SITE_T0
		for ( uint32_t e = 0; e < cnt; ++e )
		{
			uint32_t const o = out[2*e];
			uint32_t const rp = out[2*e+1];
			uint64_t const w = L.fp_adj()[o] + L.rc_w()[rp];
			if ( ncdh == 16 && w <= L.cdh()[0].w ) { if ( e == 0 ) FSTAT_ADD(20,1); SITE(4) return false; }
			uint32_t const fid = L.fp_id()[o];
			SITE(4)      // replayPair: recorded entry -> weight, comparison with the lightest kept candidate
			if ( !offerCandidate(w,fid,rp,pn) ) return false;
			SITE_RESET
		}
		return true;
	}

	// lane 0: a pair whose recorded sequence was cut (record full, or pruned at a weight that no longer prunes) goes on from the heap
	// its lane left behind in the lane scratch -- the state combinePair would reach by building every interval again and popping `cnt`
	// times (same heap mechanics: FiniteSizeHeap push / pop on the same weights).  Rounds 1-4 did exactly that: 43 k cycles per
	// continued pair, 0.7 of them per window of config 2 (site 9 of the ledger).
	DEV void continueLane(ChunkList<FNW> const & FC, uint32_t const sbase, LDSQ PSI * H, uint32_t const cnt, uint32_t pn)
	{
		uint32_t n = 0;
		while ( n < PSIQ && H[n].path != PSENT ) ++n;
		for ( uint32_t numfullpath = cnt; n && numfullpath < 16; ++numfullpath )
		{
			PSI const top = H[0];
			uint64_t const wtop = psiW(top,FC,sbase);
			if ( ncdh == 16 && wtop <= L.cdh()[0].w ) break;
			{
				--n; PSI const last = H[n]; H[0] = last;
				uint32_t i = 0, r;
				while ( (r = 2*i+2) < n )
				{
					uint32_t const m = psiW(H[r-1],FC,sbase) > psiW(H[r],FC,sbase) ? (r-1) : r;
					PSI const em = H[m], ei = H[i];
					if ( psiW(ei,FC,sbase) > psiW(em,FC,sbase) ) break;
					H[i] = em; H[m] = ei; i = m;
				}
				if ( r >= n )
				{
					uint32_t const l = 2*i+1;
					if ( l < n ) { PSI const el = H[l], ei = H[i]; if ( !(psiW(ei,FC,sbase) > psiW(el,FC,sbase)) ) { H[i] = el; H[l] = ei; } }
				}
			}
			uint32_t bi;
			if ( scoreNext(sbase,top.left,top.right,top.current,bi) )
			{
				PSI e = top; e.current = bi;
				uint32_t i = n++; H[i] = e;
				uint64_t const we = psiW(e,FC,sbase);
				while ( i )
				{
					uint32_t const p = (i-1)>>1;
					PSI const ep = H[p];
					if ( we > psiW(ep,FC,sbase) ) { H[i] = ep; H[p] = e; i = p; }
					else break;
				}
			}
			if ( !offerCandidate(wtop,L.fp_id()[clSlot<FCH>(FC,top.path)],L.rc_ord()[sbase+top.current],pn) ) return;
		}
	}

	// text: 8 byte aligned, readable up to the next multiple of 8 behind n (a row of consL).  The pattern masks and the
	// text (eight symbols per load) are fetched up front, so the column loop runs from registers.
	DEV uint32_t myersDistance(uint32_t const j, LDSQ uint8_t const * text, uint32_t const n) const
	{
		uint32_t const m = L.slen()[j];
		if ( m == 0 ) return n;
		if ( FastLds<CT>::pw == 2 ) return myersDistance2(j,text,n,m);
		LDSQ uint64_t const * PEQ = L.peq() + 4*j;
		uint64_t const e0 = PEQ[0], e1 = PEQ[1], e2 = PEQ[2], e3 = PEQ[3];
		LDSQ uint64_t const * T8 = reinterpret_cast<LDSQ uint64_t const *>(text);
		uint32_t score = m;
		uint64_t Pv = ~0ull, Mv = 0;
		uint64_t const top = 1ull<<(m-1);
		uint64_t w = n ? T8[0] : 0ull;
		for ( uint32_t c0 = 0; c0 < n; c0 += 8 )
		{
			uint64_t const wn = (c0+8 < n) ? T8[(c0>>3)+1] : 0ull;      // next word, in flight during these eight columns
			uint32_t const cnt = (n-c0 < 8) ? (n-c0) : 8u;
			for ( uint32_t u = 0; u < cnt; ++u )
			{
				uint32_t const ch = static_cast<uint32_t>(w >> (8*u)) & 3u;
				uint64_t const Eq = (ch & 2) ? ((ch & 1) ? e3 : e2) : ((ch & 1) ? e1 : e0);
				uint64_t const Xv = Eq | Mv;
				uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
				uint64_t Ph = Mv | ~(Xh | Pv);
				uint64_t Mh = Pv & Xh;
				if ( Ph & top ) ++score; else if ( Mh & top ) --score;
				Ph = (Ph<<1) | 1ull; Mh <<= 1;
				Pv = Mh | ~(Xv | Ph);
				Mv = Ph & Xv;
			}
			w = wn;
		}
		return score;
	}
	// the same for strings of up to 128 bases (tier 5): two words per pattern mask, block-wise Myers (the horizontal deltas
	// of the top row of word 0 are the carries into word 1), the score follows bit m-1
	DEV uint32_t myersDistance2(uint32_t const j, LDSQ uint8_t const * text, uint32_t const n, uint32_t const m) const
	{
		LDSQ uint64_t const * PEQ = L.peq() + 8*j;
		uint64_t const a0 = PEQ[0], a1 = PEQ[1], c0_ = PEQ[2], c1_ = PEQ[3], g0 = PEQ[4], g1 = PEQ[5], t0 = PEQ[6], t1 = PEQ[7];
		LDSQ uint64_t const * T8 = reinterpret_cast<LDSQ uint64_t const *>(text);
		bool const two = m > 64;
		uint64_t const top = two ? (1ull<<(m-65)) : (1ull<<(m-1));
		uint32_t score = m;
		uint64_t Pv0 = ~0ull, Mv0 = 0, Pv1 = ~0ull, Mv1 = 0;
		uint64_t w = n ? T8[0] : 0ull;
		for ( uint32_t c0 = 0; c0 < n; c0 += 8 )
		{
			uint64_t const wn = (c0+8 < n) ? T8[(c0>>3)+1] : 0ull;
			uint32_t const cnt = (n-c0 < 8) ? (n-c0) : 8u;
			for ( uint32_t u = 0; u < cnt; ++u )
			{
				uint32_t const ch = static_cast<uint32_t>(w >> (8*u)) & 3u;
				uint64_t const Eq0 = (ch & 2) ? ((ch & 1) ? t0 : g0) : ((ch & 1) ? c0_ : a0);
				uint64_t const Eq1 = (ch & 2) ? ((ch & 1) ? t1 : g1) : ((ch & 1) ? c1_ : a1);
				uint64_t const Xv0 = Eq0 | Mv0;
				uint64_t const Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
				uint64_t Ph0 = Mv0 | ~(Xh0 | Pv0);
				uint64_t Mh0 = Pv0 & Xh0;
				uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
				if ( !two ) { if ( Ph0 & top ) ++score; else if ( Mh0 & top ) --score; }
				Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
				Pv0 = Mh0 | ~(Xv0 | Ph0);
				Mv0 = Ph0 & Xv0;
				if ( two )
				{
					uint64_t const Eq1c = Eq1 | mhc;
					uint64_t const Xv1 = Eq1 | Mv1;
					uint64_t const Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
					uint64_t Ph1 = Mv1 | ~(Xh1 | Pv1);
					uint64_t Mh1 = Pv1 & Xh1;
					if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
					Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
					Pv1 = Mh1 | ~(Xv1 | Ph1);
					Mv1 = Ph1 & Xv1;
				}
			}
			w = wn;
		}
		return score;
	}
	DEV void buildPeq()
	{
		if constexpr ( GW ) return;      // the gather built them
		else
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t const m = L.slen()[j];
			LDSQ uint8_t const * s = L.str() + j*CT::lstr;
			enum { PW = FastLds<CT>::pw };
			#pragma unroll
			for ( uint32_t wq = 0; wq < PW; ++wq )
			{
				uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
				uint32_t const lo = 64*wq, hi = m < lo+64 ? m : lo+64;
				for ( uint32_t i = lo; i < hi; ++i )
				{
					uint32_t const c = s[i]; uint64_t const b = 1ull<<(i-lo);
					e0 |= (c == 0) ? b : 0; e1 |= (c == 1) ? b : 0; e2 |= (c == 2) ? b : 0; e3 |= (c == 3) ? b : 0;
				}
				L.peq()[4*PW*j+0*PW+wq] = e0; L.peq()[4*PW*j+1*PW+wq] = e1; L.peq()[4*PW*j+2*PW+wq] = e2; L.peq()[4*PW*j+3*PW+wq] = e3;
			}
		}
		wv_sync();
	}

	// gw layout: the enumeration pools lie over the build-phase arrays (region S); S goes to the workgroup's global slab
	// before the enumerations and comes back after the pairs (candidate errors need the pattern masks, the next
	// activation state the successor tables, the next traversal all of it)
	// Only the part of S the pools cover is copied.  The spilled image stays valid for the following traversals of the same
	// pass (only the successor flags change between them, addNextFromHeap): sfresh = the build phase has rewritten S,
	// sdirty = the successor flags have changed since the last spill.
	bool sfresh, sdirty;
	DEV void spillS()
	{
		if constexpr ( GW )
		{
			typedef FastLds<CT> LL;
			constexpr uint32_t plen = ((LL::upool - LL::sbase + 15u) & ~15u) < LL::sbytes ? ((LL::upool - LL::sbase + 15u) & ~15u) : LL::sbytes;
			constexpr uint32_t si0 = (LL::o_sinfo - LL::sbase) & ~15u, si1 = ((LL::e_sinfo - LL::sbase + 15u) & ~15u) < plen ? ((LL::e_sinfo - LL::sbase + 15u) & ~15u) : plen;
			LDSQ G4 const * const src = reinterpret_cast<LDSQ G4 const *>(L.base + LL::sbase);
			G4 * const dst = reinterpret_cast<G4 *>(gslab + LL::g_spill);
			wv_sync();
			if ( sfresh ) { for ( uint32_t i = lane; i < plen/16u; i += WSZ ) dst[i] = src[i]; }
			else if ( sdirty && si0 < si1 ) { for ( uint32_t i = si0/16u + lane; i < si1/16u; i += WSZ ) dst[i] = src[i]; }
			sfresh = false; sdirty = false;
			wv_sync();
		}
	}
	// sorted k-mer instances + last k-mer list of the current k <-> the workgroup's slab (gw tiers).  A pass that ran a
	// traversal has overwritten overlay A; the next filter frequency pass of the same k needs the very same arrays.
	DEV void saveInstances()
	{
		if constexpr ( GW )
		{
			typedef FastLds<CT> LL;
			G4 * const dst = reinterpret_cast<G4 *>(gslab + LL::g_inst);
			LDSQ G4 const * const sp = reinterpret_cast<LDSQ G4 const *>(L.pre());
			LDSQ G4 const * const sl = reinterpret_cast<LDSQ G4 const *>(L.lastk());
			for ( uint32_t i = lane; i < (npre+1)/2; i += WSZ ) dst[i] = sp[i];
			for ( uint32_t i = lane; i < (nlast+1)/2; i += WSZ ) dst[CT::precap/2 + i] = sl[i];
		}
	}
	// hand-over slots (FastBatch::hand): slot = number + 1, 0 = none.  A window keeps its slot from tier to tier (the instances of a k
	// do not change), so only the first hand-over writes.
	DEV uint32_t saveHand(FastBatch const & FB, uint32_t hslot)
	{
		if ( hslot ) return hslot;
		if ( npre + 2u*((nlast+1)/2) + 4u > FB.handwords || npre > FB.handwords ) return 0;
		uint32_t sl = 0;
		if ( lane == 0 ) { uint32_t const q = wv_atomic_add_global(FB.handctr,1u); sl = q < FB.handcap ? q+1 : 0u; }
		sl = wv_bcast(sl,0);
		if ( !sl ) return 0;
		uint64_t * const dst = FB.hand + static_cast<uint64_t>(sl-1)*FB.handwords;
		if ( lane == 0 ) { dst[0] = npre | (static_cast<uint64_t>(nlast)<<32); dst[1] = k; }
		uint32_t const po = 2, lo = 2 + ((npre+1)&~1u);
		for ( uint32_t i = lane; i < npre; i += WSZ ) dst[po+i] = L.pre()[i];
		for ( uint32_t i = lane; i < nlast; i += WSZ ) dst[lo+i] = L.lastk()[i];
		return sl;
	}
	DEV bool loadHand(FastBatch const & FB, uint32_t const hslot)
	{
		uint64_t const * const src = FB.hand + static_cast<uint64_t>(hslot-1)*FB.handwords;
		uint64_t const h0 = src[0];
		uint32_t const n = static_cast<uint32_t>(h0), nl = static_cast<uint32_t>(h0>>32);
		if ( n > CT::precap || nl > FastLds<CT>::keycap || src[1] != k ) return false;      // (cannot happen between the tiers of one batch)
		npre = n; nlast = nl;
		FSTAT_ADD(31,1);
		uint32_t const po = 2, lo = 2 + ((n+1)&~1u);
		wv_sync();
		for ( uint32_t i = lane; i < n; i += WSZ ) L.pre()[i] = src[po+i];
		for ( uint32_t i = lane; i < nl; i += WSZ ) L.lastk()[i] = src[lo+i];
		wv_sync();
		return true;
	}
	DEV void restoreInstances()
	{
		if constexpr ( GW )
		{
			typedef FastLds<CT> LL;
			G4 const * const src = reinterpret_cast<G4 const *>(gslab + LL::g_inst);
			LDSQ G4 * const dp = reinterpret_cast<LDSQ G4 *>(L.pre());
			LDSQ G4 * const dl = reinterpret_cast<LDSQ G4 *>(L.lastk());
			wv_sync();
			for ( uint32_t i = lane; i < (npre+1)/2; i += WSZ ) dp[i] = src[i];
			for ( uint32_t i = lane; i < (nlast+1)/2; i += WSZ ) dl[i] = src[CT::precap/2 + i];
			wv_sync();
		}
	}
	DEV void restoreS()
	{
		if constexpr ( GW )
		{
			typedef FastLds<CT> LL;
			constexpr uint32_t plen = ((LL::upool - LL::sbase + 15u) & ~15u) < LL::sbytes ? ((LL::upool - LL::sbase + 15u) & ~15u) : LL::sbytes;
			LDSQ G4 * const dst = reinterpret_cast<LDSQ G4 *>(L.base + LL::sbase);
			G4 const * const src = reinterpret_cast<G4 const *>(gslab + LL::g_spill);
			wv_sync();
			for ( uint32_t i = lane; i < plen/16u; i += WSZ ) dst[i] = src[i];
			wv_sync();
		}
	}

	// ================= traverse (:4496-5170) for one activation state =================
	// (first, last) candidate pairs: the reverse blocks of all last k-mers and the forward trees of a batch of first
	// k-mers are enumerated with one lane each, the score intervals of NPL pairs at a time are popped with one lane per
	// pair into recorded (path, entry) sequences, and lane 0 offers those to the candidate heap in the reference's pair
	// order.  A pair whose cached enumerations may be touched by the other k-mer's split (rare) is enumerated on its
	// exact stretch set by lane 0 at its place in that order.
	enum { NPL = 64, RPSTL = 32, PM_SKIP = 0xF0, PM_SERIAL = 0xF1, PM_EXACT = 0xE0, PM_PEND = 0x80 };      // (PM_PEND | matches: between the phases of a round only)
	static_assert(sizeof(PSI)*PSIQ*NPL <= FastLds<CT>::lscrbytes && 2u*POUTE*NPL <= 32u*32u,"score interval heaps and pop records of a round of pairs");
	enum : uint32_t { MIDCAP = 8 };
	uint32_t nmid, midbase;              // middle pieces of this activation state: pool ids midbase .. midbase+nmid-1
	// pair p of a batch -> (first k-mer candidate p / nL, last k-mer candidate p % nL): one multiplication by ceil(2^32 / nL) (exact while
	// p * nL < 2^32) instead of a 32 bit division -- some 25 VALU instructions -- per pair and phase, and per live pair on lane 0
	uint32_t nLmagic;
	DEV void pairFL(uint32_t const p, uint32_t & f, uint32_t & l) const
	{
		if ( nL <= 1 ) { f = p; l = 0; }
		else
		{
#if defined(DACC_EMUL)
			f = static_cast<uint32_t>((static_cast<uint64_t>(p) * nLmagic) >> 32);
#else
			f = __umulhi(p,nLmagic);
#endif
			l = p - f*nL;
		}
	}
	uint32_t rstop;                      // sorted reverse entries used by the cached blocks
	uint64_t roundT0;                    // lightest weight of the (full) candidate heap when the current round of pairs began
	uint32_t nsiq, ncdh, nacc;

	DEV void viewOfLast(View & V, uint32_t const li) const
	{
		viewClear(V);
		uint32_t const sl = L.parL()[li];
		if ( sl != SNONE ) { viewRemove(V,sl); viewAdd(V,L.pieL()[li]); viewAdd(V,L.pieL()[li]+1); }
	}
	DEV void viewOfFirst(View & V, uint32_t const fi) const
	{
		viewClear(V);
		uint32_t const sf = L.parF()[fi];
		if ( sf != SNONE ) { viewRemove(V,sf); viewAdd(V,L.pieF()[fi]); viewAdd(V,L.pieF()[fi]+1); }
	}
	// are the cached enumerations of (fi, li) valid for the pair?  bit 0: reverse block, bit 1: forward tree, bit 2: both
	// candidates split the same stretch
	DEV uint32_t classifyPair(uint32_t const fi, uint32_t const li, int64_t const lmax) const
	{
		uint32_t const sf = L.parF()[fi], sl = L.parL()[li];
		if ( sf != SNONE && sf == sl ) return 4;
		int32_t const firstnode = L.fnode()[fi] == 0xFFFF ? -1 : L.fnode()[fi];
		int32_t const lastnode = L.lnode()[li] == 0xFFFF ? -1 : L.lnode()[li];
		bool rcached, fcached;
		if ( sf == SNONE ) rcached = true;
		else
		{
			uint64_t const tm = L.rtmask()[li];
			uint32_t const fn = L.fnode()[fi];
			rcached = !((tm >> (L.slast()[sf]&63))&1) && !((tm >> (fn&63))&1);
			if ( !rcached ) rcached = reverseUnaffected(L.rbase()[li],L.rn()[li],lastnode,sf,fn,lmax);
		}
		if ( sl == SNONE ) fcached = true;
		else
		{
			uint64_t const tm = L.ftm()[fi];
			fcached = !((tm >> (L.sfirst()[sl]&63))&1) && !((tm >> (L.lnode()[li]&63))&1);
			if ( !fcached ) { ChunkList<FNW> FC; forwardTreeLoad(FC,fi); fcached = forwardUnaffected(FC,L.fnp()[fi],firstnode,sl,L.lnode()[li],lmax); }
		}
		return (rcached ? 1u : 0u) | (fcached ? 2u : 0u);
	}
	// lane 0: pairs [q, n) of the round starting at pair p0 of the batch that starts at candidate fstart, in order.
	// returns 0 when the round is done,
	// 2 when the forward tree of the exact pair q found no pool space next to the trees of the batch (alone: the batch
	// holds the tree of this first k-mer only)
	// live: bit q set for the pairs of the round that are not PM_SKIP (most pairs share no junction k-mer and are skipped:
	// lane 0 walks the set bits instead of reading every pair's mode)
	DEV uint32_t replayRound(uint32_t & q, uint32_t const n, uint32_t const p0, uint32_t const fstart, int64_t const lmin, int64_t const lmax, bool const alone, uint64_t const live)
	{
		for ( ; q < n; ++q )
		{
			uint64_t const rest = live >> q;
			if ( !rest ) { q = n; break; }
			q += static_cast<uint32_t>(__builtin_ctzll(rest));
			SITE_T0
			uint32_t const mode = L.poutn()[q];
			FSTAT_ADD(13,1);
			if ( mode == PM_SKIP ) { FSTAT_ADD(14,1); continue; }
			uint32_t const p = p0+q;
			uint32_t fi, li; pairFL(p,fi,li); fi += fstart;
			if ( mode < 0x40 || mode == PM_SERIAL )
			{
				// no candidate of this pair can beat the lightest kept candidate: score <= path weight + reverse weight
				if ( ncdh == 16 && L.fmx()[fi] + L.rmaxw()[li] <= L.cdh()[0].w ) { FSTAT_ADD(15,1); SITE(3) continue; }
				ChunkList<FNW> FC; forwardTreeLoad(FC,fi);
				SITE(3)      // replayRound: per live pair, up to the dispatch on its mode
				if ( mode < 0x40 )
				{
					uint32_t const cnt = mode & 0x0F; uint32_t pn;
					bool const used = replayPair(FC,L.rbase()[li],L.pout() + 2*POUTE*q,cnt,pn);
					if ( flags ) return 0;
					SITE_RESET
					// a sequence cut at weight T0 (0x20) is complete only while the candidate heap is full with a top of at least T0; one
					// cut because the record was full (0x10) always goes on
					if ( used && ( (mode & 0x10) || ((mode & 0x20) && !(ncdh == 16 && L.cdh()[0].w >= roundT0)) ) )
					{
						pcount(21,1);
						if ( !lscrdirty ) continueLane(FC,L.rbase()[li],reinterpret_cast<LDSQ PSI *>(L.lscr()) + PSIQ*q,cnt,pn);
						else combinePair(FC,L.fnp()[fi],L.rbase()[li],L.rn()[li],L.rfmask()[li],lmin,lmax,16,cnt,pn);
						SITE(9)      // replayRound: continuation of a cut sequence (from the lane's heap; combinePair with skip if the scratch is gone)
					}
				}
				else { pcount(23,1); lscrdirty = true; combinePair(FC,L.fnp()[fi],L.rbase()[li],L.rn()[li],L.rfmask()[li],lmin,lmax,16); SITE(9) }
				if ( flags ) return 0;
				continue;
			}
			// exact stretch set of the pair (split at first, then at last)
			pcount(22,1);
			lscrdirty = true;      // (its enumerations and its serial combine use the lane scratch)
			bool const rcached = mode & 1, fcached = mode & 2, same = mode & 4;
			uint32_t const sf = L.parF()[fi], sl = L.parL()[li];
			int32_t const firstnode = L.fnode()[fi] == 0xFFFF ? -1 : L.fnode()[fi];
			int32_t const lastnode = L.lnode()[li] == 0xFFFF ? -1 : L.lnode()[li];
			View V; viewClear(V);
			if ( same )
			{
				uint32_t const pf = L.posF()[fi], pl = L.posL()[li];
				viewRemove(V,sf);
				if ( pf == pl ) { viewAdd(V,L.pieF()[fi]); viewAdd(V,L.pieF()[fi]+1); }
				else
				{
					// the middle piece of this pair is in the pool (findCandidatesAndPieces)
					uint32_t const lo = pf < pl ? pf : pl, hi = pf < pl ? pl : pf;
					uint32_t m = 0;
					while ( m+1 < nmid && !(L.midpar()[m] == sf && L.midA()[m] == lo && L.midB()[m] == hi) ) ++m;
					if ( pf < pl ) { viewAdd(V,L.pieF()[fi]); viewAdd(V,L.pieL()[li]+1); } else { viewAdd(V,L.pieL()[li]); viewAdd(V,L.pieF()[fi]+1); }
					viewAdd(V,midbase+m);
				}
			}
			else
			{
				if ( sf != SNONE ) { viewRemove(V,sf); viewAdd(V,L.pieF()[fi]); viewAdd(V,L.pieF()[fi]+1); }
				if ( sl != SNONE ) { viewRemove(V,sl); viewAdd(V,L.pieL()[li]); viewAdd(V,L.pieL()[li]+1); }
			}
			uint32_t const rsave = L.ctr()[0], fsave = L.ctr()[1];
			uint32_t sbase = L.rbase()[li], nacc2 = L.rn()[li]; uint64_t rfm = L.rfmask()[li];
			if ( !rcached )
			{
				REnum RX;
				reverseEnumerateLane(RX,L.rchx(),V,L.lkmer()[li],lastnode,lmax,reinterpret_cast<LDSQ id_t *>(L.lscr()),FastLds<CT>::lscrbytes/sizeof(id_t));
				if ( flags ) return 0;
				if ( rstop + RX.narp > CT::rccap ) { over(512|0x4000); return 0; }
				reverseBlockCopy(RX,rstop);
				reverseBlockFinish(RX,rstop,FNC,lastnode,lmax);
				if ( flags ) return 0;
				sbase = rstop; nacc2 = RX.narp; rfm = L.rfmask()[FNC];
			}
			ChunkList<FNW> FC; uint32_t nfp; uint64_t ffmx;
			if ( !fcached )
			{
				FEnum FX;
				forwardEnumerateLane(FX,L.fchb() + 8*FNW*FNC,V,firstnode,lmax,reinterpret_cast<LDSQ id_t *>(L.lscr()));
				if ( flags == (512|0x10000) && FX.np < 8*FNW*FCH && !alone )
				{
					// the trees of the batch leave no room for this one: the batch is cut here and continues with the tree
					// of this first k-mer alone in the pool
					flags = 0; L.ctr()[0] = rsave; L.ctr()[1] = fsave;
					return 2;
				}
				if ( flags ) return 0;
				FC = FX.C; nfp = FX.nfpop; ffmx = FX.ffm;
			}
			else { forwardTreeLoad(FC,fi); nfp = L.fnp()[fi]; ffmx = L.ffm()[fi]; }
			if ( rfm & ffmx ) combinePair(FC,nfp,sbase,nacc2,rfm,lmin,lmax,16);   // else: no junction k-mer in common
			if ( !flags && (!rcached || !fcached) ) materializeSerial(fcached ? ~0u : fsave*FCH,rcached ? ~0u : rsave*RCH);      // candidates that name this pair's own enumerations
			L.ctr()[0] = rsave; L.ctr()[1] = fsave;
			SITE(10)      // replayRound: a pair on its exact stretch set (enumerations + serial combine)
			if ( flags ) return 0;
		}
		return 0;
	}

	DEV bool traverse(int64_t const lmin, int64_t const lmax)
	{
		PROF_T0
		cfree = 0xFFFFu; ncdh = 0; nacc = 0; nwF = 0; nwR = 0; nsiq = 0; pvpath = pvrp = pvnf = 0; pvcl = ~0u;
		LEDGER_REP(5) computeBaseStretches();
		flags = wv_or(flags); if ( flags ) return false;
		PROF(*this,8)
		LEDGER_REP(6) findCandidatesAndPieces();
		flags = wv_or(flags); if ( flags ) return false;
		nLmagic = nL > 1 ? static_cast<uint32_t>(((1ull<<32) + nL - 1u) / nL) : 0u;
#if !defined(DACC_NO_REACH)
		{
			SITE_T0
			bool reach = false;
			LEDGER_REP(7) reach = pairReachable();
			SITE(23)      // traverse: reachability prune
			if ( !reach ) { FSTAT_ADD(23,1); PROF(*this,16) return false; }
		}
#endif
		PROF(*this,16)
		loadTab();
		PROF(*this,17)
		LEDGER_REP(8) { nwF = 0; nwR = 0; computeStretchFeasLanes(0,npool); }
		flags = wv_or(flags); if ( flags ) return false;
		{ SITE_T0 bool const sf0_ = sfresh, sd0_ = sdirty; LEDGER_REP(9) { sfresh = sf0_; sdirty = sd0_; spillS(); } SITE(24) }      // traverse: build-phase arrays to the slab
		PROF(*this,9)

		// ---- reverse blocks of all last k-mer candidates, lane = candidate ----
		LEDGER_REP(10)
		{
		if ( lane == 0 ) { L.ctr()[0] = 0; L.ctr()[1] = 0; }
		wv_sync();
		{
			uint32_t sb = 0;
			for ( uint32_t c = 0; c < nL; c += WSZ )
			{
				uint32_t const li = c + lane;
				bool const ract = li < nL;
				REnum R; clInit(R.C,L.rchb() + 8*RNW*lane); R.nrp = 0; R.narp = 0; R.lastk = 0;
				int32_t lastnode = -1;
				View V; viewClear(V);
				bool rover = false;
				if ( ract )
				{
					viewOfLast(V,li);
					lastnode = L.lnode()[li] == 0xFFFF ? -1 : L.lnode()[li];
					reverseEnumerateLane(R,L.rchb() + 8*RNW*lane,V,L.lkmer()[li],lastnode,lmax,reinterpret_cast<LDSQ id_t *>(L.lscr()) + RPSTL*lane,RPSTL);
					if ( flags == (512|0x8000) ) { rover = true; flags = 0; }
				}
				// an enumeration whose heap of pending paths outgrew the lane's share of the scratch runs again with all of it
				for ( uint64_t ro = wv_ballot(rover); ro; ro &= ro-1 )
				{
					wv_sync();
					if ( lane == static_cast<int>(__builtin_ctzll(ro)) )
						reverseEnumerateLane(R,L.rchb() + 8*RNW*lane,V,L.lkmer()[li],lastnode,lmax,reinterpret_cast<LDSQ id_t *>(L.lscr()),FastLds<CT>::lscrbytes/sizeof(id_t));
				}
				wv_sync();
				flags = wv_or(flags); if ( flags ) return false;
				uint32_t tot; uint32_t const sbase = sb + wv_scan_excl(ract ? R.narp : 0u,tot);
				if ( sb + tot > CT::rccap ) { over(512|0x4000); return false; }
#if defined(DACC_EMUL)
				if ( ract ) { FILE * f = ftrav_file(); if ( f ) fprintf(f,"R %d %u %u 0\n",int(CT::maxs),R.nrp,R.narp); }
#endif
				SITE_T0
				if ( ract ) reverseBlockCopy(R,sbase);
				// blocks of more than 16 entries are sorted with an explicit stack (one lane at a time)
				uint64_t big = wv_ballot(ract && R.narp > 16);
				if ( ract && R.narp <= 16 ) reverseBlockFinish<true>(R,sbase,li,lastnode,lmax);
				while ( big )
				{
					int const b = __builtin_ctzll(big); big &= big-1;
					if ( lane == b ) reverseBlockFinish<true>(R,sbase,li,lastnode,lmax);
				}
				SITE(21)      // traverse: reverse blocks copied, sorted, ranked (lanes = last k-mer candidates)
				sb += tot;
				flags = wv_or(flags); if ( flags ) return false;
			}
			rstop = sb;
		}
		wv_sync();
		}
		PROF(*this,11)

		// ---- batches of forward trees (lane = first k-mer candidate) and their pairs ----
		uint32_t fstart = 0, bw = WSZ, pskip = 0;
		while ( fstart < nF )
		{
			materializeKept();      // the trees of the batch before are about to be overwritten
			uint32_t const fi = fstart + lane;
			bool const fact = static_cast<uint32_t>(lane) < bw && fi < nF;
			FEnum F; int32_t firstnode = -1; uint32_t fover = 0;
			LEDGER_REP(11)
			{
			if ( lane == 0 ) L.ctr()[1] = 0;
			wv_sync();
			clInit(F.C,L.fchb() + 8*FNW*(fi < nF ? fi : static_cast<uint32_t>(FNC))); F.np = 0; F.nfpop = 0; F.fmaxw = 0; F.ffm = 0; F.m0 = 0; F.m1 = 0; if constexpr ( CT::wide != 0 ) F.m2 = 0;
			firstnode = -1;
			fover = 0;
			if ( fact )
			{
				uint32_t const saved = flags; flags = 0;
				View V; viewOfFirst(V,fi);
				firstnode = L.fnode()[fi] == 0xFFFF ? -1 : L.fnode()[fi];
				forwardEnumerateLane(F,L.fchb() + 8*FNW*fi,V,firstnode,lmax,reinterpret_cast<LDSQ id_t *>(L.lscr()) + 12*lane);
				if ( flags == (512|0x10000) && F.np < 8*FNW*FCH ) { fover = 1; flags = saved; }   // no pool space left for this tree
				else flags |= saved;
#if defined(DACC_EMUL)
				{ FILE * f = ftrav_file(); if ( f ) fprintf(f,"T %d %u %u %u\n",int(CT::maxs),F.np,F.nfpop,fover); }
#endif
			}
#if defined(DACC_LEDGER)
			wv_sync();      // (the second run resets the chunk counter: every lane must have left the first)
#endif
			}
			flags = wv_or(flags); if ( flags ) return false;
			uint64_t const fo = wv_ballot(fover);
			uint32_t nb = nF - fstart; if ( nb > bw ) nb = bw;
			if ( fo ) { uint32_t const f0 = __builtin_ctzll(fo); if ( f0 < nb ) nb = f0; }
			if ( nb == 0 )
			{
				// the first tree of the batch lost the race for pool space: run it alone, and give up if even that fails
				if ( bw == 1 ) { over(512|0x10000); return false; }
				bw = 1;
				continue;
			}
			if ( fo )
			{
				// the lanes that lost the race for chunks have left holes in the pool and the chunk counter beyond its end
				if ( nb == 1 && bw != 1 ) { bw = 1; continue; }       // a single tree is kept: run it again with the pool to itself
				uint32_t mx = 0;
				if ( fact && static_cast<uint32_t>(lane) < nb ) for ( uint32_t i = 0; i < F.C.n; ++i ) { uint32_t const id = F.C.ids[i]; mx = id+1 > mx ? id+1 : mx; }
				mx = wv_max(mx);
				if ( lane == 0 ) L.ctr()[1] = mx;      // exact trees of the batch's pairs go behind the kept trees
				wv_sync();
			}
			// a batch that did not fit sets the width of the next ones
			bw = fo ? nb : ((2*bw < WSZ) ? 2*bw : static_cast<uint32_t>(WSZ));
			{ SITE_T0 if ( fact && static_cast<uint32_t>(lane) < nb ) forwardTreeFinish(F,fi,firstnode,lmax); SITE(22) }
			wv_sync();
			PROF(*this,10)
			if ( lane == 0 ) { pcount(26,1); pcount(25,nb*nL); }
			uint32_t const npairs = nb*nL;
			bool restart = false;
			for ( uint32_t p0 = pskip; p0 < npairs && !restart; p0 += NPL )
			{
				uint32_t const nround = (npairs-p0 < NPL) ? (npairs-p0) : static_cast<uint32_t>(NPL);
				bool const cfull = wv_bcast(ncdh == 16 ? 1u : 0u,0) != 0;
				roundT0 = cfull ? L.cdh()[0].w : 0ull;
				uint64_t live = 0;
				static_assert(NPL <= 64,"one bit per pair of a round");
				LEDGER_REP(12)
				{
				live = 0;
				// phase A: classification; a pair that combines cached enumerations lists its matching pops (mode PM_PEND | number of matches)
				LDSQ PSI * const HH = reinterpret_cast<LDSQ PSI *>(L.lscr());
				uint32_t ntask = 0;
				for ( uint32_t t0 = 0; t0 < nround; t0 += WSZ )
				{
					uint32_t const t = t0 + lane;
					uint32_t mode = PM_SKIP;
					if ( t < nround )
					{
						uint32_t const p = p0 + t;
						uint32_t pfi, pli; pairFL(p,pfi,pli); pfi += fstart;
						SITE_T0
						uint32_t const cl = classifyPair(pfi,pli,lmax);
						SITE(26)      // pair generation: classification of a pair (cached enumerations valid?)
						if ( cl != 3 ) mode = PM_EXACT | cl;
						else if ( (L.rfmask()[pli] & L.ffm()[pfi]) == 0 ) mode = PM_SKIP;     // the forward tree and the reverse block share no junction k-mer
						else
						{
							ChunkList<FNW> FC; forwardTreeLoad(FC,pfi);
							uint32_t const nm = matchPops(FC,L.fnp()[pfi],L.rfmask()[pli],HH + PSIQ*t);
							if ( nm == 0 ) mode = PM_SKIP;
							else mode = PM_PEND | (nm <= PSIQ ? nm : 15u);      // 15: more matches than the list holds -- the lane form, in phase B (its records share the task list's bytes)
						}
						L.poutn()[t] = mode;
					}
					uint32_t tot; wv_scan_excl(((mode & 0xF0u) == PM_PEND && (mode & 0x0Fu) != 15u) ? (mode & 0x0Fu) : 0u,tot);
					ntask += tot;
				}
				wv_sync();
				// the task list borrows the pairs' record area (written in phase B): 16 bit entries pair | match << 6
				LDSQ uint16_t * const TL = reinterpret_cast<LDSQ uint16_t *>(L.pout());
				enum : uint32_t { TLCAP = 32u*32u*static_cast<uint32_t>(sizeof(id_t))/2u };
				bool const flat = ntask != 0 && ntask <= TLCAP;
				if ( flat )
				{
					uint32_t tb = 0;
					for ( uint32_t t0 = 0; t0 < nround; t0 += WSZ )
					{
						uint32_t const t = t0 + lane;
						uint32_t const mode = t < nround ? L.poutn()[t] : static_cast<uint32_t>(PM_SKIP);
						uint32_t const nm = ((mode & 0xF0u) == PM_PEND && (mode & 0x0Fu) != 15u) ? (mode & 0x0Fu) : 0u;
						uint32_t tot; uint32_t const pre = tb + wv_scan_excl(nm,tot);
						for ( uint32_t j = 0; j < nm; ++j ) TL[pre+j] = static_cast<uint16_t>(t | (j << 6));
						tb += tot;
					}
					wv_sync();
					// phase T: one scoreInterval per lane and step
					for ( uint32_t q0 = 0; q0 < ntask; q0 += WSZ )
					{
						uint32_t const q = q0 + lane;
						if ( q < ntask )
						{
							uint32_t const en = TL[q], t = en & 63u, j = en >> 6;
							uint32_t const p = p0 + t;
							uint32_t pfi, pli; pairFL(p,pfi,pli); pfi += fstart;
							ChunkList<FNW> FC; forwardTreeLoad(FC,pfi);
							if ( !intervalTask(FC,L.rbase()[pli],L.rn()[pli],lmin,lmax,HH + PSIQ*t + j) ) L.poutn()[t] = PM_SERIAL;
						}
					}
					wv_sync();
				}
				// phase B: heap of the valid intervals and the pops of the lane form (pairs still pending; without the task list: the lane form)
				for ( uint32_t t = lane; t < nround; t += WSZ )
				{
					uint32_t mode = L.poutn()[t];
					if ( (mode & 0xF0u) == PM_PEND )
					{
						uint32_t const p = p0 + t;
						uint32_t pfi, pli; pairFL(p,pfi,pli); pfi += fstart;
						ChunkList<FNW> FC; forwardTreeLoad(FC,pfi);
						uint32_t cnt;
						if ( flat && (mode & 0x0Fu) != 15u )
						{
							uint32_t const n = heapOfList(FC,L.rbase()[pli],HH + PSIQ*t,mode & 0x0Fu);
							cnt = combineLanePops(FC,L.rbase()[pli],n,16,HH + PSIQ*t,L.pout() + 2*POUTE*t,cfull,roundT0);
						}
						else cnt = combineLane(FC,L.fnp()[pfi],L.rbase()[pli],L.rn()[pli],L.rfmask()[pli],lmin,lmax,16,HH + PSIQ*t,L.pout() + 2*POUTE*t,cfull,roundT0);
						mode = cnt == 0xFF ? static_cast<uint32_t>(PM_SERIAL) : (cnt ? cnt : static_cast<uint32_t>(PM_SKIP));
						L.poutn()[t] = mode;
					}
					if ( mode != PM_SKIP ) live |= 1ull << t;
				}
				live = wv_or64(live);
				wv_sync();
				}
				PROF(*this,5)
				if ( lane == 0 ) pcount(27,1);
				uint32_t q = 0;
				while ( true )
				{
					uint32_t req = 0;
					// (round 6, measured and dropped: the replay with every lane in uniform control flow, the candidate heap cached in registers --
					// entry i in lane i, v_readlane in the sift chains -- and a pair's recorded entries fetched by one lane each.  Bit-identical
					// on the 64-lane emulation and on tiers 0 / 1 / 6 of the device, 0.45 % faster there (725.3 -> 722.0 ms at 3000 reads,
					// profiles/r06e): the regular replay is 5 % of a window, not the 13 % the probe-inflated site ledger of round 5 showed.  And
					// WRONG in the deep tiers: a register that holds a different value in every lane and is read with v_readlane does not survive
					// the allocator's spill / reload or live-range copies inside a divergent region (they move the ACTIVE lanes only), and
					// k_window_fast<4|2|3> spill -- test_deep_batch_starts_in_the_deep_tier failed on the device while the emulation passed.)
					if ( lane == 0 ) { lscrdirty = false; req = replayRound(q,nround,p0,fstart,lmin,lmax,nb == 1,live); }
					wv_sync();
					flags = wv_bcast(flags,0); if ( flags ) return false;
					req = wv_bcast(req,0);
					if ( !req ) break;
					if ( req == 2 )
					{
						uint32_t const pq = p0 + wv_bcast(q,0);
						fstart += pq/nL; pskip = pq%nL; bw = 1; restart = true;
						if ( lane == 0 ) pcount(28,1);
						break;
					}
					break;      // (req is 0 or 2: the middle pieces of twice-split stretches are in the pool before the pairs start)
				}
				PROF(*this,7)
			}
			if ( restart ) continue;
			fstart += nb; pskip = 0;
		}
		{ SITE_T0 materializeKept(); SITE(34) }      // the pools give way to the build-phase arrays
		{ SITE_T0 LEDGER_REP(13) restoreS(); SITE(25) }
		PROF(*this,12)
		// CDH -> CH -> ACC (:5099-5136) leaves the kept candidates in descending weight order.  With pairwise distinct
		// weights that order does not depend on the heaps: one lane per candidate counts the heavier ones and stores its
		// candidate at that rank.  Equal weights (rare) take the two heaps, whose mechanics decide their order.
		wv_sync();
		uint32_t const nkept = wv_bcast(ncdh,0);
		{
			uint32_t tie = 0;
			for ( uint32_t c = lane; c < nkept; c += WSZ )
			{
				FCC const e = ldget(L.cdh()+c);
				uint32_t rank = 0;
				for ( uint32_t j = 0; j < nkept; ++j ) { uint64_t const wj = L.cdh()[j].w; rank += (wj > e.w) ? 1u : 0u; tie |= (wj == e.w && j != c) ? 1u : 0u; }
				ldput(L.acc()+rank,e);
			}
			tie = wv_any(tie);
			FSTAT_ADD(21,tie ? 1u : 0u); FSTAT_ADD(22,nkept ? 1u : 0u);
			wv_sync();
			if ( lane == 0 )
			{
				if ( tie )
				{
					uint32_t nch = 0;
					while ( ncdh ) { FCC const c = ldget(L.cdh()); spop<FCC,true>(L.cdh(),ncdh); spush<FCC,false>(L.ch(),nch,c); }
					while ( nch ) { ldput(L.acc()+nacc,ldget(L.ch())); ++nacc; spop<FCC,false>(L.ch(),nch); }
				}
				else { nacc = nkept; ncdh = 0; }
			}
		}
		wv_sync();
		uint32_t const nc = wv_bcast(nacc,0);
		nacc = nc;
		// decode the kept candidates, one lane each
		for ( uint32_t c = lane; c < nc; c += WSZ )
		{
			uint32_t const slot = L.acc()[c].o, n = L.acc()[c].l & 0xFF;
			uint32_t const len = decodeSeq(L.cseq()+FSEQCAP*slot,n,L.consL()+c*CONSROW);
			L.acc()[c].o = c*CONSROW; L.acc()[c].l = len;
		}
		wv_sync();
		LEDGER_REP(14)
		for ( uint32_t t = lane; t < nc*mao; t += WSZ )
		{
			uint32_t const c = t / mao, j = t - c*mao;
			L.canderr()[t] = myersDistance(j,L.consL() + L.acc()[c].o,L.acc()[c].l);
		}
		wv_sync();
		// error sum of candidate c on lane c, then the stable insertion sort by error on lane 0
		for ( uint32_t c = lane; c < nc; c += WSZ )
		{
			uint32_t s = 0;
			for ( uint32_t j = 0; j < mao; ++j ) s += L.canderr()[c*mao+j];
			L.accerr()[c] = s;
		}
		wv_sync();
		// std::sort of at most 16 candidates by error (:5156) = insertion sort = a stable sort: candidate c goes to the
		// number of candidates with a smaller error or the same error and a smaller index (through CH, which is free now;
		// the error sums travel in the bytes of canderr, which have been summed up)
		{
			LDSQ uint32_t * const serr = reinterpret_cast<LDSQ uint32_t *>(L.canderr());
			static_assert(16*CT::maxs >= 64 && (FastLds<CT>::o_canderr & 3) == 0,"sorted error sums borrow the candidate error bytes");
			for ( uint32_t c = lane; c < nc; c += WSZ )
			{
				uint32_t const e = L.accerr()[c];
				uint32_t rank = 0;
				for ( uint32_t j = 0; j < nc; ++j ) { uint32_t const ej = L.accerr()[j]; rank += (ej < e || (ej == e && j < c)) ? 1u : 0u; }
				ldput(L.ch()+rank,ldget(L.acc()+c));
				serr[rank] = e;
			}
			wv_sync();
			for ( uint32_t c = lane; c < nc; c += WSZ ) { ldput(L.acc()+c,ldget(L.ch()+c)); L.accerr()[c] = serr[c]; }
		}
		wv_sync();
		PROF(*this,13)
		FSTAT_MX(0,mao); FSTAT_MX(1,npre); FSTAT_MX(2,nn); FSTAT_MX(3,n0); FSTAT_MX(4,npool); FSTAT_MX(5,nlinks); FSTAT_MX(6,nwF); FSTAT_MX(7,nwR);
		FSTAT_MX(8,rstop); FSTAT_MX(9,nF); FSTAT_MX(10,nL); FSTAT_ADD(11,1); FSTAT_MX(12,nc); FSTAT_ADD(24,nc ? 0u : 1u);
#if defined(DACC_EMUL)
		if ( lane == 0 ) { FILE * f = ftrav_file(); if ( f ) { uint32_t nlv = 0; for ( uint32_t i = 0; i < nL; ++i ) nlv += L.lnode()[i] != 0xFFFF; fprintf(f,"%d %u %u %u %u %u %u %u %u %u %u %u %u\n",int(CT::maxs),mao,npre,nn,n0,npool,nF,nL,nlv,rstop,nwF,nwR,nc); fflush(f); } }
#endif
		return nc != 0;
	}

	DEV int32_t estimateLength()
	{
		int32_t maxvprodindex = -1;
		uint32_t mn = 0xFFFFFFFFu, mx = 0;
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t const len = L.slen()[j];
			uint32_t const lastpos = len ? len-1 : 0;
			mn = lastpos < mn ? lastpos : mn; mx = lastpos > mx ? lastpos : mx;
		}
		mn = ~wv_max(~mn); mx = wv_max(mx);
		uint32_t const supStart = mn < static_cast<uint32_t>(T.nsup) ? T.suplo[mn] : T.nrows;
		uint32_t const supEnd = mx < static_cast<uint32_t>(T.nsup) ? T.suphi[mx] : T.nrows;
		uint64_t bestbits = 0; uint32_t besti = 0xFFFFFFFFu;
		for ( uint32_t c = supStart; c < supEnd; c += WSZ )
		{
			uint32_t const i = c + lane;
			double vprod = 0.0;
			if ( i < supEnd )
			{
				double const * row = T.dpnorm + static_cast<uint64_t>(i)*T.nsup;
				vprod = 1.0;
				// same multiplication order as the reference; the table loads of four strings are issued together
				for ( uint32_t j = 0; j < mao; j += 4 )
				{
					uint32_t const l0 = L.slen()[j], l1 = j+1 < mao ? L.slen()[j+1] : 0, l2 = j+2 < mao ? L.slen()[j+2] : 0, l3 = j+3 < mao ? L.slen()[j+3] : 0;
					uint32_t const ns = static_cast<uint32_t>(T.nsup);
					double const r0 = (l0 && (l0-1) < ns) ? row[l0-1] : 0.0, r1 = (l1 && (l1-1) < ns) ? row[l1-1] : 0.0;
					double const r2 = (l2 && (l2-1) < ns) ? row[l2-1] : 0.0, r3 = (l3 && (l3-1) < ns) ? row[l3-1] : 0.0;
					if ( l0 ) vprod *= r0;
					if ( l1 ) vprod *= r1;
					if ( l2 ) vprod *= r2;
					if ( l3 ) vprod *= r3;
				}
			}
			union { double d; uint64_t u; } cv; cv.d = vprod;
			uint64_t const mb = wv_max64(cv.u);
			if ( mb > bestbits )
			{
				uint64_t const fi = wv_min64(cv.u == mb ? i : 0xFFFFFFFFull);
				bestbits = mb; besti = static_cast<uint32_t>(fi);
			}
		}
		union { double d; uint64_t u; } dm; dm.d = DACC_DBL_MIN;
		if ( bestbits > dm.u ) maxvprodindex = besti;
		if ( maxvprodindex == -1 )
		{
			// density fallback (HandleContext.hpp:2103-2155): histogram of lengths (count-1) against the DPnormSquare rows
			int32_t res = -1;
			if ( lane == 0 )
			{
				uint32_t maxlen = 0;
				for ( uint32_t j = 0; j < mao; ++j ) maxlen = L.slen()[j] > maxlen ? L.slen()[j] : maxlen;
				int32_t maxoff = -1; double maxoffv = DACC_DBL_MIN;
				for ( int32_t i = 0; i < T.nrows; ++i )
				{
					uint32_t const fs = T.dpsq_first[i], sz = T.dpsq_size[i];
					double const * V = T.dpsq + static_cast<uint64_t>(i)*T.nsup;
					double sdot = 0;
					for ( uint32_t t = 0; t < sz; ++t )
					{
						uint32_t const jj = fs+t;
						if ( jj < maxlen+1 )
						{
							uint32_t cnt = 0;
							for ( uint32_t j = 0; j < mao; ++j ) cnt += (L.slen()[j] == jj);
							double const o = cnt ? static_cast<double>(cnt-1) : 0.0;
							sdot += V[jj] * o;
						}
						else break;
					}
					if ( sdot > maxoffv ) { maxoff = i; maxoffv = sdot; }
				}
				if ( maxoff != -1 && maxoffv >= 1e-3 ) res = maxoff;
			}
			maxvprodindex = static_cast<int32_t>(wv_bcast(static_cast<uint32_t>(res),0));
		}
		return maxvprodindex;
	}

	// lane 0: alignment of the consensus to the A window; returns the number of steps of the edit script left in alops.
	// cons is 8 byte aligned and readable up to the next multiple of 8 behind n (bestL).  The pattern masks of the A window
	// and the consensus (2 bits per symbol) stay in registers: the forward pass only stores its columns, the traceback
	// loads a column when it moves to it (A[i-1] == cons[j-1] <=> bit i-1 of the pattern mask of cons[j-1]).
	// the same for a wide window (w in 65 ... 127, wide tier): two words per column (rows 0-63 / 64-m-1; column c at alpv[2c], alpv[2c+1]),
	// the horizontal delta leaving word 0 enters word 1 (as in the two-word distance kernel); same traceback priority
	// (diagonal > DEL > INS), same steps as the generic engine's alignAndEmitWide (dbg_window.hpp)
	DEV uint32_t alignAndEmitWide(LDSQ uint8_t const * cons, uint32_t const n)
	{
		if constexpr ( CT::wide != 0 )
		{
		uint32_t const m = P.w;
		enum { PW = FastLds<CT>::pw };
		static_assert(PW == 2,"the A window of a wide tier has two words per pattern mask");
		LDSQ uint64_t const * PEQ = L.peq();      // string 0 = the A window: word q of symbol c at PW*c + q
		uint64_t const mask1 = (m == 128) ? ~0ull : ((1ull<<(m-64))-1);
		uint64_t Pv0 = ~0ull, Mv0 = 0, Pv1 = mask1, Mv1 = 0; uint32_t score = m;
		L.alpv()[0] = Pv0; L.alpv()[1] = Pv1; L.almv()[0] = Mv0; L.almv()[1] = Mv1; L.albot()[0] = m;
		uint64_t const top = 1ull<<(m-65);
		for ( uint32_t c = 0; c < n; ++c )
		{
			uint32_t const ch = cons[c] & 3u;
			uint64_t const Eq0 = PEQ[PW*ch], Eq1 = PEQ[PW*ch+1];
			uint64_t const Xv0 = Eq0 | Mv0;
			uint64_t const Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
			uint64_t Ph0 = Mv0 | ~(Xh0 | Pv0);
			uint64_t Mh0 = Pv0 & Xh0;
			uint64_t const phc = Ph0>>63, mhc = Mh0>>63;
			Ph0 = (Ph0<<1) | 1ull; Mh0 <<= 1;
			Pv0 = Mh0 | ~(Xv0 | Ph0);
			Mv0 = Ph0 & Xv0;
			uint64_t const Eq1c = Eq1 | mhc;
			uint64_t const Xv1 = Eq1 | Mv1;
			uint64_t const Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
			uint64_t Ph1 = Mv1 | ~(Xh1 | Pv1);
			uint64_t Mh1 = Pv1 & Xh1;
			if ( Ph1 & top ) ++score; else if ( Mh1 & top ) --score;
			Ph1 = (Ph1<<1) | phc; Mh1 = (Mh1<<1) | mhc;
			Pv1 = (Mh1 | ~(Xv1 | Ph1)) & mask1;
			Mv1 = (Ph1 & Xv1) & mask1;
			L.alpv()[2*c+2] = Pv0; L.alpv()[2*c+3] = Pv1; L.almv()[2*c+2] = Mv0; L.almv()[2*c+3] = Mv1; L.albot()[c+1] = score;
		}
		uint32_t i = m, j = n; uint32_t d = score; uint32_t nops = 0;
		while ( i || j )
		{
			uint32_t op = 2; bool done = false;
			if ( i && j )
			{
				uint32_t const sh = i-1;
				uint64_t const p0 = L.alpv()[2*(j-1)], p1 = L.alpv()[2*(j-1)+1], q0 = L.almv()[2*(j-1)], q1 = L.almv()[2*(j-1)+1];
				uint32_t const np_ = sh < 64 ? (dacc_popc64(p0>>sh) + dacc_popc64(p1)) : dacc_popc64(p1>>(sh-64));
				uint32_t const nm_ = sh < 64 ? (dacc_popc64(q0>>sh) + dacc_popc64(q1)) : dacc_popc64(q1>>(sh-64));
				uint32_t const dd = L.albot()[j-1] - np_ + nm_;
				uint32_t const cj = cons[j-1] & 3u;
				uint32_t const neq = ((PEQ[PW*cj + (sh>>6)] >> (sh&63u)) & 1ull) ? 0u : 1u;      // A[i-1] != cons[j-1]
				if ( dd + neq == d ) { op = neq ? 1 : 0; --i; --j; d = dd; done = true; }
			}
			if ( !done && i )
			{
				uint32_t const r = i-1;
				uint64_t const pv = L.alpv()[2*j+(r>>6)];
				if ( (pv >> (r&63u)) & 1ull ) { op = 3; --i; d = d-1; done = true; }
			}
			if ( !done ) { op = 2; --j; d = d-1; }
			L.alops()[nops++] = op;
		}
		return nops;
		}
		else return 0;
	}
	DEV uint32_t alignAndEmit(LDSQ uint8_t const * cons, uint32_t const n)
	{
		if constexpr ( CT::wide != 0 ) { if ( DACC_WIDE_W(P.w) ) return alignAndEmitWide(cons,n); }
		uint32_t const m = P.w;
		LDSQ uint64_t const * PEQ = L.peq();
		enum { PW = FastLds<CT>::pw };      // the A window has at most 63 bases: word 0 of every symbol
		uint64_t const e0 = PEQ[0*PW], e1 = PEQ[1*PW], e2 = PEQ[2*PW], e3 = PEQ[3*PW];
		LDSQ uint64_t const * T8 = reinterpret_cast<LDSQ uint64_t const *>(cons);
		static_assert(MAXCONS <= 96 && (FastLds<CT>::o_bestL & 7) == 0,"consensus packed into three 64 bit words");
		uint64_t const mask = (m == 64) ? ~0ull : ((1ull<<m)-1);
		uint64_t Pv = mask, Mv = 0; uint32_t score = m;
		L.alpv()[0] = Pv; L.almv()[0] = Mv; L.albot()[0] = m;
		uint64_t const top = 1ull<<(m-1);
		uint64_t ck0 = 0, ck1 = 0, ck2 = 0;      // consensus symbols 0-31, 32-63, 64-95
		uint64_t w = n ? T8[0] : 0ull;
		for ( uint32_t c0 = 0; c0 < n; c0 += 8 )
		{
			uint64_t const wn = (c0+8 < n) ? T8[(c0>>3)+1] : 0ull;
			uint32_t const cnt = (n-c0 < 8) ? (n-c0) : 8u;
			for ( uint32_t u = 0; u < cnt; ++u )
			{
				uint32_t const c = c0+u;
				uint64_t const ch = (w >> (8*u)) & 3u;
				uint64_t const cs = ch << (2*(c&31));
				ck0 |= c < 32 ? cs : 0ull; ck1 |= (c >= 32 && c < 64) ? cs : 0ull; ck2 |= c >= 64 ? cs : 0ull;
				uint64_t const Eq = (ch & 2) ? ((ch & 1) ? e3 : e2) : ((ch & 1) ? e1 : e0);
				uint64_t const Xv = Eq | Mv;
				uint64_t const Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
				uint64_t Ph = Mv | ~(Xh | Pv);
				uint64_t Mh = Pv & Xh;
				if ( Ph & top ) ++score; else if ( Mh & top ) --score;
				Ph = (Ph<<1) | 1ull; Mh <<= 1;
				Pv = (Mh | ~(Xv | Ph)) & mask;
				Mv = (Ph & Xv) & mask;
				L.alpv()[c+1] = Pv; L.almv()[c+1] = Mv; L.albot()[c+1] = score;
			}
			w = wn;
		}
		uint32_t i = m, j = n; uint32_t d = score; uint32_t nops = 0;
		// column j (its Pv) and column j-1 (Pv, Mv, bottom score) in registers
		uint64_t pvj = Pv, pv1 = 0, mv1 = 0; uint32_t bot1 = 0;
		if ( j ) { pv1 = L.alpv()[j-1]; mv1 = L.almv()[j-1]; bot1 = L.albot()[j-1]; }
		while ( i || j )
		{
			uint32_t op = 2; bool done = false; bool left = false;
			if ( i && j )
			{
				uint64_t const sh = i-1;
				uint32_t const dd = bot1 - dacc_popc64(pv1>>sh) + dacc_popc64(mv1>>sh);
				uint32_t const jj = j-1;
				uint64_t const ckw = jj < 32 ? ck0 : (jj < 64 ? ck1 : ck2);
				uint32_t const ch = static_cast<uint32_t>(ckw >> (2*(jj&31))) & 3u;
				uint64_t const Eq = (ch & 2) ? ((ch & 1) ? e3 : e2) : ((ch & 1) ? e1 : e0);
				uint32_t const neq = ((Eq >> sh) & 1) ? 0u : 1u;
				if ( dd + neq == d ) { op = neq ? 1 : 0; --i; --j; d = dd; done = true; left = true; }
			}
			if ( !done && i )
			{
				uint64_t const bit = 1ull<<(i-1);
				if ( pvj & bit ) { op = 3; --i; d = d-1; done = true; }
			}
			if ( !done ) { op = 2; --j; d = d-1; left = true; }
			if ( left )
			{
				pvj = pv1;
				if ( j ) { pv1 = L.alpv()[j-1]; mv1 = L.almv()[j-1]; bot1 = L.albot()[j-1]; }
			}
			L.alops()[nops++] = op;
		}
		return nops;
	}
	// window record from the edit script alops[0..nops) (traceback order: last step first), all lanes.  Every step emits
	// exactly one symbol (INS / MATCH / MISMATCH: the next consensus symbol, DEL: 'D' = 4), so symbol q belongs to step q
	// in forward order; off[r] = number of symbols before the group of A column r = index behind the (r-1)-th step that
	// consumes an A symbol (HandleContext.hpp:2448-2489)
	DEV void emitRecord(LDSQ uint8_t const * cons, uint32_t const nops, uint8_t * rec)
	{
		uint32_t const m = P.w;
		if constexpr ( CT::wide != 0 )
		{
			if ( DACC_WIDE_W(m) )
			{
				// wide record (dev_types.hpp): 16 bit group offsets from rec[2] on, symbols behind the m+2 offsets
				uint8_t * const off = rec+2; uint8_t * const sym = rec + 2 + 2*(m+2);
				if ( lane == 0 ) { rec[0] = 1; rec[1] = 0; off[0] = 0; off[1] = 0; off[2*(m+1)] = nops & 0xFFu; off[2*(m+1)+1] = nops >> 8; }
				uint32_t cbase = 0, abase = 0;
				for ( uint32_t c0 = 0; c0 < nops; c0 += WSZ )
				{
					uint32_t const q = c0 + lane;
					bool const act = q < nops;
					uint32_t const op = act ? L.alops()[nops-1-q] : 2u;
					uint32_t ctot, atot;
					uint32_t const cpos = cbase + wv_scan_flag(act && op != 3,ctot);
					uint32_t const arank = abase + wv_scan_flag(act && op != 2,atot);
					if ( act )
					{
						sym[q] = (op == 3) ? 4 : cons[cpos];
						if ( op != 2 ) { off[2*(arank+1)] = (q+1) & 0xFFu; off[2*(arank+1)+1] = (q+1) >> 8; }
					}
					cbase += ctot; abase += atot;
				}
				return;
			}
		}
		uint8_t * off = rec+1; uint8_t * sym = rec + 1 + (m+2);
		if ( lane == 0 ) { rec[0] = 1; off[0] = 0; off[m+1] = nops; }
		uint32_t cbase = 0, abase = 0;
		for ( uint32_t c0 = 0; c0 < nops; c0 += WSZ )
		{
			uint32_t const q = c0 + lane;
			bool const act = q < nops;
			uint32_t const op = act ? L.alops()[nops-1-q] : 2u;
			uint32_t ctot, atot;
			uint32_t const cpos = cbase + wv_scan_flag(act && op != 3,ctot);
			uint32_t const arank = abase + wv_scan_flag(act && op != 2,atot);
			if ( act )
			{
				sym[q] = (op == 3) ? 4 : cons[cpos];
				if ( op != 2 ) off[arank+1] = q+1;
			}
			cbase += ctot; abase += atot;
		}
	}
};

// returns FW_DONE, FW_NEXT (does not fit this tier's capacities) or FW_GENERIC (shape no LDS tier supports)
enum { FW_DONE = 0, FW_NEXT = 1, FW_GENERIC = 2 };
// Size class of a window (pre-pass of shallow batches): 0 = small (starts in tier 0), 1 = the rest (starts in tier 1).  An
// upper bound of the number of k-mer instances at the smallest k from the window tables: every active overlap counts (the
// window may keep fewer, `maxalign`), so a window classed small can still overflow tier 0 and is then handed on like any other.
DEV uint32_t classifyWindow(WindowBatch const & B, uint64_t const widx, uint32_t const t0inst, uint32_t const t7inst = 0)
{
	// (round 6, measured and dropped: the pile index from a table filled by a pre-pass instead of this search -- no difference, profiles/r06d)
	uint32_t lo = 0, hi = B.npiles;
	while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( B.piles[mid].winbase <= widx ) lo = mid; else hi = mid; }
	DevPile const pile = B.piles[lo];
	uint32_t const y = static_cast<uint32_t>(widx - pile.winbase);
	uint32_t astart, aend;
	windowInterval(pile.l,B.P.a,B.P.w,y,astart,aend);
	uint32_t const k = B.P.klow;
	uint32_t nact = 0, inst = B.P.w >= k ? (B.P.w - k + 1) : 0u;
	DevOvl const * ov = B.ovl + pile.first_ovl;
	for ( uint32_t z = 0; z < pile.novl; ++z )
		if ( ov[z].abpos <= static_cast<int32_t>(astart) && ov[z].aepos >= static_cast<int32_t>(aend) )
		{
			uint64_t const row = ov[z].wtoff + (y - ov[z].y0);
			uint32_t const len = B.wt_e[row] - B.wt_b[row];
			++nact; inst += len >= k ? (len - k + 1) : 0u;
		}
	uint64_t const nb = (B.P.maxalign > 0) ? (B.P.maxalign-1) : 0;
	uint32_t const mao = 1 + static_cast<uint32_t>(nact < nb ? nact : nb);
	// 0: tier 0, 1: the rest (tier 1), 2: the middle class (tier 7; only with a threshold for it)
	if ( mao <= FastTier<0>::maxs && inst <= t0inst && B.P.w <= 63 ) return 0u;
	if ( t7inst && mao <= FastTier<7>::maxs && inst <= t7inst && B.P.w <= 63 ) return 2u;
	return 1u;
}

template<typename CT>
DEV int processWindowFast(FastBatch const & FB, uint64_t const widx, LDSQ uint8_t * lds, bool const resume = false)
{
	WindowBatch const & B = FB.W;
	FastEngine<CT> E;
#if defined(DACC_EMUL)
	{ char const * pz = getenv("DACC_EMUL_POISON"); if ( pz ) __builtin_memset(static_cast<void *>(&E),atoi(pz),sizeof(E)); }   // debugging aid: no member may be read before it is set
#endif
	if ( B.pregen && ((B.pregen[widx>>5] >> (widx&31)) & 1) ) return FW_DONE;      // the generic engine has this window (a string longer than 64 bases)
	E.mao = 0; E.k = 0; E.kmask = 0; E.npre = E.nlast = E.nn = E.nmfirst = E.nmlast = 0; E.n0 = E.npool = E.nlinks = E.nwF = E.nwR = 0; E.nF = E.nL = 0;
	E.nsiq = E.ncdh = E.nacc = 0; E.rstop = 0; E.roundT0 = 0; E.cfree = 0; E.pvpath = E.pvrp = E.pvnf = 0; E.pvcl = 0; E.nmid = 0; E.midbase = 0; E.nLmagic = 0;
	E.T = B.T; E.P = B.P; E.nrows = FB.F.nrows; E.nsup = FB.F.nsup; E.vst = FB.dpsq_vst;
	E.gslab = 0; E.gtab = FB.tab32; E.sfresh = true; E.sdirty = false;
	if ( CT::gw )
	{
#if defined(DACC_EMUL)
		E.gslab = FB.gslab;
#else
		E.gslab = FB.gslab + static_cast<uint64_t>(blockIdx.x)*FB.gstride;
#endif
	}
	E.lane = wv_lane(); E.flags = 0; E.prof = B.prof;
#if defined(DACC_LEDGER)
	E.ledger = FB.ledger;
#endif
#if defined(DACC_PROFILE) && !defined(DACC_EMUL)
	uint64_t * const prof = E.prof;
#endif
#if defined(DACC_EMUL)
	g_fstats.clear();
#endif
	E.L.base = lds;
	FastLds<CT> const & L = E.L;
	int const lane = E.lane;
	PROF_T0
	// A window that does not fit is handed to the next capacity tier together with the filter frequency of the pass
	// that overflowed: with a single k the passes before it ended without a consensus and left no state behind
	// (minrate, best), so the next tier starts at that pass instead of repeating them.
	int32_t curff = B.P.maxff;
	uint32_t hslot = 0;      // hand-over slot + 1 with the sorted instances of this window (FastBatch::hand), 0: none
	bool const handable = FB.hand != 0 && B.P.klow == B.P.khigh && CT::gw != 0;
	if ( resume && B.P.klow == B.P.khigh )
	{
		WindowOut const prev = B.wout[widx];
		if ( prev.status == WS_RETRY && prev.filterfreq <= B.P.maxff && prev.filterfreq >= B.P.minff )
		{
			curff = prev.filterfreq;
			if ( handable && prev.minrate <= FB.handcap ) hslot = static_cast<uint32_t>(prev.minrate);
		}
	}
	int32_t const startff = curff;
	#define FFAIL(code) { if ( lane == 0 ) { B.wout[widx].status = WS_RETRY; B.wout[widx].flags = ((code)<<24) | (E.flags & 0xFFFFFF); B.wout[widx].filterfreq = curff; B.wout[widx].minrate = hslot; } return ((code) == 1 || (code) == 4) ? FW_GENERIC : FW_NEXT; }

	// (round 6, measured and dropped: the pile index from a table filled by a pre-pass instead of this search -- no difference, profiles/r06d)
	uint32_t lo = 0, hi = B.npiles;
	while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( B.piles[mid].winbase <= widx ) lo = mid; else hi = mid; }
	DevPile const pile = B.piles[lo];
	uint32_t const y = static_cast<uint32_t>(widx - pile.winbase);
	uint32_t astart, aend;
	windowInterval(pile.l,B.P.a,B.P.w,y,astart,aend);

	WindowOut out; out.status = WS_INSUFFICIENT; out.mao = 0; out.elength = 0; out.k = 0; out.filterfreq = -1; out.conslen = 0; out.minrate = 0; out.flags = 0;
	uint8_t * rec = B.wrec + widx*(CT::wide ? DACC_WREC_OF(B.P.w) : static_cast<uint32_t>(WREC));
	if ( lane == 0 ) rec[0] = 0;
	if ( B.P.w > (CT::wide ? 127u : 63u) ) { FFAIL(1) }

	DevOvl const * ov = B.ovl + pile.first_ovl;
	uint32_t nact = 0, mao = 0, toolong = 0;
	LEDGER_REPX(E,0)
	{
	nact = 0;
	for ( uint32_t c = 0; c < pile.novl; c += WSZ )
	{
		uint32_t const z = c + lane;
		uint32_t act = 0;
		if ( z < pile.novl ) act = (ov[z].abpos <= static_cast<int32_t>(astart)) && (ov[z].aepos >= static_cast<int32_t>(aend));
		uint32_t tot; uint32_t const pre = wv_scan_excl(act,tot);
		if ( act && nact+pre < CT::precap ) L.pre()[nact+pre] = (static_cast<uint64_t>(ov[z].ekey)<<32) | z;
		nact += tot;
	}
	if ( nact > CT::precap || next_pow2(nact < 2 ? 2 : nact) > CT::precap ) { FFAIL(2) }      // (the padding below needs the next power of two to fit)
	uint32_t const ap2 = next_pow2(nact < 2 ? 2 : nact);
	for ( uint32_t i = nact + lane; i < ap2; i += WSZ ) L.pre()[i] = ~0ull;
	wv_sync();
	wv_sort_keys<CT::precap>(L.pre(),nact);
	mao = 0;
	if ( nact )
	{
		uint64_t const nb = (B.P.maxalign > 0) ? (B.P.maxalign-1) : 0;
		mao = 1 + static_cast<uint32_t>(nact < nb ? nact : nb);
	}
	if ( mao > CT::maxs ) { FFAIL(3) }
	E.mao = mao; out.mao = mao;

	toolong = 0;
	if ( mao )
	{
		// string descriptors, one lane per string (a single round of dependent HBM loads for the whole window):
		// mfirst[j] = first base | read length << 32, mlast[j] = byte offset of the read | inverse << 63
		for ( uint32_t j = lane; j < mao; j += WSZ )
		{
			uint32_t bs, len, rl; uint64_t off; bool inv;
			if ( j == 0 ) { bs = astart; len = B.P.w; off = B.boff[pile.aread]; rl = B.rlen[pile.aread]; inv = false; }
			else
			{
				uint32_t const z = static_cast<uint32_t>(L.pre()[j-1] & 0xFFFFFFFFu);
				DevOvl const & o = ov[z];
				uint64_t const row = o.wtoff + (y - o.y0);
				bs = B.wt_b[row]; len = B.wt_e[row]-bs;
				off = B.boff[o.bread]; rl = B.rlen[o.bread]; inv = o.flags & 1;
			}
			if ( len > CT::lstr ) { toolong = 1; len = 0; }
			L.slen()[j] = len;
			L.mfirst()[j] = bs | (static_cast<uint64_t>(rl)<<32);
			L.mlast()[j] = off | (static_cast<uint64_t>(inv ? 1 : 0)<<63);
		}
		toolong = wv_any(toolong);
		wv_sync();
		// bases: lane p of string j; four strings per round so that their loads are in flight together
		for ( uint32_t j0 = 0; j0 < mao; j0 += 4 )
			for ( uint32_t p = lane; p < CT::lstr; p += WSZ )
			{
				uint8_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
				bool const a0 = p < L.slen()[j0], a1 = j0+1 < mao && p < L.slen()[j0+1], a2 = j0+2 < mao && p < L.slen()[j0+2], a3 = j0+3 < mao && p < L.slen()[j0+3];
				#define DACC_LB(u) { uint64_t const d0 = L.mfirst()[j0+u], d1 = L.mlast()[j0+u]; v##u = readBase(B.bps,d1 & 0x7FFFFFFFFFFFFFFFull,static_cast<uint32_t>(d0>>32),(d1>>63) != 0,static_cast<uint32_t>(d0)+p); }
				if ( a0 ) DACC_LB(0)
				if ( a1 ) DACC_LB(1)
				if ( a2 ) DACC_LB(2)
				if ( a3 ) DACC_LB(3)
				#undef DACC_LB
				if constexpr ( CT::gw != 0 )
				{
					// pattern masks by ballot: bit p of word c of string j <=> symbol c at position p (positions behind the string: no bit)
					// (words 4*PW*j + PW*symbol + word: PW = 2 holds strings of up to 128 bases, the second round of p fills word 1)
					enum { PW = FastLds<CT>::pw };
					uint32_t const sh0 = p - static_cast<uint32_t>(lane);      // (a 64-lane wavefront: 0 or 64; the 1-lane test build walks p)
					uint32_t const wq = sh0 >> 6, sh = sh0 & 63u;
					#define DACC_PM(u) if ( j0+u < mao ) { \
						uint64_t const b0 = wv_ballot(a##u && v##u == 0) << sh, b1 = wv_ballot(a##u && v##u == 1) << sh, b2 = wv_ballot(a##u && v##u == 2) << sh, b3 = wv_ballot(a##u && v##u == 3) << sh; \
						if ( lane == 0 ) { LDSQ uint64_t * const q = L.peq() + 4*PW*(j0+u) + wq; if ( sh == 0 ) { q[0] = b0; q[PW] = b1; q[2*PW] = b2; q[3*PW] = b3; } else { q[0] |= b0; q[PW] |= b1; q[2*PW] |= b2; q[3*PW] |= b3; } } }
					DACC_PM(0) DACC_PM(1) DACC_PM(2) DACC_PM(3)
					#undef DACC_PM
				}
				else
				{
				if ( a0 ) L.str()[(j0+0)*CT::lstr+p] = v0;
				if ( a1 ) L.str()[(j0+1)*CT::lstr+p] = v1;
				if ( a2 ) L.str()[(j0+2)*CT::lstr+p] = v2;
				if ( a3 ) L.str()[(j0+3)*CT::lstr+p] = v3;
				}
			}
	}
	wv_sync();
	}
	if ( toolong ) { FFAIL(4) }
	PROF(E,0)

	int32_t elength = 0;
	if ( mao )
	{
		E.buildPeq();
		LEDGER_REPX(E,1) elength = E.estimateLength()+1;
	}
	out.elength = elength;
	PROF(E,1)

	if ( mao >= B.P.minwindowcov )
	{
		bool pathfailed = true;
		uint64_t minrate = B.P.eminrate;
		bool haveMin = false;
		uint32_t bestlen = 0;
		LDSQ uint8_t * best = L.bestL();
		for ( uint32_t k = B.P.klow; k <= B.P.khigh; ++k )
		{
			E.k = k; E.kmask = (1ull<<(2*k))-1;
			bool instvalid = false, instsaved = false, handloaded = false;
			for ( int32_t ff = startff; ff >= B.P.minff; --ff )
			{
				curff = ff;
				PROF_T0
				// the sorted instances of this k are still in place when the pass before ended without a traversal; a pass that
				// traversed has saved them to the slab first (gw tiers)
				if ( !instvalid )
				{
					SITE_T0
					if ( instsaved ) { E.restoreInstances(); SITE(27) }
					else if ( !(hslot && !handloaded && E.loadHand(FB,hslot)) ) { LEDGER_REPX(E,2) E.buildInstances(); SITE(28) }      // handed over by the tier before: no second sort
					else { SITE(30) }
					handloaded = true;
				}
				if ( E.flags ) { FFAIL(7) }     // uniform: set from wave-uniform values only
				PROF(E,2)
				LEDGER_REPX(E,3) E.buildNodes(ff > 1 ? ff : 1);
				E.flags = wv_or(E.flags);
				// the node table does not fit this tier (82 % of tier 1's hand-overs): the instances are sorted and valid, the next tier takes them
				if ( E.flags ) { if ( handable ) hslot = E.saveHand(FB,hslot); FFAIL(7) }
				PROF(E,3)
				// A pass cannot produce a candidate if no node holds position 0 of a string (no first k-mer) or no last k-mer
				// candidate is a node (every reverse enumeration is empty): its three traversals (:2270-2322) are skipped.  The
				// traversal structures then never overwrite the instance array, which the next pass takes over.
				instvalid = false;
				{ bool dead_ = false; LEDGER_REPX(E,16) dead_ = (ff != 0 && E.passIsDead()); if ( dead_ ) { instvalid = true; continue; } }
				if ( CT::gw != 0 && ff > B.P.minff && !instsaved ) { SITE_T0 LEDGER_REPX(E,17) E.saveInstances(); instsaved = true; SITE(29) }
				LEDGER_REPX(E,4) E.buildSuccessors(mao);
				PROF(E,4)
				if ( ff == 0 )
				{
					// gap filling (HandleContext.hpp:2233-2268)
					E.loadTab();
					E.levelSuccessors2();
					E.flags = wv_or(E.flags);
					if ( E.flags ) { FFAIL(6) }
					E.buildNodes(1);
					E.flags = wv_or(E.flags);
					if ( E.flags ) { FFAIL(7) }
					E.buildSuccessors(mao);
					PROF(E,6)
				}
				E.flags = wv_or(E.flags);
				if ( E.flags ) { FFAIL(7) }
				uint32_t mintry = 0; bool lconsok = false;
				while ( true )
				{
					bool const consok = E.traverse(static_cast<int64_t>(elength)-4,static_cast<int64_t>(elength)+4);
					E.flags = wv_or(E.flags);
					if ( E.flags ) { FFAIL(8) }
					if ( consok )
					{
						uint64_t const err = L.accerr()[0];
						if ( err < minrate )
						{
							lconsok = true; minrate = err; haveMin = true;
							bestlen = L.acc()[0].l;
							if ( bestlen > ((CT::wide && B.P.w > 64u) ? FastLds<CT>::consmax : static_cast<uint32_t>(MAXCONS)) ) { FFAIL(9) }
							for ( uint32_t i = lane; i < bestlen; i += WSZ ) best[i] = L.consL()[L.acc()[0].o+i];
							out.k = k; out.filterfreq = ff;
							wv_sync();
						}
						else if ( haveMin ) lconsok = true;
						break;
					}
					else
					{
						if ( ++mintry >= 3 ) break;
					}
					if ( !E.addNextFromHeap() ) break;
				}
				if ( lconsok ) { pathfailed = false; break; }
			}
		}
		if ( !pathfailed )
		{
			out.status = WS_OK; out.conslen = bestlen; out.minrate = minrate;
			PROF_T0
			uint32_t nops = 0;
			LEDGER_REPX(E,15)
			{
			nops = 0;
			wv_sync();      // the alignment scratch may lie over the candidate buffers the lanes have just read (gw layout)
			if ( lane == 0 ) nops = E.alignAndEmit(best,bestlen);
			wv_sync();
			nops = wv_bcast(nops,0);
			E.emitRecord(best,nops,rec);
			}
			PROF(E,14)
		}
		else out.status = WS_FAILED;
	}
	if ( lane == 0 ) B.wout[widx] = out;
	wv_sync();
#if defined(DACC_EMUL)
	fstats_dump(static_cast<int>(CT::maxs));
#endif
	return FW_DONE;
	#undef FFAIL
}

#if defined(DACC_EMUL)
#undef over
#endif
}
#endif
