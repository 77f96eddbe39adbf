/*
 * TEST HARNESS ONLY (never part of libdaccord_hip.so): compiles the kernel device headers
 * with g++ as a 1-lane wavefront (DACC_EMUL, see daccord_amd/csrc/wave.hpp) and drives them
 * with plain loops, so that the kernel LOGIC can be compared with the oracle in the build
 * container, which has no GPU.  The real library has no such path: it is gfx950 code only.
 */
#define DACC_EMUL 1
#include <vector>
#include <time.h>
#include <cstdio>
#include <string>
#include <cstring>
#include <cstdlib>
#include <shared_mutex>
#include "../../daccord_amd/csrc/batch_plan.hpp"
#include "../../daccord_amd/csrc/host_tables.hpp"
#include "../../daccord_amd/csrc/window_main.hpp"
#include "../../daccord_amd/csrc/fast_window.hpp"
#include "../../daccord_amd/csrc/trace_kernel.hpp"
#include "../../daccord_amd/csrc/vote_kernel.hpp"

using namespace dacc;

struct EmulCtx
{
	dacc_params par;
	HostTables H;
	bool haveprofile;
	double est_cor;
	std::vector<uint8_t> bps; std::vector<uint64_t> boff; std::vector<uint32_t> rlen;
	std::vector<dacc_fragment> frags; std::string bases;
	std::vector<dacc_window_result> windows;
	std::string err;
	bool usefast; uint64_t ntier[3], nretry, nlong, ntier0, ntier7, ntier10;
	std::vector<uint64_t> glist;   // windows that went to the generic engine: index, flags of the last tier
	uint64_t reasonsT[3][64]; uint64_t flagbitsT[3][24];
};

// arena.hpp's guard sink (arenaGuardSink) is ONE pointer per process, and every arena_carve pushes its field offsets into whatever it
// points to.  Contexts that run on several threads (bench.py's like-for-like leg: one context per thread, all reaching their sizing
// carve within microseconds of each other) would push into each other's `guards` vector -- a race on a std::vector that ends in
// "double free or corruption" once in some dozen runs (round 5, found by -fsanitize=thread).  Sizing carves (sink set) are
// exclusive; the carve at the head of every generic-engine window (window_main.hpp, sink must read 0) shares.
static std::shared_mutex g_carve_mx;
#define GENERIC_WINDOW(wdx) { std::shared_lock<std::shared_mutex> carve_(g_carve_mx); wave_run([&]() { processWindow(WB,(wdx),arena.data()); }); }

// (diagnostics: the window and tier being emulated, for a debugger or a signal handler)
extern "C" { volatile uint64_t dacc_emul_curwin = 0; volatile int dacc_emul_curtier = -1; }

static void fillDev(EmulCtx & c, DevParams & P, DevTables & T)
{
	dacc_params const & p = c.par;
	P.w = p.w; P.a = p.a; P.klow = p.klow; P.khigh = p.khigh; P.minff = p.minfilterfreq; P.maxff = p.maxfilterfreq;
	P.minwindowcov = p.minwindowcov; P.checklim = (c.est_cor != 0.0); P.maxalign = p.maxalign; P.eminrate = p.eminrate;
	P.tspace = p.tspace; P.producefull = p.producefull; P.minlen = p.minlen;
	T.nrows = c.H.nrows; T.nsup = c.H.nsup; T.kln = c.H.kln; T.pad = 0;
	T.dpnorm = c.H.dpnorm.data(); T.dpsq = c.H.dpsq.data(); T.dpsq_vs = c.H.dpsq_vs.data();
	T.dpsq_first = c.H.dpsq_first.data(); T.dpsq_size = c.H.dpsq_size.data(); T.suplo = c.H.suplo.data(); T.suphi = c.H.suphi.data();
	T.klim = c.H.klim.data();
}

extern "C" {

void * emul_create(dacc_params const * p) { EmulCtx * c = new EmulCtx; c->par = *p; c->haveprofile = false; c->est_cor = 0; c->usefast = true; c->ntier[0] = c->ntier[1] = c->ntier[2] = c->nretry = 0; c->nlong = 0; c->ntier0 = 0; c->ntier7 = 0; c->ntier10 = 0; return c; }
void emul_set_fast(void * v, int on) { static_cast<EmulCtx *>(v)->usefast = on; }
void emul_reasons_tier(void * v, int t, uint64_t * r, uint64_t * fb) { EmulCtx * c = static_cast<EmulCtx *>(v); for ( int i = 0; i < 64; ++i ) r[i] = c->reasonsT[t][i]; for ( int i = 0; i < 24; ++i ) fb[i] = c->flagbitsT[t][i]; }
void emul_reasons2(void * v, uint64_t * r, uint64_t * fb) { emul_reasons_tier(v,1,r,fb); }
void emul_reasons(void * v, uint64_t * r, uint64_t * fb) { emul_reasons_tier(v,0,r,fb); }
void emul_counts(void * v, uint64_t * nf, uint64_t * nr) { EmulCtx * c = static_cast<EmulCtx *>(v); *nf = c->ntier0+c->ntier7+c->ntier10+c->ntier[0]+c->ntier[1]+c->ntier[2]; *nr = c->nretry; }
uint64_t emul_generic_list(void * v, uint64_t * out, uint64_t cap) { EmulCtx * c = static_cast<EmulCtx *>(v); for ( uint64_t i = 0; i < c->glist.size() && i < cap; ++i ) out[i] = c->glist[i]; return c->glist.size(); }
uint64_t emul_count_long(void * v) { return static_cast<EmulCtx *>(v)->nlong; }
void emul_counts4(void * v, uint64_t * n) { EmulCtx * c = static_cast<EmulCtx *>(v); n[0] = c->ntier[0]; n[1] = c->ntier[1]; n[2] = c->ntier[2]; n[3] = c->nretry; }
uint64_t emul_count_tier0(void * v) { return static_cast<EmulCtx *>(v)->ntier0; }
uint64_t emul_count_tier7(void * v) { return static_cast<EmulCtx *>(v)->ntier7; }
uint64_t emul_count_tier10(void * v) { return static_cast<EmulCtx *>(v)->ntier10; }
void emul_destroy(void * v) { delete static_cast<EmulCtx *>(v); }
char const * emul_error(void * v) { return static_cast<EmulCtx *>(v)->err.c_str(); }

int emul_set_error_profile(void * v, double p_i, double p_d, double est_cor)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	buildHostTables(c->H,c->par.w,p_i,p_d,est_cor,c->par.klow,c->par.khigh,4200);
	c->est_cor = est_cor; c->haveprofile = true;
	return 0;
}
int emul_tables(void * v, uint64_t * out, uint64_t cap, uint64_t * n, uint64_t klimit_n)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	std::vector<uint64_t> B; serialiseHostTables(c->H,B,klimit_n);
	*n = B.size();
	if ( out ) std::memcpy(out,B.data(),8*std::min<uint64_t>(cap,B.size()));
	return 0;
}
int emul_load_db(void * v, uint8_t const * bps, uint64_t nb, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	c->bps.assign(bps,bps+nb); c->bps.resize(nb+16,0); c->boff.assign(boff,boff+nreads); c->rlen.assign(rlen,rlen+nreads);
	return 0;
}

// FNV-1a over everything BatchPlan::plan produces (tests/test_plan.py: the plan of a batch is a pure function of its input, and its
// digest for a set of inputs -- malformed piles included -- is committed; a faster planner must reproduce it)
static void fnv(uint64_t & h, void const * p, size_t n) { uint8_t const * b = static_cast<uint8_t const *>(p); for ( size_t i = 0; i < n; ++i ) { h ^= b[i]; h *= 1099511628211ull; } }
uint64_t emul_plan_digest(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl, void const * trace, uint64_t ntrace, int trace_bytes, int * rcout)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	BatchPlan BP; std::string err;
	int const rc = BP.plan(c->par,piles,npiles,ovl,novl,trace,ntrace,trace_bytes,c->rlen.data(),c->rlen.size(),err,c->H.nrows,c->H.nsup);
	if ( rcout ) *rcout = rc;
	uint64_t h = 1469598103934665603ull;
	fnv(h,&rc,sizeof(rc)); fnv(h,err.data(),err.size());
	if ( rc ) return h;
	for ( size_t i = 0; i < BP.piles.size(); ++i ) { DevPile const & d = BP.piles[i]; uint64_t f[9] = { static_cast<uint64_t>(static_cast<int64_t>(d.aread)),d.novl,d.first_ovl,d.l,d.nwin,d.winbase,d.posbase,d.rl,d.pad }; fnv(h,f,sizeof(f)); }
	for ( size_t i = 0; i < BP.ovl.size(); ++i ) { DevOvl const & o = BP.ovl[i]; int64_t f[13] = { o.bread,static_cast<int64_t>(o.flags),o.abpos,o.aepos,o.bbpos,o.bepos,static_cast<int64_t>(o.ekey),o.y0,o.ny,o.nblk,static_cast<int64_t>(o.wtoff),static_cast<int64_t>(o.blk0),static_cast<int64_t>(o.trace_off) }; fnv(h,f,sizeof(f)); }
	if ( BP.ovl_pile.size() ) fnv(h,BP.ovl_pile.data(),BP.ovl_pile.size()*sizeof(uint32_t));
	if ( BP.fragbase.size() ) fnv(h,BP.fragbase.data(),BP.fragbase.size()*sizeof(uint64_t));
	uint64_t sc[12] = { BP.nwindows,BP.nblocks,BP.nwt,BP.npos,BP.nfragslots,BP.algo_bytes,BP.maxdepth,BP.maxcols,BP.maxspan,BP.ndeepwin,BP.deep ? 1u : 0u,BP.piles.size() };
	fnv(h,sc,sizeof(sc));
	uint64_t cp[13] = { BP.caps.maxs,BP.caps.precap,BP.caps.nodecap,BP.caps.fcap,BP.caps.strcap,BP.caps.linkcap,BP.caps.sfcap,BP.caps.rlcap,BP.caps.poolcap,BP.caps.blcap,BP.caps.conscap,BP.caps.lstr,BP.caps.bytes };
	fnv(h,cp,sizeof(cp));
	if ( BP.pile_status.size() ) fnv(h,BP.pile_status.data(),BP.pile_status.size()*sizeof(int32_t));
	for ( size_t i = 0; i < BP.pile_errors.size(); ++i ) { fnv(h,BP.pile_errors[i].data(),BP.pile_errors[i].size()); fnv(h,"|",1); }
	return h;
}

// host planning of a batch alone (BatchPlan::plan, what dacc_submit_piles does before any upload): seconds of `reps` runs
double emul_plan_seconds(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl, void const * trace, uint64_t ntrace, int trace_bytes, int reps)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC,&t0);
	for ( int r = 0; r < reps; ++r ) { BatchPlan BP; if ( BP.plan(c->par,piles,npiles,ovl,novl,trace,ntrace,trace_bytes,c->rlen.data(),c->rlen.size(),c->err,c->H.nrows,c->H.nsup) ) return -1.0; }
	clock_gettime(CLOCK_MONOTONIC,&t1);
	return (t1.tv_sec-t0.tv_sec) + 1e-9*(t1.tv_nsec-t0.tv_nsec);
}

int emul_run(void * v, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl, void const * trace, uint64_t ntrace, int trace_bytes)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	if ( !c->haveprofile ) return DACC_ESTATE;
	BatchPlan BP;
	int rc = BP.plan(c->par,piles,npiles,ovl,novl,trace,ntrace,trace_bytes,c->rlen.data(),c->rlen.size(),c->err,c->H.nrows,c->H.nsup);
	if ( rc ) return rc;
	DevParams P; DevTables T; fillDev(*c,P,T);

	// block tables (prep kernel equivalent)
	std::vector<uint32_t> blk_ovl(BP.nblocks), blk_b0(BP.nblocks);
	uint8_t const * tr = static_cast<uint8_t const *>(trace);
	for ( uint64_t o = 0; o < BP.ovl.size(); ++o )
	{
		uint32_t b = BP.ovl[o].bbpos;
		for ( int32_t i = 0; i < BP.ovl[o].nblk; ++i )
		{
			blk_ovl[BP.ovl[o].blk0+i] = o; blk_b0[BP.ovl[o].blk0+i] = b;
			b += trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(trace)[BP.ovl[o].trace_off+2*i+1] : tr[BP.ovl[o].trace_off+2*i+1];
		}
	}
	std::vector<uint32_t> wt_b(BP.nwt+1,0xDEADBEEF), wt_e(BP.nwt+1,0xDEADBEEF);
	uint32_t errflag = 0;
	{
		TraceBatch TB;
		TB.P = P; TB.bps = c->bps.data(); TB.boff = c->boff.data(); TB.rlen = c->rlen.data();
		TB.piles = BP.piles.data(); TB.ovl = BP.ovl.data(); TB.ovl_pile = BP.ovl_pile.data(); TB.trace = tr;
		TB.blk_ovl = blk_ovl.data(); TB.blk_b0 = blk_b0.data(); TB.nblocks = BP.nblocks;
		TB.wt_b = wt_b.data(); TB.wt_e = wt_e.data();
		std::vector<uint64_t> colw(traceSlots(BP.maxcols)*4); std::vector<uint16_t> colsc(traceSlots(BP.maxcols));
		std::vector<TCol> cps(traceCheckpoints(BP.maxcols)); std::vector<uint64_t> segs(4*T2S);
		TraceStoreMem st; st.cp = cps.data(); st.seg = segs.data();
		TB.maxcols = BP.maxcols; TB.trace_bytes = trace_bytes; TB.errflag = &errflag; TB.work = 0; TB.slab = 0;
		if ( P.tspace <= 128 && BP.maxcols <= 928 ) for ( uint64_t t = 0; t < BP.nblocks; ++t ) traceBlock(TB,t,st);   // as the library chooses (capi.hip: tr_words)
		else
		{
			// wide blocks: the device's lane-interleaved column store over a host buffer (16 lanes, block t on lane t % 16)
			uint32_t const nl = 16, slots = traceSlots(BP.maxcols);
			if ( P.tspace <= 256 )
			{
				std::vector<uint64_t> ww(static_cast<size_t>(slots)*8*nl); std::vector<uint16_t> ws(static_cast<size_t>(slots)*nl);
				TraceStoreW<4> sw; sw.w = ww.data(); sw.sc = ws.data(); sw.nl = nl;
				for ( uint64_t t = 0; t < BP.nblocks; ++t ) { sw.lane = t % nl; traceBlockWide<4>(TB,t,sw); }
			}
			else
			{
				std::vector<uint64_t> ww(static_cast<size_t>(slots)*16*nl); std::vector<uint16_t> ws(static_cast<size_t>(slots)*nl);
				TraceStoreW<8> sw; sw.w = ww.data(); sw.sc = ws.data(); sw.nl = nl;
				for ( uint64_t t = 0; t < BP.nblocks; ++t ) { sw.lane = t % nl; traceBlockWide<8>(TB,t,sw); }
			}
		}
	}
	if ( errflag ) { c->err = "trace kernel capacity exceeded"; return DACC_ENOTSUP; }
	for ( uint64_t i = 0; i < BP.nwt; ++i ) if ( wt_b[i] == 0xDEADBEEF || wt_e[i] == 0xDEADBEEF ) { c->err = "window table entry not written"; return DACC_EHIP; }

	// window kernel
	// Device buffers are NOT zero when a kernel first sees them (an allocation may be recycled memory, a scratch arena holds what the
	// last window or the last layout left): every buffer the library does not clear is filled with a pattern here, so that code that
	// reads a field before writing it shows up on the CPU (DACC_EMUL_ARENA_FILL=<byte>, default 0xCD).
	uint8_t const arenafill = getenv("DACC_EMUL_ARENA_FILL") ? static_cast<uint8_t>(std::strtoul(getenv("DACC_EMUL_ARENA_FILL"),0,0)) : 0xCD;
	size_t const wrecb = DACC_WREC_OF(P.w);
	std::vector<uint8_t> wrec(BP.nwindows*wrecb+wrecb,arenafill);
	std::vector<WindowOut> wout(BP.nwindows+1);
	std::memset(static_cast<void *>(wout.data()),arenafill,wout.size()*sizeof(WindowOut));
	{
		Arena A; ArenaCaps caps = BP.caps;
		// guard gaps behind every arena field (arena.hpp): their offsets, and a check that they still hold the fill pattern
		std::vector<uint64_t> guards;
		{ std::unique_lock<std::shared_mutex> carve_(g_carve_mx); arenaGuardSink() = &guards; caps.bytes = arena_carve(A,0,caps,P.w); arenaGuardSink() = 0; }
		std::vector<uint8_t> arena(caps.bytes+64,arenafill);
		uint64_t guardbad = 0;
		auto checkGuards = [&](uint64_t const wdx)
		{
			for ( size_t g = 0; g < guards.size(); ++g )
				for ( uint64_t b = 0; b < ARENA_GUARD; ++b )
					if ( arena[guards[g]+b] != arenafill )
					{
						if ( !guardbad ) std::fprintf(stderr,"[emul] window %llu wrote behind arena field %zu (offset %llu + %llu)\n",static_cast<unsigned long long>(wdx),g,static_cast<unsigned long long>(guards[g]),static_cast<unsigned long long>(b));
						++guardbad; arena[guards[g]+b] = arenafill;
					}
		};
		WindowBatch WB;
		WB.P = P; WB.T = T; WB.C = caps; WB.bps = c->bps.data(); WB.boff = c->boff.data(); WB.rlen = c->rlen.data();
		WB.piles = BP.piles.data(); WB.npiles = BP.piles.size(); WB.ovl = BP.ovl.data(); WB.wt_b = wt_b.data(); WB.wt_e = wt_e.data();
		WB.nwindows = BP.nwindows; WB.wrec = wrec.data(); WB.wout = wout.data(); WB.arena = arena.data(); WB.prof = 0; WB.pregen = 0;
		FastBatch FB[3]; FastBatch FB0, FB7;
		// hand-over slots as in the library (DACC_HAND=0: off)
		uint32_t const handwords = (BP.wide ? 4096u + 128u : (BP.deep ? 2048u + 128u : 1024u + 64u)) + 4u;
		bool const handon = !(getenv("DACC_HAND") && getenv("DACC_HAND")[0] == '0') && c->par.klow == c->par.khigh;
		std::vector<uint64_t> hand(handon ? static_cast<size_t>(BP.nwindows+1)*handwords : 1); uint32_t handctr = 0;
		// (pattern in the slots that can be used first; the whole buffer is 8 KB per window of the batch and stays untouched pages otherwise)
		std::fill(hand.begin(),hand.begin()+std::min<size_t>(hand.size(),(64u<<20)/8u),0x0101010101010101ull*arenafill);
		std::vector<uint8_t> lds[3]; std::vector<uint8_t> gslab[3]; std::vector<uint8_t> lds0, gslab0, lds7, gslab7;
		bool big = false; for ( size_t i = 0; i < c->H.dpsq_vst.size(); ++i ) if ( c->H.dpsq_vst[i] >> 32 ) big = true;
		// (round 6) wide windows (w 64 ... 127: nrows = w+1 <= 128) run in tier 8, the second slot, alone; DACC_WIDE_TIER=0: generic engine only (rounds 4-5)
		bool const widetier = BP.wide && !(getenv("DACC_WIDE_TIER") && getenv("DACC_WIDE_TIER")[0] == '0');
		bool const usefast = c->usefast && !big && (widetier ? (c->H.nrows <= 128 && c->H.nsup <= FSUPCAPW) : (c->H.nrows <= 64 && c->H.nsup <= FSUPCAP && c->par.w <= 63));
		bool tierok[3];
		// the library's pre-scan (k_prescan + k_prescan_lists): windows with an active B string of more than 64 bases are flagged (the first
		// slot's tiers skip them); those with a string of more than 128 bases (second bit map: every tier skips them) are listed for the launch
		// on the second stream, the others join the list the second slot's tier reads (round 6: tiers 6 and 3 hold strings of up to 128 bases;
		// DACC_LONG128=0: all of them go to the second stream as in rounds 3-5)
		std::vector<uint32_t> pregen((BP.nwindows+31)/32+1,0), pregen2((BP.nwindows+31)/32+1,0); std::vector<uint64_t> pregenlist, slot1list;
		bool const long128 = !(getenv("DACC_LONG128") && getenv("DACC_LONG128")[0] == '0');
		if ( usefast )
		{
			for ( size_t o = 0; o < BP.ovl.size(); ++o )
			{
				DevOvl const & ov = BP.ovl[o]; uint64_t const winbase = BP.piles[BP.ovl_pile[o]].winbase;
				for ( uint32_t r = 0; r < ov.ny; ++r )
				{
					uint32_t const len = wt_e[ov.wtoff+r] - wt_b[ov.wtoff+r];
					uint64_t const w = winbase + ov.y0 + r;
					if ( len > (widetier ? 128u : 64u) ) pregen[w>>5] |= 1u << (w&31);
					if ( len > 128u ) pregen2[w>>5] |= 1u << (w&31);
				}
			}
			WB.pregen = pregen.data();
		}
		for ( int t = 0; t < 3; ++t )
		{
			FB[t].W = WB; FB[t].F = BP.ftier[t]; FB[t].dpsq_vst = c->H.dpsq_vst.data(); FB[t].retry = 0; FB[t].gearly = 0;
			gslab[t].assign(BP.ftier[t].gbytes+64,arenafill); FB[t].gslab = gslab[t].data(); FB[t].gstride = 0; FB[t].tab32 = c->H.tab32.data();
			FB[t].hand = handon ? hand.data() : 0; FB[t].handctr = &handctr; FB[t].handcap = handon ? static_cast<uint32_t>(BP.nwindows) : 0u; FB[t].handwords = handwords;
#if defined(DACC_LEDGER)
			{ char const * lm = getenv("DACC_LEDGER_MASK"); FB[t].ledger = lm ? static_cast<uint32_t>(strtoul(lm,0,0)) : 0u; }      // (ledger build: phases run twice, results must not move)
#endif
			lds[t].assign(BP.ftier[t].ldsbytes+64,arenafill);
			tierok[t] = usefast && static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= BP.ftier[t].tabcap;
			if ( widetier && t == 0 ) tierok[t] = false;
			c->ntier[t] = 0; for ( int i = 0; i < 64; ++i ) c->reasonsT[t][i] = 0; for ( int i = 0; i < 24; ++i ) c->flagbitsT[t][i] = 0;
		}
		c->nretry = 0; c->glist.clear();
		{
			bool const slot1 = usefast && tierok[0] && (tierok[1] || tierok[2]) && long128;
			for ( uint64_t w = 0; usefast && w < BP.nwindows; ++w )
				if ( (pregen[w>>5] >> (w&31)) & 1 )
				{
					wout[w].status = WS_INSUFFICIENT;
					if ( ((pregen2[w>>5] >> (w&31)) & 1) || !slot1 ) pregenlist.push_back(w); else slot1list.push_back(w);
				}
			if ( usefast && tierok[0] && long128 ) for ( int t = 1; t < 3; ++t ) FB[t].W.pregen = pregen2.data();
			if ( usefast && widetier ) { FB[1].W.pregen = pregen2.data(); FB[2].W.pregen = pregen2.data(); }
		}
		// tier 0 (size classes) in front of tier 1 of a shallow batch, as in the library (DACC_TIERS bit 3 switches it off)
		bool const tier0ok = !BP.deep && tierok[0] && !(getenv("DACC_TIERS") && !((atoi(getenv("DACC_TIERS"))>>3)&1));
		c->ntier0 = 0; c->ntier7 = 0; c->ntier10 = 0;
		// tier 7 (the middle size class) as in the library: DACC_TIERS bit 4 switches it off, DACC_T7INST is its threshold
		uint32_t const t0inst = getenv("DACC_T0INST") ? static_cast<uint32_t>(atoi(getenv("DACC_T0INST"))) : static_cast<uint32_t>(T0INST_DEFAULT);
		uint32_t const t7inst = getenv("DACC_T7INST") ? static_cast<uint32_t>(atoi(getenv("DACC_T7INST"))) : static_cast<uint32_t>(T7INST_DEFAULT);
		bool const tier7ok = tier0ok && !(getenv("DACC_TIERS") && !((atoi(getenv("DACC_TIERS"))>>4)&1)) && t7inst > t0inst;
		if ( tier7ok )
		{
			FB7.W = WB; FB7.F = BP.ftier7; FB7.dpsq_vst = c->H.dpsq_vst.data(); FB7.retry = 0; FB7.gearly = 0;
			gslab7.assign(BP.ftier7.gbytes+64,arenafill); FB7.gslab = gslab7.data(); FB7.gstride = 0; FB7.tab32 = c->H.tab32.data();
			FB7.hand = handon ? hand.data() : 0; FB7.handctr = &handctr; FB7.handcap = handon ? static_cast<uint32_t>(BP.nwindows) : 0u; FB7.handwords = handwords;
#if defined(DACC_LEDGER)
			{ char const * lm = getenv("DACC_LEDGER_MASK"); FB7.ledger = lm ? static_cast<uint32_t>(strtoul(lm,0,0)) : 0u; }
#endif
			lds7.assign(BP.ftier7.ldsbytes+64,arenafill);
			wave_run([&]() { FastLds< FastTier<7> > L; L.base = lds7.data(); fast_load_tables(L,BP.ftier7.nrows,BP.ftier7.nsup,T,c->H.dpsq_vst.data()); });
		}
		if ( tier0ok )
		{
			FB0.W = WB; FB0.F = BP.ftier0; FB0.dpsq_vst = c->H.dpsq_vst.data(); FB0.retry = 0; FB0.gearly = 0;
			gslab0.assign(BP.ftier0.gbytes+64,arenafill); FB0.gslab = gslab0.data(); FB0.gstride = 0; FB0.tab32 = c->H.tab32.data();
			FB0.hand = handon ? hand.data() : 0; FB0.handctr = &handctr; FB0.handcap = handon ? static_cast<uint32_t>(BP.nwindows) : 0u; FB0.handwords = handwords;
#if defined(DACC_LEDGER)
			{ char const * lm = getenv("DACC_LEDGER_MASK"); FB0.ledger = lm ? static_cast<uint32_t>(strtoul(lm,0,0)) : 0u; }
#endif
			lds0.assign(BP.ftier0.ldsbytes+64,arenafill);
			wave_run([&]() { FastLds< FastTier<0> > L; L.base = lds0.data(); fast_load_tables(L,BP.ftier0.nrows,BP.ftier0.nsup,T,c->H.dpsq_vst.data()); });
		}
		// tier 10 (the dense-graph tier) between the second slot's tier 6 and tier 3 of a shallow batch, as in the library (DACC_DENSE_TIER=0 switches it off)
		FastBatch FBD; std::vector<uint8_t> ldsD, gslabD;
		bool const tier10ok = usefast && !widetier && tierok[1] && tierok[2] && !(getenv("DACC_DENSE_TIER") && getenv("DACC_DENSE_TIER")[0] == '0')
			&& static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= BP.ftierD.tabcap;
		if ( tier10ok )
		{
			FBD = FB[2]; FBD.F = BP.ftierD;
			gslabD.assign(BP.ftierD.gbytes+64,arenafill); FBD.gslab = gslabD.data();
			ldsD.assign(BP.ftierD.ldsbytes+64,arenafill);
			wave_run([&]() { if ( BP.deep ) { FastLds< FastTier<11> > L; L.base = ldsD.data(); fast_load_tables(L,BP.ftierD.nrows,BP.ftierD.nsup,T,c->H.dpsq_vst.data()); } else { FastLds< FastTier<10> > L; L.base = ldsD.data(); fast_load_tables(L,BP.ftierD.nrows,BP.ftierD.nsup,T,c->H.dpsq_vst.data()); } });
		}
		auto loadTables = [&](int const t)
		{
			wave_run([&]() {
				if ( t == 0 && BP.deep ) { FastLds< FastTier<4> > L; L.base = lds[0].data(); fast_load_tables(L,BP.ftier[0].nrows,BP.ftier[0].nsup,T,c->H.dpsq_vst.data()); }
				else if ( t == 0 ) { FastLds< FastTier<1> > L; L.base = lds[0].data(); fast_load_tables(L,BP.ftier[0].nrows,BP.ftier[0].nsup,T,c->H.dpsq_vst.data()); }
				else if ( t == 1 && BP.wide ) { FastLds< FastTier<8> > L; L.base = lds[1].data(); fast_load_tables(L,BP.ftier[1].nrows,BP.ftier[1].nsup,T,c->H.dpsq_vst.data()); }
				else if ( t == 1 && BP.deep ) { FastLds< FastTier<2> > L; L.base = lds[1].data(); fast_load_tables(L,BP.ftier[1].nrows,BP.ftier[1].nsup,T,c->H.dpsq_vst.data()); }
				else if ( t == 1 ) { FastLds< FastTier<6> > L; L.base = lds[1].data(); fast_load_tables(L,BP.ftier[1].nrows,BP.ftier[1].nsup,T,c->H.dpsq_vst.data()); }
				else if ( BP.wide ) { FastLds< FastTier<9> > L; L.base = lds[2].data(); fast_load_tables(L,BP.ftier[2].nrows,BP.ftier[2].nsup,T,c->H.dpsq_vst.data()); }
				else { FastLds< FastTier<3> > L; L.base = lds[2].data(); fast_load_tables(L,BP.ftier[2].nrows,BP.ftier[2].nsup,T,c->H.dpsq_vst.data()); }
			});
		};
		for ( int t = 0; t < 3; ++t ) loadTables(t);
		// Same orchestration as the library (capi.hip): every tier is one "kernel" over the list the previous tier
		// handed over; windows only the generic engine can run go to an early list that is read ONCE, right after the
		// first tier (later tiers hand such windows on through their ordinary list); the generic engine runs the early
		// list and, at the end, what the last tier handed over.
		auto runTier = [&](int const t, uint64_t const wdx, bool const resume) -> int
		{
			dacc_emul_curwin = wdx; dacc_emul_curtier = t;
			if ( getenv("DACC_EMUL_POISON") )
			{
				// debugging aid: no window may depend on what an earlier window (or kernel) left in LDS
				std::memset(lds[t].data(),atoi(getenv("DACC_EMUL_POISON")),lds[t].size());
				std::memset(gslab[t].data(),atoi(getenv("DACC_EMUL_POISON")),gslab[t].size());
				loadTables(t);
			}
			int rc = -1;
			wave_run([&]() {
				int r;
				if ( t == 0 && BP.deep ) r = processWindowFast< FastTier<4> >(FB[0],wdx,lds[0].data(),resume);
				else if ( t == 0 ) r = processWindowFast< FastTier<1> >(FB[0],wdx,lds[0].data(),resume);
				else if ( t == 1 && BP.wide ) r = processWindowFast< FastTier<8> >(FB[1],wdx,lds[1].data(),resume);
				else if ( t == 1 && BP.deep ) r = processWindowFast< FastTier<2> >(FB[1],wdx,lds[1].data(),resume);
				else if ( t == 1 ) r = processWindowFast< FastTier<6> >(FB[1],wdx,lds[1].data(),resume);
				else if ( BP.wide ) r = processWindowFast< FastTier<9> >(FB[2],wdx,lds[2].data(),resume);
				else r = processWindowFast< FastTier<3> >(FB[2],wdx,lds[2].data(),resume);
				if ( wv_lane() == 0 ) rc = r;
			});
			return rc;
		};
		std::vector<uint64_t> cur, next, gearly, earlysnap;
		bool haveList = false, early = false;
		for ( int t = 0; t < 3; ++t )
		{
			if ( !tierok[t] ) continue;
			next.clear();
			if ( t == 0 && tier0ok )
			{
				// k_classify + k_window_fast<0> + k_window_fast<7>: the small windows first, then the middle list followed by tier 0's hand-overs;
				// tier 1 then runs the big list followed by tier 7's hand-overs (without tier 7: by tier 0's)
				std::vector<uint64_t> small, mid, big;
				for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx )
				{
					if ( WB.pregen && ((WB.pregen[wdx>>5] >> (wdx&31)) & 1) ) continue;
					uint32_t cls = 0; wave_run([&]() { uint32_t const r = classifyWindow(WB,wdx,t0inst,tier7ok ? t7inst : 0u); if ( wv_lane() == 0 ) cls = r; });
					wout[wdx].status = WS_INSUFFICIENT;
					(cls == 0 ? small : (cls == 2 ? mid : big)).push_back(wdx);
				}
				for ( size_t i = 0; i < small.size(); ++i )
				{
					uint64_t const wdx = small[i];
					if ( getenv("DACC_EMUL_POISON") ) { std::memset(lds0.data(),atoi(getenv("DACC_EMUL_POISON")),lds0.size()); std::memset(gslab0.data(),atoi(getenv("DACC_EMUL_POISON")),gslab0.size()); wave_run([&]() { FastLds< FastTier<0> > L; L.base = lds0.data(); fast_load_tables(L,BP.ftier0.nrows,BP.ftier0.nsup,T,c->H.dpsq_vst.data()); }); }
					int rc = -1;
					dacc_emul_curwin = wdx; dacc_emul_curtier = 100;
					wave_run([&]() { int const r = processWindowFast< FastTier<0> >(FB0,wdx,lds0.data(),true); if ( wv_lane() == 0 ) rc = r; });
					if ( rc == FW_DONE ) { ++c->ntier0; continue; }
					if ( rc == FW_GENERIC ) gearly.push_back(wdx); else (tier7ok ? mid : big).push_back(wdx);
				}
				for ( size_t i = 0; i < mid.size(); ++i )
				{
					uint64_t const wdx = mid[i];
					if ( getenv("DACC_EMUL_POISON") ) { std::memset(lds7.data(),atoi(getenv("DACC_EMUL_POISON")),lds7.size()); std::memset(gslab7.data(),atoi(getenv("DACC_EMUL_POISON")),gslab7.size()); wave_run([&]() { FastLds< FastTier<7> > L; L.base = lds7.data(); fast_load_tables(L,BP.ftier7.nrows,BP.ftier7.nsup,T,c->H.dpsq_vst.data()); }); }
					int rc = -1;
					dacc_emul_curwin = wdx; dacc_emul_curtier = 107;
					wave_run([&]() { int const r = processWindowFast< FastTier<7> >(FB7,wdx,lds7.data(),true); if ( wv_lane() == 0 ) rc = r; });
					if ( rc == FW_DONE ) { ++c->ntier7; continue; }
					if ( rc == FW_GENERIC ) gearly.push_back(wdx); else big.push_back(wdx);
				}
				cur.swap(big); haveList = true;
			}
			if ( t == 2 && tier10ok && haveList )
			{
				// k_window_fast<10> over the second slot's hand-overs; tier 3 then runs what it hands on
				std::vector<uint64_t> cur10;
				for ( size_t i = 0; i < cur.size(); ++i )
				{
					uint64_t const wdx = cur[i];
					if ( FBD.W.pregen && ((FBD.W.pregen[wdx>>5] >> (wdx&31)) & 1) ) { cur10.push_back(wdx); continue; }      // (skipped by tier 3 as well)
					if ( getenv("DACC_EMUL_POISON") ) { std::memset(ldsD.data(),atoi(getenv("DACC_EMUL_POISON")),ldsD.size()); std::memset(gslabD.data(),atoi(getenv("DACC_EMUL_POISON")),gslabD.size()); wave_run([&]() { if ( BP.deep ) { FastLds< FastTier<11> > L; L.base = ldsD.data(); fast_load_tables(L,BP.ftierD.nrows,BP.ftierD.nsup,T,c->H.dpsq_vst.data()); } else { FastLds< FastTier<10> > L; L.base = ldsD.data(); fast_load_tables(L,BP.ftierD.nrows,BP.ftierD.nsup,T,c->H.dpsq_vst.data()); } }); }
					int rc = -1;
					dacc_emul_curwin = wdx; dacc_emul_curtier = 110;
					wave_run([&]() { int const r = BP.deep ? processWindowFast< FastTier<11> >(FBD,wdx,ldsD.data(),true) : processWindowFast< FastTier<10> >(FBD,wdx,ldsD.data(),true); if ( wv_lane() == 0 ) rc = r; });
					if ( rc == FW_DONE ) { ++c->ntier10; continue; }
					cur10.push_back(wdx);
				}
				cur.swap(cur10);
			}
			uint64_t const n = haveList ? cur.size() : BP.nwindows;
			for ( uint64_t i = 0; i < n; ++i )
			{
				uint64_t const wdx = haveList ? cur[i] : i;
				if ( FB[t].W.pregen && ((FB[t].W.pregen[wdx>>5] >> (wdx&31)) & 1) ) continue;      // the kernels return at once for these: not counted as run by the tier
				int const rc = runTier(t,wdx,haveList);
				if ( rc == FW_DONE ) { ++c->ntier[t]; continue; }
				uint32_t const f = wout[wdx].flags; c->reasonsT[t][(f>>24)&63]++; for ( int b = 0; b < 24; ++b ) if ( (f>>b)&1 ) c->flagbitsT[t][b]++;
				if ( rc == FW_GENERIC && !early ) gearly.push_back(wdx); else next.push_back(wdx);
			}
			if ( t == 0 && !slot1list.empty() ) { next.insert(next.begin(),slot1list.begin(),slot1list.end()); slot1list.clear(); }      // (k_prescan_lists appended them before the tier ran)
			cur.swap(next); haveList = true;
			if ( !early ) { early = true; earlysnap = gearly; }       // the early generic kernel reads its list here
		}
		// the library's launch on the second stream (k_window_long): tier 5 (strings of up to 128 bases) first, the generic
		// engine for what it cannot hold
		FastBatch FBL; FBL.W = WB; FBL.W.pregen = 0; FBL.F = BP.ftierL; FBL.dpsq_vst = c->H.dpsq_vst.data(); FBL.retry = 0; FBL.gearly = 0; FBL.gslab = 0; FBL.gstride = 0; FBL.tab32 = c->H.tab32.data(); FBL.hand = 0; FBL.handctr = 0; FBL.handcap = 0; FBL.handwords = 0;
#if defined(DACC_LEDGER)
		FBL.ledger = 0;
#endif
		std::vector<uint8_t> ldsL(BP.ftierL.ldsbytes+64,arenafill);
		bool const longok = usefast && static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= BP.ftierL.tabcap;
		auto loadTablesL = [&]() { wave_run([&]() { FastLds< FastTier<5> > L; L.base = ldsL.data(); fast_load_tables(L,BP.ftierL.nrows,BP.ftierL.nsup,T,c->H.dpsq_vst.data()); }); };
		if ( longok ) loadTablesL();
		c->nlong = 0;
		earlysnap.insert(earlysnap.begin(),pregenlist.begin(),pregenlist.end());      // launch order: pre-scan list, then the first tier's list
		for ( size_t i = 0; i < earlysnap.size(); ++i )
		{
			int rc = FW_NEXT;
			if ( longok )
			{
				if ( getenv("DACC_EMUL_POISON") ) { std::memset(ldsL.data(),atoi(getenv("DACC_EMUL_POISON")),ldsL.size()); loadTablesL(); }
				wave_run([&]() { int const r = processWindowFast< FastTier<5> >(FBL,earlysnap[i],ldsL.data(),false); if ( wv_lane() == 0 ) rc = r; });
			}
			if ( rc == FW_DONE ) { ++c->nlong; continue; }
			++c->nretry; c->glist.push_back(earlysnap[i]); c->glist.push_back(wout[earlysnap[i]].flags); GENERIC_WINDOW(earlysnap[i]) checkGuards(earlysnap[i]);
		}
		{
			uint64_t const n = haveList ? cur.size() : BP.nwindows;
			for ( uint64_t i = 0; i < n; ++i )
			{
				uint64_t const wdx = haveList ? cur[i] : i;
				++c->nretry; c->glist.push_back(wdx); c->glist.push_back(wout[wdx].flags); GENERIC_WINDOW(wdx) checkGuards(wdx);
			}
		}
		for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx ) if ( wout[wdx].status == WS_RETRY ) { c->err = "internal error: a window was handed on between engines and never processed"; return DACC_EHIP; }
		// mirrors the library: windows the generic engine could not hold are run again with grown scratch capacities
		for ( int attempt = 0; attempt < 3; ++attempt )
		{
			bool any = false;
			for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx ) if ( wout[wdx].status == WS_OVERFLOW ) any = true;
			if ( !any ) break;
			if ( getenv("DACC_EMUL_VERBOSE") ) { uint64_t n = 0; uint32_t fl = 0; for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx ) if ( wout[wdx].status == WS_OVERFLOW ) { ++n; fl |= wout[wdx].flags; } std::fprintf(stderr,"[emul] scratch retry %d: %llu windows, flags 0x%x\n",attempt,static_cast<unsigned long long>(n),fl); }
			growArenaCaps(caps,P.w);
			guards.clear(); { std::unique_lock<std::shared_mutex> carve_(g_carve_mx); arenaGuardSink() = &guards; caps.bytes = arena_carve(A,0,caps,P.w); arenaGuardSink() = 0; }
			arena.assign(caps.bytes+64,arenafill);
			WB.C = caps; WB.arena = arena.data();
			for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx ) if ( wout[wdx].status == WS_OVERFLOW ) { GENERIC_WINDOW(wdx) checkGuards(wdx); }
		}
		if ( guardbad ) { c->err = "a window wrote behind the capacity of an arena field (guard gap overwritten)"; return DACC_EHIP; }
	}
	c->windows.clear();
	bool overflow = false;
	for ( uint64_t pi = 0; pi < BP.piles.size(); ++pi )
		for ( uint32_t y = 0; y < BP.piles[pi].nwin; ++y )
		{
			uint64_t const wdx = BP.piles[pi].winbase+y;
			WindowOut const & o = wout[wdx];
			dacc_window_result r; std::memset(&r,0,sizeof(r));
			r.pile = pi; r.y = y; r.status = o.status; r.mao = o.mao; r.elength = o.elength; r.k = o.k; r.filterfreq = o.filterfreq;
			r.conslen = o.conslen; r.minrate = o.minrate;
			if ( o.status == WS_OVERFLOW ) overflow = true;
			if ( o.status == WS_OK )
			{
				uint8_t const * rec = wrec.data() + wdx*wrecb;
				uint32_t const w = P.w; bool const wide = DACC_WIDE_W(w);
				uint8_t const * sym = wide ? rec+2+2*(w+2) : rec+1+(w+2);
				uint32_t const nsym = wide ? (rec[2+2*(w+1)] | (static_cast<uint32_t>(rec[3+2*(w+1)])<<8)) : rec[1+(w+1)];
				uint32_t cl = 0;
				for ( uint32_t q = 0; q < nsym && cl < 79; ++q ) if ( sym[q] < 4 ) r.cons[cl++] = "ACGT"[sym[q]];
			}
			else r.filterfreq = (o.status == WS_FAILED) ? 0 : 0;
			c->windows.push_back(r);
		}
	if ( overflow )
	{
		uint32_t fl = 0; for ( uint64_t wdx = 0; wdx < BP.nwindows; ++wdx ) if ( wout[wdx].status == WS_OVERFLOW ) fl |= wout[wdx].flags;
		char buf[96]; std::snprintf(buf,sizeof(buf),"window kernel scratch capacity exceeded (flags 0x%x)",fl);
		c->err = buf; return DACC_ENOTSUP;
	}

	// vote
	uint8_t const votefill = getenv("DACC_EMUL_ARENA_FILL") ? static_cast<uint8_t>(std::strtoul(getenv("DACC_EMUL_ARENA_FILL"),0,0)) : 0xCD;
	std::vector<uint8_t> has(BP.npos+1,votefill), oc(BP.npos+1,votefill); std::vector<uint16_t> ld0(BP.npos+1,0x0101u*votefill); std::vector<uint32_t> ocs(BP.npos+1,0x01010101u*votefill);
	std::vector<uint8_t> outsym(2*BP.npos + 64*BP.piles.size() + 64,votefill);
	std::vector<VoteFragment> vf(BP.nfragslots+1); std::memset(static_cast<void *>(vf.data()),votefill,vf.size()*sizeof(VoteFragment));
	std::vector<uint32_t> nfrag(BP.piles.size()+1,0x01010101u*votefill);
	VoteBatch VB;
	VB.P = P; VB.bps = c->bps.data(); VB.boff = c->boff.data(); VB.rlen = c->rlen.data(); VB.piles = BP.piles.data(); VB.npiles = BP.piles.size();
	VB.wrec = wrec.data(); VB.has = has.data(); VB.ld0 = ld0.data(); VB.oc = oc.data(); VB.ocs = ocs.data(); VB.outsym = outsym.data();
	VB.frags = vf.data(); VB.fragbase = BP.fragbase.data(); VB.nfrag = nfrag.data(); VB.errflag = &errflag; VB.pilebad = 0;
	c->frags.clear(); c->bases.clear();
	for ( uint64_t pi = 0; pi < BP.piles.size(); ++pi )
	{
		DevPile const & pile = BP.piles[pi];
		uint32_t const np = pileNpos(pile);
		for ( uint32_t p = 0; p < np; ++p ) votePass1(VB,pile,p);
		uint32_t run = 0;
		for ( uint32_t p = 0; p < np; ++p ) { uint32_t const n = votePass2(VB,pile,p,0); oc[pile.posbase+p] = n; ocs[pile.posbase+p] = run; run += n; }
		if ( run > 2*np+64 ) { c->err = "vote output capacity exceeded"; return DACC_ENOTSUP; }
		uint64_t const symbase = 2*pile.posbase + 64ull*pi;
		for ( uint32_t p = 0; p < np; ++p ) votePass2(VB,pile,p,outsym.data()+symbase+ocs[pile.posbase+p]);
		voteRuns(VB,pile,pi);
		for ( uint32_t f = 0; f < nfrag[pi]; ++f )
		{
			VoteFragment const & F = vf[BP.fragbase[pi]+f];
			dacc_fragment g; g.aread = pile.aread; g.first = F.first; g.last = F.last; g.len = F.len; g.seq_off = c->bases.size();
			for ( uint32_t i = 0; i < F.len; ++i ) c->bases.push_back(static_cast<char>(outsym[F.off+i]));
			c->frags.push_back(g);
		}
	}
	if ( errflag ) { c->err = "vote kernel capacity exceeded"; return DACC_ENOTSUP; }
	return 0;
}

int emul_collect(void * v, dacc_fragment const ** frags, uint64_t * nfrags, char const ** bases, uint64_t * nbases)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	*frags = c->frags.data(); *nfrags = c->frags.size(); *bases = c->bases.data(); *nbases = c->bases.size();
	return 0;
}
int emul_windows(void * v, dacc_window_result * out, uint64_t cap, uint64_t * n)
{
	EmulCtx * c = static_cast<EmulCtx *>(v);
	*n = c->windows.size();
	if ( out ) std::memcpy(out,c->windows.data(),sizeof(dacc_window_result)*std::min<uint64_t>(cap,c->windows.size()));
	return 0;
}

}

