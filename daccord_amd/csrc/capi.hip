/*
 * libdaccord_hip.so: C ABI (include/daccord_hip.h) + gfx950 kernels.
 * Host side: device buffers, batch planning, launches on one HIP stream, HIP-event timing.
 * There is no CPU fallback anywhere in this file: no device => DACC_ENODEV.
 */
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <new>
#include "../../include/daccord_hip.h"
#include "batch_plan.hpp"
#include "host_tables.hpp"
#include "window_kernels.hpp"
#include "trace_kernel.hpp"
#include "vote_kernel.hpp"

using namespace dacc;

// ---------------------------------------------------------------- kernels

struct PrepBatch
{
	DevOvl const * ovl; uint64_t novl; uint8_t const * trace; uint32_t trace_bytes;
	uint32_t * blk_ovl; uint32_t * blk_b0;
};

// per overlap: block task table (overlap id, B start of every tspace block = prefix sum of the trace B lengths)
__global__ void k_prep(PrepBatch B)
{
	uint64_t const o = static_cast<uint64_t>(blockIdx.x)*blockDim.x + threadIdx.x;
	if ( o >= B.novl ) return;
	DevOvl const ov = B.ovl[o];
	uint32_t b = ov.bbpos;
	for ( int32_t i = 0; i < ov.nblk; ++i )
	{
		B.blk_ovl[ov.blk0+i] = o; B.blk_b0[ov.blk0+i] = b;
		b += B.trace_bytes == 2 ? reinterpret_cast<uint16_t const *>(B.trace)[ov.trace_off+2*i+1] : B.trace[ov.trace_off+2*i+1];
	}
}

// one lane per tspace block, one wavefront per workgroup; the column checkpoints of every lane go to the workgroup's global
// slab, the segment the traceback is in lives in the workgroup's dynamic LDS (TRACE2_LDS = 8 KB)
__global__ void __launch_bounds__(64) k_trace(TraceBatch B)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_trace[];
	TraceStoreGL st;
	st.g = B.slab + static_cast<size_t>(blockIdx.x)*traceSlabWords(B.maxcols);
	st.w = (LDSQ uint32_t *)lds_trace; st.lane = threadIdx.x;
	if ( B.work )
	{
		// (round 6) rounds of 64 consecutive blocks drawn from a counter: with a fixed stride the workgroups whose blocks have long B spans
		// were still running when the others had left (11.8 of 18 wavefronts per CU resident on average, profiles/r06t_pmc_summary.json)
		while ( true )
		{
			uint32_t b = 0;
			if ( threadIdx.x == 0 ) b = atomicAdd(B.work,64u);
			b = __builtin_amdgcn_readfirstlane(b);
			if ( b >= B.nblocks ) break;
			uint64_t const task = static_cast<uint64_t>(b) + threadIdx.x;
			if ( task < B.nblocks ) traceBlock(B,task,st);
		}
		return;
	}
	for ( uint64_t task = static_cast<uint64_t>(blockIdx.x)*64 + threadIdx.x; task < B.nblocks; task += static_cast<uint64_t>(gridDim.x)*64 )
		traceBlock(B,task,st);
}

// tspace in (128, 64*NW]: NW words per column vector; the first nl lanes of the wavefront take blocks (nl = 64, 32, 16 or 8:
// as many as have room for their column checkpoints in LDS)
template<int NW>
__global__ void __launch_bounds__(64) k_trace_wide(TraceBatch B, uint32_t const nl)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_trace[];
	uint32_t const slots = traceSlots(B.maxcols);
	TraceStoreW<NW> st;
	st.w = (LDSQ uint64_t *)lds_trace; st.sc = (LDSQ uint16_t *)(lds_trace + static_cast<size_t>(slots)*(2*NW)*nl*8); st.lane = threadIdx.x; st.nl = nl;
	if ( threadIdx.x < nl )
		for ( uint64_t task = static_cast<uint64_t>(blockIdx.x)*nl + threadIdx.x; task < B.nblocks; task += static_cast<uint64_t>(gridDim.x)*nl )
			traceBlockWide<NW>(B,task,st);
}

// The generic engine's per-wavefront arena grows with the deepest pile of the batch (7 MB at depth 32, 200 MB at 1000,
// 800 MB at the default -D of 5000): the number of wavefronts is bounded so that one batch's arenas stay below 32 GB
// instead of failing the batch for want of 400 GB.
static inline uint32_t boundByArena(uint64_t grid, uint64_t const bytes, uint64_t const mingrid)
{
	uint64_t const budget = 32ull << 30;
	while ( grid > mingrid && grid*bytes > budget ) grid >>= 1;
	return static_cast<uint32_t>(grid < 1 ? 1 : grid);
}

// Before the LDS tiers: one wavefront per overlap scans its rows of the window tables for B strings that the first slot's tiers (strings of
// up to 64 bases) cannot hold.  Two bit maps: pregen = a string of more than 64 bases (the first slot's tiers and the size-class pre-pass skip
// the window), pregen2 = one of more than 128 (every LDS tier skips it).  k_prescan_lists then sorts the flagged windows into two lists.
// (Round 6: tiers 6 and 3 hold strings of up to 128 bases -- two words per pattern mask -- so a window with a string of 65 ... 128 bases joins
// the list the second slot reads, next to the first slot's hand-overs; rounds 3-5 ran all of them in tier 5 on the second stream, the LDS of a
// whole CU per wavefront: at -w 56 / 60 / 63 that is 21 / 65 / 88 % of the windows of PacBio-like reads, profiles/r06l.)
__global__ void __launch_bounds__(256) k_prescan(DevOvl const * ovl, uint64_t novl, uint32_t const * ovl_pile, DevPile const * piles, uint32_t const * wt_b, uint32_t const * wt_e,
	uint32_t * pregen, uint32_t * pregen2, uint32_t const thr1)
{
	// one wavefront per overlap, lanes over its rows: coalesced reads of the two tables
	uint64_t const o = static_cast<uint64_t>(blockIdx.x)*4 + (threadIdx.x>>6);
	if ( o >= novl ) return;
	DevOvl const ov = ovl[o];
	uint64_t const winbase = piles[ovl_pile[o]].winbase;
	for ( uint32_t r = threadIdx.x & 63; r < ov.ny; r += 64 )
	{
		uint32_t const len = wt_e[ov.wtoff+r] - wt_b[ov.wtoff+r];
		if ( len > thr1 )      // 64; a wide batch (tier 8: strings of up to 128 bases): 128
		{
			uint64_t const w = winbase + ov.y0 + r;
			uint32_t const bit = 1u << (w&31);
			atomicOr(pregen + (w>>5),bit);
			if ( len > 128u ) atomicOr(pregen2 + (w>>5),bit);
		}
	}
}
// one thread per word of the bit maps: a flagged window goes to the second stream's list (a string of more than 128 bases: tier 5 / the generic
// engine) or to the list the second slot's tier reads; its result record forgets what an earlier batch left (the tiers resume from it)
__global__ void __launch_bounds__(256) k_prescan_lists(uint32_t const * pregen, uint32_t const * pregen2, uint64_t nwindows, uint32_t * list128, uint32_t * slot1list, WindowOut * wout)
{
	uint64_t const wi = static_cast<uint64_t>(blockIdx.x)*256 + threadIdx.x;
	if ( wi*32 >= nwindows ) return;
	uint32_t m = pregen[wi]; uint32_t const m2 = pregen2[wi];
	while ( m )
	{
		uint32_t const b = static_cast<uint32_t>(__builtin_ctz(m)); m &= m-1;
		uint64_t const w = wi*32 + b;
		if ( w >= nwindows ) break;
		wout[w].status = WS_INSUFFICIENT;
		uint32_t * const dst = ((m2 >> b) & 1u) || !slot1list ? list128 : slot1list;
		uint32_t const q = atomicAdd(dst,1u); dst[1+q] = static_cast<uint32_t>(w);
	}
}

// after all engines: a window still marked as handed on is an internal error (errflag); a window the generic engine
// could not hold even with grown scratch drops its pile (pilebad), the batch goes on
__global__ void k_check_done(WindowOut const * wout, uint64_t n, uint32_t * errflag, DevPile const * piles, uint32_t npiles, uint8_t * pilebad)
{
	uint64_t const i = static_cast<uint64_t>(blockIdx.x)*blockDim.x + threadIdx.x;
	if ( i >= n ) return;
	uint32_t const st = wout[i].status;
	if ( st == WS_RETRY ) atomicOr(errflag,1u);
	if ( st == WS_OVERFLOW )
	{
		uint32_t lo = 0, hi = npiles;
		while ( hi-lo > 1 ) { uint32_t const mid = (lo+hi)>>1; if ( piles[mid].winbase <= i ) lo = mid; else hi = mid; }
		pilebad[lo] = 1;
	}
}

// windows the generic engine left as WS_OVERFLOW -> list (count in list[0])
__global__ void k_collect_overflow(WindowOut const * wout, uint64_t n, uint32_t * list)
{
	uint64_t const i = static_cast<uint64_t>(blockIdx.x)*blockDim.x + threadIdx.x;
	if ( i < n && wout[i].status == WS_OVERFLOW ) { uint32_t const q = atomicAdd(list,1u); list[1+q] = static_cast<uint32_t>(i); }
}

// exclusive scan of one value per thread over the 256 threads of the workgroup (4 wavefronts): DPP scan inside a
// wavefront, the four wavefront totals through LDS; MAXOP: running maximum instead of a sum
template<bool MAXOP>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t const v, uint32_t * part4, uint32_t & total)
{
	uint32_t const wave = threadIdx.x >> 6;
	uint32_t x = v;
	if ( MAXOP ) { DACC_DPP_SCAN(x,DACC_OP_MAX,0u) } else { DACC_DPP_SCAN(x,DACC_OP_ADD,0u) }
	// x = inclusive value of this lane; the exclusive one is the inclusive value of the lane before
	uint32_t excl = __shfl_up(x,1,64); if ( (threadIdx.x & 63) == 0 ) excl = 0;
	__syncthreads();
	if ( (threadIdx.x & 63) == 63 ) part4[wave] = x;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	#pragma unroll
	for ( uint32_t i = 0; i < 4; ++i )
	{
		uint32_t const t = part4[i];
		if ( i < wave ) base = MAXOP ? (t > base ? t : base) : (base + t);
		tot = MAXOP ? (t > tot ? t : tot) : (tot + t);
	}
	total = tot;
	return MAXOP ? (excl > base ? excl : base) : (base + excl);
}

// Size classes (shallow batches): one thread per window decides from the window tables whether the window starts in tier 0 (small list),
// in tier 7 (middle list = the list tier 0 appends its hand-overs to; round 6) or in tier 1 (big list = the list tier 7 appends its
// hand-overs to); windows the pre-scan set aside are in none.  Blocks reserve their range of a list with one atomic each, so a list keeps
// the window order inside a block.  t7inst == 0: no middle class.
__global__ void __launch_bounds__(256) k_classify(WindowBatch B, uint32_t * small, uint32_t * mid, uint32_t * big, uint32_t const t0inst, uint32_t const t7inst)
{
	__shared__ uint32_t part4[4]; __shared__ uint32_t base3[3];
	uint64_t const w = static_cast<uint64_t>(blockIdx.x)*256 + threadIdx.x;
	uint32_t cls = 3;
	if ( w < B.nwindows && !(B.pregen && ((B.pregen[w>>5] >> (w&31)) & 1)) )
	{
		cls = classifyWindow(B,w,t0inst,t7inst);
		B.wout[w].status = WS_INSUFFICIENT;      // never WS_RETRY from an earlier batch: the tiers resume from a hand-over record
	}
	uint32_t tot0, tot1, tot2;
	uint32_t const p0 = block_scan_excl<false>(cls == 0 ? 1u : 0u,part4,tot0);
	__syncthreads();
	uint32_t const p1 = block_scan_excl<false>(cls == 1 ? 1u : 0u,part4,tot1);
	__syncthreads();
	uint32_t const p2 = block_scan_excl<false>(cls == 2 ? 1u : 0u,part4,tot2);
	if ( threadIdx.x == 0 ) { base3[0] = tot0 ? atomicAdd(small,tot0) : 0u; base3[1] = tot1 ? atomicAdd(big,tot1) : 0u; base3[2] = tot2 ? atomicAdd(mid,tot2) : 0u; }
	__syncthreads();
	if ( cls == 0 ) small[1+base3[0]+p0] = static_cast<uint32_t>(w);
	if ( cls == 1 ) big[1+base3[1]+p1] = static_cast<uint32_t>(w);
	if ( cls == 2 ) mid[1+base3[2]+p2] = static_cast<uint32_t>(w);
}


// one workgroup per pile.  The pile is walked in tiles of 256 positions (thread = position); the records of the windows
// that cover a tile (256/a + w/a + 2 of them, 256 B each) are staged in LDS once per pass, coalesced, so that the
// byte-wise reads of the vote hit LDS and every record leaves HBM once per pass.
enum { VOTE_STAGE = 48 };
__device__ __forceinline__ VoteTile vote_stage(VoteBatch const & B, DevPile const & pile, uint32_t const p0, uint32_t const np, uint8_t * lds)
{
	VoteTile VT; VT.stage = (LDSQ uint8_t const *)lds; VT.y0 = 0; VT.n = 0;
	uint32_t const a = B.P.a, w = B.P.w, nwin = pile.nwin;
	if ( ! nwin ) return VT;
	uint32_t const plast = (p0+255 < np) ? (p0+255) : (np-1);
	uint32_t const ylo = (p0 > w) ? ((p0-w + a-1)/a) : 0;
	uint32_t yhi = plast / a; if ( yhi > nwin-1 ) yhi = nwin-1;
	__syncthreads();      // the previous tile's readers are done
	if ( ylo <= yhi && yhi-ylo+1 <= VOTE_STAGE && !DACC_WIDE_W(w) )      // (wide records, w > 64, are read where they lie)
	{
		VT.y0 = ylo; VT.n = yhi-ylo+1;
		uint4 const * src = reinterpret_cast<uint4 const *>(B.wrec + (pile.winbase+ylo)*WREC);
		uint4 * dst = reinterpret_cast<uint4 *>(lds);
		for ( uint32_t i = threadIdx.x; i < VT.n*(WREC/16); i += 256 ) dst[i] = src[i];
	}
	__syncthreads();
	return VT;
}
__global__ void __launch_bounds__(256) k_vote(VoteBatch B)
{
	__shared__ uint32_t part4[4];
	__shared__ __attribute__((aligned(16))) uint8_t stage[VOTE_STAGE*WREC];
	uint32_t const pi = blockIdx.x;
	DevPile const pile = B.piles[pi];
	uint32_t const np = pileNpos(pile);
	uint32_t const tid = threadIdx.x;
	if ( B.pilebad && B.pilebad[pi] ) { if ( tid == 0 ) B.nfrag[pi] = 0; return; }     // dropped pile: no fragments
	for ( uint32_t t0 = 0; t0 < np; t0 += 256 )
	{
		VoteTile const VT = vote_stage(B,pile,t0,np,stage);
		if ( t0+tid < np ) votePass1(B,pile,t0+tid,VT);
	}
	__syncthreads();
	uint32_t run0 = 0;
	for ( uint32_t t0 = 0; t0 < np; t0 += 256 )
	{
		VoteTile const VT = vote_stage(B,pile,t0,np,stage);
		uint32_t const p = t0+tid;
		uint32_t const n = p < np ? votePass2(B,pile,p,0,VT) : 0u;
		uint32_t tot; uint32_t const ex = block_scan_excl<false>(n,part4,tot);
		if ( p < np ) { B.oc[pile.posbase+p] = n; B.ocs[pile.posbase+p] = run0 + ex; }
		run0 += tot;
	}
	uint32_t const total = run0;
	bool const fits = total <= 2*np+64;
	if ( !fits && tid == 0 ) atomicOr(B.errflag,2u);
	uint64_t const symbase = 2*pile.posbase + 64ull*pi;
	if ( fits )
		for ( uint32_t t0 = 0; t0 < np; t0 += 256 )
		{
			VoteTile const VT = vote_stage(B,pile,t0,np,stage);
			uint32_t const p = t0+tid;
			if ( p < np ) votePass2(B,pile,p,B.outsym+symbase+B.ocs[pile.posbase+p],VT);
		}
	__syncthreads();
	uint32_t const chunk = (np + 255)/256;
	uint32_t const p0 = tid*chunk < np ? tid*chunk : np, p1 = (p0+chunk < np) ? (p0+chunk) : np;
	// runs of consecutive positions with elements, kept if last-first >= 100 (HandleContext.hpp:2590-2612): every thread
	// walks its chunk; the start of the run that is open at the chunk's first position comes from a running maximum
	// over the chunks before it (start position + 1, 0 = none)
	uint8_t const * has = B.has + pile.posbase;
	uint32_t laststart = 0; bool open = false;
	for ( uint32_t p = p0; p < p1; ++p )
		if ( has[p] && (p == 0 || !has[p-1]) ) laststart = p+1;
	uint32_t dummy;
	uint32_t const instart = block_scan_excl<true>(laststart,part4,dummy);
	(void)open;
	uint32_t cur = instart;       // start+1 of the run a position of this chunk belongs to
	uint32_t nfr = 0;
	for ( int pass = 0; pass < 2; ++pass )
	{
		uint32_t fbase = 0, ftot = 0;
		if ( pass == 1 ) { fbase = block_scan_excl<false>(nfr,part4,ftot); if ( tid == 0 ) B.nfrag[pi] = fits ? ftot : 0; if ( !fits ) break; }
		cur = instart; uint32_t k = 0;
		VoteFragment * F = B.frags + B.fragbase[pi];
		for ( uint32_t p = p0; p < p1; ++p )
		{
			if ( ! has[p] ) continue;
			if ( p == 0 || !has[p-1] ) cur = p+1;
			if ( p+1 == np || !has[p+1] )
			{
				uint32_t const first = cur-1, q = p;
				if ( q-first >= 100 )
				{
					uint32_t const so = B.ocs[pile.posbase+first];
					uint32_t const eo = B.ocs[pile.posbase+q] + B.oc[pile.posbase+q];
					uint32_t const len = eo-so;
					if ( B.P.producefull || len >= B.P.minlen )
					{
						if ( pass == 1 ) { VoteFragment & f = F[fbase+k]; f.first = first; f.last = q; f.len = len; f.pad = 0; f.off = symbase + so; }
						++k;
					}
				}
			}
		}
		nfr = k;
	}
}

// ---------------------------------------------------------------- host

namespace {

template<typename T>
struct DevBuf
{
	T * p; size_t cap;
	DevBuf() : p(0), cap(0) {}
	hipError_t ensure(size_t n)
	{
		if ( n <= cap ) return hipSuccess;
		if ( p ) { hipFree(p); p = 0; cap = 0; }
		hipError_t const e = hipMalloc(reinterpret_cast<void **>(&p),n*sizeof(T));
		if ( e == hipSuccess ) cap = n;
		return e;
	}
	void release() { if ( p ) hipFree(p); p = 0; cap = 0; }
};

// pinned host buffer (the symbol stream of a batch comes back at PCIe speed instead of through a pageable staging copy)
template<typename T>
struct HostBuf
{
	T * p; size_t cap; bool pinned;
	HostBuf() : p(0), cap(0), pinned(false) {}
	// pinned memory (the device-to-host copy of the symbol stream runs at link speed into it); where the process may not lock
	// that much memory (memlock limit) a pageable buffer does the same job through the runtime's staging copy
	hipError_t ensure(size_t n)
	{
		if ( n <= cap ) return hipSuccess;
		release();
		hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p),n*sizeof(T),hipHostMallocDefault);
		if ( e == hipSuccess ) { cap = n; pinned = true; return e; }
		(void)hipGetLastError();
		p = static_cast<T *>(std::malloc(n*sizeof(T)));
		if ( !p ) return hipErrorOutOfMemory;
		cap = n; pinned = false;
		return hipSuccess;
	}
	void release() { if ( p ) { if ( pinned ) hipHostFree(p); else std::free(p); } p = 0; cap = 0; pinned = false; }
};

}

struct dacc_ctx
{
	dacc_params par;
	int device;
	hipStream_t stream; hipStream_t stream2; hipEvent_t evFirstTier, evEarlyGeneric, evPrescan;   // stream2: generic engine for the windows no LDS tier can run, concurrent with tiers 2 and 3
	hipEvent_t ev[6]; hipEvent_t evtier[3]; hipEvent_t evT0; bool tier0_ran;
	std::string err;
	bool haveprofile, havedb, havebatch;
	double est_cor;
	HostTables H;
	DevTables T; DevParams P;
	DevBuf<double> d_dpnorm, d_dpsq; DevBuf<uint64_t> d_vs; DevBuf<uint16_t> d_first, d_size, d_suplo, d_suphi; DevBuf<uint32_t> d_klim;
	DevBuf<uint8_t> d_bps; DevBuf<uint64_t> d_boff; DevBuf<uint32_t> d_rlen;
	std::vector<uint32_t> h_rlen;
	BatchPlan BP;
	DevBuf<DevPile> d_piles; DevBuf<DevOvl> d_ovl; DevBuf<uint32_t> d_ovl_pile; DevBuf<uint8_t> d_trace;
	DevBuf<uint32_t> d_blk_ovl, d_blk_b0, d_wt_b, d_wt_e;
	DevBuf<uint8_t> d_wrec; DevBuf<WindowOut> d_wout; DevBuf<uint8_t> d_arena;
	DevBuf<uint8_t> d_has, d_oc, d_outsym, d_pilebad; DevBuf<uint16_t> d_ld0; DevBuf<uint32_t> d_ocs, d_nfrag, d_err;
	DevBuf<VoteFragment> d_frags; DevBuf<uint64_t> d_fragbase; DevBuf<uint64_t> d_prof;
	DevBuf<uint64_t> d_vst; DevBuf<uint32_t> d_tab32; DevBuf<uint8_t> d_gslab; DevBuf<uint32_t> d_retry[3], d_work, d_gearly, d_pregen, d_pregen2, d_pregenlist; DevBuf<uint8_t> d_arena2; DevBuf<uint64_t> d_trslab; DevBuf<uint32_t> d_small, d_big, d_mid; DevBuf<uint64_t> d_hand; DevBuf<uint32_t> d_handctr; uint32_t handcap, handwords; uint64_t handwant, nruns; bool nohand, oom; bool tier0_ok; uint32_t tier0_grid; uint64_t gstride0; bool tier7_ok, tier7_ran, tier7_adapt_off; int env_t7adapt; int env_trdyn; int env_widetier; bool widetier; int env_dense; bool tier10_ok, tier10_ran; uint32_t tier10_grid, tier10_out; uint64_t gstride10; DevBuf<uint32_t> d_dense; hipEvent_t evT10; uint32_t tier7_grid; uint64_t gstride7; hipEvent_t evT7;
	uint32_t nlong[2];      // windows on the two lists of the second stream in the current pass (pre-scan, first tier's generic-only windows)
	uint32_t tier_grid[3], retry_grid, early_grid; uint64_t gstride[3]; int tier_ok[3]; int tierL_ok; int usefast; int sched; uint32_t tier_out[3];
	uint32_t tr_grid, tr_lds, tr_words, tr_lanes, trace_bytes, win_grid;
	int env_nofast, env_sched, env_tiers, env_dbgretry; uint32_t env_lds_t1, env_lds_t0, env_t0inst, env_t7inst; int env_long128;     // debugging knobs, read once in dacc_create
	std::vector<uint32_t> retry_flags;                      // DACC_DEBUG_RETRY: (window, flags) of what the last LDS tier handed on
	std::vector<dacc_fragment> frags; std::string bases;
	std::vector<int32_t> pile_status; std::vector<std::string> pile_errors; std::string pile_errors_joined;
	std::vector<uint32_t> h_nfrag; std::vector<VoteFragment> h_frags; HostBuf<uint8_t> h_outsym;
	dacc_timing timing;
};

#define HIPCHK(call) do { hipError_t const e_ = (call); if ( e_ != hipSuccess ) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); if ( e_ == hipErrorOutOfMemory ) c->oom = true; return DACC_EHIP; } } while (0)

template<typename T>
static int upload(dacc_ctx * c, DevBuf<T> & b, T const * src, size_t n)
{
	HIPCHK(b.ensure(n ? n : 1));
	if ( n ) HIPCHK(hipMemcpyAsync(b.p,src,n*sizeof(T),hipMemcpyHostToDevice,c->stream));
	return DACC_OK;
}

// The host side of these entry points allocates (plans, tables, staging vectors; the planner also runs threads): nothing leaves through
// the C ABI but a return code.  Their bodies are the *_body functions below.
template<typename F> static int guarded(dacc_ctx * c, F const & f)
{
	try { return f(); }
	catch ( std::bad_alloc const & ) { if ( c ) { try { c->err = "out of host memory"; } catch ( ... ) {} } return DACC_ENOMEM; }
	//  anything that is not an allocation failure -- std::system_error of a planner thread, length_error -- is an internal error,
	// not "out of memory", and the message names it instead of leaving the previous call's text in place
	catch ( std::exception const & ex ) { if ( c ) { try { c->err = std::string("host side failed: ") + ex.what(); } catch ( ... ) {} } return DACC_EINTERNAL; }
	catch ( ... ) { if ( c ) { try { c->err = "host side failed: unknown exception"; } catch ( ... ) {} } return DACC_EINTERNAL; }
}

extern "C" {

// number of HIP devices this process sees (0 if there is none or the runtime fails): lets a caller that spreads workers over
// devices tell "no such device" from any other dacc_create failure
int dacc_device_count(void)
{
	int ndev = 0;
	if ( hipGetDeviceCount(&ndev) != hipSuccess || ndev < 0 ) return 0;
	return ndev;
}

int dacc_create(dacc_ctx ** out, dacc_params const * p)
{
	if ( !out || !p ) return DACC_EINVAL;
	*out = 0;
	if ( p->klow < 3 || p->khigh > 16 || p->klow > p->khigh || !p->w || !p->a || p->w > DACC_WMAX || p->minfilterfreq < 0 ||
	     p->maxfilterfreq < p->minfilterfreq || p->tspace <= 0 )
		return DACC_EINVAL;
	int ndev = 0;
	if ( hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev )
		return DACC_ENODEV;
	if ( hipSetDevice(p->device) != hipSuccess ) return DACC_ENODEV;
	dacc_ctx * c = new (std::nothrow) dacc_ctx;
	if ( !c ) return DACC_ENOMEM;
	c->par = *p; c->device = p->device; c->haveprofile = c->havedb = c->havebatch = false; c->est_cor = 0; c->nlong[0] = c->nlong[1] = 0; c->handcap = 0; c->handwords = 0; c->handwant = 0; c->nruns = 0; c->nohand = false; c->oom = false;
	std::memset(&c->timing,0,sizeof(c->timing));
	// launch geometry of the LDS tiers: set by every batch that uses them; a generic-only batch (DACC_NOFAST, w >= 64, a model table no
	// tier holds) reads retry_grid in its scratch retry and must not find an indeterminate value there
	c->retry_grid = c->early_grid = c->win_grid = 0; c->tier0_grid = 0; c->tier0_ok = false; c->tier0_ran = false; c->tier7_grid = 0; c->tier7_ok = false; c->tier7_ran = false; c->gstride7 = 0; c->tier7_adapt_off = false; c->tierL_ok = 0; c->usefast = 0; c->sched = 0;
	for ( int i = 0; i < 3; ++i ) { c->tier_grid[i] = 0; c->tier_ok[i] = 0; c->tier_out[i] = 0; c->gstride[i] = 0; }
	c->gstride0 = 0; c->tr_grid = c->tr_lds = c->tr_words = c->tr_lanes = c->trace_bytes = 0;
	{
		char const * e = getenv("DACC_NOFAST"); c->env_nofast = (e && e[0] == '1');
		char const * sc = getenv("DACC_SCHED"); c->env_sched = sc ? atoi(sc) : 1;      // bit 0: LDS tiers pull work from a counter, bit 1: generic engine too
		char const * tm = getenv("DACC_TIERS"); c->env_tiers = tm ? atoi(tm) : 31;      // bit t enables LDS tier t+1, bit 3 tier 0 (size classes), bit 4 tier 7 (the middle class; needs tier 0)
		char const * l8 = getenv("DACC_LONG128"); c->env_long128 = !(l8 && l8[0] == '0');      // 0: windows with a string of 65 ... 128 bases run in tier 5 on the second stream (rounds 3-5)
		{ char const * wt = getenv("DACC_WIDE_TIER"); c->env_widetier = !(wt && wt[0] == '0'); c->widetier = false; }
		{ char const * dt = getenv("DACC_DENSE_TIER"); c->env_dense = !(dt && dt[0] == '0'); c->tier10_ok = false; c->tier10_ran = false; c->tier10_grid = 0; c->tier10_out = 0; c->gstride10 = 0; }      // 0: tier 6 hands on to tier 3 directly (before round 6's dense tier)
		{ char const * td = getenv("DACC_TRACE_DYN"); c->env_trdyn = !(td && td[0] == '0'); }      // 0: k_trace walks its blocks with a fixed stride (rounds 1-5)
		char const * ta = getenv("DACC_T7_ADAPT"); c->env_t7adapt = !(ta && ta[0] == '0');      // 0: tier 7 stays on whatever it hands on
		char const * t7 = getenv("DACC_T7INST"); c->env_t7inst = t7 ? static_cast<uint32_t>(atoi(t7)) : static_cast<uint32_t>(T7INST_DEFAULT);      // size-class threshold of tier 7
		char const * l1 = getenv("DACC_LDS_T1"); c->env_lds_t1 = l1 ? static_cast<uint32_t>(atoi(l1)) : 0u;      // measurement only: LDS bytes requested for the first tier (more than it needs = fewer wavefronts per CU)
		char const * l0 = getenv("DACC_LDS_T0"); c->env_lds_t0 = l0 ? static_cast<uint32_t>(atoi(l0)) : 0u;      // the same for tier 0 (size classes)
		char const * t0 = getenv("DACC_T0INST"); c->env_t0inst = t0 ? static_cast<uint32_t>(atoi(t0)) : static_cast<uint32_t>(T0INST_DEFAULT);      // size-class threshold (k-mer instances) of tier 0
		char const * dr = getenv("DACC_DEBUG_RETRY"); c->env_dbgretry = (dr && dr[0] == '1');
	}
	if ( hipStreamCreate(&c->stream) != hipSuccess ) { delete c; return DACC_EHIP; }
	{ int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo,&hi); if ( hipStreamCreateWithPriority(&c->stream2,hipStreamNonBlocking,hi) != hipSuccess ) { delete c; return DACC_EHIP; } }
	hipEventCreateWithFlags(&c->evFirstTier,hipEventDisableTiming); hipEventCreateWithFlags(&c->evEarlyGeneric,hipEventDisableTiming); hipEventCreateWithFlags(&c->evPrescan,hipEventDisableTiming);
	for ( int i = 0; i < 6; ++i ) hipEventCreate(&c->ev[i]);
	for ( int i = 0; i < 3; ++i ) hipEventCreate(&c->evtier[i]);
	hipEventCreate(&c->evT0); c->tier0_ran = false; hipEventCreate(&c->evT7); hipEventCreate(&c->evT10);
	*out = c;
	return DACC_OK;
}

void dacc_destroy(dacc_ctx * c)
{
	if ( !c ) return;
	hipSetDevice(c->device);
	hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2);
	c->d_dpnorm.release(); c->d_dpsq.release(); c->d_vs.release(); c->d_first.release(); c->d_size.release(); c->d_suplo.release(); c->d_suphi.release(); c->d_klim.release();
	c->d_bps.release(); c->d_boff.release(); c->d_rlen.release();
	c->d_piles.release(); c->d_ovl.release(); c->d_ovl_pile.release(); c->d_trace.release(); c->d_blk_ovl.release(); c->d_blk_b0.release(); c->d_wt_b.release(); c->d_wt_e.release();
	c->d_wrec.release(); c->d_wout.release(); c->d_arena.release();
	c->h_outsym.release(); c->d_pilebad.release(); c->d_has.release(); c->d_oc.release(); c->d_outsym.release(); c->d_ld0.release(); c->d_ocs.release(); c->d_nfrag.release(); c->d_err.release(); c->d_frags.release(); c->d_fragbase.release(); c->d_prof.release(); c->d_vst.release(); c->d_tab32.release(); c->d_gslab.release(); for ( int i = 0; i < 3; ++i ) c->d_retry[i].release(); c->d_work.release(); c->d_gearly.release(); c->d_pregen.release(); c->d_pregen2.release(); c->d_pregenlist.release(); c->d_arena2.release(); c->d_trslab.release(); c->d_small.release(); c->d_big.release(); c->d_mid.release(); c->d_hand.release(); c->d_handctr.release();
	for ( int i = 0; i < 6; ++i ) hipEventDestroy(c->ev[i]);
	hipStreamDestroy(c->stream); hipStreamDestroy(c->stream2); hipEventDestroy(c->evFirstTier); hipEventDestroy(c->evEarlyGeneric); hipEventDestroy(c->evPrescan); for ( int i = 0; i < 3; ++i ) hipEventDestroy(c->evtier[i]);
	hipEventDestroy(c->evT0); hipEventDestroy(c->evT7); hipEventDestroy(c->evT10); c->d_dense.release();
	delete c;
}

char const * dacc_last_error(dacc_ctx * c) { return c ? c->err.c_str() : "null context"; }

static int dacc_set_error_profile_body(dacc_ctx * c, double p_i, double p_d, double est_cor)
{
	if ( !c ) return DACC_EINVAL;
	if ( !(p_i >= 0 && p_i < 1 && p_d >= 0 && p_d < 1 && est_cor >= 0 && est_cor <= 1) ) { c->err = "error profile out of range"; return DACC_EINVAL; }
	hipSetDevice(c->device);
	buildHostTables(c->H,c->par.w,p_i,p_d,est_cor,c->par.klow,c->par.khigh,4200);
	c->est_cor = est_cor;
	int rc;
	if ( (rc = upload(c,c->d_dpnorm,c->H.dpnorm.data(),c->H.dpnorm.size())) ) return rc;
	if ( (rc = upload(c,c->d_dpsq,c->H.dpsq.data(),c->H.dpsq.size())) ) return rc;
	if ( (rc = upload(c,c->d_vs,c->H.dpsq_vs.data(),c->H.dpsq_vs.size())) ) return rc;
	if ( (rc = upload(c,c->d_vst,c->H.dpsq_vst.data(),c->H.dpsq_vst.size())) ) return rc;
	if ( (rc = upload(c,c->d_tab32,c->H.tab32.data(),c->H.tab32.size())) ) return rc;
	if ( (rc = upload(c,c->d_first,c->H.dpsq_first.data(),c->H.dpsq_first.size())) ) return rc;
	if ( (rc = upload(c,c->d_size,c->H.dpsq_size.data(),c->H.dpsq_size.size())) ) return rc;
	if ( (rc = upload(c,c->d_suplo,c->H.suplo.data(),c->H.suplo.size())) ) return rc;
	if ( (rc = upload(c,c->d_suphi,c->H.suphi.data(),c->H.suphi.size())) ) return rc;
	if ( (rc = upload(c,c->d_klim,c->H.klim.data(),c->H.klim.size())) ) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	DevTables & T = c->T;
	T.nrows = c->H.nrows; T.nsup = c->H.nsup; T.kln = c->H.kln; T.pad = 0;
	T.dpnorm = c->d_dpnorm.p; T.dpsq = c->d_dpsq.p; T.dpsq_vs = c->d_vs.p; T.dpsq_first = c->d_first.p; T.dpsq_size = c->d_size.p;
	T.suplo = c->d_suplo.p; T.suphi = c->d_suphi.p; T.klim = c->d_klim.p;
	dacc_params const & p = c->par;
	DevParams & P = c->P;
	P.w = p.w; P.a = p.a; P.klow = p.klow; P.khigh = p.khigh; P.minff = p.minfilterfreq; P.maxff = p.maxfilterfreq;
	P.minwindowcov = p.minwindowcov; P.checklim = (est_cor != 0.0); P.maxalign = p.maxalign; P.eminrate = p.eminrate;
	P.tspace = p.tspace; P.producefull = p.producefull; P.minlen = p.minlen;
	c->haveprofile = true;
	return DACC_OK;
}

int dacc_debug_tables(dacc_ctx * c, uint64_t * out, uint64_t cap, uint64_t * n, uint64_t klimit_n)
{
	if ( !c || !n ) return DACC_EINVAL;
	if ( !c->haveprofile ) return DACC_ESTATE;
	std::vector<uint64_t> B; serialiseHostTables(c->H,B,klimit_n);
	*n = B.size();
	if ( out ) std::memcpy(out,B.data(),8*std::min<uint64_t>(cap,B.size()));
	return DACC_OK;
}

static int dacc_load_db_body(dacc_ctx * c, uint8_t const * bps, uint64_t bps_bytes, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads)
{
	if ( !c || !bps || !boff || !rlen || !nreads ) return DACC_EINVAL;
	hipSetDevice(c->device);
	for ( uint64_t i = 0; i < nreads; ++i )
		if ( boff[i] + (rlen[i]+3)/4 > bps_bytes ) { c->err = "read store offsets exceed the buffer"; return DACC_EINVAL; }
	int rc;
	HIPCHK(c->d_bps.ensure(bps_bytes+16));
	HIPCHK(hipMemsetAsync(c->d_bps.p,0,bps_bytes+16,c->stream));
	HIPCHK(hipMemcpyAsync(c->d_bps.p,bps,bps_bytes,hipMemcpyHostToDevice,c->stream));
	if ( (rc = upload(c,c->d_boff,boff,nreads)) ) return rc;
	if ( (rc = upload(c,c->d_rlen,rlen,nreads)) ) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	c->h_rlen.assign(rlen,rlen+nreads);
	c->havedb = true;
	return DACC_OK;
}

static int runDevice(dacc_ctx * c)
{
	BatchPlan & BP = c->BP;
	hipStream_t const s = c->stream;
	// DACC_DEBUG_SYNC=1: wait for every kernel of the generic-only path and say so on stderr (a device fault then names its kernel)
	static bool const dbgsync = getenv("DACC_DEBUG_SYNC") && getenv("DACC_DEBUG_SYNC")[0] == '1';
	auto const mark = [&](char const * what) { if ( dbgsync ) { hipError_t const e = hipStreamSynchronize(s); std::fprintf(stderr,"[dacc] %s: %s\n",what,hipGetErrorString(e)); std::fflush(stderr); } };
	if ( c->nruns && c->handwant > c->handcap && !c->nohand )
	{
		// second use of this context: now the hand-over buffer pays (dacc_submit_piles); an optimisation only -- if the device
		// cannot spare it, halve it, and in the end do without
		uint64_t cap = c->handwant;
		//  never more than half of what the device has free right now: the buffer is an optimisation and must not be what
		// makes a later, larger batch of this context (or of another worker on the same device) fail its mandatory buffers
		{ size_t fr = 0, tot = 0; if ( hipMemGetInfo(&fr,&tot) == hipSuccess ) { uint64_t const lim = (static_cast<uint64_t>(fr)/2) / (static_cast<uint64_t>(c->handwords)*8ull); if ( cap > lim ) cap = lim; } else (void)hipGetLastError(); }
		while ( cap && c->d_hand.ensure(static_cast<size_t>(cap)*c->handwords) != hipSuccess ) { (void)hipGetLastError(); cap = cap > 65536 ? cap/2 : 0; }
		c->handcap = static_cast<uint32_t>(cap); c->handwant = cap;
	}
	++c->nruns;
	HIPCHK(hipMemsetAsync(c->d_err.p,0,4*sizeof(uint32_t),s));
	HIPCHK(hipEventRecord(c->ev[0],s));
	if ( BP.ovl.size() )
	{
		PrepBatch PB; PB.ovl = c->d_ovl.p; PB.novl = BP.ovl.size(); PB.trace = c->d_trace.p; PB.trace_bytes = c->trace_bytes; PB.blk_ovl = c->d_blk_ovl.p; PB.blk_b0 = c->d_blk_b0.p;
		hipLaunchKernelGGL(k_prep,dim3((BP.ovl.size()+255)/256),dim3(256),0,s,PB);
	}
	if ( BP.nblocks )
	{
		TraceBatch TB;
		TB.P = c->P; TB.bps = c->d_bps.p; TB.boff = c->d_boff.p; TB.rlen = c->d_rlen.p;
		TB.piles = c->d_piles.p; TB.ovl = c->d_ovl.p; TB.ovl_pile = c->d_ovl_pile.p; TB.trace = c->d_trace.p;
		TB.blk_ovl = c->d_blk_ovl.p; TB.blk_b0 = c->d_blk_b0.p; TB.nblocks = BP.nblocks; TB.wt_b = c->d_wt_b.p; TB.wt_e = c->d_wt_e.p;
		TB.maxcols = BP.maxcols; TB.trace_bytes = c->trace_bytes; TB.errflag = c->d_err.p + 2; TB.slab = c->d_trslab.p;
		TB.work = (c->env_trdyn && BP.nblocks < 0xFFFFFF00ull) ? c->d_work.p + 60 : static_cast<uint32_t *>(0);
		if ( TB.work ) HIPCHK(hipMemsetAsync(TB.work,0,sizeof(uint32_t),s));
		if ( c->tr_words == 2 ) hipLaunchKernelGGL(k_trace,dim3(c->tr_grid),dim3(64),c->tr_lds,s,TB);
		else if ( c->tr_words == 4 ) hipLaunchKernelGGL(k_trace_wide<4>,dim3(c->tr_grid),dim3(64),c->tr_lds,s,TB,c->tr_lanes);
		else hipLaunchKernelGGL(k_trace_wide<8>,dim3(c->tr_grid),dim3(64),c->tr_lds,s,TB,c->tr_lanes);
	}
	mark("prep + trace");
	HIPCHK(hipEventRecord(c->ev[1],s));
	if ( BP.nwindows )
	{
		WindowBatch WB;
		WB.P = c->P; WB.T = c->T; WB.C = BP.caps; WB.bps = c->d_bps.p; WB.boff = c->d_boff.p; WB.rlen = c->d_rlen.p;
		WB.piles = c->d_piles.p; WB.npiles = BP.piles.size(); WB.ovl = c->d_ovl.p; WB.wt_b = c->d_wt_b.p; WB.wt_e = c->d_wt_e.p;
		WB.nwindows = BP.nwindows; WB.wrec = c->d_wrec.p; WB.wout = c->d_wout.p; WB.arena = c->d_arena.p; WB.prof = c->d_prof.p;
		WB.pregen = 0;
		// no LDS tier usable (DACC_TIERS=0 or a model table no tier's overlay holds): everything runs in the generic engine on
		// the main stream; the pre-scan / second stream would hand the same windows to two kernels
		bool const anytier = c->tier_ok[0] || c->tier_ok[1] || c->tier_ok[2];
		c->tier0_ran = false; c->tier7_ran = false; c->tier10_ran = false;
		if ( c->usefast && anytier )
		{
			// windows only the generic engine can run (a string longer than 64 bases): found by a scan of the window tables and
			// started on the second stream now, concurrently with all LDS tiers
			HIPCHK(hipMemsetAsync(c->d_pregen.p,0,((BP.nwindows+31)/32+1)*sizeof(uint32_t),s));
			HIPCHK(hipMemsetAsync(c->d_pregen2.p,0,((BP.nwindows+31)/32+1)*sizeof(uint32_t),s));
			HIPCHK(hipMemsetAsync(c->d_pregenlist.p,0,sizeof(uint32_t),s));
			for ( int i = 0; i < 3; ++i ) HIPCHK(hipMemsetAsync(c->d_retry[i].p,0,sizeof(uint32_t),s));
			if ( BP.ovl.size() )
			{
				hipLaunchKernelGGL(k_prescan,dim3((BP.ovl.size()+3)/4),dim3(256),0,s,c->d_ovl.p,static_cast<uint64_t>(BP.ovl.size()),c->d_ovl_pile.p,c->d_piles.p,c->d_wt_b.p,c->d_wt_e.p,c->d_pregen.p,c->d_pregen2.p,c->widetier ? 128u : 64u);
				// windows with a string of 65 ... 128 bases join the first slot's hand-overs when the second slot's tier holds such strings (tiers 6 / 3;
				// a deep batch's tier 2 does not: it hands them on to tier 3 at its length check) and the first slot runs at all
				bool const slot1 = c->tier_ok[0] && (c->tier_ok[1] || c->tier_ok[2]) && c->env_long128;
				hipLaunchKernelGGL(k_prescan_lists,dim3(((BP.nwindows+31)/32+255)/256),dim3(256),0,s,c->d_pregen.p,c->d_pregen2.p,static_cast<uint64_t>(BP.nwindows),c->d_pregenlist.p,
					slot1 ? c->d_retry[0].p : static_cast<uint32_t *>(0),c->d_wout.p);
			}
			HIPCHK(hipEventRecord(c->evPrescan,s));
			FastBatch FL; FL.W = WB; FL.W.arena = c->d_arena2.p; FL.W.prof = 0; FL.W.pregen = 0; FL.F = BP.ftierL; FL.dpsq_vst = c->d_vst.p; FL.retry = 0; FL.gearly = 0; FL.gslab = 0; FL.gstride = 0; FL.tab32 = c->d_tab32.p; FL.hand = 0; FL.handctr = 0; FL.handcap = 0; FL.handwords = 0;
#if defined(DACC_LEDGER)
			FL.ledger = 0;
#endif
			if ( !c->tierL_ok ) FL.F.ldsbytes = 0;
			// a launch the device refuses (the LDS of a whole CU) falls back to the generic engine alone, for good
			// (round 5) The length of a list is fetched before its launch and the grid follows it: an empty list -- the rule at the default
			// window -- is not launched at all, a list of three windows asks for three CUs' worth of LDS instead of 256.  Rounds 1-4 queued
			// the full grid twice per pass; its workgroups need the LDS of a whole CU each and sat in the queue until the tier in front retired
			// (939 ms of "duration" for one window in the round-4 kernel table).
			// (round 6) The 4 byte copy and the wait run on the SECOND stream, behind the event of the kernel that fills the
			// list, and only after every tier and the generic kernel have been queued on the main stream: round 5 synchronised the main
			// stream in the middle of the chain, which kept the host -- and a second context's batches on the same device -- out of the
			// queue until the first tier had retired.
			auto const launchLong = [&](uint32_t const * const lst, uint32_t & nlist, hipEvent_t const filled) -> int
			{
				nlist = 0;
				HIPCHK(hipStreamWaitEvent(c->stream2,filled,0));
				HIPCHK(hipMemcpyAsync(&nlist,lst,sizeof(uint32_t),hipMemcpyDeviceToHost,c->stream2));
				HIPCHK(hipStreamSynchronize(c->stream2));
				if ( !nlist ) return DACC_OK;
				uint32_t const grid = nlist < c->early_grid ? nlist : c->early_grid;
				(void)hipGetLastError();
				hipLaunchKernelGGL(k_window_long,dim3(grid),dim3(64),FL.F.ldsbytes,c->stream2,FL,c->d_err.p,lst);
				if ( FL.F.ldsbytes && hipGetLastError() != hipSuccess )
				{
					c->tierL_ok = 0; FL.F.ldsbytes = 0;
					hipLaunchKernelGGL(k_window_long,dim3(grid),dim3(64),0,c->stream2,FL,c->d_err.p,lst);
				}
				return DACC_OK;
			};
			c->nlong[0] = c->nlong[1] = 0;
			WB.pregen = c->d_pregen.p;
			HIPCHK(hipMemsetAsync(c->d_gearly.p,0,sizeof(uint32_t),s));
			HIPCHK(hipMemsetAsync(c->d_handctr.p,0,sizeof(uint32_t),s));
			bool early = false;
			HIPCHK(hipMemsetAsync(c->d_work.p,0,64*sizeof(uint32_t),s));
			// capacity tiers: every tier takes the windows the previous one handed over (list = 0: all windows)
			uint32_t const * list = 0;
			for ( int t = 0; t < 3; ++t )
			{
				if ( c->tier_ok[t] )
				{
					FastBatch FB; FB.W = WB; FB.F = BP.ftier[t]; FB.dpsq_vst = c->d_vst.p; FB.retry = c->d_retry[t].p;
					// the first slot's tiers skip every window with a string of more than 64 bases; the later ones only those the second stream has
					// (without the 128-base route: all of them, as in rounds 3-5)
					if ( t > 0 && c->tier_ok[0] && c->env_long128 ) FB.W.pregen = c->d_pregen2.p;
					if ( c->widetier ) { FB.W.pregen = c->d_pregen2.p; FB.hand = 0; }      // (its hand-overs go to the generic engine, which sorts for itself)
					FB.gslab = c->d_gslab.p; FB.gstride = c->gstride[t]; FB.tab32 = c->d_tab32.p;
					FB.hand = c->handcap ? c->d_hand.p : static_cast<uint64_t *>(0); FB.handctr = c->d_handctr.p; FB.handcap = c->handcap; FB.handwords = c->handwords;
#if defined(DACC_LEDGER)
					{ char const * lm = getenv("DACC_LEDGER_MASK"); FB.ledger = lm ? static_cast<uint32_t>(strtoul(lm,0,0)) : 0u; }      // (scripts/ledger.py)
#endif
					// only the first tier feeds the early generic list (its kernel reads the list once, right after that tier): a
					// window that reaches a later tier first (mao beyond the earlier tier) and turns out to be generic-only takes
					// the ordinary hand-over chain to the generic kernel at the end
					FB.gearly = early ? static_cast<uint32_t *>(0) : c->d_gearly.p;
					uint32_t * const work = (c->sched&1) ? c->d_work.p+8*t : static_cast<uint32_t *>(0);
					if ( t == 0 && BP.deep ) hipLaunchKernelGGL(k_window_fast<4>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( t == 0 && c->tier0_ok )
					{
						// size classes: the small windows run in tier 0 (8 wavefronts per CU), the middle class and tier 0's hand-overs in tier 7
						// (7 per CU), tier 7's hand-overs and all other windows in tier 1 (6 per CU)
						HIPCHK(hipMemsetAsync(c->d_small.p,0,sizeof(uint32_t),s)); HIPCHK(hipMemsetAsync(c->d_big.p,0,sizeof(uint32_t),s));
						// (tier 7 switches itself off for the rest of a context's batches when it hands on more than a fifth of what it runs: on the
						// ONT mix at k = 10 / 12 a third / a quarter of its windows overflow it late, on their stretches and pools, and the class costs
						// 2.6 / 1.4 % instead of gaining the 3 % it gains at k = 14 and 16 and on config 2, profiles/r06j; results do not depend on it)
						bool const use7 = c->tier7_ok && !c->tier7_adapt_off;
						if ( use7 ) HIPCHK(hipMemsetAsync(c->d_mid.p,0,sizeof(uint32_t),s));
						uint32_t * const midlist = use7 ? c->d_mid.p : c->d_big.p;
						hipLaunchKernelGGL(k_classify,dim3((BP.nwindows+255)/256),dim3(256),0,s,WB,c->d_small.p,midlist,c->d_big.p,c->env_t0inst,use7 ? c->env_t7inst : 0u);
						// (what the pre-pass itself put on the middle and the big list, before the tiers' hand-overs join them: for the counters of dacc_timing)
						if ( use7 ) { HIPCHK(hipMemcpyAsync(c->d_work.p+56,c->d_mid.p,sizeof(uint32_t),hipMemcpyDeviceToDevice,s)); HIPCHK(hipMemcpyAsync(c->d_work.p+57,c->d_big.p,sizeof(uint32_t),hipMemcpyDeviceToDevice,s)); }
						FastBatch F0 = FB; F0.F = BP.ftier0; F0.retry = midlist; F0.gstride = c->gstride0;
						hipLaunchKernelGGL(k_window_fast<0>,dim3(c->tier0_grid),dim3(64),F0.F.ldsbytes,s,F0,static_cast<uint32_t const *>(c->d_small.p),(c->sched&1) ? c->d_work.p+40 : static_cast<uint32_t *>(0));
						HIPCHK(hipEventRecord(c->evT0,s)); c->tier0_ran = true;
						if ( use7 )
						{
							FastBatch F7 = FB; F7.F = BP.ftier7; F7.retry = c->d_big.p; F7.gstride = c->gstride7;
							hipLaunchKernelGGL(k_window_fast<7>,dim3(c->tier7_grid),dim3(64),F7.F.ldsbytes,s,F7,static_cast<uint32_t const *>(c->d_mid.p),(c->sched&1) ? c->d_work.p+48 : static_cast<uint32_t *>(0));
							HIPCHK(hipEventRecord(c->evT7,s)); c->tier7_ran = true;
						}
						hipLaunchKernelGGL(k_window_fast<1>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,static_cast<uint32_t const *>(c->d_big.p),work);
					}
					else if ( t == 0 ) hipLaunchKernelGGL(k_window_fast<1>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( t == 1 && c->widetier ) hipLaunchKernelGGL(k_window_fast<8>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( t == 1 && BP.deep ) hipLaunchKernelGGL(k_window_fast<2>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( t == 1 ) hipLaunchKernelGGL(k_window_fast<6>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( c->widetier ) hipLaunchKernelGGL(k_window_fast<9>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					else if ( c->tier10_ok && list )
					{
						// (round 6) the dense-graph tier in front of tier 3: two wavefronts per CU with 16 bit path ids take what the second slot
						// handed on; tier 3 (one per CU) gets what overflows them
						HIPCHK(hipMemsetAsync(c->d_dense.p,0,sizeof(uint32_t),s));
						FastBatch FD = FB; FD.F = BP.ftierD; FD.retry = c->d_dense.p; FD.gstride = c->gstride10;
						if ( BP.deep ) hipLaunchKernelGGL(k_window_fast<11>,dim3(c->tier10_grid),dim3(64),FD.F.ldsbytes,s,FD,list,(c->sched&1) ? c->d_work.p+24 : static_cast<uint32_t *>(0));      // (deep batches: tier 11)
						else hipLaunchKernelGGL(k_window_fast<10>,dim3(c->tier10_grid),dim3(64),FD.F.ldsbytes,s,FD,list,(c->sched&1) ? c->d_work.p+24 : static_cast<uint32_t *>(0));
						HIPCHK(hipEventRecord(c->evT10,s)); c->tier10_ran = true;
						hipLaunchKernelGGL(k_window_fast<3>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,static_cast<uint32_t const *>(c->d_dense.p),work);
					}
					else hipLaunchKernelGGL(k_window_fast<3>,dim3(c->tier_grid[t]),dim3(64),FB.F.ldsbytes,s,FB,list,work);
					list = c->d_retry[t].p;
					if ( !early )
					{
						// the first tier has seen every window: what only the generic engine can run starts now, on its own
						// (high priority) stream and arena, while the remaining tiers run
						early = true;
						HIPCHK(hipEventRecord(c->evFirstTier,s));
					}
				}
				hipEventRecord(c->evtier[t],s);
			}
			if ( c->env_dbgretry && list )
			{
				// debugging: which windows did the last LDS tier hand on, and why (flags of its FFAIL)
				HIPCHK(hipStreamSynchronize(s));
				uint32_t n = 0; HIPCHK(hipMemcpy(&n,list,sizeof(uint32_t),hipMemcpyDeviceToHost));
				std::vector<uint32_t> idx(n); if ( n ) HIPCHK(hipMemcpy(idx.data(),list+1,n*sizeof(uint32_t),hipMemcpyDeviceToHost));
				c->retry_flags.clear();
				for ( uint32_t i = 0; i < n; ++i ) { WindowOut o; HIPCHK(hipMemcpy(&o,c->d_wout.p+idx[i],sizeof(WindowOut),hipMemcpyDeviceToHost)); c->retry_flags.push_back(idx[i]); c->retry_flags.push_back(o.flags); c->retry_flags.push_back(o.mao); c->retry_flags.push_back(static_cast<uint32_t>(o.filterfreq)); }
			}
			// what is left (rare shapes) goes through the generic engine
			hipLaunchKernelGGL(k_window,dim3(c->retry_grid),dim3(64),0,s,WB,c->d_err.p,list,(c->sched&2) ? c->d_work.p+32 : static_cast<uint32_t *>(0));
			// the second stream: the pre-scan's list (a string of more than 64 bases) as soon as the pre-scan is done, the first tier's
			// generic-only windows behind the first tier; the main stream's queue is full by now
			{ int const rc = launchLong(static_cast<uint32_t const *>(c->d_pregenlist.p),c->nlong[0],c->evPrescan); if ( rc ) return rc; }
			if ( early ) { int const rc = launchLong(static_cast<uint32_t const *>(c->d_gearly.p),c->nlong[1],c->evFirstTier); if ( rc ) return rc; }
			// everything the second stream ran (pre-scan list, first tier's generic-only windows) must be done before the vote
			HIPCHK(hipEventRecord(c->evEarlyGeneric,c->stream2));
			HIPCHK(hipStreamWaitEvent(s,c->evEarlyGeneric,0));
		}
		else
			{ HIPCHK(hipMemsetAsync(c->d_work.p,0,64*sizeof(uint32_t),s)); hipLaunchKernelGGL(k_window,dim3(c->win_grid),dim3(64),0,s,WB,c->d_err.p,static_cast<uint32_t const *>(0),(c->sched&2) ? c->d_work.p+32 : static_cast<uint32_t *>(0)); mark("k_window (all windows)"); }
	}
	HIPCHK(hipEventRecord(c->ev[2],s));
	uint32_t herr[4] = {0,0,0,0};
	size_t const symbytes = 2*BP.npos + 64*BP.piles.size() + 64;
	c->h_nfrag.resize(BP.piles.size()); c->h_frags.resize(BP.nfragslots+1); HIPCHK(c->h_outsym.ensure(symbytes));
	auto voteAndFetch = [&]() -> int
	{
		HIPCHK(hipMemsetAsync(c->d_pilebad.p,0,BP.piles.size()+1,s));
		if ( BP.nwindows ) hipLaunchKernelGGL(k_check_done,dim3((BP.nwindows+255)/256),dim3(256),0,s,c->d_wout.p,BP.nwindows,c->d_err.p+3,c->d_piles.p,static_cast<uint32_t>(BP.piles.size()),c->d_pilebad.p);
		if ( BP.piles.size() )
		{
			VoteBatch VB;
			VB.P = c->P; VB.bps = c->d_bps.p; VB.boff = c->d_boff.p; VB.rlen = c->d_rlen.p; VB.piles = c->d_piles.p; VB.npiles = BP.piles.size();
			VB.wrec = c->d_wrec.p; VB.has = c->d_has.p; VB.ld0 = c->d_ld0.p; VB.oc = c->d_oc.p; VB.ocs = c->d_ocs.p; VB.outsym = c->d_outsym.p;
			VB.frags = c->d_frags.p; VB.fragbase = c->d_fragbase.p; VB.nfrag = c->d_nfrag.p; VB.errflag = c->d_err.p + 1; VB.pilebad = c->d_pilebad.p;
			mark("k_check_done");
			hipLaunchKernelGGL(k_vote,dim3(BP.piles.size()),dim3(256),0,s,VB);
			mark("k_vote");
		}
		HIPCHK(hipEventRecord(c->ev[3],s));
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(herr,c->d_err.p,sizeof(herr),hipMemcpyDeviceToHost,s));
		if ( BP.piles.size() )
		{
			HIPCHK(hipMemcpyAsync(c->h_nfrag.data(),c->d_nfrag.p,BP.piles.size()*sizeof(uint32_t),hipMemcpyDeviceToHost,s));
			HIPCHK(hipMemcpyAsync(c->h_frags.data(),c->d_frags.p,BP.nfragslots*sizeof(VoteFragment),hipMemcpyDeviceToHost,s));
			HIPCHK(hipMemcpyAsync(c->h_outsym.p,c->d_outsym.p,symbytes,hipMemcpyDeviceToHost,s));
		}
		HIPCHK(hipEventRecord(c->ev[4],s));
		HIPCHK(hipStreamSynchronize(s));
		return DACC_OK;
	};
	{ int const rc = voteAndFetch(); if ( rc ) return rc; }
	for ( int i = 0; i < 3; ++i ) c->tier_out[i] = 0;
	if ( c->usefast && BP.nwindows ) for ( int i = 0; i < 3; ++i ) if ( c->tier_ok[i] ) HIPCHK(hipMemcpy(&c->tier_out[i],c->d_retry[i].p,sizeof(uint32_t),hipMemcpyDeviceToHost));
	c->tier10_out = 0;
	if ( c->tier10_ran ) HIPCHK(hipMemcpy(&c->tier10_out,c->d_dense.p,sizeof(uint32_t),hipMemcpyDeviceToHost));
	c->timing.tier0_in = 0; c->timing.tier0_out = 0; c->timing.tier7_in = 0; c->timing.tier7_out = 0;
	if ( c->usefast && BP.nwindows && (c->tier_ok[0] || c->tier_ok[1] || c->tier_ok[2]) )
	{
		// length of the two lists of the second stream (pre-scan, first tier's generic-only windows), fetched before their launches
		uint32_t const n1 = c->nlong[0], n2 = c->nlong[1];
		c->timing.long_windows = n1 + n2; c->timing.long_first_tier = n2;
		c->timing.tier7_in = 0; c->timing.tier7_out = 0;
		if ( c->tier0_ran )
		{
			// size classes: the pre-pass sent nsmall windows to tier 0, nmid0 to the middle list and nwindows - nsmall - nmid0 - n1 to the big
			// list; tier 0's hand-overs have joined the middle list since (the big list without tier 7), tier 7's the big list
			uint32_t nsmall = 0, nbig = 0, nmid = 0;
			HIPCHK(hipMemcpy(&nsmall,c->d_small.p,sizeof(uint32_t),hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&nbig,c->d_big.p,sizeof(uint32_t),hipMemcpyDeviceToHost));
			if ( c->tier7_ran ) HIPCHK(hipMemcpy(&nmid,c->d_mid.p,sizeof(uint32_t),hipMemcpyDeviceToHost));
			c->timing.tier0_in = nsmall;
			if ( !c->tier7_ran )
			{
				uint64_t const big0 = BP.nwindows - std::min<uint64_t>(BP.nwindows,static_cast<uint64_t>(nsmall) + n1);
				c->timing.tier0_out = nbig > big0 ? static_cast<uint32_t>(nbig - big0) : 0u;
			}
			else
			{
				// nmid = what the pre-pass put on the middle list + tier 0's hand-overs, nbig = the pre-pass's big windows + tier 7's hand-overs
				uint32_t pre[2] = {0,0};
				HIPCHK(hipMemcpy(pre,c->d_work.p+56,2*sizeof(uint32_t),hipMemcpyDeviceToHost));
				uint32_t const t7out = nbig > pre[1] ? nbig - pre[1] : 0u;
				c->timing.tier0_out = nmid > pre[0] ? nmid - pre[0] : 0u;
				c->timing.tier7_in = nmid; c->timing.tier7_out = t7out;
				if ( c->env_t7adapt && nmid >= 4096 && static_cast<uint64_t>(t7out)*5 > nmid ) c->tier7_adapt_off = true;
			}
		}
	}
	// a window the generic engine could not hold (dense graph at small k): grow its scratch capacities and run those
	// windows again, then the vote (rare; the capacities stay grown for the rest of the batch geometry)
	for ( int attempt = 0; herr[0] && !herr[1] && !herr[2] && attempt < 3; ++attempt )
	{
		growArenaCaps(BP.caps,c->par.w);
		Arena Atmp; BP.caps.bytes = arena_carve(Atmp,0,BP.caps,c->par.w);
		uint32_t const g = boundByArena(c->retry_grid < 64 ? c->retry_grid : 64,BP.caps.bytes,4);
		c->retry_grid = g; c->win_grid = g; if ( c->early_grid > g ) c->early_grid = g;     // later launches of this batch use the grown arenas
		HIPCHK(c->d_arena.ensure(static_cast<size_t>(g)*BP.caps.bytes));
		if ( c->usefast ) HIPCHK(c->d_arena2.ensure(static_cast<size_t>(c->early_grid)*BP.caps.bytes));
		HIPCHK(c->d_gearly.ensure(BP.nwindows+2)); HIPCHK(c->d_pregenlist.ensure(BP.nwindows+2)); HIPCHK(c->d_pregen.ensure((BP.nwindows+31)/32+2)); HIPCHK(c->d_pregen2.ensure((BP.nwindows+31)/32+2));
		HIPCHK(hipMemsetAsync(c->d_err.p,0,4*sizeof(uint32_t),s));
		HIPCHK(hipMemsetAsync(c->d_gearly.p,0,sizeof(uint32_t),s));
		hipLaunchKernelGGL(k_collect_overflow,dim3((BP.nwindows+255)/256),dim3(256),0,s,c->d_wout.p,BP.nwindows,c->d_gearly.p);
		mark("k_collect_overflow");
		if ( dbgsync ) std::fprintf(stderr,"[dacc] scratch retry %d: grid %u, arena %llu bytes per wavefront\n",attempt,g,static_cast<unsigned long long>(BP.caps.bytes));
		WindowBatch WB;
		WB.P = c->P; WB.T = c->T; WB.C = BP.caps; WB.bps = c->d_bps.p; WB.boff = c->d_boff.p; WB.rlen = c->d_rlen.p;
		WB.piles = c->d_piles.p; WB.npiles = BP.piles.size(); WB.ovl = c->d_ovl.p; WB.wt_b = c->d_wt_b.p; WB.wt_e = c->d_wt_e.p;
		WB.nwindows = BP.nwindows; WB.wrec = c->d_wrec.p; WB.wout = c->d_wout.p; WB.arena = c->d_arena.p; WB.prof = 0; WB.pregen = 0;
		hipLaunchKernelGGL(k_window,dim3(g),dim3(64),0,s,WB,c->d_err.p,static_cast<uint32_t const *>(c->d_gearly.p),static_cast<uint32_t *>(0));
		mark("k_window (scratch retry)");
		int const rc = voteAndFetch(); if ( rc ) return rc;
	}
	c->pile_status = BP.pile_status; c->pile_errors = BP.pile_errors;
	if ( herr[0] && BP.piles.size() )
	{
		// windows the generic engine could not hold even with grown scratch: their piles were dropped by the vote
		std::vector<uint8_t> bad(BP.piles.size());
		HIPCHK(hipMemcpy(bad.data(),c->d_pilebad.p,bad.size(),hipMemcpyDeviceToHost));
		for ( size_t pi = 0; pi < bad.size(); ++pi )
			if ( bad[pi] )
			{
				c->pile_status[pi] = DACC_ENOTSUP;
				if ( c->pile_errors.size() < 64 ) c->pile_errors.push_back("read " + std::to_string(BP.piles[pi].aread) + ": window kernel scratch capacity exceeded (depth / graph size), read skipped");
			}
	}
	if ( herr[3] ) { c->err = "internal error: a window was handed on between engines and never processed"; return DACC_EHIP; }
	if ( herr[1] ) { c->err = "vote kernel capacity exceeded"; return DACC_ENOTSUP; }
	if ( herr[2] ) { c->err = "trace kernel capacity exceeded (a tspace block longer than the column vectors or a B span beyond the column store)"; return DACC_ENOTSUP; }
	c->frags.clear(); c->bases.clear();
	uint64_t nbases = 0;
	for ( uint64_t pi = 0; pi < BP.piles.size(); ++pi )
		for ( uint32_t f = 0; f < c->h_nfrag[pi]; ++f ) nbases += c->h_frags[BP.fragbase[pi]+f].len;
	c->bases.reserve(nbases);
	for ( uint64_t pi = 0; pi < BP.piles.size(); ++pi )
		for ( uint32_t f = 0; f < c->h_nfrag[pi]; ++f )
		{
			VoteFragment const & F = c->h_frags[BP.fragbase[pi]+f];
			dacc_fragment g; g.aread = BP.piles[pi].aread; g.first = F.first; g.last = F.last; g.len = F.len; g.seq_off = c->bases.size();
			c->bases.append(reinterpret_cast<char const *>(c->h_outsym.p)+F.off,F.len);
			c->frags.push_back(g);
		}
	float ms = 0;
	hipEventElapsedTime(&ms,c->ev[0],c->ev[1]); c->timing.trace_ms = ms;
	hipEventElapsedTime(&ms,c->ev[1],c->ev[2]); c->timing.window_ms = ms;
	for ( int i = 0; i < 3; ++i ) { c->timing.tier_ms[i] = 0; c->timing.tier_out[i] = c->tier_out[i]; }
	if ( c->usefast && BP.nwindows ) for ( int i = 0; i < 3; ++i ) { hipEventElapsedTime(&ms,i ? c->evtier[i-1] : c->ev[1],c->evtier[i]); c->timing.tier_ms[i] = ms; }
	c->timing.tier0_ms = 0;
	if ( c->tier0_ran ) { hipEventElapsedTime(&ms,c->ev[1],c->evT0); c->timing.tier0_ms = ms; }
	c->timing.tier7_ms = 0; c->timing.pad_ = 0;
	if ( c->tier7_ran ) { hipEventElapsedTime(&ms,c->evT0,c->evT7); c->timing.tier7_ms = ms; }
	c->timing.tier10_ms = 0; c->timing.tier10_out = c->tier10_out; c->timing.tier10_ran = c->tier10_ran ? 1u : 0u;
	if ( c->tier10_ran ) { hipEventElapsedTime(&ms,c->evtier[1],c->evT10); c->timing.tier10_ms = ms; }
	hipEventElapsedTime(&ms,c->ev[2],c->ev[3]); c->timing.vote_ms = ms;
	hipEventElapsedTime(&ms,c->ev[3],c->ev[4]); c->timing.d2h_ms = ms;
	hipEventElapsedTime(&ms,c->ev[0],c->ev[3]); c->timing.total_ms = ms;
	c->timing.first_tier = (c->usefast && BP.deep) ? 4u : 1u;
	c->timing.nwindows = BP.nwindows; c->timing.nblocks = BP.nblocks; c->timing.algo_bytes = BP.algo_bytes + nbases;
	return DACC_OK;
}

static int dacc_submit_piles_body(dacc_ctx * c, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl,
	void const * trace, uint64_t ntrace, int trace_bytes)
{
	if ( !c ) return DACC_EINVAL;
	if ( !c->haveprofile || !c->havedb ) { c->err = "dacc_set_error_profile and dacc_load_db must precede dacc_submit_piles"; return DACC_ESTATE; }
	if ( (npiles && !piles) || (novl && (!ovl || !trace)) ) return DACC_EINVAL;
	hipSetDevice(c->device);
	c->havebatch = false;
	BatchPlan & BP = c->BP;
	int rc = BP.plan(c->par,piles,npiles,ovl,novl,trace,ntrace,trace_bytes,c->h_rlen.data(),c->h_rlen.size(),c->err,c->H.nrows,c->H.nsup);
	if ( rc ) return rc;
	hipStream_t const s = c->stream;
	HIPCHK(c->d_err.ensure(4));
	HIPCHK(c->d_prof.ensure(DACC_PROFW*4096)); HIPCHK(hipMemsetAsync(c->d_prof.p,0,DACC_PROFW*4096*sizeof(uint64_t),s));   // one row of counters per workgroup (no contention)
	HIPCHK(hipEventRecord(c->ev[5],s));
	if ( (rc = upload(c,c->d_piles,BP.piles.data(),BP.piles.size())) ) return rc;
	if ( (rc = upload(c,c->d_ovl,BP.ovl.data(),BP.ovl.size())) ) return rc;
	if ( (rc = upload(c,c->d_ovl_pile,BP.ovl_pile.data(),BP.ovl_pile.size())) ) return rc;
	if ( (rc = upload(c,c->d_trace,static_cast<uint8_t const *>(trace),ntrace*trace_bytes)) ) return rc;
	if ( (rc = upload(c,c->d_fragbase,BP.fragbase.data(),BP.fragbase.size())) ) return rc;
	HIPCHK(c->d_blk_ovl.ensure(BP.nblocks+1)); HIPCHK(c->d_blk_b0.ensure(BP.nblocks+1));
	HIPCHK(c->d_wt_b.ensure(BP.nwt+1)); HIPCHK(c->d_wt_e.ensure(BP.nwt+1));
	// trace kernel geometry: one wavefront per workgroup, as many workgroups per CU as their LDS column stores allow,
	// a few rounds of blocks per workgroup
	{
		c->trace_bytes = trace_bytes;
		// two word columns for tspace <= 128 while every lane's column store fits (blocks of up to 928 B bases: always with
		// one byte trace values); a batch with a longer block runs the four word kernel, which sizes its lanes to the LDS
		c->tr_words = (c->par.tspace <= 128 && BP.maxcols <= 928) ? 2 : (c->par.tspace <= 256 ? 4 : 8);
		c->tr_lanes = 64;
		if ( c->tr_words == 2 ) c->tr_lds = TRACE2_LDS;
		else
		{
			// wide blocks: as many lanes per wavefront as have room for their column checkpoints
			uint32_t const perlane = c->tr_words == 4 ? traceWideBytesPerLane<4>(BP.maxcols) : traceWideBytesPerLane<8>(BP.maxcols);
			while ( c->tr_lanes > 8 && static_cast<uint64_t>(c->tr_lanes)*perlane > 160*1024 ) c->tr_lanes >>= 1;
			c->tr_lds = c->tr_lanes*perlane;
		}
		if ( c->tr_lds > 160*1024 ) { c->err = "a trace block spans too many B bases for the trace kernel's LDS column store"; return DACC_ENOTSUP; }
		if ( c->tr_lds > 64*1024 )
		{
			if ( c->tr_words == 2 ) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trace),hipFuncAttributeMaxDynamicSharedMemorySize,c->tr_lds));
			else if ( c->tr_words == 4 ) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trace_wide<4>),hipFuncAttributeMaxDynamicSharedMemorySize,c->tr_lds));
			else HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trace_wide<8>),hipFuncAttributeMaxDynamicSharedMemorySize,c->tr_lds));
		}
		// resident workgroups per CU: LDS is handed out in granules of 1280 bytes (measured, round 5); k_trace holds 92 registers = 5 wavefronts per SIMD
		uint64_t percu = (160*1024) / ((((c->tr_lds ? c->tr_lds : 1) + 1279u)/1280u)*1280u); if ( percu > (c->tr_words == 2 ? 20u : 8u) ) percu = (c->tr_words == 2 ? 20u : 8u); if ( percu < 1 ) percu = 1;
		if ( char const * e = getenv("DACC_TR_PERCU") ) { uint64_t const v = strtoull(e,0,10); if ( v >= 1 && v <= 32 ) percu = v; }
		// two word kernel: exactly the resident workgroups (grid stride over the blocks), so that the checkpoint slabs
		// (53 KB per workgroup at a B span of 160) stay in the L2 / Infinity Cache
		uint64_t g = (BP.nblocks+c->tr_lanes-1)/c->tr_lanes, gmax = 256*percu*(c->tr_words == 2 ? 1 : 4);
		if ( g > gmax ) g = gmax;
		if ( g < 1 ) g = 1;
		c->tr_grid = g;
		if ( c->tr_words == 2 )
		{
			// the checkpoint slabs of the resident workgroups are meant to stay in the Infinity Cache (256 MB): one block with a
			// long B span makes every workgroup's slab large (300 KB at the 928 columns this kernel takes), so the grid gives way
			// before the slabs outgrow 192 MB, down to one workgroup per CU
			uint64_t const slabbytes = static_cast<uint64_t>(traceSlabWords(BP.maxcols))*sizeof(uint64_t);
			while ( g > 256 && g*slabbytes > (192ull<<20) ) g -= 256;
			c->tr_grid = g;
			HIPCHK(c->d_trslab.ensure(static_cast<size_t>(g)*traceSlabWords(BP.maxcols)));
		}
	}
	// window kernel geometry + arenas
	Arena Atmp; BP.caps.bytes = arena_carve(Atmp,0,BP.caps,c->par.w);
	uint64_t wg = ((BP.nwindows+7)/8)*8;
	uint64_t const maxwg = 256*8;
	if ( wg > maxwg ) wg = maxwg;
	if ( wg < 8 ) wg = 8;
	{
		c->usefast = !c->env_nofast;
		c->sched = c->env_sched;
		for ( size_t i = 0; i < c->H.dpsq_vst.size(); ++i ) if ( c->H.dpsq_vst[i] >> 32 ) c->usefast = 0; // table must fit 32 bits
		// (round 6) wide windows, w = 64 ... 127 (model table of up to 128 rows): tier 8 in the second slot and tier 9 in the third in front of the generic engine
		// (DACC_WIDE_TIER=0: the generic engine only, as in rounds 4-5; w = 128 always)
		c->widetier = BP.wide && c->env_widetier && c->H.nrows <= 128 && c->H.nsup <= FSUPCAPW;
		if ( !c->widetier && (c->H.nrows > 64 || c->H.nsup > FSUPCAP || c->par.w > 63) ) c->usefast = 0;
	}
	if ( c->usefast )
	{
		// LDS tiers: floor(160 KiB / ldsbytes) wavefronts per CU; a tier whose table overlay cannot hold the model table is skipped
		if ( c->env_lds_t1 > BP.ftier[0].ldsbytes && c->env_lds_t1 <= 160*1024 ) BP.ftier[0].ldsbytes = c->env_lds_t1;
		for ( int t = 0; t < 3; ++t )
		{
			FastCaps const & F = BP.ftier[t];
			c->tier_ok[t] = (static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= F.tabcap) && F.ldsbytes <= 160*1024;
			if ( !((c->env_tiers>>t)&1) ) c->tier_ok[t] = 0;
			if ( c->widetier && t == 0 ) c->tier_ok[t] = 0;
			uint64_t percu = (160*1024) / (F.ldsbytes ? F.ldsbytes : 1);
			if ( percu > 8 ) percu = 8;
			if ( percu < 1 ) percu = 1;
			uint64_t fg = ((BP.nwindows+7)/8)*8;
			if ( fg > 256*percu ) fg = 256*percu;
			if ( fg < 8 ) fg = 8;
			c->tier_grid[t] = fg;
			HIPCHK(c->d_retry[t].ensure(BP.nwindows+2));
			// gw tiers: one global slab per workgroup (weights, spill); the tiers run one after the other and share the buffer
			c->gstride[t] = F.gbytes;
			if ( F.gbytes ) HIPCHK(c->d_gslab.ensure(static_cast<size_t>(fg)*F.gbytes + 256));
		}
		{
			// tier 0 (size classes) in front of tier 1 of a shallow batch: DACC_TIERS bit 3 switches it off
			if ( c->env_lds_t0 > BP.ftier0.ldsbytes && c->env_lds_t0 <= 160*1024 ) BP.ftier0.ldsbytes = c->env_lds_t0;
			FastCaps const & F0 = BP.ftier0;
			c->tier0_ok = !BP.deep && c->tier_ok[0] && ((c->env_tiers>>3)&1) && F0.ldsbytes <= 160*1024;
			uint64_t percu0 = (160*1024) / (F0.ldsbytes ? F0.ldsbytes : 1); if ( percu0 > 8 ) percu0 = 8; if ( percu0 < 1 ) percu0 = 1;
			uint64_t fg0 = ((BP.nwindows+7)/8)*8; if ( fg0 > 256*percu0 ) fg0 = 256*percu0; if ( fg0 < 8 ) fg0 = 8;
			c->tier0_grid = static_cast<uint32_t>(fg0); c->gstride0 = F0.gbytes;
			if ( c->tier0_ok )
			{
				HIPCHK(c->d_small.ensure(BP.nwindows+2)); HIPCHK(c->d_big.ensure(BP.nwindows+2));
				HIPCHK(c->d_gslab.ensure(static_cast<size_t>(fg0)*F0.gbytes + 256));
			}
			// tier 7 (the middle size class, 7 wavefronts per CU) between them: DACC_TIERS bit 4 switches it off
			FastCaps const & F7 = BP.ftier7;
			c->tier7_ok = c->tier0_ok && ((c->env_tiers>>4)&1) && F7.ldsbytes <= 160*1024 && c->env_t7inst > c->env_t0inst;
			uint64_t percu7 = (160*1024) / (F7.ldsbytes ? F7.ldsbytes : 1); if ( percu7 > 8 ) percu7 = 8; if ( percu7 < 1 ) percu7 = 1;
			uint64_t fg7 = ((BP.nwindows+7)/8)*8; if ( fg7 > 256*percu7 ) fg7 = 256*percu7; if ( fg7 < 8 ) fg7 = 8;
			c->tier7_grid = static_cast<uint32_t>(fg7); c->gstride7 = F7.gbytes;
			if ( c->tier7_ok )
			{
				HIPCHK(c->d_mid.ensure(BP.nwindows+2));
				HIPCHK(c->d_gslab.ensure(static_cast<size_t>(fg7)*F7.gbytes + 256));
			}
			// tier 10 (the dense-graph tier, two wavefronts per CU) between the second slot's tier 6 and tier 3 of a shallow batch: DACC_DENSE_TIER=0 switches it off
			FastCaps const & FD = BP.ftierD;
			c->tier10_ok = !c->widetier && c->env_dense && c->tier_ok[1] && c->tier_ok[2] && FD.ldsbytes <= 160*1024 && (static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= FD.tabcap);
			uint64_t percuD = (160*1024) / (FD.ldsbytes ? FD.ldsbytes : 1); if ( percuD > 8 ) percuD = 8; if ( percuD < 1 ) percuD = 1;
			uint64_t fgD = ((BP.nwindows+7)/8)*8; if ( fgD > 256*percuD ) fgD = 256*percuD; if ( fgD < 8 ) fgD = 8;
			c->tier10_grid = static_cast<uint32_t>(fgD); c->gstride10 = FD.gbytes;
			if ( c->tier10_ok )
			{
				HIPCHK(c->d_dense.ensure(BP.nwindows+2));
				HIPCHK(c->d_gslab.ensure(static_cast<size_t>(fgD)*FD.gbytes + 256));
				if ( FD.ldsbytes > 64*1024 ) HIPCHK(hipFuncSetAttribute(BP.deep ? reinterpret_cast<const void *>(k_window_fast<11>) : reinterpret_cast<const void *>(k_window_fast<10>),hipFuncAttributeMaxDynamicSharedMemorySize,FD.ldsbytes));
			}
		}
		{
			// hand-over slots (sorted instances of a window that overflowed a tier's node table, picked up by the next tier): header +
			// the instance capacity of the tiers that hand on + their last k-mer lists; as many slots as a third of the windows, within
			// 16 GB (config 2: 1.6 M hand-overs of 10 M windows).  DACC_HAND=0 switches the mechanism off (every hand-over restarts).
			// The buffer is an optimisation for contexts that live: the driver clears device memory it hands out (16 GB = half a
			// second on some boxes), which a single pass does not earn back, so the FIRST pass of a context runs without it and the
			// buffer is allocated when the context is used again (runDevice).
			char const * he = getenv("DACC_HAND");
			c->handwords = (BP.deep ? 2048u + 128u : 1024u + 64u) + 4u;
			uint64_t cap = BP.nwindows/3 + 4096; uint64_t const maxcap = (16ull<<30) / (static_cast<uint64_t>(c->handwords)*8ull);
			if ( cap > maxcap ) cap = maxcap;
			if ( he && he[0] == '0' ) cap = 0;
			if ( c->par.klow != c->par.khigh ) cap = 0;
			if ( c->nohand ) cap = 0;      // a retry after an allocation failure: no optional buffer (dacc_drop_hand)
			if ( c->widetier ) cap = 0;    // a wide batch has one LDS tier: nothing to hand the sorted instances to
			c->handwant = cap;
			HIPCHK(c->d_handctr.ensure(4));
			if ( c->d_hand.cap < static_cast<size_t>(cap)*c->handwords ) c->handcap = static_cast<uint32_t>(c->d_hand.cap / c->handwords);     // what an earlier batch left
			else c->handcap = static_cast<uint32_t>(cap);
		}
		if ( BP.ftier[0].ldsbytes > 64*1024 ) HIPCHK(hipFuncSetAttribute(BP.deep ? reinterpret_cast<const void *>(k_window_fast<4>) : reinterpret_cast<const void *>(k_window_fast<1>),hipFuncAttributeMaxDynamicSharedMemorySize,BP.ftier[0].ldsbytes));
		c->tierL_ok = (static_cast<uint64_t>(c->H.nrows+1)*(c->H.nsup+1) <= BP.ftierL.tabcap) && BP.ftierL.ldsbytes <= 160*1024 && ((c->env_tiers>>2)&1);
		if ( c->tierL_ok && BP.ftierL.ldsbytes > 64*1024 ) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_window_long),hipFuncAttributeMaxDynamicSharedMemorySize,BP.ftierL.ldsbytes));
		if ( BP.ftier[1].ldsbytes > 64*1024 ) HIPCHK(hipFuncSetAttribute(c->widetier ? reinterpret_cast<const void *>(k_window_fast<8>) : (BP.deep ? reinterpret_cast<const void *>(k_window_fast<2>) : reinterpret_cast<const void *>(k_window_fast<6>)),hipFuncAttributeMaxDynamicSharedMemorySize,BP.ftier[1].ldsbytes));
		if ( BP.ftier[2].ldsbytes > 64*1024 ) HIPCHK(hipFuncSetAttribute(c->widetier ? reinterpret_cast<const void *>(k_window_fast<9>) : reinterpret_cast<const void *>(k_window_fast<3>),hipFuncAttributeMaxDynamicSharedMemorySize,BP.ftier[2].ldsbytes));
		c->retry_grid = boundByArena(wg < 512 ? wg : 512,BP.caps.bytes,8);
		c->win_grid = c->retry_grid;
		// generic engine on the second stream (windows with a string of more than 64 bases): a wavefront per window as far
		// as the arenas stay below 8 GB (the handful of such windows of config 2 took the generic engine up to 6.6 s each next to
		// the 8.2 s of the LDS tiers; tier 5 takes them now)
		c->early_grid = c->retry_grid < 256 ? c->retry_grid : 256;
		while ( c->early_grid > 16 && static_cast<uint64_t>(c->early_grid)*BP.caps.bytes > (8ull<<30) ) c->early_grid >>= 1;
		HIPCHK(c->d_gearly.ensure(BP.nwindows+2)); HIPCHK(c->d_pregenlist.ensure(BP.nwindows+2)); HIPCHK(c->d_pregen.ensure((BP.nwindows+31)/32+2)); HIPCHK(c->d_pregen2.ensure((BP.nwindows+31)/32+2));
		HIPCHK(c->d_arena2.ensure(static_cast<size_t>(c->early_grid)*BP.caps.bytes));
		HIPCHK(c->d_work.ensure(64));
		HIPCHK(c->d_arena.ensure(static_cast<size_t>(c->retry_grid)*BP.caps.bytes));
	}
	else
	{
		c->win_grid = boundByArena(wg,BP.caps.bytes,8);
		c->retry_grid = c->win_grid; c->early_grid = 0;
		for ( int t = 0; t < 3; ++t ) c->tier_ok[t] = 0;
		c->tier0_ok = false; c->tier7_ok = false; c->tierL_ok = 0;
		HIPCHK(c->d_work.ensure(64));
		HIPCHK(c->d_arena.ensure(static_cast<size_t>(c->win_grid)*BP.caps.bytes));
	}
	HIPCHK(c->d_wrec.ensure((BP.nwindows+1)*static_cast<size_t>(DACC_WREC_OF(c->par.w)))); HIPCHK(c->d_wout.ensure(BP.nwindows+1));
	HIPCHK(c->d_has.ensure(BP.npos+1)); HIPCHK(c->d_oc.ensure(BP.npos+1)); HIPCHK(c->d_ld0.ensure(BP.npos+1)); HIPCHK(c->d_ocs.ensure(BP.npos+1));
	HIPCHK(c->d_outsym.ensure(2*BP.npos + 64*BP.piles.size() + 64));
	HIPCHK(c->d_nfrag.ensure(BP.piles.size()+1)); HIPCHK(c->d_frags.ensure(BP.nfragslots+1)); HIPCHK(c->d_pilebad.ensure(BP.piles.size()+1));
	hipEvent_t h2dend; hipEventCreate(&h2dend); hipEventRecord(h2dend,s);
	rc = runDevice(c);
	float ms = 0; hipEventElapsedTime(&ms,c->ev[5],h2dend); c->timing.h2d_ms = ms; hipEventDestroy(h2dend);
	if ( rc ) return rc;
	c->havebatch = true;
	return DACC_OK;
}

static int dacc_rerun_resident_body(dacc_ctx * c)
{
	if ( !c ) return DACC_EINVAL;
	if ( !c->havebatch ) { c->err = "no resident batch"; return DACC_ESTATE; }
	hipSetDevice(c->device);
	return runDevice(c);
}

int dacc_collect(dacc_ctx * c, dacc_fragment const ** frags, uint64_t * nfrags, char const ** bases, uint64_t * nbases)
{
	if ( !c || !frags || !nfrags || !bases || !nbases ) return DACC_EINVAL;
	if ( !c->havebatch ) return DACC_ESTATE;
	*frags = c->frags.data(); *nfrags = c->frags.size(); *bases = c->bases.data(); *nbases = c->bases.size();
	return DACC_OK;
}

void dacc_release(dacc_ctx * c) { if ( c ) { c->frags.clear(); c->bases.clear(); } }

int dacc_last_timing(dacc_ctx * c, dacc_timing * t)
{
	if ( !c || !t ) return DACC_EINVAL;
	*t = c->timing;
	return DACC_OK;
}

// per-phase cycle counters of the window kernel (all zero unless built with -DDACC_PROFILE): a row of DACC_PROFW counters per
// workgroup; 0...31 the phases (scripts/prof_phases.py), 32...127 the fine sites of round 5 (site s: cycles at 32+s, visits at 80+s)
static int dacc_debug_profile_rows(dacc_ctx * c, uint64_t * out, int const from, int const n)
{
	if ( !c || !out ) return DACC_EINVAL;
	if ( !c->havebatch ) return DACC_ESTATE;
	hipSetDevice(c->device);
	{
		std::vector<uint64_t> H(DACC_PROFW*4096);
		HIPCHK(hipMemcpy(H.data(),c->d_prof.p,DACC_PROFW*4096*sizeof(uint64_t),hipMemcpyDeviceToHost));
		for ( int i = 0; i < n; ++i ) out[i] = 0;
		for ( int b = 0; b < 4096; ++b ) for ( int i = 0; i < n; ++i ) { uint64_t const v = H[DACC_PROFW*b+from+i]; if ( from+i == 29 ) out[i] = std::max(out[i],v); else out[i] += v; }
	}
	return DACC_OK;
}
int dacc_debug_profile(dacc_ctx * c, uint64_t * out32) { return dacc_debug_profile_rows(c,out32,0,32); }
int dacc_debug_profile_fine(dacc_ctx * c, uint64_t * out96) { return dacc_debug_profile_rows(c,out96,32,96); }

// DACC_DEBUG_RETRY=1: (window, flags, mao, filterfreq) of the windows the last LDS tier handed to the generic engine
int dacc_debug_retry(dacc_ctx * c, uint32_t * out, uint64_t cap, uint64_t * n)
{
	if ( !c || !n ) return DACC_EINVAL;
	*n = c->retry_flags.size();
	if ( out ) std::memcpy(out,c->retry_flags.data(),sizeof(uint32_t)*std::min<uint64_t>(cap,c->retry_flags.size()));
	return DACC_OK;
}

// per pile status of the last batch (DACC_OK or why the pile was dropped) and the messages a caller would log
int dacc_pile_status(dacc_ctx * c, int32_t * out, uint64_t cap, uint64_t * n)
{
	if ( !c || !n ) return DACC_EINVAL;
	*n = c->pile_status.size();
	if ( out ) std::memcpy(out,c->pile_status.data(),sizeof(int32_t)*std::min<uint64_t>(cap,c->pile_status.size()));
	return DACC_OK;
}
char const * dacc_pile_errors(dacc_ctx * c)
{
	if ( !c ) return "";
	c->pile_errors_joined.clear();
	for ( size_t i = 0; i < c->pile_errors.size(); ++i ) { c->pile_errors_joined += c->pile_errors[i]; c->pile_errors_joined += '\n'; }
	return c->pile_errors_joined.c_str();
}

int dacc_debug_windows(dacc_ctx * c, dacc_window_result * out, uint64_t cap, uint64_t * nwin)
{
	if ( !c || !nwin ) return DACC_EINVAL;
	if ( !c->havebatch ) return DACC_ESTATE;
	BatchPlan & BP = c->BP;
	*nwin = BP.nwindows;
	if ( !out ) return DACC_OK;
	hipSetDevice(c->device);
	size_t const wrecb = DACC_WREC_OF(c->par.w);
	std::vector<WindowOut> wout(BP.nwindows); std::vector<uint8_t> wrec(BP.nwindows*wrecb);
	if ( BP.nwindows )
	{
		HIPCHK(hipMemcpy(wout.data(),c->d_wout.p,BP.nwindows*sizeof(WindowOut),hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(wrec.data(),c->d_wrec.p,BP.nwindows*wrecb,hipMemcpyDeviceToHost));
	}
	uint64_t i = 0;
	for ( uint64_t pi = 0; pi < BP.piles.size() && i < cap; ++pi )
		for ( uint32_t y = 0; y < BP.piles[pi].nwin && i < cap; ++y, ++i )
		{
			uint64_t const wdx = BP.piles[pi].winbase+y;
			WindowOut const & o = wout[wdx];
			dacc_window_result r; std::memset(&r,0,sizeof(r));
			r.pile = pi; r.y = y; r.status = o.status; r.mao = o.mao; r.elength = o.elength; r.k = o.k;
			r.filterfreq = (o.status == WS_OK) ? o.filterfreq : 0; r.conslen = o.conslen; r.minrate = o.minrate;
			if ( o.status == WS_OK )
			{
				// (the consensus as text, cut at 79 symbols: a debugging view)
				uint8_t const * rec = wrec.data() + wdx*wrecb;
				uint32_t const w = c->P.w; bool const wide = DACC_WIDE_W(w);
				uint8_t const * sym = wide ? rec+2+2*(w+2) : rec+1+(w+2);
				uint32_t const nsym = wide ? (rec[2+2*(w+1)] | (static_cast<uint32_t>(rec[3+2*(w+1)])<<8)) : rec[1+(w+1)];
				uint32_t cl = 0;
				for ( uint32_t q = 0; q < nsym && cl < 79; ++q ) if ( sym[q] < 4 ) r.cons[cl++] = "ACGT"[sym[q]];
			}
			out[i] = r;
		}
	return DACC_OK;
}


// guarded entry points (bodies: the *_body functions above)
int dacc_set_error_profile(dacc_ctx * c, double p_i, double p_d, double est_cor) { return guarded(c,[&]() { return dacc_set_error_profile_body(c,p_i,p_d,est_cor); }); }
int dacc_load_db(dacc_ctx * c, uint8_t const * bps, uint64_t bps_bytes, uint64_t const * boff, uint32_t const * rlen, uint64_t nreads) { return guarded(c,[&]() { return dacc_load_db_body(c,bps,bps_bytes,boff,rlen,nreads); }); }
// Measurement hook, host only (no device, no context): the planner of dacc_submit_piles (BatchPlan::plan: device records, window schedule,
// active ranges, offsets -- everything a batch needs before its first byte is uploaded) on the caller's thread + the planner's own.  With
// it the front end can time the whole host side of a run (loader + selection + plan) on a box without a GPU: `daccord_hip --loaderonly`.
int dacc_plan_only(dacc_params const * par, uint32_t const * rlen, uint64_t nreads, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl,
	void const * trace, uint64_t ntrace, int trace_bytes, uint64_t * nwindows, uint64_t * nblocks)
{
	if ( !par || !rlen || (!piles && npiles) ) return DACC_EINVAL;
	try
	{
		BatchPlan BP; std::string err;
		int const rc = BP.plan(*par,piles,npiles,ovl,novl,trace,ntrace,trace_bytes,rlen,nreads,err);
		if ( nwindows ) *nwindows = BP.nwindows;
		if ( nblocks ) *nblocks = BP.nblocks;
		return rc;
	}
	catch ( std::bad_alloc const & ) { return DACC_ENOMEM; }
	catch ( ... ) { return DACC_EINTERNAL; }
}

//  a device ALLOCATION failure while the optional hand-over buffer is held (up to 16 GB) may be an allocation it
// starved: give the buffer back and run the call once more without it -- the context keeps `nohand` set, so neither the planner of the
// retried call nor runDevice's "second use" allocation brings the buffer back.  Other failures (a device fault, internal error
// flags, host allocation) are not retried: a second pass would only overwrite the first error's text.
static bool dacc_drop_hand(dacc_ctx * c)
{
	if ( !c || !c->oom ) return false;
	c->oom = false;
	if ( !c->d_hand.p ) return false;
	(void)hipGetLastError(); hipSetDevice(c->device); (void)hipDeviceSynchronize(); (void)hipGetLastError();
	c->d_hand.release(); c->handcap = 0; c->handwant = 0; c->nohand = true;
	return true;
}
int dacc_submit_piles(dacc_ctx * c, dacc_pile const * piles, uint64_t npiles, dacc_overlap const * ovl, uint64_t novl, void const * trace, uint64_t ntrace, int trace_bytes)
{
	if ( c ) c->oom = false;
	int rc = guarded(c,[&]() { return dacc_submit_piles_body(c,piles,npiles,ovl,novl,trace,ntrace,trace_bytes); });
	if ( rc == DACC_EHIP && dacc_drop_hand(c) ) rc = guarded(c,[&]() { return dacc_submit_piles_body(c,piles,npiles,ovl,novl,trace,ntrace,trace_bytes); });
	return rc;
}
int dacc_rerun_resident(dacc_ctx * c)
{
	if ( c ) c->oom = false;
	int rc = guarded(c,[&]() { return dacc_rerun_resident_body(c); });
	if ( rc == DACC_EHIP && dacc_drop_hand(c) ) rc = guarded(c,[&]() { return dacc_rerun_resident_body(c); });
	return rc;
}

}
