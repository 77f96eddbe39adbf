/* generic engine: k_window (per-wavefront arena in HBM) and k_window_long (tier 5 + generic engine on the second stream) */
#include "window_kernels.hpp"

// one wavefront per workgroup, grid-stride over windows.  Workgroup b lands on XCD b%8 (observed
// placement, used for L2 affinity only): give every XCD a contiguous run of windows so that the
// windows of one pile (which share the pile's overlaps and reads) hit one L2.
// generic engine: all windows (list == 0) or the windows the LDS fast path handed back (list[0] = count)
// Windows differ in cost by orders of magnitude, so the workgroups pull window indices from a counter (*work).
__global__ void __launch_bounds__(64) k_window(WindowBatch B, uint32_t * errflag, uint32_t const * list, uint32_t * work)
{
	uint8_t * arena = B.arena + static_cast<uint64_t>(blockIdx.x)*B.C.bytes;
	if ( B.prof ) B.prof += DACC_PROFW*(blockIdx.x & 4095);
	uint64_t const n = list ? list[0] : B.nwindows;
	uint32_t it = 0;
	while ( true )
	{
		uint32_t i = 0;
		if ( work )
		{
			if ( threadIdx.x == 0 ) i = atomicAdd(work,1u);
			i = __builtin_amdgcn_readfirstlane(i);
		}
		else { i = it*gridDim.x + blockIdx.x; ++it; }
		if ( i >= n ) break;
		uint64_t const w = list ? list[1+i] : i;
		processWindow(B,w,arena);
		if ( threadIdx.x == 0 && B.wout[w].status == WS_OVERFLOW ) atomicOr(errflag,1u);
	}
}

// Second stream: the windows the pre-scan (a B string of more than 64 bases) or the first tier (no LDS tier can run the
// shape) set aside.  One wavefront per workgroup tries tier 5 (FastTier<5>: strings of up to 128 bases, LDS of a whole CU)
// and runs the generic engine right here for what tier 5 cannot hold.  FB.F.ldsbytes == 0: tier 5 is not usable with this
// model table (then this is the generic engine alone).  Static striding over the list, like the generic launch it replaces.
__global__ void __launch_bounds__(64) k_window_long(FastBatch FB, uint32_t * errflag, uint32_t const * list)
{
	typedef FastTier<5> CT;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_generic[];
	LDSQ uint8_t * lds = (LDSQ uint8_t *)lds_generic;
	bool const tier = FB.F.ldsbytes != 0;
	if ( tier ) { FastLds<CT> L; L.base = lds; fast_load_tables(L,FB.F.nrows,FB.F.nsup,FB.W.T,FB.dpsq_vst); }
	uint8_t * arena = FB.W.arena + static_cast<uint64_t>(blockIdx.x)*FB.W.C.bytes;
	uint64_t const n = list[0];
	for ( uint32_t it = 0; ; ++it )
	{
		uint64_t const i = static_cast<uint64_t>(it)*gridDim.x + blockIdx.x;
		if ( i >= n ) break;
		uint64_t const w = list[1+i];
		int rc = FW_NEXT;
		if ( tier ) rc = processWindowFast<CT>(FB,w,lds,false);
		__syncthreads();
		if ( rc != FW_DONE )
		{
			processWindow(FB.W,w,arena);
			if ( threadIdx.x == 0 && FB.W.wout[w].status == WS_OVERFLOW ) atomicOr(errflag,1u);
		}
		__syncthreads();
	}
}

