/*
 * The window kernels of libdaccord_hip.so, one translation unit each (k_fast_<tier>.hip, k_generic.hip): the seven of them are 150-260 KB
 * of gfx950 code apiece and took 25 minutes to compile one after the other inside capi.hip; as separate objects they compile side by side
 * (daccord_amd/build.py).  This header holds what the units share: the work distribution, the kernel template of the LDS tiers (defined
 * here, instantiated explicitly in k_fast_<tier>.hip, declared `extern template` for capi.hip's launches) and the prototypes of the two
 * kernels of the generic engine (defined in k_generic.hip).  Same device code as before the split: build.kernel_isa_hashes() is unchanged.
 */
#ifndef DACC_WINDOW_KERNELS_HPP
#define DACC_WINDOW_KERNELS_HPP
#include <hip/hip_runtime.h>
#include "window_main.hpp"
#include "fast_window.hpp"

using namespace dacc;

// Work distribution of the window kernels.  Windows differ in cost by orders of magnitude, so workgroups pull indices
// from counters; workgroup b runs on XCD b%8 (observed placement), and the windows of one pile share its overlaps and
// reads, so every XCD first drains its own contiguous eighth of the index range (its L2 keeps the pile's data) and
// then steals from the other XCDs.  work[0..7] = per-XCD counters.  Returns false when nothing is left.
__device__ __forceinline__ bool next_window(uint32_t * work, uint64_t const n, uint32_t & state, uint32_t & idx)
{
	uint32_t const home = blockIdx.x & 7;
	while ( state < 8 )
	{
		uint32_t const q = (home + state) & 7;
		uint64_t const lo = (n*q)>>3, hi = (n*(q+1))>>3;
		uint32_t i = 0;
		if ( threadIdx.x == 0 ) i = atomicAdd(work+q,1u);
		i = __builtin_amdgcn_readfirstlane(i);
		if ( lo + i < hi ) { idx = static_cast<uint32_t>(lo+i); return true; }
		++state;
	}
	return false;
}

// generic engine (k_generic.hip)
__global__ void __launch_bounds__(64) k_window(WindowBatch B, uint32_t * errflag, uint32_t const * list, uint32_t * work);
__global__ void __launch_bounds__(64) k_window_long(FastBatch FB, uint32_t * errflag, uint32_t const * list);

// LDS fast path: one wavefront per workgroup, working state in the workgroup's dynamic LDS slice.
// list == 0: all windows; else the windows a smaller capacity tier handed over.  Windows that do not fit go to FB.retry.
// (tiers 0 and 1 hold 8 and 6 windows per CU in LDS: two wavefronts per SIMD need their kernels within 256 registers; the register
// allocator lands within a few registers of that bound either way, amdgpu_waves_per_eu(2) would make it a requirement (-DDACC_WPE_CAP: 248 / 253 registers, no scratch) but changes the
// scheduler's targets with it: 5 % SLOWER on config 2, profiles/r05e_ab_register_cap.log -- so the bound is kept by hand)
#if defined(DACC_WPE_CAP)
#define DACC_WPE(T) __attribute__((amdgpu_waves_per_eu((T) <= 1 ? 2 : 1)))
#elif defined(DACC_NUMVGPR_CAP)
#define DACC_WPE(T) __attribute__((amdgpu_num_vgpr((T) <= 1 ? 256 : 512)))
#elif defined(DACC_T4_NOWPE)
#define DACC_WPE(T)
#else
// (round 6) the deep tier, k_window_fast<4>: its 31.5 KB of LDS allow FIVE wavefronts per CU, its 302 + 46 registers allowed four (one per
// SIMD).  Unlike tiers 0 / 1 -- which sit a few registers below 256 on their own and lose 5 % under a cap -- it is nowhere near the bound by
// itself: amdgpu_waves_per_eu(2) brings it to 256 registers with 17 spilled dwords (72 bytes of scratch) and the fifth wavefront is worth
// -18 % on the tier, 24.4 -> 27.7 Mbase/s on the 54x shape (profiles/r06r).  amdgpu_num_vgpr(256) does not compile (allocation fails).
#define DACC_WPE(T) __attribute__((amdgpu_waves_per_eu((T) == 4 ? 2 : 1)))
#endif
template<int TIER>
__global__ void __launch_bounds__(64) DACC_WPE(TIER) k_window_fast(FastBatch FB, uint32_t const * list, uint32_t * work)
{
	typedef FastTier<TIER> CT;
	if ( FB.W.prof ) FB.W.prof += DACC_PROFW*(blockIdx.x & 4095);
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_generic[];
	LDSQ uint8_t * lds = (LDSQ uint8_t *)lds_generic;
	{ FastLds<CT> L; L.base = lds; fast_load_tables(L,FB.F.nrows,FB.F.nsup,FB.W.T,FB.dpsq_vst); }
#if defined(DACC_PROFILE)
	uint64_t const t0c = clock64(), t0w = wall_clock64();
#endif
	uint64_t const n = list ? list[0] : FB.W.nwindows;
	uint32_t it = 0, qstate = 0;
	while ( true )
	{
		uint32_t i = 0;
		if ( work ) { if ( !next_window(work,n,qstate,i) ) break; }
		else { i = it*gridDim.x + blockIdx.x; ++it; }
		if ( i >= n ) break;
		uint64_t const w = list ? list[1+i] : i;
		int const rc = processWindowFast<CT>(FB,w,lds,list != 0);
		if ( rc != FW_DONE && threadIdx.x == 0 )
		{
			uint32_t * const dst = (rc == FW_GENERIC && FB.gearly) ? FB.gearly : FB.retry;
			uint32_t const q = atomicAdd(dst,1u); dst[1+q] = static_cast<uint32_t>(w);
		}
		__syncthreads();
	}
#if defined(DACC_PROFILE)
	if ( threadIdx.x == 0 && FB.W.prof && !list ) { atomicAdd(reinterpret_cast<unsigned long long *>(FB.W.prof+30),static_cast<unsigned long long>(clock64()-t0c)); atomicAdd(reinterpret_cast<unsigned long long *>(FB.W.prof+31),static_cast<unsigned long long>(wall_clock64()-t0w)); atomicMax(reinterpret_cast<unsigned long long *>(FB.W.prof+29),static_cast<unsigned long long>(wall_clock64()-t0w)); }
#endif
}

#if !defined(DACC_INSTANTIATE_TIER)
// every unit but k_fast_<tier>.hip: the tiers' kernels are instantiated elsewhere
extern template __global__ void k_window_fast<0>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<1>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<2>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<3>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<4>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<6>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<7>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<8>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<9>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<10>(FastBatch, uint32_t const *, uint32_t *);
extern template __global__ void k_window_fast<11>(FastBatch, uint32_t const *, uint32_t *);
#endif
#endif
