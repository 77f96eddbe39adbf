/*
 * Wavefront-cooperative primitives for the gfx950 kernels (64-lane wavefronts, one wavefront
 * per workgroup in the window kernel).
 *
 * DACC_EMUL: the container this is developed in has no GPU, so the kernel LOGIC is also
 * unit-tested by compiling the same device headers with g++ as a 1-lane "wavefront"
 * (WSZ == 1, lane == 0).  That build exists only under tests/emul/ as a test harness; it is
 * never part of libdaccord_hip.so, which contains gfx950 code only and refuses to run
 * without a HIP device.
 */
#ifndef DACC_WAVE_HPP
#define DACC_WAVE_HPP
#include <stdint.h>

#if defined(DACC_EMUL) && defined(DACC_EMUL_LANES) && DACC_EMUL_LANES == 64
  // 64-lane host wavefront (coroutine per lane), see wave_emul64.hpp
  #define DEV inline
  #define HDEV inline
  #define WSZ 64
  #include "wave_emul64.hpp"
#elif defined(DACC_EMUL)
  #define DEV inline
  #define HDEV inline
  #define WSZ 1
  namespace dacc {
  static inline int wv_lane() { return 0; }
  static inline void wv_sync() {}
  static inline uint32_t wv_scan_excl(uint32_t v, uint32_t & total) { total = v; return 0; }
  static inline uint32_t wv_scan_flag(bool p, uint32_t & total) { total = p ? 1u : 0u; return 0; }
  static inline uint32_t wv_sum(uint32_t v) { return v; }
  static inline uint64_t wv_sum64(uint64_t v) { return v; }
  static inline uint32_t wv_max(uint32_t v) { return v; }
  static inline uint64_t wv_max64(uint64_t v) { return v; }
  static inline uint64_t wv_min64(uint64_t v) { return v; }
  static inline int wv_any(int p) { return p; }
  static inline uint32_t wv_or(uint32_t v) { return v; }
  static inline uint64_t wv_or64(uint64_t v) { return v; }
  static inline uint64_t wv_ballot(int p) { return p ? 1ull : 0ull; }
  static inline uint64_t wv_lanemask_lt() { return 0ull; }
  static inline uint32_t wv_bcast(uint32_t v, int) { return v; }
  static inline uint64_t wv_bcast64(uint64_t v, int) { return v; }
  static inline uint32_t wv_uni(uint32_t v) { return v; }
  static inline uint64_t wv_uni64(uint64_t v) { return v; }
  static inline uint32_t wv_shfl(uint32_t v, int) { return v; }
  static inline uint64_t wv_shfl64(uint64_t v, int) { return v; }
  static inline int dacc_popc64(uint64_t v) { return __builtin_popcountll(v); }
  static inline void atomicOrFlag(uint32_t * f) { *f |= 1u; }
  template<typename T> static inline T wv_atomic_add(T * p, T const v) { T const o = *p; *p = o + v; return o; }
  static inline uint32_t wv_atomic_add_global(uint32_t * p, uint32_t const v) { uint32_t const o = *p; *p = o + v; return o; }
  template<typename F> static inline void wave_run(F const & f) { f(); }
  }
#else
  #include <hip/hip_runtime.h>
  #define DEV __device__ __forceinline__
  #define HDEV __host__ __device__ __forceinline__
  #define WSZ 64
  namespace dacc {
  DEV int wv_lane() { return threadIdx.x & 63; }
  // one wavefront per workgroup: barrier + LDS/global visibility inside the wavefront
  DEV void wv_sync() { __syncthreads(); }
  // Cross-lane data movement inside the VALU (DPP: row_shr inside rows of 16 lanes, row_bcast15 / row_bcast31 across rows,
  // v_readlane for the result) instead of ds_bpermute round trips through the LDS pipeline: a scan or reduction is six
  // dependent VALU instructions.  `old` (the value a lane keeps when its DPP source lane does not exist) is the identity.
  #define DACC_DPP(old_,src_,ctrl_,rmask_) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(old_),static_cast<int>(src_),ctrl_,rmask_,0xF,false))
  enum { DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };
  // inclusive prefix "sums" under OP over the 64 lanes (OP associative and commutative, ID its identity)
  #define DACC_DPP_SCAN(x_,OP,ID) \
	{ uint32_t y_; \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_SHR1,0xF); x_ = OP(x_,y_); \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_SHR2,0xF); x_ = OP(x_,y_); \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_SHR4,0xF); x_ = OP(x_,y_); \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_SHR8,0xF); x_ = OP(x_,y_); \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_BCAST15,0xA); x_ = OP(x_,y_); \
	  y_ = DACC_DPP(ID,x_,DPP_ROW_BCAST31,0xC); x_ = OP(x_,y_); }
  #define DACC_OP_ADD(a_,b_) ((a_)+(b_))
  #define DACC_OP_MAX(a_,b_) ((a_) > (b_) ? (a_) : (b_))
  #define DACC_OP_OR(a_,b_) ((a_)|(b_))
  DEV uint32_t wv_scan_excl(uint32_t v, uint32_t & total)
  {
	uint32_t x = v;
	DACC_DPP_SCAN(x,DACC_OP_ADD,0u)
	total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x),63));
	return x - v;
  }
  // exclusive count of set predicates below this lane (ballot + mbcnt: no cross-lane data movement at all)
  DEV uint32_t wv_scan_flag(bool const p, uint32_t & total)
  {
	uint64_t const b = __ballot(p);
	total = static_cast<uint32_t>(__popcll(b));
	return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b>>32),__builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b),0u));
  }
  // value of the first active lane as a wave-uniform (scalar) value: keeps the control flow that depends on it uniform
  DEV uint32_t wv_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  DEV uint64_t wv_uni64(uint64_t v)
  {
	uint32_t const lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v>>32));
	return (static_cast<uint64_t>(hi)<<32) | lo;
  }
  DEV uint32_t wv_sum(uint32_t v)
  {
	DACC_DPP_SCAN(v,DACC_OP_ADD,0u)
	return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v),63));
  }
  DEV uint64_t wv_sum64(uint64_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) v += __shfl_xor(v,d,64);
	return wv_uni64(v);
  }
  DEV uint32_t wv_max(uint32_t v)
  {
	DACC_DPP_SCAN(v,DACC_OP_MAX,0u)
	return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v),63));
  }
  DEV uint64_t wv_max64(uint64_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) { uint64_t const o = __shfl_xor(v,d,64); v = o > v ? o : v; }
	return wv_uni64(v);
  }
  DEV uint64_t wv_min64(uint64_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) { uint64_t const o = __shfl_xor(v,d,64); v = o < v ? o : v; }
	return wv_uni64(v);
  }
  DEV int wv_any(int p) { return __any(p); }
  DEV uint32_t wv_or(uint32_t v)
  {
	DACC_DPP_SCAN(v,DACC_OP_OR,0u)
	return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v),63));
  }
  DEV uint64_t wv_or64(uint64_t v)
  {
	uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v>>32);
	DACC_DPP_SCAN(lo,DACC_OP_OR,0u)
	DACC_DPP_SCAN(hi,DACC_OP_OR,0u)
	uint32_t const l = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lo),63)), h = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hi),63));
	return (static_cast<uint64_t>(h)<<32) | l;
  }
  DEV uint64_t wv_ballot(int p) { return __ballot(p); }
  DEV uint64_t wv_lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
  DEV uint32_t wv_bcast(uint32_t v, int src) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v),__builtin_amdgcn_readfirstlane(src))); }   // src is wave-uniform
  DEV uint64_t wv_bcast64(uint64_t v, int src)
  {
	int const sl = __builtin_amdgcn_readfirstlane(src);
	uint32_t const lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v)),sl)), hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v>>32)),sl));
	return (static_cast<uint64_t>(hi)<<32) | lo;
  }
  DEV uint32_t wv_shfl(uint32_t v, int src) { return __shfl(v,src,64); }
  DEV uint64_t wv_shfl64(uint64_t v, int src) { return __shfl(v,src,64); }
  DEV int dacc_popc64(uint64_t v) { return __popcll(v); }
  DEV void atomicOrFlag(uint32_t * f) { atomicOr(f,1u); }
  // workgroup scope atomic add on an LDS (or global) word, returns the old value
  template<typename PT> DEV uint32_t wv_atomic_add(PT p, uint32_t const v) { return __hip_atomic_fetch_add(p,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); }
  // device scope atomic add on a word in global memory (counters shared by all workgroups of a kernel)
  DEV uint32_t wv_atomic_add_global(uint32_t * p, uint32_t const v) { return __hip_atomic_fetch_add(p,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); }
  }
#endif

// address space qualifiers: on the device LDS pointers are 32 bit ds_* addresses, on the host (planning, emulation) plain pointers
#if defined(__HIP_DEVICE_COMPILE__)
#define LDSQ __attribute__((address_space(3)))
#define GLBQ __attribute__((address_space(1)))
#else
#define LDSQ
#define GLBQ
#endif

namespace dacc {

// ascending bitonic sort of n64 (power of two) 64-bit keys in memory; all lanes call
template<typename PT>
DEV void wv_bitonic_sort(PT A, uint32_t const n)
{
	int const lane = wv_lane();
	for ( uint32_t k = 2; k <= n; k <<= 1 )
		for ( uint32_t j = k>>1; j > 0; j >>= 1 )
		{
			for ( uint32_t t = lane; t < (n>>1); t += WSZ )
			{
				// t-th compare-exchange pair of this stage
				uint32_t const i = ((t & ~(j-1)) << 1) | (t & (j-1));
				uint32_t const l = i | j;
				uint64_t const a = A[i], b = A[l];
				bool const up = ((i & k) == 0);
				if ( (a > b) == up ) { A[i] = b; A[l] = a; }
			}
			wv_sync();
		}
}

// ascending sort of n (any n) 64-bit keys: bitonic network in its all-ascending form (first sub-step of every merge is
// a flip), so that the virtual +infinity padding up to the next power of two never moves and its pairs are skipped
// The compare-exchange pairs of one step are disjoint; a lane loads up to four of them before it stores, so that the
// LDS round trips overlap.
template<typename PT>
DEV void wv_cx4(PT A, uint32_t const n, uint32_t const i0, uint32_t const l0, uint32_t const i1, uint32_t const l1,
	uint32_t const i2, uint32_t const l2, uint32_t const i3, uint32_t const l3)
{
	bool const v0 = l0 < n, v1 = l1 < n, v2 = l2 < n, v3 = l3 < n;
	uint64_t a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
	if ( v0 ) { a0 = A[i0]; b0 = A[l0]; }
	if ( v1 ) { a1 = A[i1]; b1 = A[l1]; }
	if ( v2 ) { a2 = A[i2]; b2 = A[l2]; }
	if ( v3 ) { a3 = A[i3]; b3 = A[l3]; }
	if ( v0 && a0 > b0 ) { A[i0] = b0; A[l0] = a0; }
	if ( v1 && a1 > b1 ) { A[i1] = b1; A[l1] = a1; }
	if ( v2 && a2 > b2 ) { A[i2] = b2; A[l2] = a2; }
	if ( v3 && a3 > b3 ) { A[i3] = b3; A[l3] = a3; }
}
template<typename PT>
DEV void wv_bitonic_sort_n(PT A, uint32_t const n)
{
	int const lane = wv_lane();
	uint32_t n2 = 1; while ( n2 < n ) n2 <<= 1;
	uint32_t const half = n2>>1;
	for ( uint32_t k = 2; k <= n2; k <<= 1 )
	{
		uint32_t const h = k>>1;
		for ( uint32_t t = lane; t < half; t += 4*WSZ )
		{
			uint32_t I[4], Lx[4];
			#pragma unroll
			for ( uint32_t u = 0; u < 4; ++u )
			{
				uint32_t const tt = t + u*WSZ;
				uint32_t const blk = tt / h, off = tt - blk*h;
				I[u] = blk*k + off; Lx[u] = (tt < half) ? (blk*k + (k-1-off)) : n;
			}
			wv_cx4(A,n,I[0],Lx[0],I[1],Lx[1],I[2],Lx[2],I[3],Lx[3]);
		}
		wv_sync();
		for ( uint32_t j = k>>2; j > 0; j >>= 1 )
		{
			for ( uint32_t t = lane; t < half; t += 4*WSZ )
			{
				uint32_t I[4], Lx[4];
				#pragma unroll
				for ( uint32_t u = 0; u < 4; ++u )
				{
					uint32_t const tt = t + u*WSZ;
					I[u] = ((tt & ~(j-1)) << 1) | (tt & (j-1)); Lx[u] = (tt < half) ? (I[u] | j) : n;
				}
				wv_cx4(A,n,I[0],Lx[0],I[1],Lx[1],I[2],Lx[2],I[3],Lx[3]);
			}
			wv_sync();
		}
	}
}

// ascending bitonic sort of n (power of two) indices by (key[idx], idx); 0xFFFFFFFF pads sort last
DEV void wv_bitonic_sort_idx(uint32_t * I, uint64_t const * K, uint32_t const n)
{
	int const lane = wv_lane();
	for ( uint32_t k = 2; k <= n; k <<= 1 )
		for ( uint32_t j = k>>1; j > 0; j >>= 1 )
		{
			for ( uint32_t t = lane; t < (n>>1); t += WSZ )
			{
				uint32_t const i = ((t & ~(j-1)) << 1) | (t & (j-1));
				uint32_t const l = i | j;
				uint32_t const a = I[i], b = I[l];
				bool gt;
				if ( a == 0xFFFFFFFFu ) gt = (b != 0xFFFFFFFFu);
				else if ( b == 0xFFFFFFFFu ) gt = false;
				else { uint64_t const ka = K[a], kb = K[b]; gt = (ka > kb) || (ka == kb && a > b); }
				bool const up = ((i & k) == 0);
				if ( gt == up ) { I[i] = b; I[l] = a; }
			}
			wv_sync();
		}
}

#if WSZ == 64
// ascending sort of n <= 64*R distinct 64-bit keys held in registers: lane l keeps elements l*R .. l*R+R-1 (missing ones
// are +infinity), so the compare-exchange steps at distances below R stay inside a lane and the others exchange whole
// registers with the partner lane; the memory is touched once for the load and once for the store.  The step structure
// is a run time loop (the code stays small), only the register indices are compile time constants.
template<int R, int J> DEV void wv_sort_inlane(uint64_t (&v)[R], uint32_t const base, uint32_t const k)
{
	#pragma unroll
	for ( int r = 0; r < R; ++r )
		if ( (r & J) == 0 )
		{
			uint64_t const a = v[r], b = v[r|J];
			bool const up = ((base + r) & k) == 0;
			bool const sw = (a > b) == up;
			v[r] = sw ? b : a; v[r|J] = sw ? a : b;
		}
}
template<int R, typename PT>
DEV void wv_sort_regs(PT A, uint32_t const n)
{
	static_assert(R == 2 || R == 4 || R == 8 || R == 16 || R == 32,"elements per lane");
	uint32_t const lane = wv_lane(), base = lane*R;
	uint64_t v[R];
	#pragma unroll
	for ( int r = 0; r < R; ++r ) v[r] = (base + r < n) ? A[base+r] : ~0ull;
	uint32_t n2 = 2; while ( n2 < n ) n2 <<= 1;
	for ( uint32_t k = 2; k <= n2; k <<= 1 )
	{
		for ( uint32_t j = k>>1; j >= static_cast<uint32_t>(R); j >>= 1 )
		{
			uint32_t const m = j/R;
			bool const keepmin = ((lane & m) == 0) == ((base & k) == 0);
			#pragma unroll
			for ( int r = 0; r < R; ++r )
			{
				uint64_t const p = wv_shfl64(v[r],static_cast<int>(lane ^ m));
				bool const plt = p < v[r];
				v[r] = (plt == keepmin) ? p : v[r];
			}
		}
		uint32_t const h = k>>1;
		if ( R > 16 && h >= 16 ) wv_sort_inlane<R,(R > 16 ? 16 : 1)>(v,base,k);
		if ( R > 8 && h >= 8 ) wv_sort_inlane<R,(R > 8 ? 8 : 1)>(v,base,k);
		if ( R > 4 && h >= 4 ) wv_sort_inlane<R,(R > 4 ? 4 : 1)>(v,base,k);
		if ( R > 2 && h >= 2 ) wv_sort_inlane<R,(R > 2 ? 2 : 1)>(v,base,k);
		wv_sort_inlane<R,1>(v,base,k);
	}
	wv_sync();
	#pragma unroll
	for ( int r = 0; r < R; ++r ) if ( base + r < n ) A[base+r] = v[r];
	wv_sync();
}
// (round 6) 1024 < n <= 2048 keys with SIXTEEN registers per lane instead of 32: both halves of 1024 are sorted ascending in registers one
// after the other, the first step of the last merge (element i against element 2047-i) takes its partners from memory, and the two halves --
// bitonic sequences now, the lower holding the 1024 smallest keys -- are merged in registers (the half-cleaner steps at distances 512 ... 1).
// The same number of compare-exchange steps as the 32-register network (2 x 55 + 1 + 2 x 10 passes over 16 registers against 66 over 32).
// Built to get the deep tier (k_window_fast<4>: 302 + 46 registers) under 256; measured (profiles/r06r, 54x): the allocator takes what the
// occupancy target leaves it either way (310 + 54 with this sort, uncapped), amdgpu_waves_per_eu(2) is what brings the kernel to 256, and
// under that cap the 32-register network is 1 % faster (17 spilled dwords against 13).  Kept for A/B builds (-DDACC_SORT2H), not the default.
template<int R>
DEV void wv_sort_regs_merge_steps(uint64_t (&v)[R], uint32_t const lane, uint32_t const base, uint32_t const k, uint32_t const jfirst)
{
	for ( uint32_t j = jfirst; j >= static_cast<uint32_t>(R); j >>= 1 )
	{
		uint32_t const m = j/R;
		bool const keepmin = ((lane & m) == 0) == ((base & k) == 0);
		#pragma unroll
		for ( int r = 0; r < R; ++r )
		{
			uint64_t const p = wv_shfl64(v[r],static_cast<int>(lane ^ m));
			bool const plt = p < v[r];
			v[r] = (plt == keepmin) ? p : v[r];
		}
	}
	if ( R > 16 ) wv_sort_inlane<R,(R > 16 ? 16 : 1)>(v,base,k);
	if ( R > 8 ) wv_sort_inlane<R,(R > 8 ? 8 : 1)>(v,base,k);
	if ( R > 4 ) wv_sort_inlane<R,(R > 4 ? 4 : 1)>(v,base,k);
	if ( R > 2 ) wv_sort_inlane<R,(R > 2 ? 2 : 1)>(v,base,k);
	wv_sort_inlane<R,1>(v,base,k);
}
template<typename PT>
DEV void wv_sort_two_halves(PT A, uint32_t const n)
{
	enum { R = 16, H = 64*R };
	wv_sort_regs<R>(A,H);
	wv_sort_regs<R>(A+H,n-H);
	uint32_t const lane = wv_lane(), base = lane*R;
	uint64_t v[R];
	#pragma unroll
	for ( int r = 0; r < R; ++r ) v[r] = A[base+r];
	#pragma unroll
	for ( int r = 0; r < R; ++r )
	{
		uint32_t const q = 2*H-1-(base+r);
		if ( q < n ) { uint64_t const b = A[q]; if ( v[r] > b ) { A[q] = v[r]; v[r] = b; } }
	}
	wv_sort_regs_merge_steps<R>(v,lane,base,2*H,H/2);
	wv_sync();
	#pragma unroll
	for ( int r = 0; r < R; ++r ) A[base+r] = v[r];
	#pragma unroll
	for ( int r = 0; r < R; ++r ) v[r] = (H+base+r < n) ? A[H+base+r] : ~0ull;
	wv_sort_regs_merge_steps<R>(v,lane,base,2*H,H/2);
	wv_sync();
	#pragma unroll
	for ( int r = 0; r < R; ++r ) if ( H+base+r < n ) A[H+base+r] = v[r];
	wv_sync();
}
#endif
// ascending sort of n distinct 64-bit keys in memory (all lanes call); cap = compile time bound of n
// R32: also keep up to 2048 keys in registers (32 per lane); only the deep tier asks for it
template<uint32_t CAP, bool R32 = false, typename PT>
DEV void wv_sort_keys(PT A, uint32_t const n)
{
#if WSZ == 64
	if ( CAP <= 128 || n <= 128 ) { wv_sort_regs<2>(A,n); return; }
	if ( CAP <= 256 || n <= 256 ) { wv_sort_regs<4>(A,n); return; }
	if ( CAP <= 512 || n <= 512 ) { wv_sort_regs<8>(A,n); return; }
	if ( CAP <= 1024 || n <= 1024 ) { wv_sort_regs<16>(A,n); return; }
#if defined(DACC_SORT2H)
	if ( R32 && (CAP <= 2048 || n <= 2048) ) { wv_sort_two_halves(A,n); return; }      // (A/B builds: two halves of 16 keys per lane)
#else
	if ( R32 && (CAP <= 2048 || n <= 2048) ) { wv_sort_regs<32>(A,n); return; }      // deep piles: 55 strings carry 1500 k-mer instances
#endif
#endif
	wv_bitonic_sort_n(A,n);
}

// bits of a 32 bit word in reverse order; the low 16 bits of a word spread to the even bit positions
HDEV uint32_t dacc_rev32(uint32_t x)
{
#if defined(__clang__)
	return __builtin_bitreverse32(x);
#else
	x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
	return (x >> 16) | (x << 16);
#endif
}
HDEV uint32_t dacc_spread16(uint32_t x)
{
	x &= 0xFFFFu; x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
	return x;
}
HDEV uint32_t next_pow2(uint32_t v)
{
	uint32_t p = 1;
	while ( p < v ) p <<= 1;
	return p;
}

}
#endif
