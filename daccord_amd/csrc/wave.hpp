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
  DEV uint32_t wv_scan_excl(uint32_t v, uint32_t & total)
  {
	uint32_t x = v;
	int const lane = wv_lane();
	#pragma unroll
	for ( int d = 1; d < 64; d <<= 1 )
	{
		uint32_t const y = __shfl_up(x,d,64);
		if ( lane >= d ) x += y;
	}
	total = __shfl(x,63,64);
	return x - v;
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
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) v += __shfl_xor(v,d,64);
	return wv_uni(v);
  }
  DEV uint64_t wv_sum64(uint64_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) v += __shfl_xor(v,d,64);
	return wv_uni64(v);
  }
  DEV uint32_t wv_max(uint32_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) { uint32_t const o = __shfl_xor(v,d,64); v = o > v ? o : v; }
	return wv_uni(v);
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
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) v |= __shfl_xor(v,d,64);
	return wv_uni(v);
  }
  DEV uint64_t wv_or64(uint64_t v)
  {
	#pragma unroll
	for ( int d = 32; d >= 1; d >>= 1 ) v |= __shfl_xor(v,d,64);
	return wv_uni64(v);
  }
  DEV uint64_t wv_ballot(int p) { return __ballot(p); }
  DEV uint64_t wv_lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
  DEV uint32_t wv_bcast(uint32_t v, int src) { return wv_uni(__shfl(v,src,64)); }
  DEV uint64_t wv_bcast64(uint64_t v, int src) { return wv_uni64(__shfl(v,src,64)); }
  DEV uint32_t wv_shfl(uint32_t v, int src) { return __shfl(v,src,64); }
  DEV uint64_t wv_shfl64(uint64_t v, int src) { return __shfl(v,src,64); }
  DEV int dacc_popc64(uint64_t v) { return __popcll(v); }
  DEV void atomicOrFlag(uint32_t * f) { atomicOr(f,1u); }
  // workgroup scope atomic add on an LDS (or global) word, returns the old value
  template<typename PT> DEV uint32_t wv_atomic_add(PT p, uint32_t const v) { return __hip_atomic_fetch_add(p,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); }
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

HDEV uint32_t next_pow2(uint32_t v)
{
	uint32_t p = 1;
	while ( p < v ) p <<= 1;
	return p;
}

}
#endif
