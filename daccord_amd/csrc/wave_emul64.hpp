/*
 * TEST HARNESS ONLY (DACC_EMUL with DACC_EMUL_LANES == 64): a 64-lane wavefront on the host.
 *
 * The build container has no GPU and the metered GPU minutes are too few to debug lane-parallel code there, so the
 * kernel headers can also be compiled with g++ as a REAL 64-lane wavefront: every lane is a coroutine with its own
 * stack; a lane runs until it reaches a wavefront collective (ballot, scan, shuffle, reduction, wv_sync), deposits its
 * operand and hands over to the next lane; the last lane to arrive completes the collective and lane 0 resumes.  All 64
 * lanes must reach the same collective (checked by a call-site tag): a collective inside divergent control flow, which
 * is undefined on the hardware as well, aborts the run.  Between collectives the lanes run one after the other, so a
 * missing wv_sync() between a write and a read of another lane shows up as a wrong result here.
 * Never part of libdaccord_hip.so.
 */
#ifndef DACC_WAVE_EMUL64_HPP
#define DACC_WAVE_EMUL64_HPP
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace dacc {

struct WaveEmu
{
	enum { NL = 64, STACK = 512*1024 };
	void * sp[NL]; void * mainsp;
	std::vector<uint8_t> stacks;
	uint64_t val[2][NL]; uint32_t tag[2][NL]; int line[2][NL];      // line: call site of the collective in the kernel headers (diagnostics)
	uint64_t seq[NL];
	bool done[NL];
	int cur;
	std::function<void()> const * body;
	WaveEmu() : mainsp(0), stacks(static_cast<size_t>(NL)*STACK), cur(0), body(0) {}
};

static inline WaveEmu & wave_emu() { static WaveEmu * W = new WaveEmu; return *W; }

extern "C" void dacc_emu_switch(void ** from_sp, void * to_sp);
extern "C" void dacc_emu_entry();

#if defined(DACC_EMUL_IMPL)
__asm__(
".text\n"
".globl dacc_emu_switch\n"
".type dacc_emu_switch,@function\n"
"dacc_emu_switch:\n"
"	pushq %rbp\n	pushq %rbx\n	pushq %r12\n	pushq %r13\n	pushq %r14\n	pushq %r15\n"
"	movq %rsp,(%rdi)\n"
"	movq %rsi,%rsp\n"
"	popq %r15\n	popq %r14\n	popq %r13\n	popq %r12\n	popq %rbx\n	popq %rbp\n"
"	ret\n"
".size dacc_emu_switch,.-dacc_emu_switch\n"
);
static void dacc_emu_lane_main();
extern "C" void dacc_emu_entry() { dacc_emu_lane_main(); }
#endif

// hand over to the next lane that is still running; returns when this lane is resumed
static inline void wave_emu_yield()
{
	WaveEmu & W = wave_emu();
	int const me = W.cur;
	int nx = me;
	for ( int i = 1; i <= WaveEmu::NL; ++i ) { int const c = (me+i) % WaveEmu::NL; if ( !W.done[c] ) { nx = c; break; } }
	if ( nx == me && !W.done[me] ) return;
	if ( W.done[me] && nx == me ) { W.cur = -1; dacc_emu_switch(&W.sp[me],W.mainsp); return; }
	W.cur = nx;
	dacc_emu_switch(&W.sp[me],W.sp[nx]);
}

#if defined(DACC_EMUL_IMPL)
static void dacc_emu_lane_main()
{
	WaveEmu & W = wave_emu();
	(*W.body)();
	int const me = W.cur;
	W.done[me] = true;
	// a lane that ends while others wait in a collective is a divergence error unless every lane ends
	bool alldone = true; for ( int i = 0; i < WaveEmu::NL; ++i ) alldone = alldone && W.done[i];
	if ( alldone ) { W.cur = -1; dacc_emu_switch(&W.sp[me],W.mainsp); }
	else wave_emu_yield();
	std::fprintf(stderr,"[wave emu] resumed a finished lane\n"); std::abort();
}
#endif

// run `body` once per lane as a 64-lane wavefront
static inline void wave_run(std::function<void()> const & body)
{
	WaveEmu & W = wave_emu();
	W.body = &body;
	for ( int i = 0; i < WaveEmu::NL; ++i )
	{
		W.done[i] = false; W.seq[i] = 0;
		uint8_t * top = W.stacks.data() + static_cast<size_t>(i+1)*WaveEmu::STACK;
		uintptr_t t = reinterpret_cast<uintptr_t>(top) & ~static_cast<uintptr_t>(63);
		void ** s = reinterpret_cast<void **>(t);
		// layout expected by dacc_emu_switch: r15 r14 r13 r12 rbx rbp ret ; after `ret` rsp must be 8 mod 16
		s -= 1; *s = 0;                                        // fake return address of the entry function (alignment)
		s -= 1; *s = reinterpret_cast<void *>(&dacc_emu_entry);
		for ( int r = 0; r < 6; ++r ) { s -= 1; *s = 0; }
		W.sp[i] = s;
	}
	W.cur = 0;
	dacc_emu_switch(&W.mainsp,W.sp[0]);
	for ( int i = 0; i < WaveEmu::NL; ++i ) if ( !W.done[i] ) { std::fprintf(stderr,"[wave emu] lane %d did not finish (divergent collective?)\n",i); std::abort(); }
}

static inline int wv_lane() { return wave_emu().cur; }

// rendezvous of all lanes with a 64-bit operand each; afterwards V[l] holds lane l's operand
static inline uint64_t const * wave_collect(uint64_t const v, uint32_t const tag, int const line = 0)
{
	WaveEmu & W = wave_emu();
	int const me = W.cur;
	uint64_t const q = W.seq[me]++;
	int const b = q & 1;
	W.val[b][me] = v; W.tag[b][me] = tag; W.line[b][me] = line;
	// wait until every lane has deposited for round q: lanes run in order, so after one full cycle of yields all have
	wave_emu_yield();
	for ( int i = 0; i < WaveEmu::NL; ++i )
	{
		if ( W.seq[i] < q+1 && W.done[i] ) { std::fprintf(stderr,"[wave emu] lane %d ended while lane %d waits in a collective (tag %u)\n",i,me,tag); std::abort(); }
		if ( W.seq[i] < q+1 || W.tag[b][i] != tag || (line && W.line[b][i] && W.line[b][i] != line) ) { std::fprintf(stderr,"[wave emu] divergent collective: lane %d tag %u seq %llu (source line %d) vs lane %d tag %u seq %llu (source line %d)\n",me,tag,(unsigned long long)q,line,i,W.tag[b][i],(unsigned long long)W.seq[i],W.line[b][i]); std::abort(); }
	}
	return W.val[b];
}

static inline void wv_sync(int const _ln = __builtin_LINE()) { wave_collect(0,1,_ln); }
static inline uint32_t wv_scan_excl(uint32_t v, uint32_t & total, int const _ln = __builtin_LINE())
{
	int const me = wv_lane(); uint64_t const * V = wave_collect(v,2,_ln);
	uint32_t pre = 0, tot = 0; for ( int i = 0; i < 64; ++i ) { if ( i < me ) pre += static_cast<uint32_t>(V[i]); tot += static_cast<uint32_t>(V[i]); }
	total = tot; return pre;
}
static inline uint32_t wv_scan_flag(bool p, uint32_t & total, int const _ln = __builtin_LINE()) { return wv_scan_excl(p ? 1u : 0u,total,_ln); }
static inline uint32_t wv_sum(uint32_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,3,_ln); uint32_t s = 0; for ( int i = 0; i < 64; ++i ) s += static_cast<uint32_t>(V[i]); return s; }
static inline uint64_t wv_sum64(uint64_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,4,_ln); uint64_t s = 0; for ( int i = 0; i < 64; ++i ) s += V[i]; return s; }
static inline uint32_t wv_max(uint32_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,5,_ln); uint32_t s = 0; for ( int i = 0; i < 64; ++i ) s = static_cast<uint32_t>(V[i]) > s ? static_cast<uint32_t>(V[i]) : s; return s; }
static inline uint64_t wv_max64(uint64_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,6,_ln); uint64_t s = 0; for ( int i = 0; i < 64; ++i ) s = V[i] > s ? V[i] : s; return s; }
static inline uint64_t wv_min64(uint64_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,7,_ln); uint64_t s = ~0ull; for ( int i = 0; i < 64; ++i ) s = V[i] < s ? V[i] : s; return s; }
static inline int wv_any(int p, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(p ? 1 : 0,8,_ln); int r = 0; for ( int i = 0; i < 64; ++i ) r |= (V[i] != 0); return r; }
static inline uint32_t wv_or(uint32_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,9,_ln); uint32_t s = 0; for ( int i = 0; i < 64; ++i ) s |= static_cast<uint32_t>(V[i]); return s; }
static inline uint64_t wv_or64(uint64_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,10,_ln); uint64_t s = 0; for ( int i = 0; i < 64; ++i ) s |= V[i]; return s; }
static inline uint64_t wv_ballot(int p, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(p ? 1 : 0,11,_ln); uint64_t s = 0; for ( int i = 0; i < 64; ++i ) if ( V[i] ) s |= 1ull<<i; return s; }
static inline uint64_t wv_lanemask_lt() { return (1ull << wv_lane()) - 1ull; }
static inline uint32_t wv_bcast(uint32_t v, int src, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,12,_ln); return static_cast<uint32_t>(V[src]); }
static inline uint64_t wv_bcast64(uint64_t v, int src, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,13,_ln); return V[src]; }
static inline uint32_t wv_uni(uint32_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,14,_ln); return static_cast<uint32_t>(V[0]); }
static inline uint64_t wv_uni64(uint64_t v, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,15,_ln); return V[0]; }
// value of lane `src` (per-lane source): __shfl
static inline uint32_t wv_shfl(uint32_t v, int src, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,16,_ln); return static_cast<uint32_t>(V[src & 63]); }
static inline uint64_t wv_shfl64(uint64_t v, int src, int const _ln = __builtin_LINE()) { uint64_t const * V = wave_collect(v,17,_ln); return V[src & 63]; }
static inline int dacc_popc64(uint64_t v) { return __builtin_popcountll(v); }
static inline void atomicOrFlag(uint32_t * f) { *f |= 1u; }
// LDS / global atomics of the device code (lanes never run concurrently here)
template<typename T> static inline T wv_atomic_add(T * p, T const v) { T const o = *p; *p = o + v; return o; }
static inline uint32_t wv_atomic_add_global(uint32_t * p, uint32_t const v) { uint32_t const o = *p; *p = o + v; return o; }

}
#endif
