// TEST HARNESS: the threaded host planner (BatchPlan::plan) under ThreadSanitizer on a synthetic batch, 1 / 8 / 3 / 16 threads, same digest
// (tests/test_plan.py::test_plan_has_no_data_race builds this with g++ -fsanitize=thread and runs it)
#define DACC_EMUL 1
#include "../../daccord_amd/csrc/wave.hpp"
#include "../../daccord_amd/csrc/batch_plan.hpp"
#include <random>
#include <cstring>
#include <cstdio>
using namespace dacc;
int main()
{
	uint64_t const nreads = 600, L = 3000; int const ts = 100;
	std::vector<uint32_t> rlen(nreads,L);
	std::mt19937_64 rng(5);
	std::vector<dacc_pile> P; std::vector<dacc_overlap> O; std::vector<uint8_t> T;
	for ( uint64_t a = 0; a < nreads; ++a )
	{
		dacc_pile p; std::memset(&p,0,sizeof(p)); p.aread = a; p.first_ovl = O.size(); p.novl = 0;
		uint32_t n = 5 + rng()%40; std::vector<uint32_t> st(n); for ( auto & s : st ) s = rng()%2000; std::sort(st.begin(),st.end());
		for ( uint32_t z = 0; z < n; ++z )
		{
			dacc_overlap o; std::memset(&o,0,sizeof(o));
			o.aread = a; o.bread = rng()%nreads; o.flags = rng()&1; o.abpos = st[z]; o.aepos = std::min<uint64_t>(L,st[z]+500+rng()%500);
			int64_t nblk = (o.aepos+ts-1)/ts - o.abpos/ts; o.tlen = 2*nblk; o.trace_off = T.size(); o.bbpos = rng()%100; uint64_t bs = 0;
			int64_t pos = o.abpos;
			for ( int64_t b = 0; b < nblk; ++b ) { int64_t e = std::min<int64_t>(o.aepos,(pos/ts+1)*ts); uint8_t bl = e-pos; T.push_back(rng()%5); T.push_back(bl); bs += bl; pos = e; }
			o.bepos = o.bbpos + bs; o.diffs = rng()%60;
			if ( static_cast<uint32_t>(o.bepos) > L ) { T.resize(o.trace_off); continue; }
			O.push_back(o); ++p.novl;
		}
		P.push_back(p);
	}
	dacc_params par; std::memset(&par,0,sizeof(par)); par.w = 40; par.a = 10; par.klow = par.khigh = 8; par.maxfilterfreq = 2; par.minwindowcov = 3; par.maxalign = ~0ull; par.eminrate = ~0ull; par.tspace = ts;
	std::string err; uint64_t h0 = 0;
	for ( int th : {1,8,3,16} )
	{
		setenv("DACC_PLAN_THREADS",std::to_string(th).c_str(),1);
		BatchPlan BP; int rc = BP.plan(par,P.data(),P.size(),O.data(),O.size(),T.data(),T.size(),1,rlen.data(),nreads,err,41,60);
		uint64_t h = 1469598103934665603ull; auto f = [&](void const * p, size_t n){ uint8_t const * b = (uint8_t const*)p; for ( size_t i = 0; i < n; ++i ) { h ^= b[i]; h *= 1099511628211ull; } };
		f(BP.ovl.data(),BP.ovl.size()*sizeof(DevOvl)); f(BP.ovl_pile.data(),BP.ovl_pile.size()*4); f(BP.fragbase.data(),BP.fragbase.size()*8);
		std::printf("threads %d rc %d piles %zu ovl %zu windows %llu maxdepth %u digest %016llx\n",th,rc,BP.piles.size(),BP.ovl.size(),(unsigned long long)BP.nwindows,BP.maxdepth,(unsigned long long)h);
		if ( !h0 ) h0 = h; else if ( h != h0 ) { std::printf("MISMATCH\n"); return 1; }
	}
	// an exception inside a worker thread (a bad_alloc of a per-thread buffer) must reach the caller, after all threads have been joined,
	// and never terminate the process: thrown on whichever thread takes the chunk at 640, with 8 threads and with 1
	for ( unsigned th : {8u,1u} )
	{
		std::atomic<uint64_t> done(0); bool caught = false;
		try
		{
			BatchPlan::planParallel(4096,th,[&](unsigned, uint64_t const lo, uint64_t const hi) { if ( lo <= 640 && 640 < hi ) throw std::bad_alloc(); done += hi-lo; });
		}
		catch ( std::bad_alloc const & ) { caught = true; }
		std::printf("exception threads %u caught %d chunks done before the stop %llu\n",th,int(caught),(unsigned long long)done.load());
		if ( !caught || done.load() >= 4096 ) { std::printf("MISMATCH\n"); return 1; }
	}
	return 0;
}
