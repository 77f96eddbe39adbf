// Design aid (host only): byte offsets of the LDS regions of the gw tiers, from the same constexpr layout the kernels use.
//   g++ -std=c++17 -o /tmp/lds_layout scripts/lds_layout.cpp && /tmp/lds_layout
#define DACC_EMUL 1
#include <vector>
#include <algorithm>
#include <cstdio>
#include "../daccord_amd/csrc/fast_window.hpp"
using namespace dacc;
template<int T> void dump()
{
	typedef FastLds<FastTier<T>> L;
	printf("tier %d: bytes %u = %u granules of 1280 | P [0,%u) S [%u,%u) = %u | pools [%u,%u): reverse %u forward %u tables %u | ubase %u: overlay A ..%u, overlay B ..%u (x scratch from %u) | slab %u B\n",
		T,L::uend,(L::uend+1279)/1280,L::sbase,L::sbase,L::send,L::sbytes,L::sbase,L::upool,L::e_rfmask-L::o_rc_w,L::e_fp_adj-L::o_f_w,L::upool-L::e_fp_adj,L::ubase,L::uA,L::uB,L::xbase,L::g_bytes);
}
int main() { dump<0>(); dump<7>(); dump<1>(); dump<6>(); dump<4>(); dump<2>(); dump<3>(); dump<8>(); dump<9>(); dump<10>(); dump<11>(); return 0; }
