#ifndef DACC_HOST_TABLES_HPP
#define DACC_HOST_TABLES_HPP
#include <vector>
#include <cstdint>
namespace dacc {
struct HostTables
{
	uint32_t nrows, nsup, kln, nk;
	std::vector<double> dpnorm, dpsq;
	std::vector<uint64_t> dpsq_vs, dpsq_vst;
	std::vector<uint32_t> tab32;      // [nsup+1][nrows+1] low words of dpsq_vst, zero row / zero position at the end (gw tiers read it from global memory)
	std::vector<uint16_t> dpsq_first, dpsq_size, suplo, suphi;
	std::vector<uint32_t> klim;
	std::vector<uint32_t> firsts, rowsizes;
};
void buildHostTables(HostTables & H, uint32_t w, double p_i, double p_d, double est_cor, uint32_t klow, uint32_t khigh, uint32_t kln);
void serialiseHostTables(HostTables const & H, std::vector<uint64_t> & B, uint32_t klimit_n);
}
#endif
