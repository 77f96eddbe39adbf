/* k_window_fast<11>: one LDS tier of the window kernel per translation unit (window_kernels.hpp) */
#define DACC_INSTANTIATE_TIER 11
#include "window_kernels.hpp"
template __global__ void k_window_fast<11>(FastBatch, uint32_t const *, uint32_t *);
