# Run HERE (no GPU needed) before `gpurun -- bash scripts/gpu_r3_first.sh`: profiling builds of the library with smaller code
# (the window kernels are 177-196 KB each against a 64 KB instruction cache shared by two CUs).
cd "$(dirname "$0")/.."
SRC="daccord_amd/csrc/capi.hip daccord_amd/csrc/host_tables.cpp daccord_amd/csrc/host_piles.cpp daccord_amd/csrc/host_io.cpp daccord_amd/csrc/host_eprof.cpp"
COMMON="--offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DDACC_PROFILE"
python -c "from daccord_amd import build; build.build_all(); build.build_prof()"
/opt/rocm/bin/hipcc $COMMON -Os -o daccord_amd/libvar_Os_prof.so $SRC &
/opt/rocm/bin/hipcc $COMMON -O3 -fno-unroll-loops -o daccord_amd/libvar_nounroll_prof.so $SRC &
/opt/rocm/bin/hipcc $COMMON -O2 -o daccord_amd/libvar_O2_prof.so $SRC &
# tier 2 with 1040 weight entries (81 808 bytes): do two workgroups still share a CU?  (tiers_ms[1] and tiers_out in the phase log)
/opt/rocm/bin/hipcc $COMMON -O3 -DDACC_T2_WCAP=1040 -o daccord_amd/libvar_t2w1040_prof.so $SRC &
wait
ls -la daccord_amd/*.so
