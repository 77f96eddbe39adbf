# Build HERE (no GPU): experiment variants of the library for the GPU calls of round 3 (selected with DACC_LIB=<path>).
# usage: bash scripts/r3_build_variants.sh name:"flags" ...     e.g.  noreach:"-DDACC_NO_REACH" lean:"-DDACC_T1_LEAN -DDACC_TAB_GLOBAL"
cd "$(dirname "$0")/.."
SRC="daccord_amd/csrc/capi.hip daccord_amd/csrc/host_tables.cpp daccord_amd/csrc/host_piles.cpp daccord_amd/csrc/host_io.cpp daccord_amd/csrc/host_eprof.cpp"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc $COMMON $flags -o daccord_amd/libvar_$name.so $SRC && echo "built libvar_$name.so ($flags)" ) &
done
wait
