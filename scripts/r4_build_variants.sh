# Build HERE (no GPU): experiment variants of the library for the GPU calls of round 4 (selected with DACC_LIB=<path>), with the
# product's flags plus the variant's.   usage: bash scripts/r4_build_variants.sh name:"flags" ...     e.g.  xnackoff:"--offload-arch=gfx950:xnack-"
cd "$(dirname "$0")/.."
SRC="daccord_amd/csrc/capi.hip daccord_amd/csrc/host_tables.cpp daccord_amd/csrc/host_piles.cpp daccord_amd/csrc/host_io.cpp daccord_amd/csrc/host_eprof.cpp"
COMMON="-O3 -Xarch_device -Os -mllvm -amdgpu-sched-strategy=max-ilp -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  case "$flags" in *--offload-arch*) ARCH="";; *) ARCH="--offload-arch=gfx950";; esac
  ( /opt/rocm/bin/hipcc $ARCH $COMMON $flags -o daccord_amd/libvar_$name.so $SRC 2>/dev/null && echo "built libvar_$name.so ($flags)" ) &
done
wait
