# PC sampling of the window kernels on the GPU box (round 5, VERDICT r04 task 1a):
#   bash scripts/gpu_pcsample.sh <tag> [reads=1500]   -> gpurun_out/<tag>/pcs_<method>.json (+ logs); raw sample tables stay in /tmp
# The library is the -gline-tables-only variant (instruction streams identical to the product's, build.kernel_isa_hashes), so
# Instruction_Comment carries file:line of the inlined source.
R=$GRAFT_REPO_ROOT; TAG=${1:-r05a}; READS=${2:-1500}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DACC_LIB=$R/daccord_amd/libvar_lines.so
try () {  # name method unit interval
  N=$1; rm -rf /tmp/pcs_$N
  ( timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $2 --pc-sampling-unit $3 --pc-sampling-interval $4 --kernel-trace \
      --output-format csv -d /tmp/pcs_$N -o pcs -- python $R/scripts/sweep_env.py $READS 1 "" ) > $O/pcs_$N.log 2>&1; echo "rc=$?" >> $O/pcs_$N.log
  tail -n 4 $O/pcs_$N.log | cut -c1-300
  find /tmp/pcs_$N -name "*.csv" -exec ls -la {} \; >> $O/pcs_$N.log
  ( timeout 300 python $R/scripts/pc_aggregate.py /tmp/pcs_$N $O/pcs_$N.json ) >> $O/pcs_$N.log 2>&1
  f=$(find /tmp/pcs_$N -name "*pc_sampling*.csv" | head -1); [ -n "$f" ] && head -n 300 $f > $O/pcs_${N}_head.csv
}
try stoch20 stochastic cycles 1048576
try stoch16 stochastic cycles 65536
try trap host_trap time 4000
ls -la $O
