# One GPU call, most important first (the call may be cut by the GPU-minute budget): scale parity (incl. the 54x slice),
# default bench, rocprof kernel stats, the rest of the GPU tests, smoke, phase profile, PMC passes at the headline workload.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2z
mkdir -p $R/$O
cd $R
( timeout 300 python -m pytest tests/test_gpu_scale.py -x -q -s ) > $O/scale.log 2>&1; echo "pytest rc=$?" >> $O/scale.log
( timeout 420 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
cd /tmp && export TMPDIR=/tmp
( timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/stats.log 2>&1; echo "rc=$?" >> $R/$O/stats.log
cd $R
( timeout 480 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_scale.py ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 150 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1; echo "rc=$?" >> $O/phases.log
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  ( timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/pmc_$N.log 2>&1; echo "rc=$?" >> $R/$O/pmc_$N.log
done
cd $R
grep -v amdgpu.ids $O/scale.log | tail -n 14; tail -n 2 $O/bench_default.log | cut -c1-1500; tail -n 4 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; grep "window=\|total" $O/phases.log | head; head -8 $O/stats/st_kernel_stats.csv 2>/dev/null
