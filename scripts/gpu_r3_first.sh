# First GPU call of the next round (written at the end of round 2, when the GPU minutes were spent): verifies what was
# written after the last GPU run of round 2 (tier 5 / k_window_long, positions behind the table support, 256 wavefronts for
# the generic engine on the second stream), then the default bench and the code size experiment.
# Run scripts/r3_prepare_variants.sh first (here, no GPU).  About 9 minutes on the box.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $R/$O; cd $R
[ -f probe_in/cases.txt ] && bash scripts/gpu_probe.sh   # python scripts/make_probe.py first (here): 4 cases through the C++ front end in seconds, incl. `warp` (strings of 129..256 bases, never run on a GPU)
( timeout 600 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
( timeout 420 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_54x_2000piles.log 2>&1; echo "rc=$?" >> $O/bench_54x_2000piles.log
for V in libdaccord_hip_prof libvar_Os_prof libvar_nounroll_prof libvar_O2_prof libvar_t2w1040_prof; do
  [ -f daccord_amd/$V.so ] && ( DACC_LIB=$R/daccord_amd/$V.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases_$V.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
( timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/stats.log 2>&1; echo "rc=$?" >> $R/$O/stats.log
cd $R
tail -n 4 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; tail -n 2 $O/bench_default.log | cut -c1-1200
for f in $O/phases_*.log; do echo $f; grep -v amdgpu $f | grep "total cyc" | tail -2; done
head -8 $O/stats/st_kernel_stats.csv 2>/dev/null
# NOT to be repeated: rocprofv3 --pmc with SQC_ICACHE_* / SQ_IFETCH* hung both passes on this stack (round 2, 240 s lost)
