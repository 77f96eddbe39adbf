# Round 3, GPU call 5: (1) core GPU tests on the rebuilt trace kernel (checkpoints in a global slab, 8 wavefronts per CU) and
# the ASCII symbol stream, (2) default bench, (3) occupancy curve of tier 1 (DACC_LDS_T1: 3, 4, 5, 6 wavefronts per CU on
# the same kernel), (4) code size variants (-Os, -fno-unroll-loops: the instruction cache question of DESIGN section 5),
# (5) which windows tier 3 hands to the generic engine at config 2, and why.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $R/$O; cd $R
( timeout 600 python -m pytest tests -x -q -m gpu -k "not fifty and not wide and not ranks" --durations=5 ) > $O/pytest_gpu_core.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_core.log
tail -n 9 $O/pytest_gpu_core.log
( timeout 300 python bench.py --no-cpu ) > $O/bench_default_nocpu.log 2>&1; echo "rc=$?" >> $O/bench_default_nocpu.log
for L in 0 32768 40960 54000; do
  ( DACC_LDS_T1=$L timeout 120 python bench.py --reads 3000 --steps 2 --warmup 1 --no-cpu ) > $O/occ_lds$L.log 2>&1
done
for V in Os nounroll; do
  [ -f daccord_amd/libvar_$V.so ] && ( DACC_LIB=$R/daccord_amd/libvar_$V.so timeout 120 python bench.py --reads 3000 --steps 2 --warmup 1 --no-cpu ) > $O/var_$V.log 2>&1
done
( timeout 200 python scripts/dbg_retry.py 14 10000 ) > $O/dbg_retry_cfg2.log 2>&1
for f in $O/bench_default_nocpu.log $O/occ_lds*.log $O/var_*.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16])
except Exception as e:
    print('no json', e)
"; done
grep -v amdgpu $O/dbg_retry_cfg2.log | tail -8
true
