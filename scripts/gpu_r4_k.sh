# Round 4, GPU call: the deep second tier at 2 wavefronts per CU with 2048 nodes / 96 strings / 64 candidates (54x shape), parity of the deep cases
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
( timeout 150 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_54x_2000piles.log 2>&1
grep '^{' $O/bench_54x_2000piles.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); ro = r['roofline']
print(r['value'], r['ms_per_step'], ro['kernel_ms'], ro['windows_handed_on'], r['parity']['gpu_fasta_sha256_all'][:16])"
( timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_fuzz_wide.py tests/test_gpu_parity.py -x -q -m gpu -k "cfg4 or wide or deep or random_parameter_sets" --durations=3 ) > $O/pytest_deep.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deep.log
tail -n 5 $O/pytest_deep.log
