# Round 3, GPU call 17: a bench line in the shape of BASELINE config 3's reads (20x, 14 kb reads; 10 000 A reads on one GPU)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3n; mkdir -p $R/$O; cd $R
( timeout 150 python bench.py --readlen 14000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_cfg3shape.log 2>&1; echo "rc=$?" >> $O/bench_cfg3shape.log
grep '^{' $O/bench_cfg3shape.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['value'], r['ms_per_step'], r['config']['workload'], r['config']['windows_rank0'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['parity']['gpu_fasta_sha256_all'][:16], r['accuracy'].get('erate'))
"
tail -n 2 $O/bench_cfg3shape.log | cut -c1-300
