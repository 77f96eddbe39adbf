# Round 3, GPU call 19: the size-class prototype with T0INST = 472 (3 % hand-overs from tier 0 instead of 35 %)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3p; mkdir -p $R/$O; cd $R
( DACC_LIB=$R/daccord_amd/libvar_sc472.so timeout 50 python bench.py --reads 3000 --steps 3 --warmup 1 --no-cpu ) > $O/var_sc472.log 2>&1
grep '^{' $O/var_sc472.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['parity']['gpu_fasta_sha256_all'][:16])
"
