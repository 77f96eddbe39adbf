# Round 3, GPU call 18: the size-class prototype (tier 0 in front of tier 1; scripts/next/size_classes.patch, not part of the build)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3o; mkdir -p $R/$O; cd $R
( DACC_LIB=$R/daccord_amd/libvar_sizeclasses.so timeout 60 python bench.py --reads 3000 --steps 3 --warmup 1 --no-cpu ) > $O/var_sizeclasses.log 2>&1
( timeout 60 python bench.py --reads 3000 --steps 3 --warmup 1 --no-cpu ) > $O/var_default.log 2>&1
for f in $O/var_*.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['parity']['gpu_fasta_sha256_all'][:16])
except Exception as e:
    print('no json', e)
"; tail -n 2 $f | cut -c1-200; done
