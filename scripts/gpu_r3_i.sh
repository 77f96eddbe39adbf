# Round 3, GPU calls 9 to 11: machine scheduler strategies for the device code (flags only), 3000 reads of the default workload each
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3i; mkdir -p $R/$O; cd $R
for V in maxilp itilp itminreg; do
  L=$R/daccord_amd/libvar_$V.so; [ -f $L ] || L=$R/daccord_amd/libdaccord_hip.so
  ( DACC_LIB=$L timeout 120 python bench.py --reads 3000 --steps 3 --warmup 1 --no-cpu ) > $O/var_$V.log 2>&1
done
for f in $O/var_*.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['parity']['gpu_fasta_sha256_all'][:16])
except Exception as e:
    print('no json', e)
"; done
true
