R=$GRAFT_REPO_ROOT; TAG=r05ag; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_step9.so default
( timeout 200 python scripts/sweep_env.py 1500 2 "" ) > /dev/null 2>&1
for lib in libvar_step9.so libdaccord_hip.so; do ( DACC_LIB=$R/daccord_amd/$lib timeout 300 python bench.py --ont --reads 2000 --steps 2 --warmup 1 --no-cpu --live-parity 0 ) > $O/ont_$lib.log 2>&1; echo "ONT $lib: $(grep '^{' $O/ont_$lib.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'])")"; done
