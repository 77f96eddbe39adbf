# instruction cache counters of the window kernels + phase profiles of smaller-code builds
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2i; mkdir -p $R/$O; cd /tmp; export TMPDIR=/tmp
( timeout 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc_icache -o pmc -- python $R/bench.py --reads 1000 --steps 1 --warmup 0 --no-cpu ) > $R/$O/pmc_icache.log 2>&1; echo "rc=$?" >> $R/$O/pmc_icache.log
( timeout 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $R/$O/pmc_ifetch -o pmc -- python $R/bench.py --reads 1000 --steps 1 --warmup 0 --no-cpu ) > $R/$O/pmc_ifetch.log 2>&1; echo "rc=$?" >> $R/$O/pmc_ifetch.log
cd $R
for V in libdaccord_hip_prof libvar_Os_prof libvar_nounroll_prof; do
  ( DACC_LIB=$R/daccord_amd/$V.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases_$V.log 2>&1
done
for f in $O/phases_*.log; do echo $f; grep -v amdgpu $f | grep "total cyc" | tail -2; done
tail -n 3 $O/pmc_icache.log | cut -c1-200; tail -n 3 $O/pmc_ifetch.log | cut -c1-200
python3 - <<'PY'
import csv, collections, os
for d in ("pmc_icache","pmc_ifetch"):
    fn = os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r2i",d,"pmc_counter_collection.csv")
    if not os.path.exists(fn):
        print("missing", fn); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "k_window_fast" in k or "k_trace" in k: print(d, k, dict(v))
PY
