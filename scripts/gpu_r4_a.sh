# Round 4, GPU call 1: tier 0 merged -- threshold sweep; PMC passes with lane utilisation / in-flight levels / L2 hit rate
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
( timeout 400 python scripts/sweep_env.py 3000 3 "DACC_TIERS=7" "" "DACC_T0INST=440" "DACC_T0INST=456" "DACC_T0INST=488" "DACC_T0INST=504" "DACC_T0INST=520" ) > $O/sweep_t0.log 2>&1
cat $O/sweep_t0.log | grep '^{' | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  N=$(echo $C | cut -d' ' -f2)
  ( timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --reads 3000 --steps 1 --warmup 0 --no-cpu ) > $O/pmc_$N.log 2>&1; echo "rc=$?" >> $O/pmc_$N.log
  f=$(find $O/pmc_$N -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dacc::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    if k.startswith("k_window") or k.startswith("k_trace"):
        print(k, len(cnt[k]), {c: v / len(cnt[k]) for c, v in acc[k].items()})
P
  find $O/pmc_$N -name "*.csv" ! -name "*counter_collection.csv" -delete
done
