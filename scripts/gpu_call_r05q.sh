R=$GRAFT_REPO_ROOT; TAG=r05q; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_step5.so default daccord_amd/libvar_ntslab.so
cd /tmp && export TMPDIR=/tmp
for v in default ntslab; do
  if [ "$v" = "default" ]; then unset DACC_LIB; else export DACC_LIB=$R/daccord_amd/libvar_$v.so; fi
  for cn in WRITE_SIZE FETCH_SIZE; do
  ( timeout 200 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d /tmp/pmc_${v}_$cn -o pmc -- python $R/scripts/sweep_env.py 3000 1 "" ) > $O/pmc_${v}_$cn.log 2>&1
  f=$(find /tmp/pmc_${v}_$cn -name "*counter_collection.csv" | head -1)
  python - "$f" "$v" "$cn" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ","").replace("dacc::","")
    acc[k] += float(r["Counter_Value"]); 
    if (k,r["Dispatch_Id"]) not in seen: seen.add((k,r["Dispatch_Id"])); n[k]+=1
for k in sorted(acc):
    if "window_fast" in k: print(sys.argv[2], sys.argv[3], k, "launches", n[k], "KB per launch %.0f" % (acc[k]/n[k]))
PY
  done
done
