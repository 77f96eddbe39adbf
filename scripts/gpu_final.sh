# last GPU call of a round: gpu_evidence.sh on the final sources, then the k sweep of config 5 as bench lines with live parity, the driver-shape bench
# and the C++ front end from files to FASTA on 30 000 reads
R=$GRAFT_REPO_ROOT; TAG=${1:?tag}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_evidence.sh $TAG
for k in 10 12 14 16; do
  ( timeout 300 python bench.py --ont --reads 4000 --k $k --steps 3 --warmup 2 --no-cpu --live-parity 8 ) > $O/bench_ont_k$k.log 2>&1
  echo "ONT k=$k"; grep '^{' $O/bench_ont_k$k.log | tail -n 1 | python scripts/bench_brief.py
done
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu ) > $O/bench_default_20_steps_5_warmups.log 2>&1; grep '^{' $O/bench_default_20_steps_5_warmups.log | tail -n 1 | python scripts/bench_brief.py
( timeout 500 python scripts/cli_end_to_end.py 30000 /tmp/dacc_e2e ) > $O/cli_end_to_end_30000.log 2>&1; tail -n 6 $O/cli_end_to_end_30000.log | cut -c1-300
true
