# Closing measurements of round 5 on its final build: the driver's own invocation shape (20 timed steps, 5 warm-ups), the front end from files to FASTA,
# the fine-site ledger of the final build
R=$GRAFT_REPO_ROOT; TAG=r05ah; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu ) > $O/bench_default_20steps.log 2>&1; grep '^{' $O/bench_default_20steps.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['steps'], r['warmup'], r['parity']['identical'], r['parity']['piles_compared_distinct'], r['roofline']['kernel_ms'])"
( timeout 600 python scripts/cli_end_to_end.py ) > $O/cli_end_to_end.log 2>&1; grep -E "front end|no profile" $O/cli_end_to_end.log | cut -c1-200 | head -4
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; head -n 24 $O/sites_cfg2_256piles.log | cut -c1-170
