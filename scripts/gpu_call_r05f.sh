R=$GRAFT_REPO_ROOT; TAG=r05f; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=6 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 12 $O/pytest_gpu.log
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu --live-parity 8 ) > $O/bench_54x_2000piles.log 2>&1
( timeout 300 python bench.py --ont --reads 4000 --steps 2 --warmup 1 --no-cpu --live-parity 16 ) > $O/bench_ont_4000piles.log 2>&1
for f in $O/bench_54x_2000piles.log $O/bench_ont_4000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); ro = r['roofline']
print(r['value'], r['ms_per_step'], ro['kernel_ms'], ro['kernel'], ro['achieved'], ro['frac'], ro['frac_step'], r['parity'].get('live'), r.get('post_loop_s'))
"; done
