# One parameterised entry point for the GPU calls of a round (replaces the per-call wrappers of round 5).
#   gpurun --timeout T -- 'bash scripts/gpu_steps.sh <tag> <step> [<step> ...]'      -> gpurun_out/<tag>/
# steps:  quick          a parity subset of the GPU suite + smoke (is the library sound?)
#         suite          the whole GPU suite + smoke
#         sweep:<reads>:<steps>:<setting>[;<setting>...]   scripts/sweep_env.py in one process ("" = defaults; settings are K=V,K=V)
#         ab:<reads>:<steps>:<lib>[;<lib>...]   one process per library (DACC_LIB), twice, "default" = the product
#         pmc:<name>:<reads>:<env or ->:<counter>[,<counter>...]   one rocprofv3 --pmc pass of bench.py (--no-cpu, one step)
#         counters       list of the counters the box offers
#         ledger[:reads]  scripts/ledger.py: instruction / wave-time ledger by phase on daccord_amd/libvar_ledger.so
#         bench[:args]   python bench.py [args with , for spaces]  -> bench_<n>.log
#         stats          rocprofv3 --kernel-trace --stats of the default bench command
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
nb=0; ns=0
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  case $kind in
    quick)
      ( timeout 600 python -m pytest tests/test_abi.py tests/test_gpu_parity.py -x -q -m gpu -k "loaded or tables or windows_and_fragments or size_class or deep_batch or long_strings or empty_shallow or golden" ) > $O/pytest_quick.log 2>&1; echo "pytest rc=$?" >> $O/pytest_quick.log; tail -n 4 $O/pytest_quick.log
      ( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log ;;
    suite)
      ( timeout 1200 python -m pytest tests -x -q -m gpu -rs --durations=6 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -n 12 $O/pytest_gpu.log
      ( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log ;;
    sweep)
      IFS=: read -r reads steps settings <<< "$rest"
      IFS=';' read -r -a S <<< "$settings"; [ ${#S[@]} -eq 0 ] && S=("")
      ns=$((ns+1)); ( timeout 900 python scripts/sweep_env.py $reads $steps "${S[@]}" ) > $O/sweep_${reads}_$ns.log 2>&1; grep '^{' $O/sweep_${reads}_$ns.log | cut -c1-330 ;;
    ab)
      IFS=: read -r reads steps libs <<< "$rest"
      IFS=';' read -r -a Lb <<< "$libs"
      bash scripts/gpu_ab.sh $TAG $reads $steps "${Lb[@]}" ;;
    info)
      ( nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -3; lscpu | grep -i "model name\|^CPU(s)\|thread\|socket"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; rocm-smi --showmeminfo vram 2>/dev/null | head -8 ) > $O/box_info.log 2>&1; cat $O/box_info.log ;;
    counters)
      ( rocprofv3-avail list 2>&1 || rocprofv3 -L 2>&1 ) > $O/counters_avail.log; grep -c . $O/counters_avail.log ;;
    pmc)
      IFS=: read -r name reads envs ctrs <<< "$rest"
      [ "$envs" = "-" ] && envs=""
      ( cd /tmp && export TMPDIR=/tmp && env $(echo $envs | tr ',' ' ') timeout 400 rocprofv3 --pmc $(echo $ctrs | tr ',' ' ') --kernel-trace --output-format csv -d $O/pmc_$name -o pmc -- python $R/bench.py --reads $reads --steps 1 --warmup 0 --no-cpu ) > $O/pmc_$name.log 2>&1; echo "rc=$?" >> $O/pmc_$name.log
      f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ]; then python scripts/pmc_by_kernel.py $f > $O/pmc_$name.txt; cat $O/pmc_$name.txt | cut -c1-220; gzip -c $f > $O/pmc_$name.csv.gz; fi
      rm -rf $O/pmc_$name; tail -n 2 $O/pmc_$name.log | cut -c1-300 ;;
    bench)
      nb=$((nb+1)); a=$(echo "$rest" | tr ',' ' ')
      ( timeout 900 python bench.py $a ) > $O/bench_$nb.log 2>&1; echo "rc=$? args=$a" >> $O/bench_$nb.log
      grep '^{' $O/bench_$nb.log | tail -n 1 | python scripts/bench_brief.py ;;
    ledger)
      IFS=: read -r reads extra <<< "$rest"
      ( timeout 1500 python scripts/ledger.py $O/ledger ${reads:-1500} $(echo $extra | tr ',' ' ') ) > $O/ledger.log 2>&1; echo "rc=$?" >> $O/ledger.log; tail -n 75 $O/ledger.log | cut -c1-230 ;;
    stats)
      ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $O/stats.log 2>&1; echo "rc=$?" >> $O/stats.log
      find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats.csv \; ; rm -rf $O/stats; head -8 $O/rocprof_kernel_stats.csv | cut -c1-200 ;;
    *) echo "unknown step $step" ;;
  esac
done
true
