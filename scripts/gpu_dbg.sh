mkdir -p gpurun_out
T='tests/test_gpu_parity.py -m gpu -x -q -k test_windows_and_fragments'
( DACC_SCHED=1 timeout 60 python -m pytest $T ) > gpurun_out/dbg1.log 2>&1; A=$?; echo "A(fast dyn, generic static) rc=$A" >> gpurun_out/dbg1.log
( DACC_SCHED=0 DACC_NOFAST=1 timeout 90 python -m pytest $T ) > gpurun_out/dbg2.log 2>&1; B=$?; echo "B(nofast static) rc=$B" >> gpurun_out/dbg2.log
( DACC_SCHED=0 timeout 60 python -m pytest $T ) > gpurun_out/dbg3.log 2>&1; C=$?; echo "C(all static) rc=$C" >> gpurun_out/dbg3.log
tail -n 2 gpurun_out/dbg1.log; tail -n 2 gpurun_out/dbg2.log; tail -n 2 gpurun_out/dbg3.log
