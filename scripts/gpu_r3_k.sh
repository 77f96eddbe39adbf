# Round 3, GPU call 13: the front end's --gpus mode (three device workers wrapping around on the one device) through the CLI test
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3k; mkdir -p $R/$O; cd $R
( timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli" --durations=3 ) > $O/pytest_cli.log 2>&1; echo "pytest rc=$?" >> $O/pytest_cli.log
tail -n 8 $O/pytest_cli.log
