"""The N > 1 path of bench.py itself (VERDICT r02 task 6): two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...), in weak and in strong scaling mode, must return
the FASTA of the N = 1 run over the same reads: per-rank overlap generation, -J sharding, gather in rank order, sequential
well numbers.  The test box has ONE GPU, so the ranks share device 0 and talk over gloo (DACC_BENCH_ONE_DEVICE=1, a hook that
only changes the device ordinal and the backend); on a multi-GPU node the same script runs one rank per GPU over RCCL."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, reads, scaling, extra=()):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--reads", str(reads), "--readlen", "5000", "--steps", "1", "--warmup", "0", "--no-cpu", "--scaling", scaling] + list(extra)
    env = dict(os.environ, DACC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_weak_and_strong_equal_one_rank():
    one = _run(1, 240, "weak")
    weak = _run(2, 120, "weak")            # 2 x 120 reads = the same 240 read set
    strong = _run(2, 240, "strong")
    assert one["n_gpus"] == 1 and weak["n_gpus"] == 2 and weak["ranks"] == 2 and strong["ranks"] == 2
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    d1 = one["parity"]["gpu_fasta_sha256_all"]
    assert weak["parity"]["gpu_fasta_sha256_all"] == d1 and strong["parity"]["gpu_fasta_sha256_all"] == d1
    assert weak["config"]["corrected_bases_total"] == one["config"]["corrected_bases_total"] == strong["config"]["corrected_bases_total"]
    assert weak["config"]["piles_rank0"] == 120 and strong["config"]["piles_rank0"] == 120 and one["config"]["piles_rank0"] == 240


def test_eight_ranks_on_one_device_weak():
    """the shape of the driver's 8-GPU run (--gpus 8, weak scaling: --reads per rank) end to end on the one device of the test box:
    eight ranks, -J g,8 sharding of one data set, gather in rank order, rank 0's digests and accuracy block; equals the N = 1 run over
    the same 8 x 40 reads and reports which gather ran and how long rank 0 worked behind the timed loop"""
    one = _run(1, 320, "weak")
    eight = _run(8, 40, "weak", extra=["--live-parity", "4"])
    # round 5: a line the committed digests do not cover compares a sample of rank 0's own shard with the oracle and the reference build
    live = eight["parity"]["live"]
    assert live["identical"] is True and live["oracle"]["identical"] is True and live["oracle"]["piles"] == 4, live
    assert live["reference_build"].get("identical", True) is True, live
    # round 6: every rank samples ITS OWN shard, rank 0 collects the verdicts
    assert live["all_ranks"] == {"ranks_reporting": 8, "ranks_identical": 8, "piles_total": 32, "identical_on_every_rank": True}, live
    assert eight["preflight"]["ranks"] == list(range(8)) and eight["preflight"]["backend"] == "gloo", eight["preflight"]
    assert eight["n_gpus"] == 8 and eight["ranks"] == 8 and eight["gather"] == "p2p" and one["gather"] is None
    assert eight["parity"]["gpu_fasta_sha256_all"] == one["parity"]["gpu_fasta_sha256_all"]
    assert eight["config"]["corrected_bases_total"] == one["config"]["corrected_bases_total"] and eight["config"]["piles_rank0"] == 40
    assert eight["post_loop_s"] < 60 and "accuracy" in eight
