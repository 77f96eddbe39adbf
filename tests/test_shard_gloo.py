"""N>1 path on CPU: two gloo ranks shard the piles like -J g,G, produce fragments with the CPU oracle (the test
checker; no GPU here), gather to rank 0, and rank 0 compares with the unsharded run."""
import os
import sys
import socket
import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from daccord_amd import shard
    from daccord_amd._structs import default_params
    from daccord_amd.synth import SynthData
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = SynthData(60000, 120, 3000, seed=5)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    piles = piles[:10]
    p = default_params(k=8)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    mine = shard.shard_piles(piles, rank, world)
    f, b = O.run(mine, ovl, d.trace, nthreads=2)
    F, B = shard.gather_fragments(f, b, device="cpu")
    if rank == 0:
        fo, bo = O.run(piles, ovl, d.trace, nthreads=2)
        q.put((pyoracle.fasta(F, B) == pyoracle.fasta(fo, bo), len(F), len(fo), int(len(mine))))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_matches_reference_partition():
    from daccord_amd.shard import shard_range
    # src/daccord.cpp:1156-1183: partsize = ceil(n/G), last parts may be short or empty
    assert [shard_range(0, 10, g, 4) for g in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(5, 7, g, 4) for g in range(4)] == [(5, 6), (6, 7), (7, 7), (7, 7)]
    cover = [shard_range(100, 1123, g, 8) for g in range(8)]
    assert cover[0][0] == 100 and cover[-1][1] == 1123 and all(cover[i][1] == cover[i + 1][0] for i in range(7))


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    same, n, nref, nmine = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert same and n == nref and 0 < nmine < 10
