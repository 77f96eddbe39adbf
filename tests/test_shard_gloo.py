"""N>1 path on CPU: two gloo ranks shard the piles like -J g,G, produce fragments with the CPU oracle (the test
checker; no GPU here), gather to rank 0, and rank 0 compares with the unsharded run."""
import os
import sys
import socket
import numpy as np
import pytest
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


def _worker_edge(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from daccord_amd import shard
    from daccord_amd._structs import DaccFragment
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = np.dtype(DaccFragment)
    # the preflight bench.py runs before it generates data: every rank sees every rank, on the backend's own primitives
    pf = shard.preflight()
    assert pf["ranks"] == list(range(world)) and pf["backend"] == "gloo", pf
    # rank 1 has nothing to contribute; the others have r+1 fragments of different lengths
    nf = 0 if rank == 1 else rank + 1
    f = np.zeros(nf, dtype=dt); bases = b""
    for i in range(nf):
        s = bytes([65 + rank]) * (10 + i)
        f[i]["aread"] = 100 * rank + i; f[i]["first"] = i; f[i]["last"] = i + len(s); f[i]["len"] = len(s); f[i]["seq_off"] = len(bases)
        bases += s
    F, B = shard.gather_fragments(f, bases)            # device from the backend: host arrays for gloo
    assert shard.last_transport == os.environ.get("DACC_GATHER", "p2p")
    if rank == 0:
        # a second gather overwrites the buffer the first result aliases unless a copy was asked for (documented)
        F2, B2 = shard.gather_fragments(f, bases, copy=True)
        assert isinstance(B2, bytes) and bytes(B) == B2 and [int(x) for x in F2["aread"]] == [int(x) for x in F["aread"]]
        ok = len(F) == 1 + 0 + 3 and [int(x) for x in F["aread"]] == [0, 200, 201, 202]
        ok = ok and all(B[int(x["seq_off"]):int(x["seq_off"]) + int(x["len"])] == bytes([65 + int(x["aread"]) // 100]) * int(x["len"]) for x in F)
        q.put(ok)
    else:
        assert F is None and B is None
        assert shard.gather_fragments(f, bases, copy=True) == (None, None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["p2p", "padded"])
def test_gather_with_an_empty_rank_and_uneven_sizes(transport, monkeypatch):
    monkeypatch.setenv("DACC_GATHER", transport)      # read by every rank: the transport is chosen by configuration, collectively
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, 3, port, q)) for r in range(3)]
    for pr in procs:
        pr.start()
    ok = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert ok


def _worker_hip(rank, world, port, q):
    """two ranks on the one GPU of the test box: each corrects its -J part with the HIP engine, gloo gather to rank 0"""
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from daccord_amd import shard, engine
    from daccord_amd._structs import default_params
    from daccord_amd.synth import SynthData
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = SynthData(60000, 120, 3000, seed=5)
    ovl, piles = engine.pile_select(d.ovl, d.piles)
    piles = piles[:12]
    p = default_params(k=8, device=0)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    mine = shard.shard_piles(piles, rank, world)
    f, b = E(mine, ovl, d.trace)
    F, B = shard.gather_fragments(f, b, device="cpu")
    if rank == 0:
        O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
        fo, bo = O.run(piles, ovl, d.trace, nthreads=4)
        q.put((engine.fasta(F, B) == pyoracle.fasta(fo, bo), len(F), int(len(mine))))
    dist.barrier()
    dist.destroy_process_group()



@pytest.mark.gpu
def test_two_ranks_with_the_hip_engine():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hip, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    same, n, nmine = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert same and n > 0 and 0 < nmine < 12


def _worker_rccl1(q):
    """ONE rank on the one GPU of the test box over the REAL backend (nccl = RCCL): what an 8-GPU node adds to the gloo tests above and no
    CPU test can execute -- backend initialisation with a device id, device-resident message tensors, the all_gather of the counts on
    RCCL, the device receive buffers, the copy into the pinned host buffer and the seq_off rebasing (gather_fragments(force=True));
    the point-to-point sends need a second device and stay untested here."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from daccord_amd import shard
    from daccord_amd._structs import DaccFragment
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        pf = shard.preflight("cuda", device_index=0)
        dt = np.dtype(DaccFragment)
        f = np.zeros(3, dtype=dt); bases = b""
        for i in range(3):
            s = bytes([65 + i]) * (100 + 7 * i)
            f[i]["aread"] = i; f[i]["first"] = i; f[i]["last"] = i + len(s); f[i]["len"] = len(s); f[i]["seq_off"] = len(bases)
            bases += s
        F, B = shard.gather_fragments(f, bases, force=True)
        ok = (shard.last_transport == "p2p" and bytes(B) == bases and F.tobytes() == f.tobytes())
        F2, B2 = shard.gather_fragments(f[:0], b"", force=True)      # an empty shard
        ok = ok and len(F2) == 0 and len(B2) == 0
        q.put((ok, pf))
        dist.destroy_process_group()
    except Exception as ex:      # reported, not hung
        q.put((False, repr(ex)))


@pytest.mark.gpu
def test_rccl_backend_with_device_buffers_on_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        pr = ctx.Process(target=_worker_rccl1, args=(q,))
        pr.start()
        ok, pf = q.get(timeout=300)
        pr.join(timeout=120)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ok, pf
    assert pf["ranks"] == [0] and pf["devices"] == [0] and pf["backend"] == "nccl"
