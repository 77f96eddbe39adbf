"""Multi-GPU layer of the path: static sharding of A-read piles and the final gather of corrected fragments.

Piles are independent (src/daccord.cpp:2107-2112), so there is no data-path collective: rank g of G takes the A-read
range the reference's `-J g,G` option names (src/daccord.cpp:1156-1183), runs its piles on its own GPU with the read
store replicated, and the fragments are gathered to rank 0 in rank order (= ascending read order, the order of
src/daccord.cpp:2491-2534).  The gather uses torch.distributed (RCCL on the GPU box, gloo in the CPU tests); it is the
only communication of a step.
"""
import numpy as np


def shard_range(lo, hi, g, G):
    """A-read interval [lo,hi) -> the part of rank g of G (src/daccord.cpp:1156-1183: partsize = ceil(n/G))."""
    n = max(0, hi - lo)
    part = (n + G - 1) // G if G > 0 else n
    a = min(hi, lo + g * part)
    b = min(hi, lo + (g + 1) * part)
    return a, b


def shard_piles(piles, g, G):
    """Piles (sorted by aread) of rank g of G: split by A-read id range like -J g,G."""
    if len(piles) == 0:
        return piles
    lo, hi = int(piles["aread"].min()), int(piles["aread"].max()) + 1
    a, b = shard_range(lo, hi, g, G)
    return piles[(piles["aread"] >= a) & (piles["aread"] < b)]


def preflight(device=None, device_index=None, dst=0):
    """All ranks call, right after init_process_group and BEFORE anything expensive: one tiny all_gather and one point-to-point round on
    the real backend -- the two primitives gather_fragments uses -- so that a missing rank, two ranks on one device, or a backend that
    cannot send fails in seconds and says what it saw, not after minutes of data generation inside the first timed step.  Returns
    {"ranks": [...], "devices": [...], "backend": ...} on every rank; raises RuntimeError with the ranks / devices seen otherwise."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks": [0], "devices": [device_index], "backend": None}
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    me = torch.tensor([rank, -1 if device_index is None else int(device_index)], dtype=torch.int64, device=device)
    seen = [torch.full((2,), -7, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(seen, me)
    seen = torch.stack(seen).cpu().numpy()
    ranks, devs = [int(x) for x in seen[:, 0]], [int(x) for x in seen[:, 1]]
    if ranks != list(range(world)):
        raise RuntimeError("preflight: all_gather over %s returned ranks %r, expected 0..%d" % (dist.get_backend(), ranks, world - 1))
    if device_index is not None and str(device) != "cpu" and len(set(devs)) != world:
        raise RuntimeError("preflight: %d ranks share devices %r -- RCCL needs one device per rank" % (world, devs))
    # every rank sends 8 bytes (its rank, twice) to rank dst with batch_isend_irecv, as the gather does
    msg = torch.tensor([rank, rank], dtype=torch.int32, device=device)
    if rank != dst:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, msg, dst)]):
            q.wait()
    else:
        got = torch.full((world, 2), -1, dtype=torch.int32, device=device)
        ops = [dist.P2POp(dist.irecv, got[r], r) for r in range(world) if r != dst]
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        got = got.cpu().numpy()
        bad = [r for r in range(world) if r != dst and (int(got[r][0]) != r or int(got[r][1]) != r)]
        if bad:
            raise RuntimeError("preflight: point-to-point messages of ranks %r did not arrive intact at rank %d (got %r)" % (bad, dst, got.tolist()))
    dist.barrier()
    return {"ranks": ranks, "devices": devs, "backend": dist.get_backend(), "primitives": ["all_gather", "batch_isend_irecv"]}


_pinned = {}


def _host_buffer(n, device):
    """A reusable host buffer of at least n bytes (pinned when the data comes from a GPU: the device-to-host copy of the
    gathered bases then runs at PCIe speed instead of through a pageable staging copy)."""
    import torch
    key = str(device)
    t = _pinned.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=(key != "cpu" and torch.cuda.is_available()))
        _pinned[key] = t
    return t


# which gather ran last ("p2p" | "padded" | "local"); bench.py echoes it in its JSON line
last_transport = "local"


def _transport():
    """The gather's transport, decided ONCE and by configuration alone -- never by catching an exception on one rank (a rank that
    fell back to a collective while its peers had finished their sends would wait for them for ever): point-to-point sends of
    exactly the bytes each rank has ("p2p": RCCL / NCCL and gloo both implement batch_isend_irecv) unless DACC_GATHER=padded asks
    for the padded collective (debugging, or a backend without point-to-point operations).  Every rank reads the same
    environment, so the choice is collective."""
    import os
    t = os.environ.get("DACC_GATHER", "p2p")
    if t not in ("p2p", "padded"):
        raise ValueError("DACC_GATHER must be p2p or padded")
    return t


def gather_fragments(frags, bases, device=None, dst=0, copy=False, force=False):
    """All ranks call.  Returns (frags, bases) of the whole job on rank `dst` (fragments in rank order, seq_off rebased
    onto the concatenated base buffer) and (None, None) elsewhere.  Without an initialised process group: identity.

    Every rank sends exactly its bytes (no padding to the largest rank) and rank `dst` receives them at their final offsets
    of ONE fragment buffer and ONE base buffer, so nothing is re-sliced or concatenated afterwards.  `device`: where the
    message buffers live; default = what the backend needs ("cuda" for nccl = RCCL, whose operands must be device tensors;
    "cpu" for gloo, whose operands are the host arrays themselves -- no host -> device -> host round trip).

    ALIASING: the gathered `bases` is a memoryview of a module-level host buffer that the NEXT call overwrites; pass
    copy=True to get an owned bytes object when a result must outlive the next call.  Errors of the transport are raised, not
    swallowed.  `force`: do not short-cut a group of one rank."""
    global last_transport
    import torch
    import torch.distributed as dist
    # (force: run the whole path -- counts, message buffers, receive buffers, pinned copy, rebasing -- on a group of ONE rank too; the
    # GPU suite drives the RCCL / device-buffer branch that way on a box with a single device, tests/test_shard_gloo.py)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        last_transport = "local"
        return frags, bases
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    fb = np.ascontiguousarray(frags).view(np.uint8).reshape(-1)
    bb = np.frombuffer(bases, dtype=np.uint8)
    cnt = torch.tensor([fb.size, bb.size], dtype=torch.int64, device=device)
    allc = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allc, cnt)
    allc = torch.stack(allc).cpu().numpy()
    nf, nb = allc[:, 0].astype(np.int64), allc[:, 1].astype(np.int64)
    foff = np.concatenate([np.zeros(1, np.int64), np.cumsum(nf)]); boff = np.concatenate([np.zeros(1, np.int64), np.cumsum(nb)])
    last_transport = _transport()
    if last_transport == "padded":
        F2, B2 = _gather_padded(frags, fb, bb, allc, device, dst)
        return (F2, bytes(B2)) if (copy and B2 is not None) else (F2, B2)

    def msg(a):
        # the host array itself on a CPU backend (torch.from_numpy shares its memory; it is only read), a device copy for RCCL
        t = torch.from_numpy(a if a.flags.writeable else a.copy())
        return t if str(device) == "cpu" else t.to(device)

    if rank != dst:
        ops = []
        if fb.size:
            ops.append(dist.P2POp(dist.isend, msg(fb), dst))
        if bb.size:
            ops.append(dist.P2POp(dist.isend, msg(bb), dst))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return None, None
    F = torch.empty(max(int(foff[-1]), 1), dtype=torch.uint8, device=device)
    if str(device) == "cpu":
        B = _host_buffer(int(boff[-1]), device)      # received straight into the reused host buffer
    else:
        B = torch.empty(max(int(boff[-1]), 1), dtype=torch.uint8, device=device)
    ops = []
    for r in range(world):
        if r == dst:
            if fb.size:
                F[foff[r]:foff[r + 1]] = msg(fb)
            if bb.size:
                B[boff[r]:boff[r + 1]] = msg(bb)
            continue
        if nf[r]:
            ops.append(dist.P2POp(dist.irecv, F[foff[r]:foff[r + 1]], r))
        if nb[r]:
            ops.append(dist.P2POp(dist.irecv, B[boff[r]:boff[r + 1]], r))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    allf = F[:int(foff[-1])].cpu().numpy().view(frags.dtype).copy() if foff[-1] else frags[:0].copy()
    # rebase seq_off: the fragments of rank r start at fragment index foff[r] / itemsize and their bases at boff[r]
    isz = frags.dtype.itemsize
    for r in range(world):
        if nf[r]:
            allf["seq_off"][int(foff[r]) // isz:int(foff[r + 1]) // isz] += allf["seq_off"].dtype.type(int(boff[r]))
    if str(device) == "cpu":
        host = B
    else:
        host = _host_buffer(int(boff[-1]), device)
        host[:int(boff[-1])].copy_(B[:int(boff[-1])])
    view = memoryview(host.numpy())[:int(boff[-1])]
    return allf, (bytes(view) if copy else view)


def _gather_padded(frags, fb, bb, allc, device, dst):
    """DACC_GATHER=padded: every rank padded to the largest, one all_gather (for a backend without point-to-point operations)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    mx = int(allc.sum(axis=1).max())
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if fb.size:
        buf[:fb.size] = torch.from_numpy(fb.copy()).to(device)
    if bb.size:
        buf[fb.size:fb.size + bb.size] = torch.from_numpy(bb.copy()).to(device)
    # all_gather: implemented by every backend (one code path on all ranks; the padded transport is a debugging aid)
    out = [torch.zeros(max(mx, 1), dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, buf)
    if rank != dst:
        return None, None
    allf, allb, off = [], [], 0
    for r in range(world):
        raw = out[r].cpu().numpy()
        nf, nb = int(allc[r][0]), int(allc[r][1])
        f = raw[:nf].view(frags.dtype).copy()
        f["seq_off"] += off
        allf.append(f)
        allb.append(raw[nf:nf + nb].tobytes())
        off += nb
    return np.concatenate(allf) if allf else frags, b"".join(allb)
