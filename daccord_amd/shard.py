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


def gather_fragments(frags, bases, device="cpu", dst=0):
    """All ranks call.  Returns (frags, bases) of the whole job on rank `dst` (fragments in rank order, seq_off rebased
    onto the concatenated base buffer) and (None, None) elsewhere.  Without an initialised process group: identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return frags, bases
    world, rank = dist.get_world_size(), dist.get_rank()
    fb = np.ascontiguousarray(frags).view(np.uint8).reshape(-1)
    bb = np.frombuffer(bases, dtype=np.uint8)
    cnt = torch.tensor([fb.size, bb.size], dtype=torch.int64, device=device)
    allc = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allc, cnt)
    allc = torch.stack(allc).cpu().numpy()
    mx = int(allc.sum(axis=1).max())
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if fb.size:
        buf[:fb.size] = torch.from_numpy(fb.copy()).to(device)
    if bb.size:
        buf[fb.size:fb.size + bb.size] = torch.from_numpy(bb.copy()).to(device)
    out = [torch.zeros(max(mx, 1), dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    try:
        dist.gather(buf, out, dst=dst)
    except (RuntimeError, NotImplementedError):
        # backend without gather: every rank collects (same result on dst, a little more traffic)
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
