"""Multi-GPU sharding of the MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

Mirrors the reference's only parallel strategy -- contiguous split of the (scalar, point) pairs
over workers, then gather + serial EC sum of the 96/192-byte Jacobian partials on the main thread
(/root/reference src/bn128.js:353-383, 385-415) -- with workers = GPUs and the gather = ONE
all_gather of 96 (G1) or 192 (G2) bytes per rank.  RCCL has no elliptic-curve reduction operator,
so the 'all-reduce' of partial sums is all_gather + a local W-way EC sum on every rank
(wsnark_g1_sum / wsnark_g2_sum), which is bit-identical on all ranks.

Alternative (the north star's wording): `bn.set_window_shard(rank, world)` makes every rank compute,
over ALL pairs, only the Pippenger windows w % world == rank; partials are pre-scaled by 2^(c w) and
combine with the same all_gather + EC sum.  That divides one MSM's latency by the number of GPUs
(strong scaling) at the price of every GPU reading all points; `bench.py --shard windows` measures it.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """floor(n/W) pairs per worker, remainder to the last (src/bn128.js:354-361)."""
    per = n // world
    lo = rank * per
    hi = n if rank == world - 1 else lo + per
    return lo, hi


def allgather_partials(partial, device=None):
    """partial: bytes (96 or 192, or a 576-byte prove record).  Returns the concatenation over ranks, in rank
    order.  One collective into one contiguous tensor and one copy back: the per-step cost that the multi-GPU
    bench adds to every MSM."""
    world = dist.get_world_size()
    t = torch.frombuffer(bytearray(partial), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t)
    except (RuntimeError, NotImplementedError, AttributeError):      # backend without the fused form
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        out = torch.cat(parts)
    return out.cpu().numpy().tobytes()


def sharded_msm(bn, g, local_partial, device=None):
    """Combine per-rank partial MSM results (Jacobian-Montgomery bytes) into the full sum."""
    allp = allgather_partials(local_partial, device)
    return bn.g1_sum(allp) if g == 1 else bn.g2_sum(allp)


def sharded_prove(bn, key, witness, r=None, s=None, device=None):
    """Groth16 proof with the MSM windows sharded over the ranks (one process per GPU, same key and
    witness everywhere): every rank computes its 576-byte record of partial sums, ONE all_gather, then
    every rank assembles the identical proof on the host.  r, s must be the same on all ranks."""
    rank, world = dist.get_rank(), dist.get_world_size()
    bn.set_window_shard(rank, world)
    try:
        part = bn.groth16_prove_partial(witness, key)
    finally:
        bn.set_window_shard(0, 1)
    return bn.groth16_prove_finish(key, allgather_partials(part, device), r=r, s=s)
