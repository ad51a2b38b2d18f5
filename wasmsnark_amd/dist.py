"""Multi-GPU sharding of the MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

Mirrors the reference's only parallel strategy -- contiguous split of the (scalar, point) pairs
over workers, then gather + serial EC sum of the 96/192-byte Jacobian partials on the main thread
(/root/reference src/bn128.js:353-383, 385-415) -- with workers = GPUs and the gather = ONE
all_gather of 96 (G1) or 192 (G2) bytes per rank.  RCCL has no elliptic-curve reduction operator,
so the 'all-reduce' of partial sums is all_gather + a local W-way EC sum on every rank
(wsnark_g1_sum / wsnark_g2_sum), which is bit-identical on all ranks.

Alternative (the north star's wording): `shard=(rank, world)` on the multiexp / prove_partial calls makes every rank
compute, over ALL pairs, only the Pippenger windows w % world == rank; partials are pre-scaled by 2^(c w) and
combine with the same all_gather + EC sum.  That divides one MSM's latency by the number of GPUs
(strong scaling) at the price of every GPU reading all points; `bench.py --gpus N` measures it on whole proofs.
The shard is a per-call argument: there is no process-global mode to set and restore.
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


def sharded_prove(bn, key, witness, r=None, s=None, device=None, d_witness=None):
    """Groth16 proof with the MSM windows sharded over the ranks (one process per GPU, same key and
    witness everywhere): every rank computes its 576-byte record of partial sums, ONE all_gather, then
    every rank assembles the identical proof on the host.
    r, s = None: rank 0 draws the blinding values and they travel with its record's all_gather slot (64 extra
    bytes per rank; the other ranks send zeros), so that all ranks still return the same proof.
    d_witness: (device pointer, byte length) of a witness already resident on this rank's GPU."""
    import os
    rank, world = dist.get_rank(), dist.get_world_size()
    if d_witness is not None:
        part = bn.groth16_prove_partial_dev(d_witness[0], d_witness[1], key, shard=(rank, world))
    else:
        part = bn.groth16_prove_partial(witness, key, shard=(rank, world))
    draw = r is None or s is None
    if draw:
        part += (os.urandom(64) if rank == 0 else bytes(64))
    allp = allgather_partials(part, device)
    if draw:
        rec = len(part)
        r0, s0 = allp[576:608], allp[608:640]
        r, s = (r if r is not None else r0), (s if s is not None else s0)
        allp = b"".join(allp[i * rec:i * rec + 576] for i in range(world))
    return bn.groth16_prove_finish(key, allp, r=r, s=s)
