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


# ---------------------------------------------------------------------------------------------------------------
# Distributed (four-step) NTT over the ranks of one node: SURVEY.md section 8f row 3.
#
# The reference never parallelises a transform (src/bn128.js:126-166 runs CALC_H inside ONE worker; src/build_fft.js:
# 223-372 is a single-threaded loop), and a radix-2 transform does not shard without an exchange step.  n = n1 * n2:
#   X[k2 + n2 k1] = sum_{i1} w_n1^(i1 k1) * [ w_n^(i1 k2) * sum_{i2} x[i1 + n1 i2] w_n2^(i2 k2) ]
# Layout "n1-interleaved": rank p holds the sub-sequences i1 in [p n1/P, (p+1) n1/P), each complete (all i2), as a row-major
# (n1/P) x n2 block.  Then the column step (length-n2 transforms over i2) and the twiddle are local, ONE all-to-all
# turns the (n1/P) x n2 blocks into (n2/P) x n1 blocks, and the row step (length-n1 transforms over i1) is local
# again.  The result is n2-interleaved (rank p holds the k2 in its range, all k1): with n1 == n2 that IS the input
# layout of the next transform, so chains of transforms (CALC_H: iNTT -> coset NTT -> iNTT) need no re-layout.
# Bytes exchanged per transform: every rank sends and receives (P-1)/P * (n/P) * 32 B  (2^24, P = 8: 56 MiB per rank,
# 8 MiB per xGMI link; 2^20: 3.5 MiB per rank).  Backend: torch.distributed all_to_all_single ("nccl" = RCCL on the
# GPUs; the CPU tests run the same code on gloo with the thread-emulator build, where all_to_all falls back to
# all_gather when the backend lacks it).  All arithmetic is exact: results are bit-identical to wsnark_fr_ntt.
# ---------------------------------------------------------------------------------------------------------------
def ntt_layout_split(log_n, world):
    """(log_n1, log_n2) of the four-step split used by dist_ntt: n1 >= n2, both divisible by the world size."""
    log_n1 = (log_n + 1) // 2
    log_n2 = log_n - log_n1
    if world & (world - 1) or (1 << log_n2) < world:
        raise ValueError("dist_ntt needs a power-of-two world size <= n2 = 2^%d" % log_n2)
    return log_n1, log_n2


def to_interleaved(x_full, log_m, rank, world):
    """Slice of a full vector (uint8 tensor of n*32 bytes, any device) in the m-interleaved layout (m = 2^log_m):
    rows = the rank's residues i mod m, each row the complete sub-sequence x[res + m*j], j = 0 .. n/m - 1."""
    m = 1 << log_m
    n = x_full.numel() // 32
    per = m // world
    v = x_full.view(n // m, m, 32)[:, rank * per:(rank + 1) * per, :]
    return v.permute(1, 0, 2).contiguous().view(-1)


def from_interleaved(parts, log_m):
    """Inverse of to_interleaved for the concatenation (rank order) of all ranks' slices."""
    m = 1 << log_m
    n = parts.numel() // 32
    return parts.view(m, n // m, 32).permute(1, 0, 2).contiguous().view(-1)


def _all_to_all(out, inp, group=None):
    try:
        dist.all_to_all_single(out, inp, group=group)
    except (RuntimeError, NotImplementedError):      # backend without all-to-all (older gloo): emulate with all_gather
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        parts = [torch.empty_like(inp) for _ in range(world)]
        dist.all_gather(parts, inp, group=group)
        chunk = inp.numel() // world
        for q in range(world):
            out[q * chunk:(q + 1) * chunk] = parts[q][rank * chunk:(rank + 1) * chunk]


def dist_ntt(bn, x_local, log_n, odd=0, inverse=False, group=None):
    """fft_fft / fft_ifft (src/build_fft.js:159-221) of a length-2^log_n vector of Montgomery Fr elements spread over the
    ranks.  x_local: this rank's uint8 tensor in the n1-interleaved layout (to_interleaved(x, log_n1, ...)) on the
    device the library runs on; it is overwritten.  Returns the rank's slice of the result in the n2-interleaved layout
    (from_interleaved(all slices, log_n2) is the natural-order vector).  One all-to-all."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0            # a world of one: the same four steps, the exchange is the identity
    log_n1, log_n2 = ntt_layout_split(log_n, world)
    n1, n2 = 1 << log_n1, 1 << log_n2
    r1, r2 = n1 // world, n2 // world
    if x_local.numel() != r1 * n2 * 32 or not x_local.is_contiguous():
        raise ValueError("x_local must be the rank's contiguous (n1/P) x n2 block")
    c = bn.lib.c
    inv = 1 if inverse else 0
    sync = (lambda: torch.cuda.synchronize()) if x_local.is_cuda else (lambda: None)
    ptr = x_local.data_ptr()
    if odd:      # x[t] *= w_2n^t (also for the inverse: the reference's rawfft scales before its index flip)
        bn.lib.check(c.wsnark_fr_dist_scale_dev(ptr, r1, n2, rank * r1, log_n1, log_n, 1, 0, None))
    if log_n2 >= 1:
        bn.lib.check(c.wsnark_fr_ntt_batch_dev(ptr, n2, r1, inv, None))          # column step: r1 transforms over i2
    bn.lib.check(c.wsnark_fr_dist_scale_dev(ptr, r1, n2, rank * r1, log_n1, log_n, 0, inv, None))
    bn.lib.c.wsnark_timing_report(None, 0)                                        # the library's queues have drained
    sync()
    send = x_local.view(r1, world, r2, 32).permute(1, 0, 2, 3).contiguous().view(-1)   # block q = my rows x rank q's columns
    if world == 1:
        recv = send
    else:
        recv = torch.empty_like(send)
        _all_to_all(recv, send, group)
    # received block q = rank q's rows (its i1 range) x my columns  ->  (my k2) x (all i1)
    y = recv.view(world, r1, r2, 32).permute(2, 0, 1, 3).contiguous().view(-1)
    sync()
    bn.lib.check(c.wsnark_fr_ntt_batch_dev(y.data_ptr(), n1, r2, inv, None))       # row step: r2 transforms over i1
    bn.lib.c.wsnark_timing_report(None, 0)
    sync()
    return y
