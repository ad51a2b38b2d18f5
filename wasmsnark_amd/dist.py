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
    if device is not None and dist.get_backend() != "gloo":      # (gloo moves host memory only)
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
# Bytes exchanged per vector and transform: every rank sends and receives (P-1)/P * (n/P) * 32 B  (2^24, P = 8: 56 MiB per
# rank, 8 MiB per xGMI link; 2^20: 3.5 MiB per rank); `stack` vectors share one exchange.  Backend: torch.distributed all_to_all_single ("nccl" = RCCL on the
# GPUs; the CPU tests run the same code on gloo with the thread-emulator build, where all_to_all falls back to
# all_gather when the backend lacks it).  All arithmetic is exact: results are bit-identical to wsnark_fr_ntt.
# ---------------------------------------------------------------------------------------------------------------
def ntt_layout_split(log_n, world, flip=False):
    """(log_n1, log_n2) of the four-step split used by dist_ntt: n1 >= n2 (flip: n1 <= n2), both divisible by the world
    size.  A transform maps the n1-interleaved layout to the n2-interleaved one, so a chain of transforms alternates
    flip = False, True, False, ... (for even log_n both are the same split and nothing alternates)."""
    log_n1 = (log_n + 1) // 2
    log_n2 = log_n - log_n1
    if world & (world - 1) or (1 << log_n2) < world:
        raise ValueError("dist_ntt needs a power-of-two world size <= n2 = 2^%d" % log_n2)
    return (log_n2, log_n1) if flip else (log_n1, log_n2)


def to_interleaved(x_full, log_m, rank, world, elem=32):
    """Slice of a full vector (uint8 tensor of n*elem bytes, any device) in the m-interleaved layout (m = 2^log_m):
    rows = the rank's residues i mod m, each row the complete sub-sequence x[res + m*j], j = 0 .. n/m - 1."""
    m = 1 << log_m
    n = x_full.numel() // elem
    per = m // world
    v = x_full.view(n // m, m, elem)[:, rank * per:(rank + 1) * per, :]
    return v.permute(1, 0, 2).contiguous().view(-1)


def from_interleaved(parts, log_m, elem=32):
    """Inverse of to_interleaved for the concatenation (rank order) of all ranks' slices."""
    m = 1 << log_m
    n = parts.numel() // elem
    return parts.view(m, n // m, elem).permute(1, 0, 2).contiguous().view(-1)


def gather_all(t, group=None):
    """Concatenation (rank order) of every rank's tensor `t`, on t's device; staged through the host on gloo."""
    world = dist.get_world_size(group)
    src = t.cpu() if (t.is_cuda and dist.get_backend(group) == "gloo") else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src.contiguous(), group=group)
    return torch.cat(parts).to(t.device)


_a2a_native = {}


def _all_to_all(out, inp, group=None):
    """all_to_all_single where the backend has it (nccl = RCCL always; gloo in recent torch builds); otherwise -- decided ONCE
    per backend, on the first call -- the same exchange through all_gather (CPU tests only)."""
    backend = dist.get_backend(group)
    if backend == "gloo" and inp.is_cuda:       # gloo moves host memory only: stage through the host (tests: two ranks on one GPU)
        h_out = torch.empty(out.shape, dtype=out.dtype)
        _all_to_all(h_out, inp.cpu(), group)
        out.copy_(h_out)
        return
    if _a2a_native.get(backend, True):
        try:
            dist.all_to_all_single(out, inp, group=group)
            _a2a_native[backend] = True
            return
        except (RuntimeError, NotImplementedError):
            if backend == "nccl" or _a2a_native.get(backend):
                raise                                   # a real failure, not a missing feature
            _a2a_native[backend] = False
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [torch.empty_like(inp) for _ in range(world)]
    dist.all_gather(parts, inp, group=group)
    chunk = inp.numel() // world
    for q in range(world):
        out[q * chunk:(q + 1) * chunk] = parts[q][rank * chunk:(rank + 1) * chunk]


_side_streams = {}


class _on_side_stream:
    """Runs a block on a dedicated (non-default) torch stream of the tensor's device and hands that stream's handle to the
    library: torch's default stream is the NULL stream, which the C ABI reads as "use the library's own queue" -- the kernels
    would then not be ordered with the torch ops around them.  Entry and exit are ordered with the caller's current
    stream by events (wait_stream), not by host synchronisation.  CPU tensors (emulator tests): a no-op."""

    def __init__(self, t):
        self.cuda = t.is_cuda
        self.handle = None
        if self.cuda:
            dev = t.device
            if dev not in _side_streams:
                _side_streams[dev] = torch.cuda.Stream(device=dev)
            self.side = _side_streams[dev]
            self.cur = torch.cuda.current_stream(dev)

    def __enter__(self):
        if self.cuda and self.cur.cuda_stream != self.side.cuda_stream:
            self.side.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
            self.handle = self.side.cuda_stream
        elif self.cuda:
            self.ctx = None
            self.handle = self.side.cuda_stream
        return self

    def __exit__(self, *exc):
        if self.cuda and self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.cur.wait_stream(self.side)
        return False


def dist_ntt(bn, x_local, log_n, odd=0, inverse=False, group=None, flip=False, stack=1):
    with _on_side_stream(x_local) as ss:
        y = _dist_ntt(bn, x_local, log_n, odd, inverse, group, flip, ss.handle, stack)
        if x_local.is_cuda:
            x_local.record_stream(torch.cuda.current_stream(x_local.device))
            y.record_stream(torch.cuda.current_stream(x_local.device))
    return y


def _dist_ntt(bn, x_local, log_n, odd, inverse, group, flip, st, stack=1):
    """fft_fft / fft_ifft (src/build_fft.js:159-221) of a length-2^log_n vector of Montgomery Fr elements spread over the
    ranks.  x_local: this rank's uint8 tensor in the n1-interleaved layout (to_interleaved(x, log_n1, ...)) on the
    device the library runs on; it is overwritten.  Returns the rank's slice of the result in the n2-interleaved layout
    (from_interleaved(all slices, log_n2) is the natural-order vector).  One all-to-all.
    stack = k: x_local holds the slices of k vectors one after the other; they go through the same transform together --
    one set of kernel launches and ONE exchange for all of them (the exchanges are latency-bound at these sizes)."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0            # a world of one: the same four steps, the exchange is the identity
    log_n1, log_n2 = ntt_layout_split(log_n, world, flip)
    n1, n2 = 1 << log_n1, 1 << log_n2
    r1, r2 = n1 // world, n2 // world
    k = int(stack)
    if k < 1 or x_local.numel() != k * r1 * n2 * 32 or not x_local.is_contiguous():
        raise ValueError("x_local must be `stack` contiguous (n1/P) x n2 blocks")
    c = bn.lib.c
    inv = 1 if inverse else 0
    # everything is enqueued on ONE (non-default) torch stream: the library kernels (stream argument `st`), the layout
    # permutes and the collective are ordered by the stream itself -- no host synchronisation inside a transform
    ptr = x_local.data_ptr()
    if odd:      # x[t] *= w_2n^t (also for the inverse: the reference's rawfft scales before its index flip)
        bn.lib.check(c.wsnark_fr_dist_scale_dev(ptr, k, r1, n2, rank * r1, log_n1, log_n, 1, 0, st))
    if log_n2 >= 1:
        bn.lib.check(c.wsnark_fr_ntt_batch_dev(ptr, n2, k * r1, inv, st))        # column step: r1 transforms over i2 per vector
    bn.lib.check(c.wsnark_fr_dist_scale_dev(ptr, k, r1, n2, rank * r1, log_n1, log_n, 0, inv, st))
    send = x_local.view(k, r1, world, r2, 32).permute(2, 0, 1, 3, 4).contiguous().view(-1)   # block q = (all vectors) my rows x rank q's columns
    if world == 1:
        recv = send
    else:
        recv = torch.empty_like(send)
        _all_to_all(recv, send, group)
    # received block q = rank q's rows (its i1 range) x my columns  ->  per vector (my k2) x (all i1)
    y = recv.view(world, k, r1, r2, 32).permute(1, 3, 0, 2, 4).contiguous().view(-1)
    bn.lib.check(c.wsnark_fr_ntt_batch_dev(y.data_ptr(), n1, k * r2, inv, st))     # row step: r2 transforms over i1 per vector
    return y


class DistProver:
    """Groth16 proving over the ranks of one node with NOTHING replicated but the cheap parts: the four sums over the
    witness are window-sharded (wsnark_groth16_prove_partial, WSNARK_PARTIAL_SKIP_H), CALC_H runs on the distributed
    four-step transform (six transforms in three batches: three all-to-alls per proof), and every rank sums ITS slice of h against ITS slice of
    the key's H points -- a points-sharded partial sum that goes into the H slot of the rank's 576-byte record.  One
    all_gather of the records and the host-side finish as in sharded_prove.  The two linear combinations a = A w, b = B w
    are still evaluated in full on every rank (0.3 ms at 2^20: one sparse product each), then sliced.
    Reference shape: src/bn128.js:126-166 (CALC_H on one worker) + :353-415 (sums split over workers)."""

    def __init__(self, bn, key, points_h, device=None, group=None):
        """points_h: the key's hExps section (domain x 64 B, reference format: proving_key.bin from offset pHExps)."""
        self.bn, self.key, self.group, self.device = bn, key, group, device
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        dom = key.domain
        self.log_n = dom.bit_length() - 1
        if len(points_h) < dom * 64:
            raise ValueError("points_h shorter than domain * 64 bytes")
        # h comes out of three chained transforms (l1 -> l2 -> l1 -> l2): it is n2-interleaved, n2 of the UNflipped split
        self.l1, self.l2 = ntt_layout_split(self.log_n, self.world)
        full = torch.frombuffer(bytearray(points_h[:dom * 64]), dtype=torch.uint8)
        mine = to_interleaved(full, self.l2, self.rank, self.world, elem=64).clone()
        self.h_points = mine.to(device) if device is not None else mine
        self.n_local = dom // self.world

    def _calc_h_local(self, d_witness, witness_len):
        """This rank's slice of h (plain form, n2-interleaved), computed with the distributed transform."""
        with _on_side_stream(self.h_points) as ss:
            h = self._calc_h_on(d_witness, witness_len, ss.handle)
            if h.is_cuda:
                torch.cuda.current_stream(h.device).synchronize()      # the H sum runs on the library's own queue
        return h

    def _calc_h_on(self, d_witness, witness_len, st):
        """CALC_H (src/bn128.js:139-164) on this rank's slices with THREE exchanges: the inverse transforms of A, B and
        E = A.B travel together, then the coset transforms of A and B, then the inverse transform of O."""
        bn, c, dom = self.bn, self.bn.lib.c, self.key.domain
        dev = self.h_points.device
        ab = torch.empty(2 * dom * 32, dtype=torch.uint8, device=dev)
        bn.lib.check(c.wsnark_pkey_eval_ab_dev(self.key._h, d_witness, witness_len, ab.data_ptr(), ab.data_ptr() + dom * 32, st))
        n_loc = dom // self.world
        abe = torch.empty(3 * n_loc * 32, dtype=torch.uint8, device=dev)          # slices of A, B, E one after the other
        abe[:n_loc * 32] = to_interleaved(ab[:dom * 32], self.l1, self.rank, self.world)
        abe[n_loc * 32:2 * n_loc * 32] = to_interleaved(ab[dom * 32:], self.l1, self.rank, self.world)
        p = abe.data_ptr()
        bn.lib.check(c.wsnark_fr_mul_dev(p, p + n_loc * 32, p + 2 * n_loc * 32, n_loc, st))            # E = A.B on the domain
        abe = _dist_ntt(bn, abe, self.log_n, 0, True, self.group, False, st, 3)                          # coefficients of A, B; e (l2-interleaved)
        e = abe[2 * n_loc * 32:]
        ab2 = _dist_ntt(bn, abe[:2 * n_loc * 32].contiguous(), self.log_n, 1, False, self.group, True, st, 2)   # odd-coset evaluations (l1)
        q = ab2.data_ptr()
        o = torch.empty(n_loc * 32, dtype=torch.uint8, device=dev)
        bn.lib.check(c.wsnark_fr_mul_dev(q, q + n_loc * 32, o.data_ptr(), n_loc, st))                   # O = A.B on the coset
        o = _dist_ntt(bn, o, self.log_n, 0, True, self.group, False, st, 1)                              # l2-interleaved, like e
        rows = (1 << self.l2) // self.world
        h = torch.empty(n_loc * 32, dtype=torch.uint8, device=dev)
        e = e.contiguous()
        bn.lib.check(c.wsnark_fr_dist_combine_dev(e.data_ptr(), o.data_ptr(), h.data_ptr(), rows, 1 << (self.log_n - self.l2),
                                                  self.rank * rows, self.l2, self.log_n, st))
        return h

    def prove(self, d_witness, witness_len, r=None, s=None):
        """d_witness: device pointer of the plain witness on this rank's GPU.  r, s: as in sharded_prove."""
        import os
        import threading
        bn, rank, world = self.bn, self.rank, self.world
        box = {}

        def sums():      # A, B1, C, B2 on this rank's windows: a blocking call, so it gets its own host thread (and lane)
            try:
                box["rec"] = bn.groth16_prove_partial_dev(d_witness, witness_len, self.key, shard=(rank, world), skip_h=True)
            except Exception as ex:  # noqa: BLE001
                box["err"] = ex

        th = threading.Thread(target=sums)
        th.start()
        try:
            h = self._calc_h_local(d_witness, witness_len)
            hpart = bn.g1_multiexp_dev(h.data_ptr(), self.h_points.data_ptr(), self.n_local)    # points-sharded partial of the H sum
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        part = box["rec"][:288] + hpart + box["rec"][384:]
        draw = r is None or s is None
        if draw:
            part += (os.urandom(64) if rank == 0 else bytes(64))
        allp = allgather_partials(part, self.device) if world > 1 else part
        if draw:
            rec = len(part)
            r0, s0 = allp[576:608], allp[608:640]
            r, s = (r if r is not None else r0), (s if s is not None else s0)
            allp = b"".join(allp[i * rec:i * rec + 576] for i in range(world))
        return bn.groth16_prove_finish(self.key, allp, r=r, s=s)


# ---------------------------------------------------------------------------------------------------------------
# The same prover with NOTHING in Python between the kernels: wsnark_groth16_prove_dist (csrc/dist.hip) runs the whole
# proof of a rank behind the C ABI -- partial sums over the rank's POINTS shard of the key (wsnark_pkey_load_shard: 1 / world
# of the key resident, the reference's own worker split src/bn128.js:353-361), row-sharded sparse products written directly
# in the transform's layout, three exchanges, the H sum over the rank's hExps share, one all-gather -- and calls back into
# the host only for the transport.  DistProver above stays as the reference orchestration the tests compare it with.
# ---------------------------------------------------------------------------------------------------------------
import ctypes as _C


class _Comm(_C.Structure):     # wsnark_comm_t
    _fields_ = [("rank", _C.c_uint32), ("world", _C.c_uint32), ("d_send", _C.c_void_p), ("d_recv", _C.c_void_p),
                ("buf_bytes", _C.c_uint64), ("all_to_all", _C.c_void_p), ("all_gather", _C.c_void_p), ("user", _C.c_void_p)]


_A2A = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_uint64, _C.c_void_p)
_AG = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_uint64)


class NativeDistProver:
    """One rank of the multi-GPU prover over wsnark_groth16_prove_dist.  `sections`: the key's sections (every rank passes
    the whole key and keeps its share) -- or `path`: a key FILE (proving_key.bin / the WSNARK64 container), of which the rank
    maps and reads only its share (wsnark_pkey_load_file: N ranks do not each hold the whole key in host memory).  The transport callbacks run torch.distributed collectives: "nccl" (= RCCL over
    xGMI) enqueued on the library's own queue (torch.cuda.ExternalStream), so no host synchronisation separates kernels and
    exchanges; "gloo" (CPU tests, ranks sharing a GPU) staged through the host."""

    def __init__(self, bn, sections=None, device=None, group=None, path=None):
        self.bn, self.group, self.device = bn, group, device
        if (sections is None) == (path is None):
            raise ValueError("NativeDistProver: pass the key's sections or the path of a key file")
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        dom = sections["domain"] if path is None else bn.key_file_info(path)["domain"]
        self.log_n = dom.bit_length() - 1
        self.l2 = self.log_n // 2
        if self.world & (self.world - 1) or (1 << self.l2) < self.world:
            raise ValueError("the distributed prover needs a power-of-two world size <= 2^%d" % self.l2)
        self.key = bn.load_key(sections=sections, path=path, shard=(self.rank, self.world), h_interleave_log=self.l2)
        nbytes = 3 * (dom // self.world) * 32
        self.send = torch.empty(nbytes, dtype=torch.uint8, device=device if device is not None else "cpu")
        self.recv = torch.empty(nbytes, dtype=torch.uint8, device=self.send.device)
        self.errors = []
        self._a2a = _A2A(self._all_to_all_cb)
        self._ag = _AG(self._all_gather_cb)
        self.comm = _Comm(self.rank, self.world, self.send.data_ptr(), self.recv.data_ptr(), nbytes,
                          _C.cast(self._a2a, _C.c_void_p), _C.cast(self._ag, _C.c_void_p), None)

    def _all_to_all_cb(self, user, bytes_per_rank, stream):
        try:
            n = int(bytes_per_rank) * self.world
            send, recv = self.send[:n], self.recv[:n]
            if send.is_cuda:
                ext = torch.cuda.ExternalStream(int(stream), device=send.device) if stream else torch.cuda.current_stream(send.device)
                with torch.cuda.stream(ext):      # the collective is ordered on the queue the library's kernels run on
                    _all_to_all(recv, send, self.group)
            else:
                _all_to_all(recv, send, self.group)
            return 0
        except Exception as ex:  # noqa: BLE001 - must not unwind through the C frame
            self.errors.append(ex)
            return 1

    def _all_gather_cb(self, user, send, recv, nbytes):
        try:
            allp = allgather_partials(_C.string_at(send, int(nbytes)), self.device)
            _C.memmove(recv, allp, len(allp))
            return 0
        except Exception as ex:  # noqa: BLE001
            self.errors.append(ex)
            return 1

    def prove(self, d_witness, witness_len, r=None, s=None, stream=None):
        """d_witness: device pointer of the plain witness on this rank's GPU (all nVars signals).  r, s = None: rank 0 draws
        the blinding values; every rank returns the same proof."""
        from .bn128 import proof_from_bytes
        out = (_C.c_uint8 * 384)()
        del self.errors[:]
        rc = self.bn.lib.c.wsnark_groth16_prove_dist(self.key._h, d_witness, witness_len, _C.byref(self.comm), r, s, out, stream)
        if rc and self.errors:
            raise self.errors[0]
        self.bn.lib.check(rc)
        return proof_from_bytes(bytes(out))
