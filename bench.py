#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
one BN128 G1 Pippenger MSM over 2^20 (scalar, point) pairs (BASELINE.json configs[1]).
  python bench.py [--gpus N --steps K --warmup W]
For N > 1 the driver launches one rank per GPU (torch.distributed.run, backend nccl = RCCL): every
rank owns its own 2^20-pair shard (weak scaling: the reference's contiguous split over workers,
src/bn128.js:353-383), and each step ends with the single exchange the path has: one all_gather
of the 96-byte Jacobian partial sums + a local EC sum (RCCL has no EC reduction operator).

Rank 0 prints ONE JSON line.  `value` = total pairs/s over all ranks / 1e6.  Extra objects:
  roofline      -- dominant kernel (msm_accumulate_g1): algorithmic bytes (96 B/pair) per launch /
                   its mean launch duration from HIP events on the launching stream
  cpu_baseline  -- the oracle's restatement of the reference's multiexp (w=7 subset tables, 256
                   accumulators, contiguous split over threads) on a bounded sample, rank 0 only
  extras        -- NTT 2^22 (config 3) and full Groth16 prove 2^20 (config 4) timings
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-extras", action="store_true", help="skip the NTT / prove extras")
    ap.add_argument("--extras", default="ntt,skewed,prove", help="comma list of extras to run: ntt, skewed, prove")
    ap.add_argument("--prove-log-domain", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["msm", "prove"], default="msm",
                    help="'msm' (default, BASELINE configs[1]); 'prove' = full Groth16 prove at --prove-log-domain, "
                         "window-sharded over the ranks with one all_gather of 576-byte records (strong scaling)")
    ap.add_argument("--shard", choices=["points", "windows"], default="points",
                    help="N>1: 'points' = every rank its own 2^log_n pairs (weak scaling, the reference's split); "
                         "'windows' = one 2^log_n MSM, rank g computes windows w %% N == g (strong scaling)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if local_rank == 0:
        import __graft_entry__
        __graft_entry__.ensure_built()      # in-tree hipcc build if the library is not there yet
    if world > 1:
        dist.barrier()
    import wasmsnark_amd
    from wasmsnark_amd import dist as wdist
    bn = wasmsnark_amd.build(device=local_rank)

    if args.workload == "prove":
        return bench_prove(args, bn, rank, world, dev)

    n = 1 << args.log_n
    windows = args.shard == "windows" and world > 1
    if windows:
        bn.set_window_shard(rank, world)
    rng = np.random.default_rng(1234 + (0 if windows else rank))
    # scalars: uniform 253-bit (< r); points: k_i * G for uniform k_i (distinct valid curve points)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    pts = bn.mul_base(1, ks.tobytes())
    d_s = torch.from_numpy(sc.reshape(-1)).to(dev)
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    def step():
        part = bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
        if world > 1:
            return wdist.sharded_msm(bn, 1, part, dev)
        return part

    for _ in range(args.warmup):
        step()
    # timed region: HIP events bracket ONLY the dominant kernel (mode 2), on the library's own stream
    bn.lib.c.wsnark_timing_reset()
    bn.lib.c.wsnark_timing_enable(2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    bn.lib.c.wsnark_timing_enable(0)
    kt = bn.lib.timing_report()
    # per-kernel breakdown: a few more (untimed) steps with every kernel bracketed
    bn.lib.c.wsnark_timing_reset()
    bn.lib.c.wsnark_timing_enable(1)
    for _ in range(min(args.steps, 3)):
        bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
    torch.cuda.synchronize()
    bn.lib.c.wsnark_timing_enable(0)
    kt_all = bn.lib.timing_report()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = (1 if windows else world) * n / (dt / args.steps) / 1e6
    acc_ms, acc_cnt = kt.get("msm_accumulate_g1", (0.0, 0))
    kernel_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt_all.items()}
    # HBM-side traffic of the dominant kernel: measured out of band (rocprofv3 cannot wrap itself) by
    # tools/gpu_session.sh with two separate --pmc passes on this same workload, committed under profiles/
    traffic, traffic_src = None, None
    try:
        pj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        pm = json.load(open(pj))["kernels"]["msm_accumulate"]
        if args.log_n == LOG_N:
            traffic = pm["fetch_bytes"] + pm["write_bytes"]
            traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; bytes per launch)"
    except Exception:  # noqa: BLE001
        pass
    roof = None
    if acc_cnt:
        avg_s = acc_ms / acc_cnt / 1e3
        achieved = 96.0 * n / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": "msm_accumulate_g1", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(acc_ms / acc_cnt, 4), "algorithmic_bytes_per_launch": 96 * n,
                "modmul_per_launch": 10 * 16 * n,
                "achieved_Gmodmul_s": round(10 * 16 * n / avg_s / 1e9, 1), "peak_Gmodmul_s_microbench": 172.0,
                "frac_of_multiplier_peak": round(10 * 16 * n / avg_s / 1e9 / 172.0, 4),
                "note": "integer-ALU bound (256-bit modmul): 16 windows x 10 products per pair; the multiplier's "
                        "measured chip peak is 172 G modmul/s (tools/microbench.hip). Traffic is ~15x the algorithmic "
                        "bytes because a windowed MSM gathers every point once per window (16 x 64 B), mostly from the "
                        "256 MiB Infinity Cache"}

    out = {"metric": "BN128 G1 MSM Mpoints/s (2^%d pairs/GPU)" % args.log_n, "value": round(value, 3),
           "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if windows else "weak",
           "vs_baseline": None, "dtype": "u256 (Montgomery; 9x29-bit limbs, v_mad_u64_u32)", "data": "synthetic",
           "config": {"workload": "BN128 G1 Pippenger MSM, 2^%d random (scalar,point) pairs per GPU, inputs resident in HBM" % args.log_n,
                      "pairs_per_gpu": n, "parallelism": ("windows-sharded x%d" if windows else "points-sharded x%d") % world + ", 1 all_gather of 96 B partials",
                      "device": bn.device_info},
           "roofline": roof, "kernel_ms": kernel_ms}

    # ---------------- extras: NTT 2^22 and full prove ----------------
    want = set() if args.no_extras else set(x for x in args.extras.split(",") if x)
    if want and world == 1:
        extras = {}
        try:
            if "ntt" not in want:
                raise KeyError("skip")
            m = 1 << 22
            x = torch.from_numpy(rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
            x[:, 31] &= 0x1F
            dx = x.reshape(-1).to(dev)
            for _ in range(2):
                bn.fft_dev(dx.data_ptr(), m, 0)
            bn.lib.c.wsnark_timing_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                bn.fft_dev(dx.data_ptr(), m, 0)
                bn.fft_dev(dx.data_ptr(), m, 0, inverse=True)
            bn.lib.c.wsnark_timing_report(None, 0)   # syncs the library stream
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
            bn.lib.c.wsnark_timing_enable(1)         # per-kernel brackets in a separate, untimed repetition
            bn.fft_dev(dx.data_ptr(), m, 0)
            bn.fft_dev(dx.data_ptr(), m, 0, inverse=True)
            bn.lib.c.wsnark_timing_enable(0)
            extras["ntt_2p22_fwd_plus_inv_ms"] = round(t * 1e3, 4)
            extras["ntt_2p22_algorithmic_GBps"] = round(2 * 64.0 * m / t / 1e9, 2)   # 64 B/coef/transform
            extras["ntt_2p22_hbm_frac"] = round(2 * 64.0 * m / t / 1e9 / HBM_PEAK_GBS, 5)
            extras["ntt_kernel_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in bn.lib.timing_report().items()}
            del dx
        except KeyError:
            pass
        except Exception as e:  # noqa: BLE001
            extras["ntt_error"] = repr(e)
        try:
            if "skewed" not in want:
                raise KeyError("skip")
            # the same MSM with a circuit-like scalar distribution (example-witness
            # histogram: 6.7 % zeros, 3.1 % ones, 10 % < 2^32, the rest full width): exercises hot buckets
            u = rng.random(n)
            sk = sc.copy()
            sk[u < 0.067] = 0
            ones = (u >= 0.067) & (u < 0.098)
            sk[ones] = 0
            sk[ones, 0] = 1
            small = (u >= 0.098) & (u < 0.2)
            sk[small, 4:] = 0
            d_sk = torch.from_numpy(sk.reshape(-1)).to(dev)
            torch.cuda.synchronize()
            bn.g1_multiexp_dev(d_sk.data_ptr(), d_p.data_ptr(), n)
            bn.lib.c.wsnark_timing_reset()
            t0 = time.perf_counter()
            for _ in range(5):
                bn.g1_multiexp_dev(d_sk.data_ptr(), d_p.data_ptr(), n)
            t = (time.perf_counter() - t0) / 5
            bn.lib.c.wsnark_timing_enable(1)
            bn.g1_multiexp_dev(d_sk.data_ptr(), d_p.data_ptr(), n)
            bn.lib.c.wsnark_timing_enable(0)
            extras["msm_circuit_like_scalars_ms"] = round(t * 1e3, 4)
            extras["msm_circuit_like_kernel_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in bn.lib.timing_report().items()}
            del d_sk
        except KeyError:
            pass
        except Exception as e:  # noqa: BLE001
            extras["msm_circuit_like_error"] = repr(e)
        if args.prove_log_domain and "prove" in want:
            try:
                from wasmsnark_amd import synth
                t0 = time.perf_counter()
                circ = synth.make_circuit(args.prove_log_domain, n_public=5, seed=1)
                S = synth.setup(circ, seed=2)
                pkey, _ = synth.build_key(circ, S, bn.mul_base)
                key = bn.load_key(pkey)
                wit = synth.witness_bin(circ)
                extras["prove_setup_s"] = round(time.perf_counter() - t0, 2)
                d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to(dev)
                r32, s32 = bytes(range(32)), bytes(range(32, 64))
                proof = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)   # warm
                ok = proof == synth.expected_proof(circ, S, r32, s32, bn.mul_base)
                for _ in range(2):                                                            # (clocks, allocations)
                    bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
                bn.lib.c.wsnark_timing_reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 8
                for _ in range(reps):
                    bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
                bn.lib.c.wsnark_timing_enable(1)
                bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
                bn.lib.c.wsnark_timing_enable(0)
                extras["prove_ms"] = round(t * 1e3, 3)
                import struct as _st
                _nv, _pb1 = _st.unpack_from("<I", pkey, 0)[0], _st.unpack_from("<I", pkey, 24)[0]
                _b1x = np.frombuffer(pkey, dtype=np.uint8, count=_nv * 64, offset=_pb1).reshape(_nv, 64)[:, :32]
                extras["prove_config"] = {"log_domain": args.prove_log_domain, "n_vars": circ.n_vars,
                                          "key_bytes": len(pkey), "key_resident": True,
                                          # variables absent from matrix B (B1 = B2 = infinity): the prover leaves
                                          # them out of the two B sums (WSNARK_PROVE_SPARSE=0 turns that off)
                                          "b_points_at_infinity_frac": round(float((~_b1x.any(axis=1)).mean()), 4)}
                extras["prove_matches_toxic_waste_closed_form"] = bool(ok)
                # the same proofs with every pair in every sum (no plan variants for the variables absent from A / B)
                os.environ["WSNARK_PROVE_SPARSE"] = "0"
                try:
                    key_dense = bn.load_key(pkey)
                    for _ in range(3):
                        p2 = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key_dense, r=r32, s=s32)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(reps):
                        bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key_dense, r=r32, s=s32)
                    torch.cuda.synchronize()
                    extras["prove_ms_all_pairs_in_every_sum"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
                    extras["prove_all_pairs_same_proof"] = bool(p2 == proof)
                    key_dense.free()
                finally:
                    del os.environ["WSNARK_PROVE_SPARSE"]
                extras["prove_kernel_ms_total"] = {k: round(v[0], 4) for k, v in bn.lib.timing_report().items()}
                extras["reference_wasm_8_workers_prove_2p20_s"] = 132.6   # BASELINE.md (survey container, other hardware)
            except Exception as e:  # noqa: BLE001
                extras["prove_error"] = repr(e)
        out["extras"] = extras

    # ---------------- CPU baseline (oracle = port of the reference algorithm), bounded sample ----------------
    if not args.no_cpu_baseline and world == 1:
        try:
            from oracle import pyoracle as orc
            cores = os.cpu_count() or 1
            threads = min(cores, 64)
            ns = min(n, 1 << 19)   # ~34 CPU-seconds of the reference algorithm
            t0 = time.perf_counter()
            got = orc.multiexp(1, "workers%d" % threads, sc[:ns].tobytes(), pts[: ns * 64], ns)
            tc = time.perf_counter() - t0
            chk = bn.g1_multiexp(sc[:ns].tobytes(), pts[: ns * 64])
            out["cpu_baseline"] = {"value": round(ns / tc / 1e6, 5), "unit": "Mpoints/s", "cores": threads,
                                   "host_cores": cores, "kind": "port",
                                   "sample": "first 2^%d pairs of the same workload, oracle g1m_multiexp2 restatement (w=7), "
                                             "contiguous split over %d threads" % (ns.bit_length() - 1, threads),
                                   "seconds": round(tc, 2), "gpu_result_matches": orc.g_affine(1, got) == chk,
                                   "reference_wasm_8_workers_2p20_Mpoints_s": 0.0695}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}

    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_prove(args, bn, rank, world, dev):
    """Full Groth16 prove (BASELINE config 4; north-star target at 1/2/4/8 GPUs): same key and witness on
    every rank, MSM windows sharded (w % N == rank), one all_gather of 576 B per rank, host finish."""
    import torch
    import torch.distributed as dist
    from wasmsnark_amd import dist as wdist, synth
    logd = args.prove_log_domain or 20
    circ = synth.make_circuit(logd, n_public=5, seed=1)
    S = synth.setup(circ, seed=2)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    r32, s32 = bytes(range(32)), bytes(range(32, 64))

    def step():
        if world > 1:
            return wdist.sharded_prove(bn, key, wit, r=r32, s=s32, device=dev)
        return bn.groth16GenProof(wit, key, r=r32, s=s32)

    for _ in range(args.warmup):
        proof = step()
    ok = proof == synth.expected_proof(circ, S, r32, s32, bn.mul_base) if args.warmup else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        print(json.dumps({"metric": "BN128 Groth16 prove ms @ 2^%d constraints" % logd, "value": round(ms, 3), "unit": "ms",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                          "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
                          "dtype": "u256 (Montgomery; 9x29-bit limbs, v_mad_u64_u32)", "data": "synthetic",
                          "config": {"workload": "BN128 Groth16 prove, synthetic R1CS, domain 2^%d, nVars %d, witness from host each step, key resident" % (logd, circ.n_vars),
                                     "parallelism": "MSM windows sharded x%d, 1 all_gather of 576 B records" % world,
                                     "device": bn.device_info},
                          "matches_toxic_waste_closed_form": ok,
                          "reference_wasm_8_workers_prove_2p20_s": 132.6}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
