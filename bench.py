#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: BN128 Groth16 prove ms @ 2^20 constraints (configs[3]), with the G1 MSM
(configs[1]) and the NTT (configs[2]) figures of the same metric string carried in the same line.

  python bench.py [--gpus N --steps K --warmup W]

A "step" is one whole proof (CALC_H + 5 MSMs + assembly) of the SURVEY.md section 8d C4 circuit -- synthetic R1CS, domain
2^20, 1-3 non-zeros per COLUMN of A and B (every variable present), valid witness, key from known toxic waste -- with
the key and the witness already resident in HBM.  Every timed proof is checked (first and last) against the toxic-waste
closed form.  For N > 1 the driver launches one rank per GPU (torch.distributed.run, backend nccl = RCCL): the same proof
strong-scaled -- every rank holds a POINTS shard of the key (1/N of every section, wsnark_pkey_load_shard), CALC_H runs on the
distributed four-step transform (three all-to-alls), ONE all_gather of the 592-byte records of partial sums per proof, host-side
EC sum + assembly on every rank (RCCL has no elliptic-curve reduction); a mode that fails or disagrees with the closed form on
any rank falls through to the next (native -> Python orchestration -> replicated CALC_H) and the line says so.  N = 1 is the
one-call prover, no collective.

Rank 0 prints ONE JSON line.  `value` = ms per proof (max over ranks), witness resident in HBM (the contract keeps PCIe out of
`value`; the drop-in call from a HOST witness sits beside it).  Objects beside the contract's fields:
  roofline          dominant kernel msm_accumulate_g1: algorithmic bytes (96 B/pair) per launch / mean launch duration from HIP events
                    on the launching queue over the timed region (IN SITU: two-queue schedule), plus the same kernel ALONE (a few
                    more proofs on one queue, same run) and what the committed rocprofv3 summaries of this command say (in situ under
                    the profiler, and serialised -- where profiler and events agree); the HBM bound says little here --
  roofline_int_alu  -- the governing bound: 256-bit modmul/s against the multiplier's peak measured in this run
  roofline_proof    the whole proof: issue cycles of its VALU instruction stream (committed rocprofv3 --pmc SQ_INSTS_VALU over exactly
                    P proofs) over the cycles one measured proof takes
  cpu_baseline      the oracle (C port of the reference's groth16GenProof: w=7 subset-table multiexp over worker threads,
                    single-thread CALC_H) on the benchmark's own circuit, timed on this box's host cores, rank 0 at N=1 (an N > 1
                    line carries the last N = 1 measurement and says so)
  extras            G1 MSM 2^20 (one at a time / two in flight / host pointers), NTT 2^22 (odd 0, odd 1, inverse),
                    two proofs in flight, the sparse (round-1) circuit
--workload msm keeps round 1's line (one G1 MSM of 2^20 pairs per step, weak scaling over ranks).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
MODMUL_PEAK_G = 172.0            # tools/microbench.hip on MI355X (profiles/r01_session17_microbench.jsonl): radix-2^29 product chain
DTYPE = "u256 (Montgomery; 9x29-bit limbs, v_mad_u64_u32)"
REF_WASM_PROVE_2P20_S = 132.6    # the reference (Node + WASM, 8 workers) in the survey container -- OTHER hardware
REF_WASM_MSM_MPTS = 0.0695
EXTRA_REPS = int(os.environ.get("WSNARK_BENCH_EXTRA_REPS", "20"))   # repetitions inside the extras (the CPU dry run of tests/ lowers it)


def kernel_ms(report, per=1):
    return {k: round(v[0] / max(per, 1), 4) for k, v in sorted(report.items())}


def build_prover(bn, logd, style, seed=1, keep_h=False, container="auto", load=True, cold_probe=False, sections=True):
    """Synthetic circuit + key (device-resident) + witness bytes.  Keys past the 4 GiB of proving_key.bin's u32 offsets
    (2^23 constraints and up) go through the sections loader.  The circuit comes from the library's host-side generator
    (csrc/synth.hip; same family as wasmsnark_amd/synth.py's Python generator, seconds instead of minutes)."""
    from wasmsnark_amd import synth
    t0 = time.perf_counter()
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=seed, style=style)
    if not sections:      # (--key-file, ranks > 0: the key comes from the file rank 0 writes; only the witness and the closed form are needed here)
        info = {"log_domain": logd, "n_vars": circ.n_vars, "n_public": circ.n_public, "nnz_A_plus_B": int(circ.nnz), "style": style,
                "vars_absent_from_A_B": [int(x) for x in circ.absent], "key_bytes": None, "key_container": "WSNARK64 file written by rank 0",
                "generator": "csrc/synth.hip (wsnark_synth_*)", "setup_s": round(time.perf_counter() - t0, 1), "_cold": {}, "_sections": None}
        return circ, None, circ.witness_bin(), info
    sec, _ = circ.build_sections()
    use_sections = container == "sections" or (container == "auto" and logd >= 23)
    t1 = time.perf_counter()
    if not load:      # N > 1: every rank loads only ITS shard of these sections (NativeDistProver); nothing whole is resident
        wit = circ.witness_bin()
        info = {"log_domain": logd, "n_vars": circ.n_vars, "n_public": circ.n_public, "nnz_A_plus_B": int(circ.nnz), "style": style,
                "vars_absent_from_A_B": [int(x) for x in circ.absent],
                "key_bytes": sum(len(v) for v in sec.values() if isinstance(v, (bytes, bytearray))), "key_container": "sections, points-sharded per rank",
                "generator": "csrc/synth.hip (wsnark_synth_*)", "setup_s": round(time.perf_counter() - t0, 1), "_cold": {}, "_sections": sec}
        return circ, None, wit, info
    # (wait_tables=False: the load returns as the C call does, with the table rows still being built in the background; bench_prove
    #  times the first proof right behind it and then waits for the build before anything steady-state is timed)
    wit = circ.witness_bin()            # (before the load: nothing may stand between the load's return and the first proof)
    if use_sections:
        key = bn.load_key(sections=sec, wait_tables=not cold_probe)
        key_bytes = sum(len(v) for v in sec.values() if isinstance(v, (bytes, bytearray)))
    else:
        pkey = synth.sections_to_pkey(sec)
        t1 = time.perf_counter()
        key = bn.load_key(pkey, wait_tables=not cold_probe)
        key_bytes = len(pkey)
    t_load = time.perf_counter() - t1
    t_loaded = time.perf_counter()
    cold_runs = None
    if cold_probe:
        # cold: the very first proofs of this process on the freshly loaded key, from a host witness, right behind the load's return
        # (transform plans and their twiddle tables, lane buffers are created inside the first one; the table rows are still being
        # built under all of them) -- the reference's timing hook wraps the whole call, key parsing included
        # (example/bn128/index.html:39-49, src/bn128.js:581-604)
        r32, s32 = bytes(range(32)), bytes(range(32, 64))
        cold_runs = {"ms": [], "proofs": []}
        for _ in range(4):
            tq = time.perf_counter()
            cold_runs["proofs"].append(bn.groth16GenProof(wit, key, r=r32, s=s32))
            cold_runs["ms"].append(round((time.perf_counter() - tq) * 1e3, 2))
        key.wait_tables()                                   # everything after this is steady state: tables resident
        cold_runs["tables_ready_ms"] = round((time.perf_counter() - t_loaded) * 1e3, 2)
    info = {"log_domain": logd, "n_vars": circ.n_vars, "n_public": circ.n_public, "nnz_A_plus_B": int(circ.nnz), "style": style,
            "vars_absent_from_A_B": [int(x) for x in circ.absent], "key_bytes": key_bytes,
            "key_container": "sections" if use_sections else "proving_key.bin", "generator": "csrc/synth.hip (wsnark_synth_*)",
            "setup_s": round(time.perf_counter() - t0, 1)}
    cold = {"key_load_ms": {k: round(v, 2) for k, v in key.load_ms.items()}, "key_load_wall_ms_incl_binding": round(t_load * 1e3, 2),
            "key_bytes": key_bytes, "table_bytes": int(key.table["bytes"]), "table_rows": [key.table["rows_w"], key.table["rows_h"]],
            "table_window_bits": [key.table["c_w"], key.table["c_h"]]}
    if keep_h:
        info["h_points"] = bytes(sec["pointsH"])
    cold["_runs"] = cold_runs
    info["_cold"] = cold
    info["_sections"] = sec
    return circ, key, wit, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["prove", "msm"], default="prove")
    ap.add_argument("--prove-log-domain", type=int, default=20, help="20 = BASELINE config 4; 24 = config 5 (sections loader)")
    ap.add_argument("--circuit", choices=["columns", "rows"], default="columns")
    ap.add_argument("--key-container", choices=["auto", "file", "sections"], default="auto",
                    help="proving_key.bin (u32 offsets: up to 4 GiB) or the section container (wsnark_pkey_load_sections); auto = sections from 2^23")
    ap.add_argument("--key-file", default="", help="N>1 (per-process ranks): rank 0 writes the key ONCE as a WSNARK64 container file (the u64-offset "
                                                     "form of proving_key.bin) and every rank maps it and reads only its shard (wsnark_pkey_load_file) -- instead "
                                                     "of every rank generating and holding the whole key in host memory.  'auto' = a file under $WSNARK_BENCH_KEY_DIR "
                                                     "or the temp directory; or a path on a file system all ranks see")
    ap.add_argument("--log-n", type=int, default=20, help="--workload msm / extras: pairs per MSM")
    ap.add_argument("--extras", default="msm,ntt,cold,inflight,power,sparse,node", help="comma list (N=1 only): msm, ntt, cold, inflight, power, sparse, node")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone-pass", action="store_true",
                    help="skip the one-queue proofs that time the dominant kernel alone (the in-situ rocprofv3 pass of tools/gpu_session.sh: "
                         "its kernel summary must hold two-queue launches only)")
    ap.add_argument("--shard", choices=["points", "windows"], default="points", help="--workload msm, N>1")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N>1 transport: nccl (= RCCL over xGMI; one GPU per rank) or gloo with host staging -- the latter lets "
                         "several ranks SHARE one GPU (functional check of the N>1 path on a one-GPU box; not a scaling measurement)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N > 1 WITHOUT torch.distributed: this ONE process drives N devices through wsnark_group_* (csrc/group.hip: "
                         "a context and a host thread per device, points shards, the transport inside the library) -- what the Node drop-in's "
                         "buildBn128({devices}) uses.  Run it plainly: python bench.py --gpus N --single-process")
    ap.add_argument("--group-devices", default="", help="--single-process: comma list of device ordinals (default 0..N-1; an ordinal may repeat: "
                                                        "several contexts on one GPU, a functional check)")
    ap.add_argument("--calc-h", choices=["replicated", "dist", "native"], default="native",
                    help="N>1: 'native' = wsnark_groth16_prove_dist: points-sharded key (1/N resident per rank), distributed CALC_H, one C call "
                         "per proof with the transport as callbacks; 'dist' = the same algorithm orchestrated from Python on window shards of "
                         "the whole key (DistProver); 'replicated' = every rank repeats the whole CALC_H (round 1).  A mode that fails or "
                         "disagrees with the closed form on any rank falls through to the next one and the line says so")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # ranks may share a GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if rank == 0:
        import __graft_entry__
        __graft_entry__.ensure_built()      # in-tree hipcc build if the library is not there yet
    if world > 1:
        dist.barrier()
    import wasmsnark_amd
    bn = wasmsnark_amd.build(device=local_rank)
    ctx = {"args": args, "bn": bn, "rank": rank, "world": world, "dev": dev, "torch": torch, "dist": dist}
    if args.single_process and world == 1 and args.gpus > 1:
        out = bench_group(ctx)
    else:
        out = bench_msm(ctx) if args.workload == "msm" else bench_prove(ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def timed(ctx, step, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    torch, dist, world, dev = ctx["torch"], ctx["dist"], ctx["world"], ctx["dev"]
    res = None
    for _ in range(warmup):
        res = step()
    lib = ctx["bn"].lib
    lib.c.wsnark_timing_reset()
    lib.c.wsnark_timing_enable(2)          # HIP events around the dominant kernel only, on its own stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    lib.c.wsnark_timing_enable(0)
    kt = lib.timing_report()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=(dev if dist.get_backend() == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, kt, res


def pmc_traffic(kernel, pairs_per_launch, windows):
    """HBM-side bytes per launch of `kernel`, measured out of band by tools/gpu_session.sh (rocprofv3 cannot wrap
    itself): two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over this same command, summary committed."""
    # (two committed summaries: proofs on the table key -- 13 passes per launch at 2^20 -- and the plain 16-window launches)
    latest = _latest_profile("*pmc_traffic_table.json")
    calib = _latest_profile("*pmc_calibration.json")
    names = ([os.path.basename(latest)] if latest else []) + ["r03_final_pmc_traffic_table.json", "r02_pmc_traffic_table.json", "r02_pmc_traffic.json"]
    for name in names:
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            k = pm["kernels"][kernel]
            shape = pm.get("msm_accumulate_shape") or {"pairs_per_launch": 1 << 20, "windows_per_launch": 16}
            if abs(pairs_per_launch - shape["pairs_per_launch"]) > 16 or windows != shape["windows_per_launch"]:
                continue        # the committed summary was taken on another launch shape: not quoted
            return k["fetch_bytes"] + k["write_bytes"], ("profiles/%s: %s; FETCH_SIZE calibrated on this access pattern (64-B point gathers out of a 1 GiB table: counter / known bytes = 1.00, "
                                                         "profiles/%s)" % (name, pm.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes per launch"),
                                                                         os.path.basename(calib) if calib else "r03_s1_pmc_calibration.json"))
        except Exception:  # noqa: BLE001
            continue
    return None, None


def _latest_profile(pattern):
    """Newest committed profiles/rNN_<pattern> (by round tag, then name)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return c[-1] if c else None


def ntt_issue_note():
    """The transform passes' issue floor by instruction class over their duration, from the newest committed SQ-counter summary."""
    f = _latest_profile("*_pmc_sq_counters.json")
    try:
        d = json.load(open(f))
        k = d.get("kernels", d)["ntt_pass_kernel"]
        if "issue_floor_frac" in k:
            return "issue floor by instruction class / duration = %.2f, profiles/%s" % (k["issue_floor_frac"], os.path.basename(f))
        return "%.2f of the VALU slots under the flat 4-cycle model of rounds 3-5, profiles/%s" % (k["valu_issue_frac"], os.path.basename(f))
    except Exception:  # noqa: BLE001
        return "about 0.6-0.7 of their issue floor by instruction class (profiles/r06_ntt_experiments.txt)"


def rocprof_kernel_mean(csv_path, must_contain, must_not_contain=()):
    """Mean duration (ms), calls, min, max of one kernel in a committed `rocprofv3 --kernel-trace --stats` summary."""
    import csv
    try:
        with open(csv_path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name", "")
                if all(x in name for x in must_contain) and not any(x in name for x in must_not_contain):
                    return {"mean_ms": round(float(row["AverageNs"]) / 1e6, 4), "calls": int(row["Calls"]),
                            "min_ms": round(float(row["MinNs"]) / 1e6, 4), "max_ms": round(float(row["MaxNs"]) / 1e6, 4),
                            "file": "profiles/" + os.path.basename(csv_path)}
    except Exception:  # noqa: BLE001
        pass
    return None


def rocprof_citation(alg_bytes):
    """What the committed rocprofv3 summaries of THIS command say about the dominant kernel: (a) proofs only, the shipped two-queue
    schedule -- the profiler's per-dispatch instrumentation stretches a proof by ~0.5 ms and changes which kernels share the SIMDs,
    so launches that run beside the other queue's full-width kernels are longer than in the unprofiled run; (b) proofs only with
    WSNARK_PROVE_OVERLAP=0 -- one queue, every kernel alone: there the profiler and HIP events must agree."""
    g1 = (["msm_accumulate", "Field29"], ["Fp2T"])
    out = {}
    for key, pat in (("in_situ_two_queues", "*kernel_stats_proofs_only.csv"), ("serialised_one_queue", "*kernel_stats_proofs_only_serialised.csv")):
        f = _latest_profile(pat)
        m = rocprof_kernel_mean(f, *g1) if f else None
        if m:
            m["GBps"] = round(alg_bytes / (m["mean_ms"] / 1e3) / 1e9, 2)
            m["hbm_frac"] = round(m["GBps"] / HBM_PEAK_GBS, 5)
            out[key] = m
    if out:
        out["how_to_read"] = ("serialised_one_queue is the same command with WSNARK_PROVE_OVERLAP=0: every kernel alone -- compare with "
                              "roofline.avg_launch_ms (basis alone); in_situ_two_queues is the shipped schedule UNDER the profiler, whose per-dispatch "
                              "instrumentation stretches a proof by ~0.2-0.5 ms and changes the overlap -- compare with the events of THAT run "
                              "(profiles/*bench_under_rocprof.json), not with roofline.in_situ of an unprofiled run")
    return out or None


def proof_issue_roofline(ms_per_proof, rates):
    """Whole-proof VALU-issue roofline BY INSTRUCTION CLASS (round 6; tools/issue_model.py): for every kernel of a proof its DYNAMIC
    wave-instruction count (rocprofv3 --pmc SQ_INSTS_VALU over exactly P proofs, committed summary) x its STATIC class mix
    (profiles/rNN_isa_classes.json, tools/isa_histogram.py) / the class rates THIS run measured on THIS box (wsnark_peak_probe 6..,
    `issue_classes`).  floor = the time a chip that did nothing but issue the proof's instruction stream would need, whatever the
    schedule; frac = floor / one measured proof.  No clock and no "4 cycles per instruction" enter (rounds 3-5 priced every VALU
    instruction at 4 cycles, under which mul_base_kernel read 1.34 of the chip: plain 32-bit instructions issue in 2)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import issue_model
    f = _latest_profile("*proof_issue_budget.json")
    classes = issue_model.load_classes()
    if not f or not classes or not rates:
        return None
    try:
        b = json.load(open(f))
        insts = float(b["valu_insts_per_proof"])
    except Exception:  # noqa: BLE001
        return None
    per_kernel, floor_s, unclassified = {}, 0.0, 0.0
    for k, v in b.get("kernels", {}).items():
        fs, how = issue_model.kernel_floor_s(k, float(v["valu_insts_per_proof"]), classes, rates, v.get("dynamic_int64_share"))
        floor_s += fs
        per_kernel[k] = fs
        if how.startswith("unclassified"):
            unclassified += float(v["valu_insts_per_proof"])
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]
    cyc, clock = issue_model.equivalent_cycles(rates)
    return {"bound": "valu-issue (by instruction class)", "achieved": round(floor_s * 1e3, 3), "peak": round(ms_per_proof, 3), "unit": "ms per proof",
            "frac": round(floor_s * 1e3 / ms_per_proof, 4), "valu_wave_insts_per_proof": int(insts), "issue_floor_ms": round(floor_s * 1e3, 3),
            "ms_per_proof": round(ms_per_proof, 3),
            "issue_floor_ms_by_kernel": {k: round(v * 1e3, 4) for k, v in top},
            "share_of_instructions_unclassified": round(unclassified / insts, 5) if insts else None,
            "class_rates_G_lane_ops_per_s": {k: rates.get(v) for k, v in classes["probe_of_class"].items()},
            "class_cycles_per_wave_instruction_if_v_add_u32_is_2": {k: cyc.get(v) for k, v in classes["probe_of_class"].items()},
            "clock_GHz_implied": clock,
            "flat_4_cycle_model_of_rounds_3_to_5": {"issue_floor_ms": round(4.0 * insts / 1024 / 2.4e9 * 1e3, 3),
                                                     "frac": round(4.0 * insts / 1024 / 2.4e9 * 1e3 / ms_per_proof, 4),
                                                     "note": "every VALU instruction at 4 cycles, 2.4 GHz: too high a floor (kept for comparison with BENCH_r03..r05)"},
            "source": "instructions: profiles/%s (%s); class mix: profiles/%s; class rates: this run (issue_classes)" % (
                os.path.basename(f), b.get("how", ""), os.path.basename(classes["_path"])),
            "note": "achieved = sum over kernels of SQ_INSTS_VALU x 64 x sum(class share / class rate): the issue time of one proof's "
                    "instruction stream; peak = one measured proof; frac = the share of the chip's VALU issue capacity a whole proof uses"}


def measure_peaks(bn):
    """The integer roofline's peak from THIS box and THIS run (wsnark_peak_probe, ~10 ms each): a dependent chain of the
    library's own radix-2^29 Montgomery product on every lane, and the raw v_mad_u64_u32 rate."""
    import ctypes as C
    out = {}
    for name, probe in (("modmul_G_per_s", 0), ("modmul_inlined_G_per_s", 1), ("mad_u64_u32_G_per_s", 2)):
        v = C.c_double(0)
        try:
            bn.lib.check(bn.lib.c.wsnark_peak_probe(probe, C.byref(v)))
            out[name] = round(v.value, 1)
        except Exception as e:  # noqa: BLE001
            out[name] = None
            out["error"] = repr(e)
    try:      # VALU issue rate per instruction class (wsnark_peak_probe 6..28, ~10 ms each): what prices roofline_proof
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import issue_probe
        out["issue_classes"] = issue_probe.measure(bn, reps=2)
    except Exception as e:  # noqa: BLE001
        out["issue_classes_error"] = repr(e)
    return out


def rooflines(kt, kernel, pairs_per_launch, windows_owned, bytes_per_pair, modmul_per_add, peak=None, alone_ms=None, cite_rocprof=False):
    """Both rooflines of the dominant kernel.  Basis of `achieved` / `frac`: the launch ALONE when this run measured it (HIP events
    around the kernel in proofs that run on one queue, right after the timed region) -- a kernel's roofline is the kernel's, and it
    is the figure a profiler reproduces (rocprofv3 instruments every dispatch and thereby changes which kernels of a two-queue
    proof share the SIMDs; with one queue it agrees with the events: profiles/*kernel_stats_proofs_only_serialised.csv) -- with the
    launch IN SITU (events over the timed region, two-queue schedule, no profiler) beside it in `in_situ`.  Without an alone
    pass (N > 1, --no-alone-pass, stand-alone MSM) the in-situ figure is the basis and `basis` says so."""
    ms, cnt = kt.get(kernel, (0.0, 0))
    if not cnt or ms <= 0:
        return None, None
    situ_ms = ms / cnt
    alg = bytes_per_pair * pairs_per_launch
    modmul = modmul_per_add * windows_owned * pairs_per_launch
    base_ms = alone_ms if alone_ms else situ_ms
    basis = ("alone: HIP events around the kernel, same run, proofs on ONE queue (WSNARK_PROVE_OVERLAP=0 through wsnark_tuning_set) "
             "right after the timed region" if alone_ms else
             "in situ: HIP events on the launching queue over the timed region (whatever the lane's other queue runs beside the kernel is inside)")
    achieved = alg / (base_ms / 1e3) / 1e9
    # the committed PMC summary names the launch shape it was taken on: only quoted for that shape
    traffic, src = pmc_traffic(kernel, pairs_per_launch, windows_owned)
    hbm = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
           "avg_launch_ms": round(base_ms, 4), "basis": basis, "launches_timed": cnt, "algorithmic_bytes_per_launch": int(alg),
           "in_situ": {"avg_launch_ms": round(situ_ms, 4), "GBps": round(alg / (situ_ms / 1e3) / 1e9, 2),
                       "frac": round(alg / (situ_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5), "launches_timed": cnt,
                       "what": "HIP events on the launching queue over the timed region: the two-queue schedule, no profiler"},
           "note": "reported because the contract asks for it; the kernel is integer-ALU bound (see roofline_int_alu): "
                   "~%d modmul per %d bytes" % (modmul_per_add * windows_owned, bytes_per_pair)}
    if cite_rocprof:
        hbm["rocprofv3"] = rocprof_citation(alg)
    live = max([v for k, v in (peak or {}).items() if k.startswith("modmul") and v] or [0])
    pk = live or MODMUL_PEAK_G
    g = modmul / (base_ms / 1e3) / 1e9
    alu = {"bound": "int-alu", "kernel": kernel, "achieved": round(g, 1), "peak": pk, "unit": "Gmodmul/s",
           "frac": round(g / pk, 4), "basis": "alone" if alone_ms else "in situ",
           "in_situ": {"achieved": round(modmul / (situ_ms / 1e3) / 1e9, 1), "frac": round(modmul / (situ_ms / 1e3) / 1e9 / pk, 4)},
           "modmul_per_launch": int(modmul), "windows_per_launch": windows_owned,
           "peak_source": ("wsnark_peak_probe in this run on this box: dependent chain of the library's radix-2^29 Montgomery product "
                           "(205 instructions, 162 of them v_mad_u64_u32) on 8 x 256 lanes per CU; best of 3 after a warm-up launch" if live else
                           "tools/microbench.hip on MI355X, round 1 (profiles/r01_session17_microbench.jsonl): not re-measured in this run"),
           "peaks_measured_in_this_run": peak}
    return hbm, alu


def bench_prove(ctx):
    args, bn, rank, world, dev, torch = ctx["args"], ctx["bn"], ctx["rank"], ctx["world"], ctx["dev"], ctx["torch"]
    from wasmsnark_amd import dist as wdist, synth
    logd = args.prove_log_domain
    use_file = bool(args.key_file) and world > 1
    circ, key, wit, info = build_prover(bn, logd, args.circuit, container=args.key_container, load=(world == 1), cold_probe=(world == 1),
                                        sections=not (use_file and rank != 0))
    info.pop("h_points", None)
    cold = info.pop("_cold")
    sec = info.pop("_sections", None)
    r32, s32 = bytes(range(32)), bytes(range(32, 64))
    want = circ.expected_proof(r32, s32)
    runs = cold.pop("_runs", None)
    if world == 1 and runs:
        cold["first_proof_ms"] = runs["ms"][0]
        cold["next_proofs_ms"] = runs["ms"][1:]
        cold["tables_ready_ms_after_the_load_returned"] = runs["tables_ready_ms"]
        cold["first_proof_matches_closed_form"] = bool(all(p == want for p in runs["proofs"]))
        cold["first_proof_ran_on"] = "the plain sections, beside the background build of the table rows (wsnark_pkey_wait_tables not called yet)"
        cold["key_load_plus_first_proof_ms"] = round(cold["key_load_ms"]["total"] + cold["first_proof_ms"], 2)
        cold["key_load_ms"] = {k: round(v, 2) for k, v in key.load_ms.items()}
        cold["key_load_ms_note"] = ("total = what the load call took (points_h2d + masks_convert + pols_to_csr); table_build = the background build's own "
                                    "duration on the GPU, with the first four proofs running beside it")
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    # N > 1: a ladder of orchestrations, each checked against the closed form on EVERY rank before it is timed.  None of them
    # had run on RCCL when this was written (the build's boxes have one GPU; they are tested on gloo, with ranks sharing a
    # GPU, and with a world of one): a mode that fails or disagrees anywhere falls through to the next and the line says so.
    holder = {"key": key, "npv": None, "dp": None, "sec": sec}
    key_file = None
    if use_file:
        import tempfile
        from wasmsnark_amd import formats
        key_file = args.key_file if args.key_file != "auto" else os.path.join(os.environ.get("WSNARK_BENCH_KEY_DIR") or tempfile.gettempdir(),
                                                                              "wsnark_bench_key_2p%d_%s.wsnark64" % (logd, args.circuit))
        if rank == 0:
            t_w = time.perf_counter()
            nbytes = formats.write_key_container(sec, key_file)
            info["key_file"] = {"path": key_file, "bytes": nbytes, "write_s": round(time.perf_counter() - t_w, 2),
                                "what": "WSNARK64 container written once by rank 0; every rank maps it and reads only its shard (wsnark_pkey_load_file)"}
            info["key_container"] = "WSNARK64 file, points-sharded per rank"
        ctx["dist"].barrier()

    def sections():      # (the fall-back orchestrations want the whole key: ranks that never generated it read the file)
        if holder["sec"] is None:
            from wasmsnark_amd import formats
            holder["sec"] = formats.read_key_container(key_file)
        return holder["sec"]

    def whole_key():
        if holder["key"] is None:
            holder["key"] = bn.load_key(sections=sections())
        return holder["key"]

    def mode_native():
        if "native" in os.environ.get("WSNARK_BENCH_FAIL", "").split(","):       # (tests: exercise the fall-through)
            raise RuntimeError("forced by WSNARK_BENCH_FAIL")
        holder["npv"] = wdist.NativeDistProver(bn, device=dev, path=key_file) if key_file else wdist.NativeDistProver(bn, sec, device=dev)
        return lambda: holder["npv"].prove(d_w.data_ptr(), len(wit), r=r32, s=s32)

    def mode_dist():
        if "dist" in os.environ.get("WSNARK_BENCH_FAIL", "").split(","):
            raise RuntimeError("forced by WSNARK_BENCH_FAIL")
        holder["dp"] = wdist.DistProver(bn, whole_key(), bytes(sections()["pointsH"]), device=dev)
        return lambda: holder["dp"].prove(d_w.data_ptr(), len(wit), r=r32, s=s32)

    def mode_replicated():
        k = whole_key()
        return lambda: wdist.sharded_prove(bn, k, None, r=r32, s=s32, device=dev, d_witness=(d_w.data_ptr(), len(wit)))

    calc_h_mode, fell = "single GPU", []
    if world == 1:
        step = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
    else:
        ladder = [("native", mode_native), ("dist", mode_dist), ("replicated", mode_replicated)]
        ladder = ladder[[m for m, _ in ladder].index(args.calc_h):]
        step = None
        for name, make in ladder:
            ok_here, why = 1, ""
            try:
                cand = make()
                ok_here = int(cand() == want)
                why = "" if ok_here else "proof != closed form"
            except Exception as ex:  # noqa: BLE001
                ok_here, why = 0, repr(ex)[:300]
            flag = torch.tensor([ok_here], dtype=torch.int32, device=(dev if ctx["dist"].get_backend() == "nccl" else "cpu"))
            ctx["dist"].all_reduce(flag, op=ctx["dist"].ReduceOp.MIN)
            if int(flag.item()) == 1:
                step, calc_h_mode = cand, name
                break
            fell.append("%s: %s" % (name, why or "another rank failed"))
        if step is None:
            raise RuntimeError("no N > 1 orchestration produced the closed-form proof on every rank: %s" % "; ".join(fell))
    key = holder["key"]
    first = step()
    dt, kt, last = timed(ctx, step, args.steps, args.warmup)
    ok = bool(first == want and last == want)
    # per-kernel breakdown: two more (untimed) proofs with every kernel bracketed
    bn.lib.c.wsnark_timing_reset()
    bn.lib.c.wsnark_timing_enable(1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    bn.lib.c.wsnark_timing_enable(0)
    kt_all = bn.lib.timing_report()
    # the dominant kernel ALONE, and a proof with nothing overlapped: a few more proofs on ONE queue (WSNARK_PROVE_OVERLAP=0 through the
    # library's A/B switch), HIP events around the accumulations only.  (single GPU: the N > 1 modes bring their own schedule)
    alone_ms = serial_ms = kt_serial = None
    if world == 1 and not args.no_alone_pass:
        bn.lib.tune("PROVE_OVERLAP", 0)
        try:
            for _ in range(2):
                step()
            bn.lib.c.wsnark_timing_reset()
            bn.lib.c.wsnark_timing_enable(2)
            n_alone = max(4, args.steps // 2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n_alone):
                step()
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - t0) / n_alone * 1e3
            bn.lib.c.wsnark_timing_enable(0)
            ka = bn.lib.timing_report().get("msm_accumulate_g1")
            alone_ms = ka[0] / ka[1] if ka and ka[1] else None
            # every kernel ALONE, by HIP events on the one queue: the per-kernel means the bench line quotes (they add up to at most
            # one serialised proof; the rocprofv3 summary of WSNARK_PROVE_OVERLAP=0 must agree with them)
            bn.lib.c.wsnark_timing_reset()
            bn.lib.c.wsnark_timing_enable(1)
            n_k = 4
            for _ in range(n_k):
                step()
            torch.cuda.synchronize()
            bn.lib.c.wsnark_timing_enable(0)
            kt_serial = kernel_ms(bn.lib.timing_report(), n_k)
        finally:
            bn.lib.tune("PROVE_OVERLAP", None)
    # the drop-in call itself: genZKSnarkProof(witness, provingKey) hands over a HOST witness (32 B x nVars of H2D inside
    # every proof).  Timed like the headline, same steps; reported beside `value` (the contract keeps PCIe out of `value`)
    host_ms = host_ok = None
    if world == 1:
        for _ in range(max(2, args.warmup // 2)):
            hp = bn.groth16GenProof(wit, key, r=r32, s=s32)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            hp = bn.groth16GenProof(wit, key, r=r32, s=s32)
        torch.cuda.synchronize()
        host_ms = (time.perf_counter() - t0) / args.steps * 1e3
        host_ok = bool(hp == want)
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    nv, dom = circ.n_vars, circ.domain
    peak = measure_peaks(bn)                  # (rank 0, any world size: ~30 ms after the timed region)
    # passes over the points per sum: the rows of the key's fixed-base tables (13 at 2^20), or the windows of the plain method
    if calc_h_mode == "native":
        # points shards: every rank sums its own n / N pairs over ALL rows of its shard's tables
        sk = holder["npv"].key
        W_own = sk.table["rows_w"] if sk.table["rows_w"] > 1 else (255 + 15) // 16
        pairs = (3 * sk.shard["n_signals"] + sk.shard["n_hexps"]) / 4.0
        shard_info = {"resident_table_bytes_this_rank": int(sk.table["bytes"]), "pairs_this_rank": int(sk.shard["n_signals"]),
                      "hexps_this_rank": int(sk.shard["n_hexps"]), "table_rows": [sk.table["rows_w"], sk.table["rows_h"]]}
    else:
        if key.table["rows_w"] > 1:
            W_all = key.table["rows_w"]
        else:
            c_win = 16 if logd >= 20 else max(4, logd - 4)
            W_all = (255 + c_win - 1) // c_win
        W_own = len(range(rank, W_all, world))
        pairs = (3 * nv + dom) / 4.0                   # msm_accumulate_g1 launches per proof: A, B1, C (nVars pairs) and H (domain pairs)
        shard_info = None
    hbm, alu = rooflines(kt, "msm_accumulate_g1", pairs, W_own, 96, 10, peak, alone_ms=alone_ms, cite_rocprof=(world == 1 and logd == 20))
    g2 = kt.get("msm_accumulate_g2")
    # algorithmic bytes of one proof
    alg_bytes = 32 * nv + 8 * nv + 36 * info["nnz_A_plus_B"] + 64 * nv * 2 + 128 * nv + 64 * (nv - circ.n_public - 1) + 64 * dom + 64 * 6 * dom
    out = {"metric": "BN128 Groth16 prove ms @ 2^%d constraints" % logd, "value": round(ms, 3), "unit": "ms", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": False,
           "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": "BN128 full Groth16 prove, synthetic 2^%d-constraint R1CS (1-3 non-zeros per column of A and B, "
                                  "every variable present), key and witness resident in HBM, r and s injected" % logd
                                  if args.circuit == "columns" else
                                  "BN128 full Groth16 prove, synthetic 2^%d-constraint R1CS, round-1 sparse generator (1-2 terms per row)" % logd,
                      "circuit": info, "parallelism": ((("key POINTS-sharded: 1/%d of every section resident per rank (wsnark_pkey_load_shard), one C call per proof "
                                                         "(wsnark_groth16_prove_dist), " % world) if calc_h_mode == "native" else ("MSM windows sharded w %% %d == rank, " % world))
                                                       + ("CALC_H on the distributed four-step NTT (3 all-to-alls per proof: 3 + 2 + 1 vectors of %d B per rank and peer), H sum points-sharded, "
                                                          % ((circ.domain // world // world) * 32) if calc_h_mode in ("dist", "native") else "CALC_H %s, " % calc_h_mode)
                                                       + "1 all_gather of 576 B records per proof"
                                                       + ("; fell through: " + "; ".join(fell) if fell else ""))
                      + ("" if args.backend == "nccl" else " [transport gloo with host staging: ranks may share a GPU -- functional check, not a scaling figure]")
                      if world > 1 else "1 GPU, no collective", "lanes": int(os.environ.get("WSNARK_LANES", "2")), "device": bn.device_info},
           "proofs_match_toxic_waste_closed_form": ok,
           "proofs_per_s": round(1e3 / ms, 2),
           "drop_in_call": None if host_ms is None else {
               "what": "groth16GenProof(witness, provingKey) / genZKSnarkProof with the witness in (pageable) HOST memory, as the reference's "
                       "callers hold it (src/bn128.js:580; key handle resident): the %d-byte H2D copy is inside every proof" % len(wit),
               "ms": round(host_ms, 3), "proofs_per_s": round(1e3 / host_ms, 2), "steps": args.steps, "same_proof": host_ok,
               "over_resident_witness_ms": round(host_ms - ms, 3)},
           "drop_in_call_ms": None if host_ms is None else round(host_ms, 3),      # (= drop_in_call.ms: the call the north star names, PCIe included)
           "js_drop_in_call_ms": None,                                            # filled by the `node` extra: the same call through the Node.js addon
           "table_memory": table_memory(key, nv, dom) if world == 1 else None,
           "cold": cold if world == 1 else None,
           "shard": shard_info,
           "int_alu_peaks_this_run": peak,
           "prove_algorithmic_bytes": int(alg_bytes), "prove_algorithmic_GBps": round(alg_bytes / (ms / 1e3) / 1e9, 1),
           "prove_hbm_frac": round(alg_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
           "roofline": hbm, "roofline_int_alu": alu,
           "roofline_proof": proof_issue_roofline(ms, ((peak or {}).get("issue_classes") or {}).get("G_lane_ops_per_s")) if (world == 1 and logd == 20 and args.circuit == "columns") else None,
           "serialised_one_queue_ms_per_proof": round(serial_ms, 3) if serial_ms else None,
           "msm_accumulate_g2_avg_launch_ms": round(g2[0] / g2[1], 4) if g2 and g2[1] else None,
           "kernel_ms_per_proof": None if kt_serial is None else dict(kt_serial, _sum=round(sum(kt_serial.values()), 3),
                                                                      _what="each kernel ALONE: HIP events on the one queue of WSNARK_PROVE_OVERLAP=0 proofs (4 proofs right "
                                                                            "after the timed region); _sum <= serialised_one_queue_ms_per_proof + the events' own cost"),
           "kernel_event_spans_two_queues_ms_per_proof": dict(kernel_ms(kt_all, 2), _what="event-bracket SPANS on the launching queue under the shipped two-queue "
                                                              "schedule: a span contains whatever the other queue's kernels took from the chip meanwhile -- NOT kernel times, "
                                                              "they add up to more than a proof"),
           "reference_wasm_8_workers_prove_2p20_s": {"value": REF_WASM_PROVE_2P20_S, "where": "BASELINE.md: survey container (8 vCPU), NOT this box: the reference may not travel"}}
    want_extras = set() if (args.no_extras or world > 1) else set(x for x in args.extras.split(",") if x)
    extras = {}
    if "inflight" in want_extras:
        run_extra(extras, "two_proofs_in_flight", lambda: extra_prove_inflight(ctx, key, d_w, len(wit), r32, s32, want, ms))
    if "power" in want_extras:
        run_extra(extras, "clock_and_power", lambda: extra_power(ctx, key, d_w, len(wit), r32, s32))
    del d_w
    if "msm" in want_extras:
        run_extra(extras, "g1_msm_2p%d" % args.log_n, lambda: extra_msm(ctx, "cold" in want_extras))
    if "ntt" in want_extras:
        run_extra(extras, "ntt_2p22", lambda: extra_ntt(ctx))
        nt = extras.get("ntt_2p22", {})
        live_peak = max([v for k, v in (peak or {}).items() if k.startswith("modmul") and v] or [0])
        if "fwd_plus_inv_ms" in nt:
            # BASELINE config 3 as an object of its own: the one sub-path where the HBM fraction is informative
            m = 1 << 22
            out["roofline_c3"] = {"workload": "BN128 Fr NTT + iNTT, 2^22 coefficients, in place, resident", "bound": "hbm",
                                  "achieved": nt["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nt["hbm_frac"], "traffic": None,
                                  "ms": {"forward_odd0": nt["fwd_odd0_ms"], "forward_odd1": nt["fwd_odd1_ms"], "inverse": nt["inv_ms"], "forward_plus_inverse": nt["fwd_plus_inv_ms"]},
                                  "algorithmic_bytes": 2 * 64 * m,
                                  "int_alu": {"products": int(2 * 13.0 * m),
                                              "note": "11 butterfly + 2 inter-digit twiddle products per coefficient and transform (three passes): the passes are "
                                                      "issue-bound (%s), not HBM-bound" % ntt_issue_note(),
                                              "G_modmul_per_s": round(2 * 13.0 * m / (nt["fwd_plus_inv_ms"] / 1e3) / 1e9, 1),
                                              "frac_of_product_peak": round(2 * 13.0 * m / (nt["fwd_plus_inv_ms"] / 1e3) / 1e9 / live_peak, 4) if live_peak else None}}
    if "sparse" in want_extras and args.circuit == "columns" and logd <= 20:
        key.free()
        run_extra(extras, "prove_sparse_rows_circuit", lambda: extra_prove_sparse(ctx, logd))
        run_extra(extras, "prove_boolean_circuit", lambda: extra_prove_sparse(ctx, logd, "boolean"))
    if "node" in want_extras and logd == 20:
        run_extra(extras, "node_drop_in", lambda: extra_node(ctx, circ, wit, sec))
        nd = extras.get("node_drop_in", {})
        if "key_bytes_call_ms" in nd:
            out["js_drop_in_call_ms"] = nd["key_bytes_call_ms"]
    if extras:
        out["extras"] = extras
    # cpu_baseline: timed on rank 0 at N = 1 only (the contract); an N > 1 line carries the last N = 1 measurement of this
    # checkout on this box if there is one (.bench_cache/, written below), else the committed one, and says which
    cache = os.path.join(ROOT, ".bench_cache", "cpu_baseline_2p%d.json" % logd)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_prove(ctx, logd, circ, wit, sec)
        try:
            if "value" in out["cpu_baseline"]:
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                json.dump(dict(out["cpu_baseline"], measured_by="bench.py --gpus 1 on this box", device=bn.device_info), open(cache, "w"))
        except Exception:  # noqa: BLE001
            pass
    elif world > 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cached_cpu_baseline(cache, logd)
    return out


def table_rows_bytes(n, h):
    """the library's table geometry (csrc/msm.hip: msm_table_window / msm_table_rows): bytes of the five sections as fixed-base tables"""
    def c(n):
        lg = max(n, 1).bit_length() - 1
        w = lg + 1 if n * 4 >= (5 << lg) else lg
        return min(max(w, 4), 20)
    rows = lambda n: -(-255 // c(n))
    return n * 320 * rows(n) + h * 64 * rows(h), rows(n), rows(h)


def table_memory(key, nv, dom):
    """What the resident key costs and what it buys: this key's bytes as tables / as plain sections, the 2^24 figures on
    one GPU and per points shard of 8, and the measured sweep over which sections are tables (profiles/r05_table_sweep.txt)."""
    gb = lambda b: round(b / 2.0**30, 2)
    b24, _, _ = table_rows_bytes((1 << 24) + 2, 1 << 24)
    b24s, _, _ = table_rows_bytes(((1 << 24) + 2) // 8, (1 << 24) // 8)
    return {"this_key_table_GiB": gb(key.table["bytes"]), "this_key_plain_sections_GiB": gb(nv * 320 + dom * 64),
            "rows": [key.table["rows_w"], key.table["rows_h"]],
            "2p24_one_gpu_table_GiB": gb(b24), "2p24_per_shard_of_8_table_GiB": gb(b24s), "2p24_plain_sections_GiB": gb(((1 << 24) + 2) * 320 + (1 << 24) * 64),
            "choice": "WSNARK_KEY_TABLE: 1 = all five sections (default), 0 = none, 2 = hExps only, 3 = A / B1 / B2 / C only; measured ms per proof and GiB "
                      "for each at 2^20: profiles/r05_table_sweep.txt"}


def extra_node(ctx, circ, wit, sec):
    """The drop-in call as a Node.js caller makes it (wasmsnark_amd/js: groth16GenProof(witness, keyBytes)): tools/node_bench.js in a
    process of its own on the same GPU, on the benchmark's own key and witness written to a scratch directory."""
    import shutil
    import subprocess
    import tempfile
    from wasmsnark_amd import synth
    if shutil.which("node") is None:
        return {"skipped": "no node on this box"}
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wasmsnark_amd", "js"), "-s"])
    d = tempfile.mkdtemp(prefix="wsnark_bench_node_")
    try:
        kp, wp = os.path.join(d, "proving_key.bin"), os.path.join(d, "witness.bin")
        open(kp, "wb").write(synth.sections_to_pkey(sec))
        open(wp, "wb").write(wit)
        r = subprocess.run(["node", os.path.join(ROOT, "tools", "node_bench.js"), kp, wp, str(min(20, EXTRA_REPS))], capture_output=True, text=True, timeout=600)
        line = [x for x in r.stdout.splitlines() if x.startswith("NODE_BENCH ")]
        if r.returncode or not line:
            return {"error": (r.stdout + r.stderr)[-400:]}
        nb = json.loads(line[0][len("NODE_BENCH "):])
        keep = ("key_bytes_call_ms", "key_bytes_call_trusted_ms", "key_handle_call_ms", "pinned_witness_call_ms", "first_call_ms", "whole_buffer_digest_ms", "reps")
        out = {k: nb[k] for k in keep if k in nb}
        out["what"] = ("key_bytes_call_ms = groth16GenProof(witness, keyBytes) in steady state, all key bytes digested beside every proof (the default); "
                       "..._trusted_ms = {trustCache: true}; key_handle = a loadKey() handle; pinned = the witness in an allocInput() buffer; first_call = key load + first proof of a fresh process")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def bench_group(ctx):
    """--gpus N --single-process: ONE process, N devices, wsnark_group_* (csrc/group.hip).  A step is one proof of the 2^logd circuit
    from a witness in HOST memory (the group call's boundary: every device uploads the whole witness inside the call -- there is no
    resident-witness form of it), so the figure compares with drop_in_call_ms of the N = 1 line, not with its `value`."""
    args, bn, torch = ctx["args"], ctx["bn"], ctx["torch"]
    from wasmsnark_amd import bn128, synth
    logd = args.prove_log_domain
    devs = [int(x) for x in args.group_devices.split(",") if x] or list(range(args.gpus))
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1, style=args.circuit)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    r32, s32 = bytes(range(32)), bytes(range(32, 64))
    want = circ.expected_proof(r32, s32)
    g = bn128.Group(lib=bn.lib, devices=devs)
    t0 = time.perf_counter()
    key = g.load_key(sections=sec)
    t_load = time.perf_counter() - t0
    step = lambda: g.groth16GenProof(wit, key, r=r32, s=s32)
    first = step()
    for _ in range(args.warmup):
        last = step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    # the one-GPU call on the same inputs, same process, for the ratio
    one = bn.load_key(sections=sec)
    for _ in range(3):
        p1 = bn.groth16GenProof(wit, one, r=r32, s=s32)
    t0 = time.perf_counter()
    for _ in range(max(4, args.steps // 2)):
        p1 = bn.groth16GenProof(wit, one, r=r32, s=s32)
    ms1 = (time.perf_counter() - t0) / max(4, args.steps // 2) * 1e3
    out = {"metric": "BN128 Groth16 prove ms @ 2^%d constraints" % logd, "value": round(ms, 3), "unit": "ms", "n_gpus": len(devs),
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": False, "scaling": "strong",
           "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": "BN128 full Groth16 prove, synthetic 2^%d-constraint R1CS, key resident as one points shard per device, witness in HOST "
                                  "memory (uploaded to every device inside the call), r and s injected" % logd,
                      "parallelism": "ONE process, %d devices %s: wsnark_group_prove (a context + host thread per device, %s, 1 gather of 576 B records in host memory)"
                                     % (len(devs), devs, "CALC_H on the distributed four-step NTT: 3 device-to-device exchanges (hipMemcpyPeerAsync) per proof"
                                        if key.distributed_calc_h else "CALC_H complete on every device"),
                      "device": bn.device_info},
           "proofs_match_toxic_waste_closed_form": bool(first == want and last == want),
           "proofs_per_s": round(1e3 / ms, 2), "group_key_load_s": round(t_load, 2),
           "one_gpu_same_call_ms": round(ms1, 3), "one_gpu_proof_matches": bool(p1 == want), "speedup_over_one_gpu_same_call": round(ms1 / ms, 3),
           "note": ("several contexts share device(s) %s: a functional check, not a scaling figure" % sorted(set(devs))) if len(set(devs)) < len(devs) else None}
    key.free(); one.free(); g.terminate()
    return out


def cached_cpu_baseline(cache, logd):
    try:
        cb = json.load(open(cache))
        cb["from"] = "the N = 1 run of this checkout on this box (.bench_cache/): the CPU leg is timed at N = 1 only"
        return cb
    except Exception:  # noqa: BLE001
        pass
    import glob
    import re
    c = sorted(x for x in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*bench.json"))
               if re.fullmatch(r"r\d\d_s\d+_bench\.json", os.path.basename(x)))          # (rNN_sM_bench.json: not ..._node_bench.json)
    f = c[-1] if (c and logd == 20) else None
    try:
        cb = json.load(open(f))["cpu_baseline"]
        cb["from"] = "profiles/%s: ANOTHER box (no N = 1 run of this checkout was found here); the CPU leg is timed at N = 1 only" % os.path.basename(f)
        return cb
    except Exception:  # noqa: BLE001
        return None


def run_extra(extras, name, fn):
    try:
        extras[name] = fn()
    except Exception as e:  # noqa: BLE001
        extras[name] = {"error": repr(e)}


def extra_prove_inflight(ctx, key, d_w, wlen, r32, s32, want, ms_single):
    """Two host threads, one key handle: each proof holds a lane; one proof's reduction tails leave SIMDs idle that the
    other's full-width kernels take."""
    bn, torch = ctx["bn"], ctx["torch"]
    reps, bad = min(10, EXTRA_REPS), []

    def worker():
        for _ in range(reps):
            if bn.groth16GenProof_dev(d_w.data_ptr(), wlen, key, r=r32, s=s32) != want:
                bad.append(1)

    lanes = int(os.environ.get("WSNARK_LANES", "2"))
    res = {"one_at_a_time_per_s": round(1e3 / ms_single, 2)}
    for nthreads in sorted({2, lanes} - {1}):
        th = [threading.Thread(target=worker) for _ in range(nthreads)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        key_ = "" if nthreads == 2 else "_%d_threads" % nthreads
        res["prove_throughput_per_s" + key_] = round(nthreads * reps / dt, 2)
        res["ms_per_proof_amortised" + key_] = round(dt / (nthreads * reps) * 1e3, 3)
    res["all_proofs_identical_to_closed_form"] = not bad
    return res


def _gpu_power_sampler():
    """-> a function that returns (sclk_MHz, socket_W) of GPU 0 now, or None: rocm-smi's own figures (the firmware's metrics table;
    an ordinary-user read, no setting touched).  (The amdgpu hwmon node's freq1_input / power1_average read 1 034 MHz / 296 W under a
    load that rocm-smi -- and the proof rate -- put at 2 230 MHz / 1 300 W: gpurun_out r06_s3; not used.)"""
    import subprocess

    def smi():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            mhz = [ln for ln in t.splitlines() if "sclk" in ln][0].split("(")[1].split("Mhz")[0]
            w = [ln for ln in t.splitlines() if "ower" in ln and "(W)" in ln][0].split(":")[-1]
            return float(mhz), float(w)
        except Exception:  # noqa: BLE001
            return None
    return smi if smi() is not None else None


def extra_power(ctx, key, d_w, wlen, r32, s32):
    """What clock and socket power the chip runs proofs at, against the clock / power of the probes that define the integer peak
    (wsnark_peak_probe 1: the product chain).  A power-managed part does not run every instruction mix at one clock: the peak
    (G products/s) is measured at the probe's clock, the proof runs at its own."""
    bn = ctx["bn"]
    import ctypes as C
    read = _gpu_power_sampler()
    if read is None:
        return {"error": "no readable clock / power source (sysfs hwmon, rocm-smi)"}

    def sample_while(fn, seconds):
        got, stop = [], threading.Event()

        def sampler():
            time.sleep(0.5)                                 # (the firmware's averaging window)
            while not stop.is_set():
                v = read()                                  # (~0.2 s per call)
                if v:
                    got.append(v)
        th = threading.Thread(target=sampler); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            fn(); n += 1
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        if not got:
            return None
        return {"sclk_MHz": round(sum(g[0] for g in got) / len(got)), "socket_W": round(sum(g[1] for g in got) / len(got)),
                "samples": len(got), "calls": n, "ms_per_call": round(dt / n * 1e3, 3)}

    def probe(which):
        v = C.c_double(0)
        bn.lib.check(bn.lib.c.wsnark_peak_probe(which, C.byref(v)))
    res = {"idle": dict(zip(("sclk_MHz", "socket_W"), [round(x) for x in read()])),
           "proofs_back_to_back": sample_while(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), wlen, key, r=r32, s=s32), 4.0),
           "product_chain_probe": sample_while(lambda: probe(1), 3.0),
           "multiply_add_probe": sample_while(lambda: probe(10), 3.0)}
    a, b = res["proofs_back_to_back"], res["product_chain_probe"]
    if a and b and b["sclk_MHz"]:
        res["proof_clock_over_probe_clock"] = round(a["sclk_MHz"] / b["sclk_MHz"], 4)
        res["note"] = ("the issue rates and the product peak of this line are measured by probes that run at product_chain_probe.sclk_MHz; whole "
                       "proofs hold the socket at its power limit and run at proofs_back_to_back.sclk_MHz: every `frac` against those peaks "
                       "contains that clock ratio (profiles/r06_clock_power_probe.txt)")
    return res


def msm_inputs(ctx, log_n, seed):
    import numpy as np
    bn, torch, dev = ctx["bn"], ctx["torch"], ctx["dev"]
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    # scalars: uniform 253-bit (< r); points: k_i * G for uniform k_i (distinct valid curve points)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    pts = bn.mul_base(1, ks.tobytes())
    d_s = torch.from_numpy(sc.reshape(-1)).to(dev)
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    return n, rng, sc, pts, d_s, d_p


def extra_msm(ctx, cold):
    """BASELINE configs[1]: one G1 MSM over 2^20 resident pairs per call."""
    bn, torch, args = ctx["bn"], ctx["torch"], ctx["args"]
    n, rng, sc, pts, d_s, d_p = msm_inputs(ctx, args.log_n, 1234)
    call = lambda: bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
    for _ in range(min(10, EXTRA_REPS)):
        ref = call()
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = EXTRA_REPS
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    bn.lib.c.wsnark_timing_enable(0)
    kt = bn.lib.timing_report()
    peak = measure_peaks(bn)
    live = max([v for k, v in peak.items() if k.startswith("modmul") and v] or [0]) or MODMUL_PEAK_G
    hbm, alu = rooflines(kt, "msm_accumulate_g1", n, 16 if args.log_n >= 20 else (255 + args.log_n - 5) // (args.log_n - 4), 96, 10, peak)
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
    for _ in range(3):
        call()
    torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
    res = {"ms": round(t * 1e3, 4), "Mpoints_per_s": round(n / t / 1e6, 2), "kernel_ms": kernel_ms(bn.lib.timing_report(), 3),
           "roofline": hbm, "roofline_int_alu": alu,
           "whole_msm_frac_of_multiplier_peak": round(10 * 16 * n / t / 1e9 / live, 4) if args.log_n == 20 else None}
    # the same bases made resident once (wsnark_points_load: fixed-base window tables) -- what a caller that sums over one base
    # set again and again pays per call: no points upload / preparation, ceil(254 / table c) windows instead of 16
    if args.log_n >= 16:
        t0 = time.perf_counter()
        rp = bn.load_points(1, pts)
        t_load = time.perf_counter() - t0
        rcall = lambda: rp.multiexp_dev(d_s.data_ptr(), n)
        same = all(rcall() == ref for _ in range(min(10, EXTRA_REPS)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            rcall()
        torch.cuda.synchronize()
        tr = (time.perf_counter() - t0) / reps
        rows = rp.table["rows"]
        res["resident_bases"] = {"ms": round(tr * 1e3, 4), "Mpoints_per_s": round(n / tr / 1e6, 2), "same_result_as_per_call": same,
                                 "table": rp.table, "load_ms": round(t_load * 1e3, 2),
                                 "frac_of_multiplier_peak": round(10 * rows * n / tr / 1e9 / live, 4),
                                 "frac_priced_as_the_per_call_sum": round(10 * 16 * n / tr / 1e9 / live, 4) if args.log_n == 20 else None,
                                 "note": "frac_of_multiplier_peak counts the additions this path performs (rows windows x n x 10 products); "
                                         "frac_priced_as_the_per_call_sum prices the same call at the 16 windows the per-call sum needs, "
                                         "i.e. whole_msm_frac_of_multiplier_peak's numerator over this path's time"}
        rp.free()
    # two MSMs in flight (two host threads, two lanes)
    bad = []

    def worker():
        for _ in range(reps):
            if call() != ref:
                bad.append(1)

    th = [threading.Thread(target=worker) for _ in range(2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / (2 * reps)
    res["two_in_flight"] = {"ms_amortised": round(t2 * 1e3, 4), "Mpoints_per_s": round(n / t2 / 1e6, 2), "same_results": not bad}
    # circuit-like scalar histogram (6.7 % zeros, 3.1 % ones, 10 % < 2^32): hot buckets
    u = rng.random(n)
    sk = sc.copy()
    sk[u < 0.067] = 0
    ones = (u >= 0.067) & (u < 0.098)
    sk[ones] = 0
    sk[ones, 0] = 1
    small = (u >= 0.098) & (u < 0.2)
    sk[small, 4:] = 0
    d_sk = torch.from_numpy(sk.reshape(-1)).to(ctx["dev"])
    torch.cuda.synchronize()
    bn.g1_multiexp_dev(d_sk.data_ptr(), d_p.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(5):
        bn.g1_multiexp_dev(d_sk.data_ptr(), d_p.data_ptr(), n)
    res["circuit_like_scalars_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 4)
    if cold:   # PCIe-inclusive: host pointers, 96 MB H2D per call (never `value`)
        sb, pb = sc.tobytes(), pts
        for _ in range(2):
            c = bn.g1_multiexp(sb, pb)
        t0 = time.perf_counter()
        for _ in range(5):
            bn.g1_multiexp(sb, pb)
        res["from_host_pointers_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        res["from_host_pointers_same_result"] = bool(c == ref)
    return res


def extra_ntt(ctx):
    """BASELINE configs[2]: 2^22 coefficients in place; forward odd=0, forward odd=1 and inverse each on their own."""
    import numpy as np
    bn, torch, dev = ctx["bn"], ctx["torch"], ctx["dev"]
    m = 1 << 22
    x = torch.from_numpy(np.random.default_rng(7).integers(0, 256, size=(m, 32), dtype=np.uint8))
    x[:, 31] &= 0x1F
    dx = x.reshape(-1).to(dev)
    for _ in range(12):      # tables + clocks: the first ~40 transforms after an idle gap run 15-20 % slower (tools/ntt_probe.py)
        for odd, inv in ((0, False), (1, False), (0, True)):
            bn.fft_dev(dx.data_ptr(), m, odd, inverse=inv)
    bn.lib.c.wsnark_timing_report(None, 0)
    res, reps = {}, 20
    for name, odd, inv in (("fwd_odd0_ms", 0, False), ("fwd_odd1_ms", 1, False), ("inv_ms", 0, True)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            bn.fft_dev(dx.data_ptr(), m, odd, inverse=inv)
        bn.lib.c.wsnark_timing_report(None, 0)   # syncs the library's streams
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    pair = res["fwd_odd0_ms"] + res["inv_ms"]
    res["fwd_plus_inv_ms"] = round(pair, 4)
    res["algorithmic_GBps"] = round(2 * 64.0 * m / (pair / 1e3) / 1e9, 1)   # 64 B / coefficient / transform
    res["hbm_frac"] = round(2 * 64.0 * m / (pair / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
    bn.fft_dev(dx.data_ptr(), m, 0)
    bn.fft_dev(dx.data_ptr(), m, 0, inverse=True)
    bn.lib.c.wsnark_timing_enable(0)
    kt = bn.lib.timing_report()
    res["kernel_ms_per_launch"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}
    return res


def extra_prove_sparse(ctx, logd, style="rows"):
    """Key-dependent extras.  "rows": round 1's generator (1-2 terms per ROW) leaves ~40 % of the variables out of A and of B;
    their key points are infinity and the sums run on plan variants that skip them.  "boolean" (round 6): a VALID bit-decomposition
    circuit -- 87.5 % of the witness is 0 / 1 (csrc/synth.hip style 2): the digit-0 / digit-1 buckets of every window row are very
    hot (msm_plan_emit_hot, the hot role of msm_combine_all), zero scalars drop out of the sums, one B column has 2^20 entries."""
    bn, torch, dev = ctx["bn"], ctx["torch"], ctx["dev"]
    from wasmsnark_amd import synth
    circ, key, wit, info = build_prover(bn, logd, style)
    info.pop("_cold", None); info.pop("_sections", None)
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to(dev)
    r32, s32 = bytes(range(32)), bytes(range(32, 64))
    for _ in range(3):
        p = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    ok = p == circ.expected_proof(r32, s32)
    key.free()
    return {"prove_ms": round(t * 1e3, 3), "matches_closed_form": bool(ok), "circuit": info}


def cpu_baseline_prove(ctx, logd, circ, wit, sec):
    """The oracle's restatement of the reference prover (w = 7 subset-table multiexp split over worker threads, CALC_H on one
    thread like the reference's worker, src/bn128.js:126-166, 353-415) timed on THIS box's host cores on THE benchmark's own
    circuit and witness -- no scaling factor.  Two figures: all host cores (the reference's worker-pool shape widened to the
    box), and one thread (CALC_H at full size; the sums on a stated sample of the same pairs, because a one-thread 2^20 G2
    sum alone would take minutes)."""
    import ctypes as C
    bn = ctx["bn"]
    try:
        from oracle import pyoracle as orc
        from wasmsnark_amd import synth
        if logd > 21:
            return {"skipped": "the oracle takes proving_key.bin (u32 offsets); the CPU leg is timed on the 2^20 workload only"}
        host_threads = os.cpu_count() or 1
        # one worker per PHYSICAL core: the sums are integer-multiply bound, SMT siblings only add contention (measured on the
        # 2 x 64-core box: 256 threads 39.2 s, 64 threads 24.0 s)
        cores = max(1, host_threads // 2) if host_threads >= 16 else host_threads
        pkey = bytearray(synth.sections_to_pkey(sec))
        nv, dom = circ.n_vars, circ.domain
        r32, s32 = bytes(range(32)), bytes(range(32, 64))
        L = orc.lib()
        cb = lambda b: (C.c_uint8 * len(b)).from_buffer(b)
        wbuf, kbuf, out = cb(bytearray(wit)), cb(pkey), (C.c_uint8 * 384)()
        t0 = time.perf_counter()
        rc = L.orc_groth16_prove(wbuf, C.c_size_t(len(wit)), kbuf, C.c_size_t(len(pkey)), cb(bytearray(r32)), cb(bytearray(s32)), cores, out)
        t_all = time.perf_counter() - t0
        got = orc.proof_from_bytes(bytes(out)) if rc == 0 else None
        match = bool(got == circ.expected_proof(r32, s32))
        # one thread: CALC_H at full size, the G1 / G2 sums on the first 2^16 / 2^14 pairs of the same key and witness
        t0 = time.perf_counter()
        orc.calc_h(wit, sec["polsA"], sec["polsB"], nv, dom)
        t_h = time.perf_counter() - t0
        n1, n2 = min(nv, 1 << 16), min(nv, 1 << 14)
        t0 = time.perf_counter()
        orc.multiexp(1, "multiexp2", wit[:n1 * 32], bytes(sec["pointsA"][:n1 * 64]), n1)
        t_g1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.multiexp(2, "multiexp", wit[:n2 * 32], bytes(sec["pointsB2"][:n2 * 128]), n2)
        t_g2 = time.perf_counter() - t0
        # the reference's own shape -- 8 workers (src/bn128.js:214) -- on a bounded sample: a whole proof of the same generator's 2^17
        # circuit (1/8 of the pairs: ~20 s here; the 2^20 proof on 8 threads would take minutes)
        eight = None
        try:
            c17 = synth.NativeCircuit(bn.lib, max(4, logd - 3), n_public=5 if logd >= 8 else 1, seed=1, style="columns")
            s17, _ = c17.build_sections()
            k17, w17 = bytearray(synth.sections_to_pkey(s17)), bytearray(c17.witness_bin())
            o17 = (C.c_uint8 * 384)()
            t0 = time.perf_counter()
            rc17 = L.orc_groth16_prove(cb(w17), C.c_size_t(len(w17)), cb(k17), C.c_size_t(len(k17)), cb(bytearray(r32)), cb(bytearray(s32)), 8, o17)
            t17 = time.perf_counter() - t0
            eight = {"cores": 8, "ms": round(t17 * 1e3, 1), "sample": "one whole proof of a smaller circuit of the same generator (n_vars %d) on 8 threads: the "
                                                                      "reference's worker count, src/bn128.js:214 (2^%d: 1/8 of the benchmark's pairs)" % (c17.n_vars, max(4, logd - 3)),
                     "cpu_proof_matches_closed_form": bool(rc17 == 0 and orc.proof_from_bytes(bytes(o17)) == c17.expected_proof(r32, s32)),
                     "note": "the sums scale with the pairs (x8 to 2^20), CALC_H with n log n: the survey's reference run at 2^20 on 8 vCPU was %.1f s" % REF_WASM_PROVE_2P20_S}
        except Exception as e17:  # noqa: BLE001
            eight = {"error": repr(e17)}
        return {"value": round(t_all * 1e3, 1), "unit": "ms", "cores": cores, "host_cores": host_threads, "kind": "port", "eight_threads_sample": eight,
                "sample": "one whole proof of the benchmark's own 2^%d circuit and witness (n_vars %d, nnz %d) by the oracle's groth16GenProof "
                          "restatement: every sum split over %d threads, CALC_H on one thread like the reference's worker; %.2f s"
                          % (logd, nv, circ.nnz, cores, t_all),
                "sample_seconds": round(t_all, 2), "cpu_proof_matches_closed_form": match,
                "one_thread": {"cores": 1, "calc_h_full_size_ms": round(t_h * 1e3, 1),
                               "g1_multiexp2_first_2p%d_pairs_ms" % (n1.bit_length() - 1): round(t_g1 * 1e3, 1),
                               "g2_multiexp_first_2p%d_pairs_ms" % (n2.bit_length() - 1): round(t_g2 * 1e3, 1),
                               "us_per_g1_pair": round(t_g1 / n1 * 1e6, 2), "us_per_g2_pair": round(t_g2 / n2 * 1e6, 2),
                               "note": "a stated sample for the sums (same pairs as the proof's A and B2 sums), CALC_H at full size; nothing is multiplied up"},
                "reference_wasm_8_workers_prove_2p20_s": REF_WASM_PROVE_2P20_S,
                "reference_note": "the reference's own figure was recorded in the survey container (8 vCPU), other hardware: "
                                  "nothing of /root/reference may travel to this box"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def bench_msm(ctx):
    """Round 1's line: one G1 MSM of 2^log_n resident pairs per step; N > 1: every rank its own pairs (weak scaling, the
    reference's contiguous split, src/bn128.js:353-383) or --shard windows (one MSM, strong scaling); one all_gather of
    the 96-byte partials + local EC sum per step."""
    args, bn, rank, world, dev, torch = ctx["args"], ctx["bn"], ctx["rank"], ctx["world"], ctx["dev"], ctx["torch"]
    from wasmsnark_amd import dist as wdist
    windows = args.shard == "windows" and world > 1
    n, rng, sc, pts, d_s, d_p = msm_inputs(ctx, args.log_n, 1234 + (0 if windows else rank))
    shard = (rank, world) if windows else None

    def step():
        part = bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n, shard=shard)
        return wdist.sharded_msm(bn, 1, part, dev) if world > 1 else part

    dt, kt, _ = timed(ctx, step, args.steps, args.warmup)
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    W_all = 16 if args.log_n >= 20 else (255 + args.log_n - 5) // max(args.log_n - 4, 4)
    hbm, alu = rooflines(kt, "msm_accumulate_g1", n, len(range(rank, W_all, world)) if windows else W_all, 96, 10)
    out = {"metric": "BN128 G1 MSM Mpoints/s (2^%d pairs/GPU)" % args.log_n,
           "value": round((1 if windows else world) * n / (dt / args.steps) / 1e6, 3), "unit": "Mpoints/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
           "scaling": "strong" if windows else "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": "BN128 G1 Pippenger MSM, 2^%d random (scalar,point) pairs per GPU, inputs resident in HBM" % args.log_n,
                      "pairs_per_gpu": n, "parallelism": ("windows-sharded x%d" if windows else "points-sharded x%d") % world + ", 1 all_gather of 96 B partials",
                      "device": bn.device_info},
           "roofline": hbm, "roofline_int_alu": alu}
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
            out["cpu_baseline"] = {"value": round(ns / tc / 1e6, 5), "unit": "Mpoints/s", "cores": threads, "host_cores": cores, "kind": "port",
                                   "sample": "first 2^%d pairs of the same workload, oracle g1m_multiexp2 restatement (w=7), contiguous split over %d threads" % (ns.bit_length() - 1, threads),
                                   "seconds": round(tc, 2), "gpu_result_matches": orc.g_affine(1, got) == chk,
                                   "reference_wasm_8_workers_2p20_Mpoints_s": REF_WASM_MSM_MPTS,
                                   "reference_note": "recorded in the survey container, other hardware"}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
