"""bench.py's control flow on CPU (tools/bench_dryrun.py: the thread-emulator build at toy sizes, gloo instead of nccl):
the N = 1 line with every extra, and the N = 2 launch the driver uses for the scaling run -- window-sharded sums, the
distributed CALC_H and the points-sharded H sum through DistProver, the max-over-ranks timing, one JSON line from rank 0
with the contract's fields.  Not a measurement: a guard against Python-side mistakes in a path that only runs on GPUs."""
import json
import os
import subprocess
import sys

from conftest import ROOT

FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "data", "config", "roofline", "roofline_int_alu")


def _line(out):
    rows = [l for l in out.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert out.returncode == 0 and len(rows) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(rows[0])


def test_bench_single_gpu_line():
    from emul_util import emul_bn128
    emul_bn128()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_dryrun.py"), "--prove-log-domain", "6", "--log-n", "8",
                          "--steps", "2", "--warmup", "1", "--extras", "msm,inflight"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, WSNARK_BENCH_EXTRA_REPS="2"))
    d = _line(out)
    assert all(k in d for k in FIELDS) and d["n_gpus"] == 1 and d["higher_is_better"] is False and d["unit"] == "ms"
    assert d["proofs_match_toxic_waste_closed_form"] is True
    assert d["config"]["workload"].startswith("BN128 full Groth16 prove") and "model" not in d["config"]
    ex = d["extras"]
    assert ex["two_proofs_in_flight"]["all_proofs_identical_to_closed_form"]
    assert ex["g1_msm_2p8"]["two_in_flight"]["same_results"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cpu_proof_matches_closed_form"] is True
    assert "x 16" not in d["cpu_baseline"]["sample"] and d["cpu_baseline"]["one_thread"]["cores"] == 1
    e8 = d["cpu_baseline"]["eight_threads_sample"]                  # the reference's worker count, on a stated smaller sample
    assert e8["cores"] == 8 and e8["cpu_proof_matches_closed_form"] is True and e8["ms"] > 0
    # round 5: the drop-in call's figure at the top level beside `value`, the table memory, the slot of the JS figure
    assert d["drop_in_call_ms"] == d["drop_in_call"]["ms"] and "js_drop_in_call_ms" in d
    assert d["table_memory"]["2p24_one_gpu_table_GiB"] > 70 and d["table_memory"]["2p24_per_shard_of_8_table_GiB"] < 12
    assert d["drop_in_call"]["same_proof"] is True and d["drop_in_call"]["ms"] > 0
    cold = d["cold"]
    assert cold["first_proof_matches_closed_form"] is True and cold["table_bytes"] > 0
    assert set(cold["key_load_ms"]) == {"pols_to_csr", "points_h2d", "masks_convert", "table_build", "total"}
    assert set(d["int_alu_peaks_this_run"]) >= {"modmul_G_per_s", "modmul_inlined_G_per_s", "mad_u64_u32_G_per_s"}   # (values need a GPU clock)
    # round 4: the one-queue pass (the dominant kernel alone) ran, and the whole-proof issue roofline has its slot in the line
    assert d["serialised_one_queue_ms_per_proof"] > 0 and "roofline_proof" in d


def test_bench_two_ranks_line():
    from emul_util import emul_bn128
    emul_bn128()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29657", os.path.join(ROOT, "tools", "bench_dryrun.py"), "--gpus", "2", "--prove-log-domain", "6",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", WSNARK_EMUL_DEVICES="2"))
    d = _line(out)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["proofs_match_toxic_waste_closed_form"] is True
    assert "distributed four-step NTT" in d["config"]["parallelism"] and "extras" not in d
    # the default N > 1 orchestration is the native one (points-sharded key, one C call per proof) and it did not fall through
    assert "wsnark_groth16_prove_dist" in d["config"]["parallelism"] and "fell through" not in d["config"]["parallelism"]
    assert d["shard"]["pairs_this_rank"] > 0 and d["shard"]["resident_table_bytes_this_rank"] > 0
    assert "modmul_G_per_s" in d["int_alu_peaks_this_run"]          # the N > 1 line keeps the integer peak of its own run


def test_bench_two_ranks_from_a_key_file(tmp_path):
    """--key-file: rank 0 writes the key once as a WSNARK64 container, every rank maps it and reads only its shard
    (wsnark_pkey_load_file) -- the N > 1 launch that does not need the whole key in every rank's host memory."""
    from emul_util import emul_bn128
    emul_bn128()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29661", os.path.join(ROOT, "tools", "bench_dryrun.py"), "--gpus", "2", "--prove-log-domain", "6",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--key-file", "auto"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", WSNARK_EMUL_DEVICES="2", WSNARK_BENCH_KEY_DIR=str(tmp_path)))
    d = _line(out)
    assert d["n_gpus"] == 2 and d["proofs_match_toxic_waste_closed_form"] is True
    assert "wsnark_groth16_prove_dist" in d["config"]["parallelism"] and "fell through" not in d["config"]["parallelism"]
    kf = d["config"]["circuit"]["key_file"]
    assert os.path.getsize(kf["path"]) == kf["bytes"] and d["config"]["circuit"]["key_container"].startswith("WSNARK64 file")


def test_bench_two_ranks_falls_through_when_an_orchestration_fails():
    """None of the N > 1 orchestrations has run on RCCL in the build container: bench.py checks each one against the closed form on
    every rank and falls through native -> Python (DistProver) -> replicated CALC_H.  Forced here: the native mode fails on every
    rank, then the Python one too; the line must say which ran and why the others did not."""
    from emul_util import emul_bn128
    emul_bn128()
    for fail, ran in (("native", "CALC_H on the distributed four-step NTT"), ("native,dist", "CALC_H replicated")):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", "29659", os.path.join(ROOT, "tools", "bench_dryrun.py"), "--gpus", "2", "--prove-log-domain", "6",
                              "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                             capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", WSNARK_BENCH_FAIL=fail))
        d = _line(out)
        par = d["config"]["parallelism"]
        assert d["proofs_match_toxic_waste_closed_form"] is True and "fell through: native: RuntimeError" in par and ran in par, par
        assert "wsnark_groth16_prove_dist" not in par and d["shard"] is None


def test_bench_single_process_group_line():
    """python bench.py --gpus 2 --single-process: ONE process drives the devices through wsnark_group_* (two contexts on the
    emulator's one device, then one context on each of two emulated devices: tests/emul/hip_emul.h)"""
    from emul_util import emul_bn128
    emul_bn128()
    for devs, ndev in (("0,0", "1"), ("0,1", "2")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_dryrun.py"), "--gpus", "2", "--single-process", "--group-devices", devs,
                              "--prove-log-domain", "6", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, WSNARK_EMUL_DEVICES=ndev))
        d = _line(out)
        check_group_line(d, shared=(devs == "0,0"))


def check_group_line(d, shared):
    assert d["n_gpus"] == 2 and d["proofs_match_toxic_waste_closed_form"] is True and d["one_gpu_proof_matches"] is True
    assert "wsnark_group_prove" in d["config"]["parallelism"] and "four-step" in d["config"]["parallelism"]
    assert ("functional check" in d["note"]) if shared else d["note"] is None      # (two contexts on ONE device is not a scaling figure)
