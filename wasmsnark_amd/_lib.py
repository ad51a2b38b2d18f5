"""ctypes loader for libwsnark.so (the hand-written HIP library behind include/wsnark.h).

There is deliberately NO fallback: if the shared library is missing, or no GPU is visible,
loading / init fails loudly.  (tests/emul builds a CPU thread-emulator of the same kernel
sources for index-math tests; it is only ever loaded through `load(path=...)` by tests.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, "libwsnark.so")

ERRORS = {1: "WSNARK_ERR_SIZE", 2: "WSNARK_ERR_FORMAT", 3: "WSNARK_ERR_HIP", 4: "WSNARK_ERR_ARG", 5: "WSNARK_ERR_NOINIT"}

# every symbol include/wsnark.h declares
SYMBOLS = [
    "wsnark_init", "wsnark_shutdown", "wsnark_last_error", "wsnark_device_info",
    "wsnark_g1_msm", "wsnark_g2_msm", "wsnark_g1_msm_dev", "wsnark_g2_msm_dev",
    "wsnark_g1_msm_windows", "wsnark_g2_msm_windows", "wsnark_g1_msm_windows_dev", "wsnark_g2_msm_windows_dev",
    "wsnark_g1_sum", "wsnark_g2_sum",
    "wsnark_fr_ntt", "wsnark_fr_ntt_dev", "wsnark_fr_ntt_batch_dev", "wsnark_fr_dist_scale_dev",
    "wsnark_pkey_eval_ab_dev", "wsnark_fr_mul_dev", "wsnark_fr_dist_combine_dev", "wsnark_fr_to_montgomery", "wsnark_fr_from_montgomery",
    "wsnark_calc_h", "wsnark_pkey_load", "wsnark_pkey_free", "wsnark_pkey_info", "wsnark_pkey_table_info",
    "wsnark_groth16_prove", "wsnark_groth16_prove_dev", "wsnark_pkey_load_sections", "wsnark_pkey_load_shard", "wsnark_pkey_load_file", "wsnark_pkey_file_info", "wsnark_pkey_shard_info", "wsnark_pkey_load_stats", "wsnark_pkey_wait_tables", "wsnark_pkey_h_msm_dev", "wsnark_last_blinding", "wsnark_groth16_verify",
    "wsnark_groth16_prove_partial", "wsnark_groth16_prove_partial_dev", "wsnark_groth16_prove_finish", "wsnark_groth16_prove_dist",
    "wsnark_g1_mul_base_batch", "wsnark_g2_mul_base_batch",
    "wsnark_synth_new", "wsnark_synth_free", "wsnark_synth_info", "wsnark_synth_witness", "wsnark_synth_pols",
    "wsnark_synth_key_scalars", "wsnark_synth_expected",
    "wsnark_selftest_field", "wsnark_selftest_curve",
    "wsnark_timing_enable", "wsnark_timing_reset", "wsnark_timing_report", "wsnark_peak_probe", "wsnark_tuning_set", "wsnark_host_alloc", "wsnark_host_free",
    "wsnark_points_load", "wsnark_points_free", "wsnark_points_info", "wsnark_points_msm", "wsnark_points_msm_dev",
    "wsnark_group_create", "wsnark_group_free", "wsnark_group_size", "wsnark_group_pkey_load", "wsnark_group_pkey_load_sections", "wsnark_group_pkey_load_file", "wsnark_group_pkey_free",
    "wsnark_group_pkey_info", "wsnark_group_pkey_wait_tables", "wsnark_group_prove", "wsnark_group_last_blinding", "wsnark_group_g1_msm", "wsnark_group_g2_msm",
]


class WsnarkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


class Lib:
    # The product binds ITS OWN library and nothing else: no path argument, no environment override.  (The test-suite's CPU
    # thread-emulator build of the same ABI is bound by a subclass that lives under tests/: tests/emul_util.py.)
    SO = DEFAULT_SO

    def __init__(self):
        path = type(self).SO
        if not os.path.exists(path):
            raise ImportError(
                "libwsnark.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        # One HIP runtime per process.  The Python host side keeps device buffers in PyTorch tensors (and uses
        # torch.distributed for N > 1), and PyTorch-ROCm bundles its own libamdhip64: whichever library is loaded FIRST
        # decides which runtime the soname resolves to.  If this library came first, its /opt/rocm runtime and PyTorch's
        # bundled one would both open the device and the second to initialise fails ("no ROCm-capable device is
        # detected" -- seen on the MI355X when build() loaded the library before smoke() imported torch).  So PyTorch's
        # goes in first whenever PyTorch is installed; hosts without it (the Node addon) use /opt/rocm's alone.
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
        self.c = C.CDLL(path)
        for s in SYMBOLS:
            getattr(self.c, s)  # AttributeError if the ABI is incomplete
        c = self.c
        c.wsnark_last_error.restype = C.c_char_p
        c.wsnark_device_info.restype = C.c_char_p
        c.wsnark_timing_report.restype = C.c_size_t
        c.wsnark_timing_report.argtypes = [C.c_void_p, C.c_size_t]
        u64, vp, sz, u32 = C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint32
        c.wsnark_g1_msm.argtypes = [vp, vp, u64, vp]
        c.wsnark_g2_msm.argtypes = [vp, vp, u64, vp]
        c.wsnark_g1_msm_dev.argtypes = [vp, vp, u64, vp, vp]
        c.wsnark_g2_msm_dev.argtypes = [vp, vp, u64, vp, vp]
        c.wsnark_g1_msm_windows.argtypes = [vp, vp, u64, u32, u32, vp]
        c.wsnark_g2_msm_windows.argtypes = [vp, vp, u64, u32, u32, vp]
        c.wsnark_g1_msm_windows_dev.argtypes = [vp, vp, u64, u32, u32, vp, vp]
        c.wsnark_g2_msm_windows_dev.argtypes = [vp, vp, u64, u32, u32, vp, vp]
        c.wsnark_g1_sum.argtypes = [vp, u64, vp]
        c.wsnark_g2_sum.argtypes = [vp, u64, vp]
        c.wsnark_fr_ntt.argtypes = [vp, u64, C.c_int, C.c_int]
        c.wsnark_fr_ntt_dev.argtypes = [vp, u64, C.c_int, C.c_int, vp]
        c.wsnark_fr_ntt_batch_dev.argtypes = [vp, u64, u64, C.c_int, vp]
        c.wsnark_fr_dist_scale_dev.argtypes = [vp, u64, u64, u64, u64, u32, u32, C.c_int, C.c_int, vp]
        c.wsnark_fr_to_montgomery.argtypes = [vp, vp, u64]
        c.wsnark_fr_from_montgomery.argtypes = [vp, vp, u64]
        c.wsnark_calc_h.argtypes = [vp, vp, sz, vp, sz, u32, u32, vp]
        c.wsnark_pkey_load.argtypes = [vp, sz, C.POINTER(vp)]
        c.wsnark_pkey_free.argtypes = [vp]
        c.wsnark_pkey_free.restype = None
        c.wsnark_pkey_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        c.wsnark_pkey_table_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)]
        c.wsnark_groth16_prove.argtypes = [vp, vp, sz, vp, vp, vp]
        c.wsnark_groth16_prove_dev.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        c.wsnark_pkey_load_sections.argtypes = [vp, C.POINTER(vp)]
        c.wsnark_pkey_load_shard.argtypes = [vp, u32, u32, u32, C.POINTER(vp)]
        c.wsnark_pkey_load_file.argtypes = [C.c_char_p, u32, u32, u32, C.POINTER(vp)]
        c.wsnark_pkey_file_info.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(C.c_int)]
        c.wsnark_group_pkey_load_file.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        c.wsnark_pkey_shard_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
        c.wsnark_peak_probe.argtypes = [C.c_int, C.POINTER(C.c_double)]
        c.wsnark_tuning_set.argtypes = [C.c_char_p, C.c_int64]
        c.wsnark_host_alloc.argtypes = [sz, C.POINTER(vp)]
        c.wsnark_host_free.argtypes = [vp]
        c.wsnark_host_free.restype = None
        c.wsnark_pkey_load_stats.argtypes = [vp, C.POINTER(C.c_double)]
        c.wsnark_pkey_wait_tables.argtypes = [vp]
        c.wsnark_pkey_h_msm_dev.argtypes = [vp, vp, u64, vp, vp]
        c.wsnark_groth16_prove_partial.argtypes = [vp, vp, sz, u32, u32, u32, vp]
        c.wsnark_groth16_prove_partial_dev.argtypes = [vp, vp, sz, u32, u32, u32, vp, vp]
        c.wsnark_groth16_prove_dist.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
        c.wsnark_pkey_eval_ab_dev.argtypes = [vp, vp, sz, vp, vp, vp]
        c.wsnark_fr_mul_dev.argtypes = [vp, vp, vp, u64, vp]
        c.wsnark_fr_dist_combine_dev.argtypes = [vp, vp, vp, u64, u64, u64, u32, u32, vp]
        c.wsnark_last_blinding.argtypes = [vp, vp]
        c.wsnark_groth16_verify.argtypes = [vp, sz, vp, u64, vp, C.POINTER(C.c_int)]
        c.wsnark_selftest_field.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, u64]
        c.wsnark_selftest_curve.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, u64]
        c.wsnark_groth16_prove_finish.argtypes = [vp, vp, u64, vp, vp, vp]
        c.wsnark_g1_mul_base_batch.argtypes = [vp, vp, u64, vp]
        c.wsnark_g2_mul_base_batch.argtypes = [vp, vp, u64, vp]
        c.wsnark_synth_new.argtypes = [u32, u32, u64, u64, C.c_int, C.POINTER(vp)]
        c.wsnark_synth_free.argtypes = [vp]
        c.wsnark_synth_free.restype = None
        c.wsnark_synth_info.argtypes = [vp, vp]
        c.wsnark_synth_witness.argtypes = [vp, vp]
        c.wsnark_synth_pols.argtypes = [vp, C.c_int, vp, u64]
        c.wsnark_synth_key_scalars.argtypes = [vp, C.c_int, vp]
        c.wsnark_synth_expected.argtypes = [vp, vp, vp, vp]
        c.wsnark_points_load.argtypes = [C.c_int, vp, u64, C.POINTER(vp)]
        c.wsnark_points_free.argtypes = [vp]
        c.wsnark_points_free.restype = None
        c.wsnark_points_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
        c.wsnark_points_msm.argtypes = [vp, vp, u64, vp]
        c.wsnark_points_msm_dev.argtypes = [vp, vp, u64, vp, vp]
        c.wsnark_group_create.argtypes = [C.POINTER(C.c_int), u32, C.POINTER(vp)]
        c.wsnark_group_free.argtypes = [vp]
        c.wsnark_group_free.restype = None
        c.wsnark_group_size.argtypes = [vp]
        c.wsnark_group_size.restype = u32
        c.wsnark_group_pkey_load.argtypes = [vp, vp, sz, C.POINTER(vp)]
        c.wsnark_group_pkey_load_sections.argtypes = [vp, vp, C.POINTER(vp)]
        c.wsnark_group_pkey_free.argtypes = [vp]
        c.wsnark_group_pkey_free.restype = None
        c.wsnark_group_pkey_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_int)]
        c.wsnark_group_pkey_wait_tables.argtypes = [vp]
        c.wsnark_group_prove.argtypes = [vp, vp, sz, vp, vp, vp]
        c.wsnark_group_last_blinding.argtypes = [vp, vp, vp]
        c.wsnark_group_g1_msm.argtypes = [vp, vp, vp, u64, vp]
        c.wsnark_group_g2_msm.argtypes = [vp, vp, vp, u64, vp]
        self.initialised = False

    def check(self, rc):
        if rc != 0:
            raise WsnarkError(rc, (self.c.wsnark_last_error() or b"").decode())

    def init(self, device=-1):
        self.check(self.c.wsnark_init(device))
        self.initialised = True
        return (self.c.wsnark_device_info() or b"").decode()

    def shutdown(self):
        self.c.wsnark_shutdown()
        self.initialised = False

    def tune(self, name, value=None):
        """A/B switch `name` (the WSNARK_<name> environment variable without its prefix): override it for every later call;
        None forgets the override."""
        self.check(self.c.wsnark_tuning_set(name.encode(), -(1 << 63) if value is None else int(value)))

    def timing_report(self):
        n = self.c.wsnark_timing_report(None, 0)
        buf = C.create_string_buffer(n + 1)
        self.c.wsnark_timing_report(buf, n + 1)
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.split()
            out[name] = (float(ms), int(cnt))
        return out


_default = None


def load():
    """Returns the process-wide Lib (wasmsnark_amd/libwsnark.so)."""
    global _default
    if _default is None:
        _default = Lib()
    return _default
