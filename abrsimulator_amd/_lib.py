"""ctypes binding of the C ABI in include/abr_env.h (libabr_hip.so).

There is no CPU fallback: if the HIP library is missing or does not export the
full ABI, importing the product fails loudly.  torch is imported first so that
the library's libamdhip64.so.7 dependency resolves to the HIP runtime PyTorch
already loaded (one runtime per process: streams and pointers are shared).
"""
import ctypes as C
import os
import subprocess

# Kernel arguments in device memory rather than host-coherent memory: the role-split kernels re-read their parameter block
# inside the iteration loop (csrc/abr_env_roles.h: fresh_params), and on MI355X that measures +2.5 % at 48 decisions per launch
# and +4.5 % at 20 (same box, interleaved: profiles/r06_ab_dev_kernarg.txt).  The HIP runtime reads the variable when it
# initialises (the first HIP call of the process, not `import torch`), so this default only takes effect when the package is
# imported before anything touched the GPU; a value the caller has set stands.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: F401,E402  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# diagnostic builds of the same ABI (rejected kernels, instrumentation) live OUTSIDE the package: tools/diag/csrc builds them
# into tools/diag/lib.  The package never loads one by itself: ABR_HIP_LIB / BatchedABREnv(library=...) name them explicitly.
DIAG_LIB_DIR = os.path.join(os.path.dirname(_HERE), "tools", "diag", "lib")


def resolve(name):
    """Path of a library given by name: a path as it is; a bare file name in tools/diag/lib, then in csrc/."""
    if os.sep in name or os.path.isabs(name):
        return os.path.abspath(name)
    for d in (DIAG_LIB_DIR, CSRC):
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return os.path.join(DIAG_LIB_DIR, name)


SO_PATH = os.path.join(CSRC, "libabr_hip.so")
if os.environ.get("ABR_HIP_LIB"):          # run everything on a diagnostic build (tools/, A/B scripts)
    SO_PATH = resolve(os.environ["ABR_HIP_LIB"])

ABI_VERSION = 4
MAX_RATES = 16
MAX_HORIZON = 8
OBS_DIM = 8
F64_DIM = 16

OBS_ROWS = ["chunk_id", "last_bitrate", "last_bandwidth", "buffer_level", "global_time",
            "play_time", "rebuffer_time", "start_up_time"]
F64_ROWS = ["global_time", "rebuffer_time", "start_up_time", "play_time", "average_latency",
            "buffer_level", "play_length", "last_bandwidth", "chunk_id", "play_id",
            "last_bitrate", "flags", "hist_n", "hist_sum_inv", "tick", "download_time"]

DONE_EPISODE, DONE_TIMEOUT, DONE_BADACT, DONE_BADARG, DONE_INTERNAL = 1, 2, 4, 8, 16


class EnvConfig(C.Structure):
    _fields_ = [("n_rates", C.c_int32), ("video_length", C.c_int32),
                ("chunk_length", C.c_double), ("max_buffer", C.c_double),
                ("start_up_length", C.c_double), ("interval", C.c_double),
                ("rebuffer_weight", C.c_double), ("variance_weight", C.c_double),
                ("startup_weight", C.c_double), ("latency_weight", C.c_double),
                ("speed", C.c_double), ("ladder", C.c_double * MAX_RATES),
                ("max_ticks", C.c_int32), ("auto_reset", C.c_int32)]


class MpcConfig(C.Structure):
    _fields_ = [("n_rates", C.c_int32), ("horizon", C.c_int32), ("video_length", C.c_int32),
                ("clip_horizon", C.c_int32), ("chunk_length", C.c_double),
                ("max_buffer", C.c_double), ("variance_weight", C.c_double),
                ("rebuffer_weight", C.c_double), ("startup_weight", C.c_double)]


class MpcOptions(C.Structure):
    _fields_ = [("predictor", C.c_int32), ("utility", C.c_int32), ("hist_dev", C.c_void_p),
                ("hist_stride", C.c_int64), ("hist_len_dev", C.c_void_p),
                ("scratch_dev", C.c_void_p), ("scratch_bytes", C.c_size_t),
                ("mask_is_done", C.c_int32), ("reserved_", C.c_int32)]


class StateView(C.Structure):
    _fields_ = [("n_lanes", C.c_int64), ("chunk_id", C.c_void_p), ("last_bitrate", C.c_void_p),
                ("buffer_level", C.c_void_p), ("hist_n", C.c_void_p), ("hist_sum_inv", C.c_void_p),
                ("done", C.c_void_p), ("action_hist", C.c_void_p), ("bw_hist", C.c_void_p)]


# every symbol include/abr_env.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("abr_abi_version", C.c_int, []),
    ("abr_last_error", C.c_char_p, []),
    ("abr_env_workspace_bytes", C.c_int, [C.POINTER(EnvConfig), C.c_int64, C.POINTER(C.c_size_t)]),
    ("abr_env_create", C.c_int, [C.POINTER(EnvConfig), _P, _P, _P, C.c_int32, C.c_int64, _P,
                                 C.c_size_t, _P, C.POINTER(_P)]),
    ("abr_env_destroy", C.c_int, [_P]),
    ("abr_env_set_lane_id_base", C.c_int, [_P, C.c_int64]),
    ("abr_env_set_impl", C.c_int, [_P, C.c_int32]),
    ("abr_env_has_impl", C.c_int, [C.c_int32]),
    ("abr_env_set_lane_speeds", C.c_int, [_P, _P]),
    ("abr_env_set_bitrate_table", C.c_int, [_P, _P]),
    ("abr_env_set_speed_schedule", C.c_int, [_P, _P, C.c_int32]),
    ("abr_env_reset", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("abr_env_step", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("abr_env_step_random", C.c_int, [_P, C.c_int32, C.c_uint64, _P, _P, _P, _P, _P]),
    ("abr_env_step_script", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    ("abr_env_get_effective_impl", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    ("abr_env_notify_restore", C.c_int, [_P]),
    ("abr_env_episode_qoe", C.c_int, [_P, _P, _P]),
    ("abr_env_observe_f64", C.c_int, [_P, _P, _P]),
    ("abr_env_get_state", C.c_int, [_P, C.POINTER(StateView)]),
    ("abr_mpc_select", C.c_int, [C.POINTER(MpcConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 C.c_int64, _P]),
    ("abr_mpc_scratch_bytes", C.c_int, [C.POINTER(MpcConfig), C.c_int64, C.POINTER(C.c_size_t)]),
    ("abr_mpc_select_opt", C.c_int, [C.POINTER(MpcConfig), C.POINTER(MpcOptions), _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, C.c_int64, _P]),
    ("abr_env_step_mpc", C.c_int, [_P, C.POINTER(MpcConfig), _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    ("abr_debug_chain", C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    ("abr_debug_selfcheck", C.c_int, [_P, _P, _P]),
    ("abr_debug_drain", C.c_int, [C.c_double, C.c_double, _P, _P, C.c_int64, _P, _P, _P, C.POINTER(C.c_int32), _P]),
    ("abr_mpc_objective_grid", C.c_int, [C.POINTER(MpcConfig), C.c_int32, C.c_int32, C.c_double,
                                         _P, _P, _P, _P, _P]),
]

_libs = {}


class AbrError(RuntimeError):
    pass


def build(force=False, target="libabr_hip.so"):
    """Compile csrc/abr_env.hip for gfx950 (hipcc cross-compiles without a GPU).  The Makefile's
    dependency list decides whether anything has to be rebuilt."""
    subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []) + [target])
    return os.path.join(CSRC, target)


def lib(name=None):
    """The loaded C-ABI library.  Raises ImportError if it is absent -- by design.  `name`: a
    diagnostic build of the same ABI (tools/diag/csrc/Makefile), by path or file name; default: the product."""
    path = resolve(name) if name else SO_PATH
    L = _libs.get(path)
    if L is None:
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "abrsimulator_amd/csrc`). There is deliberately no CPU fallback.")
        L = C.CDLL(path)
        for sym, res, args in SYMBOLS:
            try:
                fn = getattr(L, sym)
            except AttributeError as e:
                raise ImportError(f"{path} does not export {sym}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        if L.abr_abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {L.abr_abi_version()} != {ABI_VERSION}")
        _libs[path] = L
    return L


def check(rc, L=None):
    if rc != 0:
        raise AbrError(f"abr C-ABI error {rc}: {(L or lib()).abr_last_error().decode()}")


def ptr(t):
    """data_ptr of a tensor (or None) after checking it is a contiguous device tensor."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor handed to the C ABI must be contiguous")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device):
    """The caller's current HIP stream on `device` as a void* (the fast raw accessor when torch has it)."""
    if _raw_stream is not None and device.index is not None:
        return C.c_void_p(_raw_stream(device.index))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
