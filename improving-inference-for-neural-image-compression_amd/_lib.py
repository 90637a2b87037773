"""ctypes binding of libsga_hip.so (C ABI: include/sga_hip.h).

There is no CPU fallback: if the library is missing or no gfx950 device is visible the
import of the product path fails loudly (`load_library` raises).
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libsga_hip.so")
# the laboratory build (`make EXPERIMENTS=1`): the same library plus every ablation / placement switch of
# DESIGN_EXPERIMENTS.md (SGA_FUSED_GDN, SGA_KEEP_U, SGA_GS3_GEMM, SGA_FORK_AT, ...).  The product library contains none of them.
LAB_LIB_PATH = os.path.join(_PKG_DIR, "libsga_hip_lab.so")
# the run-time knobs the PRODUCT library reads from the environment; tests/test_host.py checks
# the strings of the built library against this list
PRODUCT_ENV_KNOBS = ("SGA_PRECISION", "SGA_NO_GRAPH", "SGA_NO_OVERLAP", "SGA_NO_SPLITK", "SGA_FORK_NAME", "SGA_FORK_VERBOSE",
                     "SGA_PROFILE_BY_LAYER", "SGA_X3_FORK", "SGA_X3_VARIANTS")

SGA_ABI_VERSION = 6

STATUS = {
    0: "SGA_OK", -1: "SGA_ERR_BAD_ARG", -2: "SGA_ERR_BAD_SHAPE", -3: "SGA_ERR_UNSUPPORTED",
    -4: "SGA_ERR_HIP", -5: "SGA_ERR_NO_DEVICE", -6: "SGA_ERR_NOMEM",
}

# sga_layer enum (include/sga_hip.h)
LAYERS = {name: i for i, name in enumerate(
    ["GA0", "GA1", "GA2", "GA3", "GS0", "GS1", "GS2", "GS3", "HA0", "HA1", "HA2", "HS0", "HS1", "HS2"])}


class SgaConfig(C.Structure):
    _fields_ = [("num_filters", C.c_int32), ("max_batch", C.c_int32), ("max_height", C.c_int32),
                ("max_width", C.c_int32), ("bits_back", C.c_int32), ("precision", C.c_int32),
                ("scale_bound", C.c_float), ("reserved", C.c_int32)]


_FP = C.POINTER(C.c_float)


class SgaWeights(C.Structure):
    _fields_ = [
        ("ga_kernel", _FP * 4), ("ga_bias", _FP * 4), ("ga_beta", _FP * 3), ("ga_gamma", _FP * 3),
        ("gs_kernel", _FP * 4), ("gs_bias", _FP * 4), ("gs_beta", _FP * 3), ("gs_gamma", _FP * 3),
        ("ha_kernel", _FP * 3), ("ha_bias", _FP * 3),
        ("hs_kernel", _FP * 3), ("hs_bias", _FP * 3),
        ("eb_matrix", _FP * 4), ("eb_bias", _FP * 4), ("eb_factor", _FP * 3),
    ]


# every symbol include/sga_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_F = C.c_float
_D = C.c_double
_I = C.c_int
_I64 = C.c_int64
SYMBOLS = {
    "sga_abi_version": (_I, []),
    "sga_create": (_I, [C.POINTER(_P), C.POINTER(SgaConfig), C.POINTER(SgaWeights)]),
    "sga_destroy": (_I, [_P]),
    "sga_last_error": (_I, [_P, C.c_char_p, _I]),
    "sga_latent_shape": (_I, [_P, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "sga_encode": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "sga_step_grads": (_I, [_P, _P, _I, _I, _I, _P, _P, _F, _F, _F, C.c_uint64, C.c_uint32,
                            _P, _P, _P, _P, _P, _P, _P]),
    "sga_adam": (_I, [_P, _P, _P, _P, _P, _I64, _I, _D, _D, _D, _D, _P]),
    "sga_run": (_I, [_P, _P, _I, _I, _I, _F, _F, _I, _D, _D, _I, _D, C.c_uint64,
                     _P, _P, _P, _P, _P, _P, _P]),
    "sga_run_begin": (_I, [_P, _P, _I, _I, _I, _F, _F, _I, _D, _D, _I, _D, C.c_uint64, _P, _P, _P]),
    "sga_run_steps": (_I, [_P, _I, _P]),
    "sga_run_state": (_I, [_P, _I, _P, _P, _P, _P]),
    "sga_quantize_centered": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "sga_eval": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "sga_base_compress": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "sga_base_compress_bound": (_I, [_P, _P, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "sga_op_gaussian_likelihood_bound": (_I, [_P, _P, _P, _P, _I64, _F, _P, _P, _P, _P, _P]),
    "sga_op_layer_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "sga_op_layer_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "sga_op_sample": (_I, [_P, _P, _P, _I64, _F, _P, _P, _P]),
    "sga_op_factorized_likelihood": (_I, [_P, _P, _I64, _P, _P, _P]),
    "sga_op_gaussian_likelihood": (_I, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P]),
}



class SgaKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms_total", C.c_double),
                ("flops_total", C.c_double)]


SYMBOLS["sga_bb_init_z"] = (_I, [_P, _P, _I, _I, _I, _P, _P])
SYMBOLS["sga_bb_step_grads"] = (_I, [_P, _P, _I, _I, _I, _P, _P, _F, _F, _F, C.c_uint64, C.c_uint32,
                                     _P, _P, _I, _P, _P, _P, _P, _P])
SYMBOLS["sga_bb_run"] = (_I, [_P, _P, _I, _I, _I, _F, _F, _I, _I, _D, _D, _D, _I, _D, C.c_uint64,
                              _P, _P, _P, _P, _P, _P])
SYMBOLS["sga_bb_refine"] = (_I, [_P, _P, _I, _I, _I, _F, _I, _D, C.c_uint64, _P, _P])
SYMBOLS["sga_bb_eval"] = (_I, [_P, _P, _I, _I, _I, _P, _P, _P, C.c_uint64, _P, _P])
SYMBOLS["sga_op_factorized_density"] = (_I, [_P, _P, _I64, _P, _P, _P])
SYMBOLS["sga_set_relaxation"] = (_I, [_P, _I, _I])
SYMBOLS["sga_set_scale_bound"] = (_I, [_P, _F])
# sga_config.scale_bound (include/sga_hip.h): NONE mirrors sga.py:130-133 (tfc layer never built), BUILT mbt2018.py:77-80
SCALE_BOUND_NONE, SCALE_BOUND_BUILT = 0.0, 0.11
SYMBOLS["sga_op_rate_terms"] = (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P])
SYMBOLS["sga_set_image_ids"] = (_I, [_P, C.POINTER(C.c_int32), _I])
SYMBOLS["sga_set_image_seeds"] = (_I, [_P, C.POINTER(C.c_uint64), _I])
PRECISIONS = {"default": 0, "f32": 1, "bf16x3": 2, "bf16x2": 3}
RELAXATIONS = {"sga": 0, "danneal": 1, "unoise": 2, "ste": 3, "none": 4}
SCHEDULES = {"exp0": 0, "exp": 1}
SYMBOLS["sga_profile_begin"] = (_I, [_P])
SYMBOLS["sga_profile_end"] = (_I, [_P, C.POINTER(SgaKernelStat), _I, C.POINTER(_I)])
SYMBOLS["sga_profile_graph_begin"] = (_I, [_P, C.c_char_p])
SYMBOLS["sga_profile_graph_end"] = (_I, [_P, C.POINTER(SgaKernelStat)])
SYMBOLS["sga_get_fork_point"] = (_I, [_P, C.c_char_p, _I])
SYMBOLS["sga_debug_counter"] = (_I, [_P, _I, C.POINTER(C.c_longlong)])

SYMBOLS["sga_ec_y_symbols"] = (_I, [_P, _P, _P, _I64, _P, _I, _I, _I, _P, _P, _P, _P, _P])
SYMBOLS["sga_ec_z_symbols"] = (_I, [_P, _I64, _I, _P, _P, _P, _P])
SYMBOLS["sga_ec_y_symbols_centred"] = (_I, [_P, _P, _P, _I64, _P, _I, _I, _P, _P, _P, _P])
SYMBOLS["sga_ec_z_symbols_centred"] = (_I, [_P, _P, _I64, _I, _P, _P, _P, _P])
SYMBOLS["sga_ec_encode"] = (_I, [_P, _P, _I64, _I, _P, _P, _P, _I, _P, _I, _P, _P])
SYMBOLS["sga_ec_compact"] = (_I, [_P, _I, _P, _P, _I, _P, _P])
SYMBOLS["sga_ec_decode"] = (_I, [_P, _P, _P, _I, _P, _I64, _I, _P, _P, _P, _I, _P, _P, _P])

_lib = None
_lab = None


def load_lab_library():
    """The laboratory build, loaded beside the product library (its own kernels and statics; the same HIP runtime)."""
    global _lab
    if _lab is None:
        _lab = load_library(LAB_LIB_PATH)
    return _lab


def load_library(path: str | None = None):
    """dlopen libsga_hip.so and type every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SGA_LIB") or LIB_PATH      # SGA_LIB: A/B of two builds (scripts/ab_iter.py)
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the SGA hot path.")
    # PyTorch-ROCm bundles its own libamdhip64; libsga_hip.so links the system one by soname.  The
    # process must end up with ONE HIP runtime (device pointers and streams cross this boundary), so
    # torch's is loaded first and the dynamic linker resolves ours to it.  Loading libsga_hip.so
    # before torch gave two runtimes and `sga_create` -> SGA_ERR_NO_DEVICE on a GPU box.
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("SGA_CALL_LOG"):
        lib = _CallLog(lib, os.environ["SGA_CALL_LOG"])
    ver = lib.sga_abi_version()
    if ver != SGA_ABI_VERSION:
        raise RuntimeError(f"libsga_hip ABI version {ver} != expected {SGA_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


class _CallLog:
    """Debugging aid (SGA_CALL_LOG=<file>): every C-ABI call of this process as one text line -- the entry point, then its
    arguments in order: integers and floats by value, a handle as `h<n>` (n-th successful sga_create of the process, from 0), any other pointer as `P` / `0`
    (non-null / null), the members of sga_config for sga_create -- flushed BEFORE the call is made, so the last line of a
    crashed process is the call it died in.  `tests/c_client/sga_replay.cpp` replays such a file against the library from
    a process without Python or PyTorch (round 6: the hunt for the host SIGSEGV after mid-life hipGraphExecDestroy)."""

    def __init__(self, lib, path):
        self._lib, self._f, self._handles, self._created = lib, open(path, "a"), {}, 0

    def _fmt(self, v):
        if isinstance(v, bool):
            return str(int(v))
        if isinstance(v, int):
            return str(v)
        if isinstance(v, float):
            return repr(v)
        if isinstance(v, bytes):
            return "S:" + v.decode(errors="replace").replace(" ", "_")
        if v is None:
            return "0"
        val = getattr(v, "value", None)
        if isinstance(v, C.c_void_p):
            if val in self._handles:
                return "h%d" % self._handles[val]
            return "P" if val else "0"
        if isinstance(val, (int, float)) and not isinstance(v, (C.c_char_p,)):
            return self._fmt(val)
        return "P"

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("sga_"):
            return fn

        def call(*args):
            if name == "sga_create":
                cfg = args[1]._obj
                line = "sga_create %d %d %d %d %d %d %r" % (cfg.num_filters, cfg.max_batch, cfg.max_height, cfg.max_width,
                                                          cfg.bits_back, cfg.precision, cfg.scale_bound)
            else:
                line = name + " " + " ".join(self._fmt(a) for a in args)
            self._f.write(line + "\n")
            self._f.flush()
            rc = fn(*args)
            if name == "sga_create" and rc == 0:      # h<n> = the n-th handle this process created (never reused)
                self._handles[args[0]._obj.value] = self._created
                self._created += 1
            if name == "sga_destroy":
                self._handles.pop(getattr(args[0], "value", None), None)
            return rc
        return call


class SgaError(RuntimeError):
    pass


def check(lib, handle, status: int, what: str):
    if status == 0:
        return
    msg = STATUS.get(status, str(status))
    detail = ""
    if handle:
        buf = C.create_string_buffer(256)
        hip_err = lib.sga_last_error(handle, buf, 256)
        if hip_err:
            detail = f" (hipError {hip_err}: {buf.value.decode(errors='replace')})"
    raise SgaError(f"{what} failed: {msg}{detail}")
