"""ctypes binding of libirsde_hip.so (C ABI: include/irsde_hip.h).

The HIP library IS the product: there is no PyTorch / CPU fallback.  If the shared library is
missing or fails to load, every entry point raises `IrsdeLibraryError`.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IRSDE_LIB_PATH: load another build of the SAME library (A/B measurements of two builds inside one GPU call, tools/ab_bench.sh);
# there is still no fallback — a missing file fails exactly like a missing default library.
LIB_PATH = os.environ.get("IRSDE_LIB_PATH") or os.path.join(_HERE, "libirsde_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MODE = {"sde": 0, "ode": 1, "posterior": 2, "dsde_sde": 3, "dsde_ode": 4}
COEF_STRIDE = 12
FLAG_KEEP_ACTIVATIONS = 1
FLAG_NAIVE_CONV = 2
FLAG_NO_WINOGRAD = 4
FLAG_NO_WINOGRAD_F43 = 8
FLAG_UNCOND_FULLATTN = 16
FLAG_BF16 = 32
FLAG_NAF_INTRO_SKIP = 64
FLAG_BF16_ACT = 128
FLAG_NO_FUSED_LN = 256
FLAG_NAF_LENS = 512
FLAG_FP16 = 1024
FLAG_NO_WINOGRAD_FUSED = 2048
FLAG_NO_FUSED_ATTN = 4096
FLAG_NO_NAF_CHAIN = 8192
FLAG_SPLIT_BF16X2 = 16384
FLAG_SPLIT_F16X2 = 32768
SAMPLE_GRAPH = 1
SAMPLE_PROFILE = 2

# every symbol include/irsde_hip.h declares (checked by tests/test_cabi.py)
SYMBOLS = [
    "irsde_last_error", "irsde_version", "irsde_create", "irsde_create_nafnet", "irsde_destroy", "irsde_num_weights",
    "irsde_weight_name", "irsde_weight_shape", "irsde_load_weight", "irsde_finalize_weights",
    "irsde_set_schedule", "irsde_unet_forward", "irsde_sample", "irsde_sde_step", "irsde_philox_normal",
    "irsde_get_profile", "irsde_debug_tap", "irsde_work_model", "irsde_debug_conv", "irsde_plan_describe", "irsde_bench_conv", "irsde_op_profile", "irsde_debug_split_gemm", "irsde_bench_naf_chain", "irsde_debug_force_subbatches", "irsde_debug_force_chain_groups",
    "irsde_eval_metrics", "irsde_tensor2img",
    "irsde_set_lens_info", "irsde_create_latent_unet", "irsde_latent_shapes", "irsde_latent_encode", "irsde_latent_decode", "irsde_latent_hidden",
]


class IrsdeLibraryError(RuntimeError):
    pass


class IrsdeError(RuntimeError):
    pass


class Config(ctypes.Structure):
    _fields_ = [("in_nc", ctypes.c_int32), ("out_nc", ctypes.c_int32), ("nf", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("device", ctypes.c_int32), ("flags", ctypes.c_int32)]


class NafConfig(ctypes.Structure):
    _fields_ = [("img_channel", ctypes.c_int32), ("width", ctypes.c_int32), ("middle_blk_num", ctypes.c_int32),
                ("n_enc", ctypes.c_int32), ("enc_blk_nums", ctypes.c_int32 * 8), ("n_dec", ctypes.c_int32),
                ("dec_blk_nums", ctypes.c_int32 * 8), ("device", ctypes.c_int32), ("flags", ctypes.c_int32)]


class LatentConfig(ctypes.Structure):
    _fields_ = [("in_ch", ctypes.c_int32), ("out_ch", ctypes.c_int32), ("ch", ctypes.c_int32), ("n_mult", ctypes.c_int32),
                ("ch_mult", ctypes.c_int32 * 8), ("embed_dim", ctypes.c_int32), ("device", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


_lib = None
_lock = threading.Lock()


PROBES_LIB_PATH = os.path.join(_HERE, "libirsde_hip_probes.so")


def build_library(force=False, verbose=False, probes=True):
    """Compile csrc/*.hip for gfx950 into libirsde_hip.so (hipcc cross-compiles without a GPU) and, with probes=True, the
    measurement build libirsde_hip_probes.so (`make PROBES=1`: the product library + superseded kernel generations, cycle-stamp /
    ablation / tuning twins — what tools/ and the kernel-generation A/B tests load; the product never does)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    for extra in ([], ["PROBES=1"]) if probes else ([],):
        r = subprocess.run(["make", "-j8", "-C", CSRC] + extra, capture_output=True, text=True)
        if r.returncode != 0:
            raise IrsdeLibraryError("building libirsde_hip%s.so failed:\n" % ("_probes" if extra else "") + r.stdout + r.stderr)
        if verbose:
            print(r.stdout)
    return LIB_PATH


def _declare(lib):
    c = ctypes
    P = c.c_void_p
    lib.irsde_last_error.restype = c.c_char_p
    lib.irsde_last_error.argtypes = []
    lib.irsde_version.restype = c.c_int
    lib.irsde_create.argtypes = [c.POINTER(Config), c.POINTER(P)]
    lib.irsde_create_nafnet.argtypes = [c.POINTER(NafConfig), c.POINTER(P)]
    lib.irsde_destroy.argtypes = [P]
    lib.irsde_destroy.restype = None
    lib.irsde_num_weights.argtypes = [P]
    lib.irsde_weight_name.argtypes = [P, c.c_int]
    lib.irsde_weight_name.restype = c.c_char_p
    lib.irsde_weight_shape.argtypes = [P, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_int)]
    lib.irsde_load_weight.argtypes = [P, c.c_char_p, P, c.POINTER(c.c_int64), c.c_int]
    lib.irsde_finalize_weights.argtypes = [P]
    lib.irsde_set_schedule.argtypes = [P, c.c_int, P]
    lib.irsde_unet_forward.argtypes = [P, P, P, c.POINTER(c.c_int64), c.c_int, c.c_int, c.c_int, c.c_int, P, P]
    lib.irsde_sample.argtypes = [P, c.c_int, P, P, P, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int,
                                 c.c_int, c.c_int, P, P, c.c_int]  # ..., B, H, W, T, t_stop, out, stream, flags
    lib.irsde_sde_step.argtypes = [c.c_int, c.c_int, P, P, P, P, P, c.c_uint64, c.c_uint64, c.c_int, c.c_int,
                                   c.c_int, c.c_int, P]
    lib.irsde_philox_normal.argtypes = [P, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_uint64, P]
    lib.irsde_get_profile.argtypes = [P, c.POINTER(c.c_double)]
    lib.irsde_debug_tap.argtypes = [P, c.c_char_p, P, c.POINTER(c.c_int64)]
    lib.irsde_work_model.argtypes = [P, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double)]
    lib.irsde_debug_conv.argtypes = [P, c.c_int, P, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, P, c.c_int, c.c_int,
                                     c.c_int, c.c_int, c.c_int, P, P, c.c_int, c.c_int, P, P, c.c_int, c.c_int, P]
    lib.irsde_plan_describe.argtypes = [P, c.c_int, c.c_int, c.c_int, c.c_char_p, c.c_int]
    lib.irsde_op_profile.argtypes = [P, c.c_char_p, c.c_int]
    lib.irsde_bench_conv.argtypes = [c.c_int] * 11 + [c.POINTER(c.c_double)]
    lib.irsde_bench_naf_chain.argtypes = [c.c_int] * 4 + [c.POINTER(c.c_double)]
    lib.irsde_debug_force_subbatches.argtypes = [c.c_int]
    lib.irsde_debug_force_chain_groups.argtypes = [c.c_int]
    lib.irsde_debug_split_gemm.argtypes = [P, P, P, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, P]
    lib.irsde_eval_metrics.argtypes = [P, P, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double), P]
    lib.irsde_tensor2img.argtypes = [P, P, c.c_int, c.c_int, c.c_int, c.c_int, P]
    lib.irsde_set_lens_info.argtypes = [P, c.POINTER(c.c_float), c.c_int]
    lib.irsde_create_latent_unet.argtypes = [c.POINTER(LatentConfig), c.POINTER(P)]
    lib.irsde_latent_shapes.argtypes = [P, c.c_int, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_int64), c.POINTER(c.c_int)]
    lib.irsde_latent_encode.argtypes = [P, P, c.c_int, c.c_int, c.c_int, P, c.POINTER(P), P]
    lib.irsde_latent_decode.argtypes = [P, P, c.POINTER(P), c.c_int, c.c_int, c.c_int, P, P]
    lib.irsde_latent_hidden.argtypes = [P, c.c_int, c.c_int, c.c_int, c.c_int, P, P]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError here = the .so does not export what the header declares
    return lib


def lib():
    """The loaded library; raises IrsdeLibraryError (never falls back) when it is unavailable."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise IrsdeLibraryError(
                    "libirsde_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` or `make -C %s`. There is no CPU/PyTorch fallback for this path." % (LIB_PATH, CSRC))
            try:
                _lib = _declare(ctypes.CDLL(LIB_PATH))
            except OSError as ex:
                raise IrsdeLibraryError("failed to load %s: %s" % (LIB_PATH, ex)) from ex
        return _lib


_probes = None


def probes_lib():
    """The measurement build (`make PROBES=1`), for tools/ and the tests that compare kernel generations; raises IrsdeLibraryError when it has not
    been built.  A second copy of the library in the process: its engines, error string and tuning state are its own."""
    global _probes
    with _lock:
        if _probes is None:
            if not os.path.exists(PROBES_LIB_PATH):
                raise IrsdeLibraryError("libirsde_hip_probes.so not found at %s — build it with `make -C %s PROBES=1`" % (PROBES_LIB_PATH, CSRC))
            try:
                _probes = _declare(ctypes.CDLL(PROBES_LIB_PATH))
            except OSError as ex:
                raise IrsdeLibraryError("failed to load %s: %s" % (PROBES_LIB_PATH, ex)) from ex
        return _probes


def check(rc, L=None):
    if rc != 0:
        msg = (L or lib()).irsde_last_error()
        raise IrsdeError("libirsde_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def stream_ptr(torch_stream=None):
    """hipStream_t of torch's current stream (so calls are ordered with the caller's torch work)."""
    import torch
    s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
