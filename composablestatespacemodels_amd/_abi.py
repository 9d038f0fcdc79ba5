"""ctypes view of include/cssm_pf.h and the loader of libcssm_pf.so.

The shared library is the product: hand-written HIP kernels for gfx950 behind a C ABI.  There is
no CPU fallback -- if the library has not been built (``python -c 'import __graft_entry__ as g;
g.build()'``) every entry point of this package raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSSM_PF_LIB") or os.path.join(HERE, "csrc", "libcssm_pf.so")   # (CSSM_PF_LIB: an experiment build, tools/)

# ---- constants (include/cssm_pf.h) ---------------------------------------------------------
CSSM_OK = 0
CSSM_EINVAL_DESC = -1
CSSM_EHIP = -2
CSSM_ESHARD = -3
CSSM_ENOMEM = -4
CSSM_ENONFINITE = -5
CSSM_EINVAL_ARG = -6
CSSM_ESTATE = -7

SDE_BROWNIAN, SDE_GEN_BROWNIAN, SDE_OU, SDE_EULER_AFFINE = 0, 1, 2, 3
F_FIRST, F_SEASONAL = 0, 1
OBS_POISSON, OBS_GAUSSIAN, OBS_LGCP = 0, 1, 2
OBS_NEGBIN, OBS_ZIP, OBS_BERNOULLI, OBS_STUDENT_T, OBS_BETA = 3, 4, 5, 6, 7
MAX_DIM = 16
MAX_LEAVES = 16

_dp = C.POINTER(C.c_double)


class LeafDesc(C.Structure):
    _fields_ = [
        ("sde_kind", C.c_int32), ("dim", C.c_int32), ("f_kind", C.c_int32), ("period", C.c_int32),
        ("harmonics", C.c_int32), ("has_scale", C.c_int32), ("scale", C.c_double),
        ("n_m0", C.c_int32), ("n_c0", C.c_int32), ("n_mu", C.c_int32), ("n_phi", C.c_int32),
        ("n_sigma", C.c_int32), ("reserved", C.c_int32),
        ("m0", _dp), ("c0", _dp), ("mu", _dp), ("phi", _dp), ("sigma", _dp),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_leaves", C.c_int32), ("obs_kind", C.c_int32), ("lgcp_precision", C.c_int32),
        ("obs_df", C.c_int32), ("leaves", C.POINTER(LeafDesc)),
    ]


class PeerHandle(C.Structure):
    """cssm_peer_handle: what a rank hands the others so that they can map its receive windows (peer-written exchange)."""
    _fields_ = [("ipc", C.c_char * 64), ("pid", C.c_uint64), ("local_ptr", C.c_uint64), ("bytes", C.c_uint64),
                ("device", C.c_int32), ("has_ipc", C.c_int32)]


class CssmError(RuntimeError):
    """Raised for a non-zero status of a cssm_* call (the reference throws from stepFilter)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"cssm error {code}: {msg}")
        self.code = code


# Every symbol include/cssm_pf.h declares: (name, restype, argtypes)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_descp = C.POINTER(ModelDesc)
_h = C.c_void_p

SYMBOLS = [
    ("cssm_pf_create", C.c_int, [_descp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(_h)]),
    ("cssm_pf_create_shard", C.c_int, [_descp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(_h)]),
    ("cssm_pf_destroy", None, [_h]),
    ("cssm_pf_set_params", C.c_int, [_h, _descp]),
    ("cssm_pf_reseed", C.c_int, [_h, C.c_uint64]),
    ("cssm_pf_run_key", C.c_uint64, [C.c_uint64, C.c_uint64]),
    ("cssm_pf_get_weights", C.c_int, [_h, _dp, _dp]),
    ("cssm_pf_init", C.c_int, [_h, C.c_double]),
    ("cssm_pf_init_from", C.c_int, [_h, C.c_double, _dp]),
    ("cssm_pf_step", C.c_int, [_h, C.c_double, C.c_double, C.c_int, _dp, _i32p]),
    ("cssm_pf_propagate", C.c_int, [_h, C.c_double, C.c_double, C.c_int]),
    ("cssm_pf_adopt", C.c_int, [_h, _dp, C.c_double, C.c_int32]),
    ("cssm_pf_ll_filter", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t, _dp, _dp, _i32p]),
    ("cssm_pf_ll_filter_more", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t, _dp, _dp, _i32p]),
    ("cssm_pf_filter", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t, _dp, _dp, _i32p, _dp]),
    ("cssm_pf_last_loop_ms", C.c_int, [_h, C.POINTER(C.c_float)]),
    ("cssm_pf_last_device_us", C.c_int, [_h, _dp]),
    ("cssm_pf_stream_idle", C.c_int, [_h]),
    ("cssm_pf_set_option", C.c_int, [_h, C.c_int, C.c_int]),
    ("cssm_rtc_info", C.c_int, [_u64p]),
    ("cssm_pf_profile", C.c_int, [_h, C.c_int]),
    ("cssm_pf_profile_read", C.c_int, [_h, _dp, _u64p]),
    ("cssm_pf_num_particles", C.c_uint64, [_h]),
    ("cssm_pf_dim", C.c_int32, [_h]),
    ("cssm_pf_get_particles", C.c_int, [_h, _dp]),
    ("cssm_pf_get_ancestors", C.c_int, [_h, _u32p]),
    ("cssm_pf_get_logw", C.c_int, [_h, _dp]),
    ("cssm_pf_get_proposed", C.c_int, [_h, _dp]),
    ("cssm_pf_summary", C.c_int, [_h, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp]),
    ("cssm_pf_interpolate", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    ("cssm_resample_systematic", C.c_int, [_dp, C.c_size_t, C.c_double, _u32p, C.c_int]),
    ("cssm_resample", C.c_int, [C.c_int, _dp, C.c_size_t, C.c_double, C.c_uint64, C.c_uint32, _u32p, C.c_int]),
    ("cssm_resample_residual", C.c_int, [_dp, C.c_size_t, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.c_int]),
    ("cssm_pf_shard_init", C.c_int, [_h, C.c_double]),
    ("cssm_pf_shard_propagate", C.c_int, [_h, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    ("cssm_pf_shard_sums", C.c_int, [_h, C.c_void_p, C.c_int, C.c_void_p]),
    ("cssm_pf_shard_offspring", C.c_int, [_h, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cssm_pf_shard_begin", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t]),
    ("cssm_pf_shard_continue", C.c_int, [_h, _dp, _dp, _u8p, C.c_size_t]),
    ("cssm_pf_shard_propagate_at", C.c_int, [_h, C.c_size_t, C.c_void_p]),
    ("cssm_pf_shard_wait_stats", C.c_int, [_h, _u64p]),
    ("cssm_pf_shard_status", C.c_int, [_h, _dp, _i32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]),
    ("cssm_pf_shard_want_path", C.c_int, [_h, C.c_int]),
    ("cssm_pf_shard_get_path", C.c_int, [_h, _dp, C.c_size_t]),
    ("cssm_model_structure", C.c_int, [_descp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    ("cssm_pf_shard_summary_begin", C.c_int, [_h, C.c_double, C.c_void_p, C.c_void_p]),
    ("cssm_pf_shard_summary_hist", C.c_int, [_h, C.c_int, C.c_void_p]),
    ("cssm_pf_shard_summary_pick", C.c_int, [_h, C.c_int, C.c_void_p]),
    ("cssm_pf_shard_summary_finish", C.c_int, [_h, C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]),
    ("cssm_pf_shard_pack", C.c_int, [_h, C.c_int, _i64p, _i64p, C.c_int, C.c_void_p]),
    ("cssm_pf_shard_adopt", C.c_int, [_h, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    ("cssm_pf_shard_result", C.c_int, [_h, _dp, _i32p]),
    ("cssm_rccl_available", C.c_int, []),
    ("cssm_rccl_library", C.c_char_p, []),
    ("cssm_rccl_unique_id", C.c_int, [C.c_void_p]),
    ("cssm_rccl_comm_create", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("cssm_rccl_comm_destroy", None, [C.c_void_p]),
    ("cssm_pf_shard_series_rccl", C.c_int, [_h, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, _u8p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("cssm_pf_shard_spec_segment", C.c_int64, [_h, C.c_int64]),
    ("cssm_pf_shard_unit", C.c_int64, [_h]),
    ("cssm_pf_shard_boundary_pack", C.c_int, [_h, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    ("cssm_pf_shard_adopt_spec", C.c_int, [_h, C.c_void_p, C.c_int, C.c_int, C.c_int64]),
    ("cssm_pf_shard_resume", C.c_int, [_h, C.POINTER(C.c_uint32)]),
    ("cssm_pf_shard_resume_level", C.c_int, [_h, C.POINTER(C.c_uint32)]),
    ("cssm_pf_shard_peer_setup", C.c_int, [_h, C.c_int, C.c_int, C.c_int64, C.POINTER(PeerHandle)]),
    ("cssm_pf_shard_peer_connect", C.c_int, [_h, C.POINTER(PeerHandle), C.c_int]),
    ("cssm_pf_shard_peer_handshake", C.c_int, [_h, C.c_uint32]),
    ("cssm_pf_shard_peer_probe_stale", C.c_uint32, [_h]),
    ("cssm_pf_shard_peer_close", None, [_h]),
    ("cssm_pf_shard_pack_peer", C.c_int, [_h, C.c_int, C.c_int, C.c_int64]),
    ("cssm_pf_shard_pack_rows_peer", C.c_int, [_h, C.c_int, C.c_int, C.c_int64]),
    ("cssm_pf_shard_peer_rows", C.c_int, [_h, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("cssm_pf_shard_peer_debug", C.c_int64, [_h, C.POINTER(C.c_uint32), C.c_size_t]),
    ("cssm_pf_shard_adopt_peer", C.c_int, [_h, C.c_int, C.c_int, C.c_int64]),
    ("cssm_pf_shard_exchange_peer", C.c_int, [_h, C.c_int, C.c_int, C.c_int64]),
    ("cssm_pf_shard_series_peer", C.c_int, [_h, C.c_int, C.c_int, C.c_size_t, C.c_size_t, _u8p, C.c_int64]),
    ("cssm_pmmh_run", C.c_int, [_h, _descp, _dp, C.c_size_t, C.c_double, _dp, _dp, _u8p, C.c_size_t,
                                C.c_uint64, C.c_size_t, _dp, _dp, _i32p, _dp]),
    ("cssm_pfb_create", C.c_int, [_descp, C.c_uint64, C.c_int, C.c_int, C.POINTER(_h)]),
    ("cssm_pfb_destroy", None, [_h]),
    ("cssm_pfb_num_chains", C.c_int, [_h]),
    ("cssm_pfb_chain", _h, [_h, C.c_int]),
    ("cssm_pfb_filter", C.c_int, [_h, C.POINTER(_descp), _u64p, _dp, _dp, _u8p, C.c_size_t, _dp, _dp, C.POINTER(C.c_int)]),
    ("cssm_pmmh_run_batched", C.c_int, [_h, _descp, _dp, C.c_size_t, C.c_double, _dp, _dp, _u8p, C.c_size_t, _u64p, C.c_size_t, _dp, _dp, _i32p, _dp]),
    ("cssm_pmmh_run_speculative", C.c_int, [_h, _descp, _dp, C.c_size_t, C.c_double, _dp, _dp, _u8p, C.c_size_t, C.c_uint64, C.c_size_t, _dp, _dp, _i32p, _dp]),
    ("cssm_diag_copy_ceiling", C.c_int, [C.c_int, C.c_size_t, C.c_int, _dp]),
    ("cssm_contract_eval", C.c_int, [C.c_int, C.c_int, _dp, C.c_size_t, _dp, C.c_size_t]),
    ("cssm_desc_flatten", C.c_int, [_descp, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("cssm_last_error", C.c_char_p, []),
    ("cssm_version", C.c_char_p, []),
]

_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libcssm_pf.so and bind every declared symbol.  Fails loudly if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ "
            "as g; g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def last_error() -> str:
    msg = load_library().cssm_last_error()
    return msg.decode() if msg else ""


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().cssm_last_error()
        raise CssmError(rc, msg.decode() if msg else "")
