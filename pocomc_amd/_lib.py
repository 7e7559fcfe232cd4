"""ctypes binding of ``libpocomc_amd.so`` (the C ABI in ``include/pocomc_amd.h``).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible, every entry point raises.  PyTorch is used for device memory and
streams only; all arithmetic happens in the HIP kernels behind this ABI.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMC_LIBRARY") or os.path.join(_HERE, "libpocomc_amd.so")    # (PMC_LIBRARY: A/B builds)

c_p = C.c_void_p


class pmc_maf_t(C.Structure):
    _fields_ = [("packed", c_p), ("meta", c_p),
                ("D", C.c_int32), ("H", C.c_int32), ("T", C.c_int32),
                ("Hp", C.c_int32), ("Dp", C.c_int32),
                ("nT", C.c_int32), ("nXT", C.c_int32), ("nOT", C.c_int32),
                ("pk_per_transform", C.c_int64),
                ("tri_ok", C.c_int32), ("n_out", C.c_int32),
                ("lane16", c_p), ("lane16_fmt", C.c_int32), ("reserved", C.c_int32)]


class pmc_maf_train_t(C.Structure):
    _fields_ = [("packedT", c_p), ("gmap", c_p), ("pkT_per_transform", C.c_int64),
                ("gmap_per_transform", C.c_int64),
                ("jobs", c_p), ("n_jobs", C.c_int32), ("max_sets", C.c_int32), ("n_sq_partial", C.c_int32),
                ("table_waves", C.c_int32), ("tables", c_p),
                ("xt_scratch", c_p), ("act_scratch", c_p), ("delta_scratch", c_p), ("par_scratch", c_p),
                ("par_per_transform", C.c_int64), ("loss_partial", c_p), ("sq_partial", c_p), ("wsum", c_p)]


class pmc_maf_wide_t(C.Structure):
    _fields_ = [("image", c_p), ("image_idx", c_p), ("image_per_transform", C.c_int64),
                ("bias", c_p), ("bias_idx", c_p), ("bias_per_transform", C.c_int64),
                ("scratch", c_p), ("scratch_bytes", C.c_int64), ("wsum", c_p)]


class pmc_adamw_t(C.Structure):
    _fields_ = [("params", c_p), ("grad", c_p), ("exp_avg", c_p), ("exp_avg_sq", c_p), ("n_params", C.c_int64),
                ("pack_idx", c_p), ("packed", c_p), ("n_packed", C.c_int64),
                ("packT_idx", c_p), ("packedT", c_p), ("n_packedT", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("max_norm", C.c_double), ("step", C.c_int64),
                ("scatter_ptr", c_p), ("scatter_dst", c_p), ("snapshot", c_p)]


class pmc_scaler_t(C.Structure):
    _fields_ = [("low", c_p), ("high", c_p), ("mu", c_p), ("sigma", c_p),
                ("kind", c_p), ("bc", c_p), ("log_width", c_p),
                ("D", C.c_int32), ("logit", C.c_int32), ("scale", C.c_int32),
                ("reserved", C.c_int32), ("sum_log_sigma", C.c_double)]


class pmc_done_t(C.Structure):
    _fields_ = [("flag", c_p), ("value", C.c_int64), ("ticket", c_p)]


class pmc_prior_t(C.Structure):
    _fields_ = [("family", c_p), ("loc", c_p), ("scale", c_p), ("D", C.c_int32), ("reserved", C.c_int32)]


class pmc_rng_t(C.Structure):
    _fields_ = [("gamma", c_p), ("normal", c_p), ("uniform", c_p),
                ("seed", C.c_uint64), ("step", C.c_uint64), ("offset", C.c_uint64)]


class pmc_state_t(C.Structure):
    _fields_ = [("theta32", c_p), ("u", c_p), ("x", c_p), ("logdetj", c_p),
                ("logl", c_p), ("logp", c_p), ("logdetj_flow", c_p)]


class pmc_proposal_t(C.Structure):
    _fields_ = [("theta64", c_p), ("u", c_p), ("x", c_p), ("logdetj", c_p),
                ("logl", c_p), ("logp", c_p), ("logdetj_flow", c_p),
                ("quad", c_p), ("quad_prop", c_p)]


class pmc_step_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("preconditioned", C.c_int32), ("n", C.c_int64), ("D", C.c_int32),
                ("inverse_algo", C.c_int32), ("maf", c_p), ("scaler", c_p), ("cur", pmc_state_t),
                ("mu", c_p), ("inv_cov", c_p), ("chol", c_p),
                ("p_theta64", c_p), ("p_theta32", c_p), ("p_u32", c_p), ("p_ldjf", c_p), ("p_u", c_p), ("p_x", c_p),
                ("p_xT", c_p), ("p_logdetj", c_p), ("p_fin", c_p), ("quad", c_p), ("p_quad", c_p), ("p_logl", c_p),
                ("p_logp", c_p), ("alpha", c_p), ("accept", c_p), ("sums", c_p), ("ws", c_p),
                ("h_mu", c_p), ("h_x", c_p), ("h_fin", c_p), ("h_logl", c_p), ("h_logp", c_p), ("h_sums", c_p),
                ("h_accept", c_p), ("ev_inv0", c_p), ("ev_inv1", c_p),
                ("prior", c_p), ("h_logp_out", c_p),
                ("rng_normal", c_p * 2), ("rng_gamma", c_p * 2), ("rng_uniform", c_p * 2), ("rng_ready", c_p),
                ("ev_pre_done", c_p), ("h_done", c_p), ("done_ticket", c_p),
                ("no_fuse", C.c_int32), ("host_direct", C.c_int32),
                ("adapt_state", C.c_void_p), ("adapt_mode", C.c_int32), ("adapt_pad", C.c_int32),
                ("adapt_c_sigma", C.c_double), ("adapt_c_mu", C.c_double), ("adapt_cap", C.c_double),
                ("adapt_n_total", C.c_double), ("adapt_other", C.c_void_p * 7), ("adapt_n_other", C.c_int32),
                ("adapt_pad2", C.c_int32), ("h_clean", c_p), ("clean_count", c_p),
                ("fill_rejected", C.c_int32), ("fill_pad", C.c_int32)]


# name -> (restype, argtypes); every symbol include/pocomc_amd.h declares
i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
P = C.POINTER
SIGNATURES = {
    "pmc_last_error": (C.c_char_p, []),
    "pmc_adapt_update": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "pmc_maf_inverse_auto_is_duo": (C.c_int, [C.POINTER(pmc_maf_t), C.c_int64]),
    "pmc_maf_inverse_auto_is_lane": (C.c_int, [C.POINTER(pmc_maf_t)]),
    "pmc_maf_inverse_auto_is_nsf2": (C.c_int, [C.POINTER(pmc_maf_t)]),
    "pmc_maf_train_waves": (C.c_int, [C.POINTER(pmc_maf_t)]),
    "pmc_abi_version": (C.c_int, []),
    "pmc_build_id": (C.c_char_p, []),
    "pmc_maf_pack": (C.c_int, [c_p, c_p, c_p, i64, c_p]),
    "pmc_maf_forward": (C.c_int, [P(pmc_maf_t), c_p, c_p, c_p, c_p, i64, c_p]),
    "pmc_maf_pack_bf16": (C.c_int, [c_p, c_p, c_p, i64, c_p]),
    "pmc_maf_forward_bf16": (C.c_int, [P(pmc_maf_t), c_p, i64, c_p, c_p, c_p, c_p, i64, c_p, c_p]),
    "pmc_maf_inverse": (C.c_int, [P(pmc_maf_t), c_p, c_p, c_p, i64, C.c_int, c_p]),
    "pmc_maf_lane16_elems": (C.c_int64, [P(pmc_maf_t)]),
    "pmc_maf_pack_lane16": (C.c_int, [P(pmc_maf_t), C.c_int, c_p, c_p]),
    "pmc_maf_loss_grad": (C.c_int, [P(pmc_maf_t), P(pmc_maf_train_t), c_p, c_p, c_p, C.c_float, c_p, c_p, i64, c_p]),
    "pmc_maf_train_epoch": (C.c_int, [P(pmc_maf_t), P(pmc_maf_train_t), P(pmc_adamw_t), c_p, c_p, c_p, i64, i64, c_p,
                                      c_p]),
    "pmc_maf_train_epoch_gated": (C.c_int, [P(pmc_maf_t), P(pmc_maf_train_t), P(pmc_adamw_t), c_p, c_p, c_p, i64, i64, c_p,
                                            c_p, c_p]),
    "pmc_maf_wide_scratch_bytes": (C.c_int64, [P(pmc_maf_t)]),
    "pmc_maf_wide_refresh": (C.c_int, [P(pmc_maf_t), P(pmc_maf_wide_t), c_p, c_p]),
    "pmc_maf_loss_grad_bf16": (C.c_int, [P(pmc_maf_t), P(pmc_maf_wide_t), c_p, c_p, c_p, C.c_float, c_p, c_p, i64, c_p]),
    "pmc_maf_train_epoch_bf16": (C.c_int, [P(pmc_maf_t), P(pmc_maf_wide_t), P(pmc_adamw_t), c_p, c_p, c_p, i64, i64, c_p,
                                           c_p, c_p]),
    "pmc_prefetcher_create": (C.c_void_p, [C.c_int32, c_p]),
    "pmc_prefetcher_submit": (C.c_int, [c_p, c_p, i64, c_p, i64, f64]),
    "pmc_prefetcher_destroy": (None, [c_p]),
    "pmc_maf_valid_epoch": (C.c_int, [P(pmc_maf_t), c_p, c_p, c_p, i64, i64, c_p, c_p, c_p]),
    "pmc_neg_weighted_sum": (C.c_int, [c_p, c_p, c_p, C.c_float, c_p, i64, c_p]),
    "pmc_sum_f32": (C.c_int, [c_p, c_p, i64, c_p]),
    "pmc_adamw_step": (C.c_int, [c_p, c_p, c_p, c_p, i64, f64, f64, f64, f64, f64, f64, i64, c_p, c_p]),
    "pmc_scaler_inverse": (C.c_int, [P(pmc_scaler_t), c_p, c_p, c_p, c_p, c_p, c_p, c_p, i64, c_p]),
    "pmc_scaler_forward": (C.c_int, [P(pmc_scaler_t), c_p, c_p, i64, c_p]),
    "pmc_prior_logpdf": (C.c_int, [P(pmc_prior_t), c_p, c_p, c_p, i64, c_p]),
    "pmc_scaler_inverse_prior": (C.c_int, [P(pmc_scaler_t), P(pmc_prior_t), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                           c_p, c_p, P(pmc_done_t), i64, c_p]),
    "pmc_wait_flag": (C.c_int, [c_p, i64, f64]),
    "pmc_propose": (C.c_int, [C.c_int, c_p, c_p, c_p, c_p, c_p, f64, f64, f64, P(pmc_rng_t),
                              c_p, c_p, c_p, c_p, i64, i32, c_p]),
    "pmc_accept_workspace_bytes": (i64, [i64, i32]),
    "pmc_accept": (C.c_int, [C.c_int, C.c_int, P(pmc_state_t), P(pmc_proposal_t), f64, f64,
                             P(pmc_rng_t), c_p, c_p, c_p, c_p, i64, i32, c_p]),
    "pmc_accept_armed": (C.c_int, [C.c_int, C.c_int, P(pmc_state_t), P(pmc_proposal_t), f64, f64,
                                   P(pmc_rng_t), c_p, c_p, c_p, c_p, P(pmc_done_t), c_p, i64, i32, c_p]),
    "pmc_propose_inverse": (C.c_int, [C.c_int, c_p, c_p, c_p, c_p, f64, f64, f64, P(pmc_rng_t), c_p, c_p, c_p,
                                      P(pmc_maf_t), c_p, c_p, i64, c_p]),
    "pmc_step_pre": (C.c_int, [P(pmc_step_t), P(pmc_rng_t), f64, f64, f64, c_p]),
    "pmc_step_post": (C.c_int, [P(pmc_step_t), P(pmc_rng_t), f64, f64, C.c_int, C.c_int, c_p]),
    "pmc_stream_synchronize": (C.c_int, [c_p]),
    "pmc_pipeline_create": (C.c_void_p, [C.POINTER(C.c_void_p), i32, C.c_uint64, C.POINTER(C.c_uint64), c_p, f64, c_p]),
    "pmc_pipeline_set_comm": (C.c_int, [c_p, c_p]),
    "pmc_comm_create": (C.c_void_p, [i32, i32, i32]),
    "pmc_comm_create_host": (C.c_void_p, [i32, i32, i32]),
    "pmc_comm_kind": (C.c_int, [c_p]),
    "pmc_comm_unlink": (C.c_int, [c_p]),
    "pmc_comm_handle": (C.c_int, [c_p, c_p]),
    "pmc_comm_connect": (C.c_int, [c_p, c_p]),
    "pmc_comm_destroy": (None, [c_p]),
    "pmc_comm_adapt_update": (C.c_int, [c_p, C.POINTER(C.c_void_p), i32, i32, c_p, c_p, c_p, i32, f64, f64, f64, f64, P(pmc_done_t), f64,
                                        c_p]),
    "pmc_pipeline_destroy": (None, [c_p]),
    "pmc_pipeline_start": (C.c_int, [c_p, f64, i64]),
    "pmc_pipeline_next": (C.c_int, [c_p, i32, f64, f64, i32, f64, f64, f64, f64, i32]),
    "pmc_pipeline_stats": (C.c_int, [c_p, C.POINTER(C.c_double), i32]),
    "pmc_event_create": (c_p, []),
    "pmc_event_record": (C.c_int, [c_p, c_p]),
    "pmc_event_elapsed_ms": (C.c_float, [c_p, c_p]),
    "pmc_event_synchronize": (C.c_int, [c_p]),
    "pmc_weight_penalty": (C.c_int, [c_p, c_p, c_p, i64, f64, f64, C.c_float, c_p, c_p, c_p]),
    "pmc_add_noise_f32": (C.c_int, [c_p, i64, i32, C.c_float, C.c_uint64, C.c_uint64, c_p, c_p]),
    "pmc_add_noise_rows_f32": (C.c_int, [c_p, i64, i32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint64, c_p, c_p]),
    "pmc_mean_distance_f32": (C.c_int, [c_p, i64, i32, i64, c_p, c_p]),
    "pmc_affine_rows": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, i32, c_p]),
    "pmc_sum_f64": (C.c_int, [c_p, i64, c_p, c_p]),
    "pmc_weights_from_logw": (C.c_int, [c_p, i64, c_p, c_p, c_p]),
    "pmc_trim_select_workspace_bytes": (C.c_int64, [i64]),
    "pmc_trim_select": (C.c_int, [c_p, i64, c_p, c_p, c_p, c_p, c_p, i64, c_p]),
    "pmc_moments_workspace_bytes": (C.c_int64, [i32]),
    "pmc_moments": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, c_p, c_p, c_p, c_p, i64, c_p]),
    "pmc_column_medians_workspace_bytes": (C.c_int64, [i64, i32, i32]),
    "pmc_column_medians": (C.c_int, [c_p, c_p, c_p, i64, i32, c_p, c_p, c_p, i64, c_p]),
    "pmc_bootstrap_logz": (C.c_int, [c_p, i64, c_p, i64, C.c_uint64, c_p, c_p]),
    "pmc_bootstrap_logz_replay": (C.c_int, [c_p, i64, c_p, i64, c_p, c_p, c_p]),
    "pmc_rng_fill": (C.c_int, [P(pmc_rng_t), f64, c_p, c_p, c_p, i64, i32, c_p]),
    "pmc_event_destroy": (None, [c_p]),
    "pmc_logw": (C.c_int, [c_p, c_p, c_p, f64, c_p, i32, i64, c_p]),
    "pmc_reduce_workspace_bytes": (i64, [i64]),
    "pmc_logw_stats": (C.c_int, [c_p, i64, i64, c_p, c_p, c_p]),
    "pmc_trim_workspace_bytes": (i64, [i64]),
    "pmc_trim_threshold": (C.c_int, [c_p, i64, f64, i32, c_p, c_p, i64, c_p]),
    "pmc_gather": (C.c_int, [c_p, i64, i32] + [c_p] * 10 + [c_p]),
    "pmc_resample_multinomial": (C.c_int, [c_p, i64, c_p, i64, c_p, c_p, c_p]),
    "pmc_resample_systematic": (C.c_int, [c_p, i64, f64, i64, c_p, c_p, c_p]),
}

_lib = None


class PocomcAmdError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for this) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PocomcAmdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pocomc_amd has no CPU fallback.")
    # torch's HIP runtime first: a process that loads this library (its fat binaries register with the runtime at
    # load time) before torch has initialised the device ends up with "no ROCm-capable device" at the first launch
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pmc_abi_version() != 9:
        raise PocomcAmdError("libpocomc_amd.so: ABI version mismatch")
    # timing-only ablation builds (scripts/abl_*.sh: TRI5_ABL / TRI6_ABL / NSF2_ABL != 0) compute WRONG results by design and
    # carry a marker symbol: never loaded by accident
    for marker in ("pmc_ablation_tri5", "pmc_ablation_tri6", "pmc_ablation_nsf2"):
        if hasattr(lib, marker) and not os.environ.get("PMC_ALLOW_ABLATION"):
            raise PocomcAmdError(f"{LIB_PATH} is a timing-only ablation build ({marker}): its results are wrong by design; "
                                 "set PMC_ALLOW_ABLATION=1 to time it")
    built, tree = lib.pmc_build_id().decode(), source_build_id()
    if tree is not None and built.split("+")[0] != tree:
        raise PocomcAmdError(f"{LIB_PATH} was built from other sources than the ones next to it (library id {built}, "
                             f"tree id {tree}): rebuild it with `make -C pocomc_amd/csrc` (or __graft_entry__.build())")
    _lib = lib
    return lib


def source_build_id():
    """The id ``csrc/Makefile`` stamps into the library (``pmc_build_id``): sha256 over Makefile, ``*.hip``, ``*.h`` of
    ``csrc/`` and the public header, sorted by name within each group -- or None where the sources do not travel with the
    package (an installed copy of the library alone)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "pocomc_amd.h")
    if not (os.path.isfile(os.path.join(csrc, "Makefile")) and os.path.isfile(hdr)):
        return None
    files = ([os.path.join(csrc, "Makefile")] + sorted(glob.glob(os.path.join(csrc, "*.hip")))
             + sorted(glob.glob(os.path.join(csrc, "*.h"))) + [hdr])
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_gpu_seen = False


def require_gpu():
    global _gpu_seen
    import torch
    if not _gpu_seen:                      # (is_available() re-counts the devices on every call)
        if not torch.cuda.is_available():
            raise PocomcAmdError("pocomc_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
                                 "there is no CPU fallback")
        _gpu_seen = True
    return torch.device("cuda", torch.cuda.current_device())


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pmc_last_error()
        raise PocomcAmdError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
