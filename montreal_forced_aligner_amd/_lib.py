"""ctypes binding of libmfa_hip.so (the C ABI declared in include/mfa_hip.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded this module raises —
nothing in this package imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_SO = Path(os.environ["MFA_HIP_SO"]).resolve() if os.environ.get("MFA_HIP_SO") else _PKG / "libmfa_hip.so"   # (override: A/B builds)
_SOURCES = ["api.hip", "mfcc.hip", "feats.hip", "gmm.hip", "viterbi.hip", "viterbi_general.hip", "fmllr.hip"]
_LIB = None


class MfaHipError(RuntimeError):
    pass


# d_status codes beyond 0 / 1 (include/mfa_hip.h), as the failure reasons the host layers report
STATUS_REASONS = {
    2: "no alignment within the retry beam",
    3: "token capacity exceeded (device decoder status 3)",
    4: "back-pointer capacity exceeded (device decoder status 4)",
    5: "unsupported graph: a state with more than 64 arcs of a kind (device decoder status 5)",
    6: "internal consistency check failed (device decoder status 6)",
    7: "the best path carries more word labels than the utterance has frames: output labels on epsilon arcs (device decoder status 7)",
}


def status_reason(code: int) -> str:
    return STATUS_REASONS.get(int(code), f"device decoder status {int(code)} (include/mfa_hip.h)")


class MfccOpts(C.Structure):
    _fields_ = [
        ("sample_frequency", C.c_float), ("frame_length_ms", C.c_float), ("frame_shift_ms", C.c_float),
        ("preemphasis", C.c_float), ("low_frequency", C.c_float), ("high_frequency", C.c_float),
        ("cepstral_lifter", C.c_float), ("energy_floor", C.c_float),
        ("num_mel_bins", C.c_int32), ("num_coefficients", C.c_int32), ("snip_edges", C.c_int32),
        ("remove_dc_offset", C.c_int32), ("use_energy", C.c_int32), ("raw_energy", C.c_int32),
    ]


class GraphBatch(C.Structure):
    _fields_ = [
        ("n_utt", C.c_int32),
        ("d_state_off", C.c_void_p), ("d_arc_base", C.c_void_p), ("d_start", C.c_void_p), ("d_arc_off", C.c_void_p),
        ("d_final", C.c_void_p), ("d_arc_next", C.c_void_p), ("d_arc_weight", C.c_void_p), ("d_arc_col", C.c_void_p),
        ("d_arc_ilabel", C.c_void_p), ("d_arc_olabel", C.c_void_p), ("d_state_nemit", C.c_void_p),
    ]


class ScorePlan(C.Structure):
    _fields_ = [
        ("d_pdf_list", C.c_void_p), ("d_pdf_off", C.c_void_p), ("d_class_counts", C.c_void_p),
        ("d_pdf_first_frame", C.c_void_p), ("d_pdf_last_depth", C.c_void_p), ("d_state_depth", C.c_void_p),
        ("max_cols", C.c_int32), ("groups", C.c_int32), ("d_group_counts", C.c_void_p),
    ]


class AlignOpts(C.Structure):
    _fields_ = [
        ("beam", C.c_float), ("retry_beam", C.c_float), ("acoustic_scale", C.c_float),
        ("max_tokens", C.c_int32), ("bp_tokens_per_frame", C.c_int32),
    ]


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into libmfa_hip.so next to this file (hipcc cross-compiles without a GPU)."""
    src_dir = _PKG / "csrc"
    srcs = [src_dir / s for s in _SOURCES]
    deps = srcs + [src_dir / "ctx.hpp", _PKG.parent / "include" / "mfa_hip.h"]
    if not force and _SO.exists() and all(_SO.stat().st_mtime >= d.stat().st_mtime for d in deps if d.exists()):
        return _SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
           "-fvisibility=hidden"] + os.environ.get("MFA_HIPCC_FLAGS", "").split() + ["-o", str(_SO)] + [str(s) for s in srcs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


# name → (restype, argtypes); mirrors include/mfa_hip.h one to one
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "mfa_create": (_vp, [C.c_int]),
    "mfa_destroy": (None, [_vp]),
    "mfa_last_error": (C.c_char_p, [_vp]),
    "mfa_version": (C.c_int, []),
    "mfa_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "mfa_synchronize": (C.c_int, [_vp]),
    "mfa_device_alloc": (_vp, [_vp, C.c_size_t]),
    "mfa_device_free": (C.c_int, [_vp, _vp]),
    "mfa_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mfa_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mfa_timer_begin": (C.c_int, [_vp]),
    "mfa_timer_end_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "mfa_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "mfa_kernel_time_ms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mfa_kernel_time_reset": (C.c_int, [_vp]),
    "mfa_mfcc_configure": (C.c_int, [_vp, C.POINTER(MfccOpts)]),
    "mfa_mfcc_num_frames": (_i32, [_vp, _i64]),
    "mfa_mfcc_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mfa_gather_pcm": (C.c_int, [_i32, _vp, _vp, _vp, _i32]),
    "mfa_cmvn_stats": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "mfa_feats_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "mfa_load_gmm": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mfa_gmm_slot": (_i32, [_vp, _i32]),
    "mfa_gmm_sort_pdf_list": (C.c_int, [_vp, _vp, _i32, _vp]),
    "mfa_gmm_sort_pdf_list_keyed": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "mfa_fst_first_frames": (C.c_int, [_i32, _vp, _vp, _i32, _vp]),
    "mfa_debug_gmm_trace": (C.c_int, [_vp, _vp]),
    "mfa_debug_viterbi_stamps": (C.c_int, [_vp, _vp]),
    "mfa_gmm_score_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_align_batch": (C.c_int, [_vp, C.POINTER(GraphBatch), _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, C.POINTER(AlignOpts),
                                  _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_align_general_batch": (C.c_int, [_vp, C.POINTER(GraphBatch), _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(AlignOpts),
                                          _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_align_features_batch": (C.c_int, [_vp, C.POINTER(GraphBatch), C.POINTER(ScorePlan), _vp, _vp, _i32, _i64, _i64, _i32, _i32,
                                           C.POINTER(AlignOpts), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_fst_last_depths": (C.c_int, [_i32, _vp, _vp, _i32, _vp, _vp]),
    "mfa_build_score_plan": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_build_score_plan_grouped": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_build_score_plans_batch": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _vp, _vp]),
    "mfa_fmllr_acc_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "mfa_fmllr_acc_ali_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "mfa_fmllr_stats_model": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "mfa_align_workspace_bytes": (C.c_size_t, [_vp, _i32, _i64, C.POINTER(AlignOpts)]),
}


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not _SO.exists():
            raise MfaHipError(
                f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the alignment path has no CPU fallback)"
            )
        # torch first: it brings its own HIP runtime, and the process must end up with ONE (device buffers and streams
        # are shared with torch); loading libmfa_hip.so first would bind it to a second copy of libamdhip64
        import torch  # noqa: F401

        try:
            handle = C.CDLL(str(_SO))
        except OSError as e:  # e.g. no ROCm runtime on this machine
            raise MfaHipError(f"cannot load {_SO}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mfa_last_error(ctx)
        raise MfaHipError(f"{what}: {msg.decode() if msg else 'error'}")
