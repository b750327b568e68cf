"""ctypes binding of libctsi.so (the C-ABI declared in include/ctsi.h).

The library is built in-tree (``video-to-video-diffusion_amd/libctsi.so``) by
``__graft_entry__.build()`` / ``make -C video-to-video-diffusion_amd/csrc``.  There is no
fallback: if the shared object is missing or a call fails, a ``CtsiError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "libctsi.so"
CSRC_DIR = _PKG_DIR / "csrc"
HEADER_PATH = _PKG_DIR.parent / "include" / "ctsi.h"


class CtsiError(RuntimeError):
    """Raised when libctsi is unavailable or one of its entry points reports an error."""


class ConvDesc(C.Structure):
    _fields_ = [
        ("transposed", C.c_int),
        ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
        ("sh", C.c_int), ("sw", C.c_int),
        ("pd", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
        ("n", C.c_int), ("c1", C.c_int), ("c2", C.c_int),
        ("cout", C.c_int),
        ("di", C.c_int), ("hi", C.c_int), ("wi", C.c_int),
        ("halo_d", C.c_int),
    ]


class ConvOut(C.Structure):
    _fields_ = [
        ("y", C.c_void_p),
        ("mode", C.c_int),
        ("cout_stride", C.c_int),
        ("c_off", C.c_int),
        ("sn", C.c_longlong), ("sc", C.c_longlong), ("sd", C.c_longlong),
        ("sh", C.c_longlong), ("sw", C.c_longlong),
        ("act", C.c_int),
        ("colsum", C.c_void_p),
        ("gn_x", C.c_void_p), ("gn_sums", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),
        ("gn_groups", C.c_int), ("gn_eps", C.c_float), ("gn_count", C.c_longlong), ("gn_silu", C.c_int),
        ("workspace", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
        ("sh", C.c_int), ("sw", C.c_int),
        ("pd", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
        ("n", C.c_int),
        ("dr", C.c_int), ("hr", C.c_int), ("wr", C.c_int),
        ("dg", C.c_int), ("hg", C.c_int), ("wg", C.c_int),
        ("cr", C.c_int), ("cr_stride", C.c_int),
        ("cg", C.c_int), ("cg_stride", C.c_int),
    ]


_vp, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes, returns_status)
SIGNATURES = {
    "ctsi_version": (_i, [], False),
    "ctsi_ablation_build": (_i, [], False),
    "ctsi_last_error": (C.c_char_p, [], False),
    "ctsi_device_available": (_i, [], False),
    "ctsi_ncdhw_f32_to_ndhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_ndhwc_bf16_to_ncdhw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_ncdhw_f32_to_ndhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_ndhwc_f32_to_ncdhw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_conv_plan_create": (_i, [C.POINTER(_vp), C.POINTER(ConvDesc)], True),
    "ctsi_conv_plan_set_weight_cin": (_i, [_vp, _i], True),
    "ctsi_conv_plan_set_stream_tail": (_i, [_vp, _i], True),
    "ctsi_conv_plan_destroy": (None, [_vp], False),
    "ctsi_conv_plan_out_dims": (_i, [_vp, _ip, _ip, _ip], True),
    "ctsi_conv_plan_weight_bytes": (_sz, [_vp], False),
    "ctsi_conv_plan_workspace_bytes": (_sz, [_vp], False),
    "ctsi_conv_plan_tiles": (_i, [_vp], False),
    "ctsi_conv_plan_tiles_per_sample": (_i, [_vp], False),
    "ctsi_conv_plan_cout_pad": (_i, [_vp], False),
    "ctsi_conv_plan_flops": (C.c_double, [_vp], False),
    "ctsi_conv_plan_config": (_i, [_vp, _ip, _ip, _ip], True),
    "ctsi_conv_plan_pack_weights": (_i, [_vp, _vp, _vp, _vp], True),
    "ctsi_conv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(ConvOut), _vp], True),
    "ctsi_gn_colsum": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _ip, _vp], True),
    "ctsi_gn_colsum_tiles": (_i, [_i, _i, _i], False),
    "ctsi_gn_finalize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_gn_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp,
                           _i, _vp], True),
    "ctsi_attn_depthsum": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_attn_depthsum_tiles": (_i, [_i, _i, _i], False),
    "ctsi_attn_normsum": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp], True),
    "ctsi_attn_pv_supported": (_i, [_i, _i], False),
    "ctsi_attn_pv": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp], True),
    "ctsi_attn_broadcast_add": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_attn_softmax_rowsum": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_time_embed_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp], True),
    "ctsi_trilinear_depth_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp], True),
    "ctsi_ddim_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], True),
    "ctsi_ddpm_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_ddpm_posterior": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _vp], True),
    "ctsi_step_advance": (_i, [_vp, _vp], True),
    "ctsi_nan_to_num_f32": (_i, [_vp, _ll, _vp], True),
    "ctsi_count_nonfinite_f32": (_i, [_vp, _ll, _i, _vp, _vp], True),
    "ctsi_wgrad_workspace_bytes": (_sz, [C.POINTER(WgradDesc)], False),
    "ctsi_wgrad_flops": (C.c_double, [C.POINTER(WgradDesc)], False),
    "ctsi_wgrad": (_i, [C.POINTER(WgradDesc), _vp, _vp, _vp, _sz, _vp, _ll, _ll, _ll, _f, _vp], True),
    "ctsi_weight_dgrad_layout": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_linear_wgrad_multi": (_i, [_vp, _vp, _i, _vp], True),
    "ctsi_gn_bwd_tiles": (_i, [_i, _i, _i], False),
    "ctsi_gn_bwd_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i], False),
    "ctsi_gn_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp, _vp, _vp,
                         _vp, _vp, _vp, _ll, _vp, _vp], True),
    "ctsi_channel_sum_workspace_floats": (_sz, [_ll, _i], False),
    "ctsi_channel_sum": (_i, [_vp, _ll, _i, _i, _vp, _vp, _f, _vp], True),
    "ctsi_add_bf16": (_i, [_vp, _vp, _ll, _vp], True),
    "ctsi_f32_to_bf16": (_i, [_vp, _vp, _ll, _vp], True),
    "ctsi_q_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_mse_loss_workspace_doubles": (_sz, [_i], False),
    "ctsi_mse_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], True),
    "ctsi_mse_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp], True),
    "ctsi_time_embed_train_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp], True),
    "ctsi_linear_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp], True),
    "ctsi_blend_accumulate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], True),
    "ctsi_blend_normalize": (_i, [_vp, _vp, _ll, _vp], True),
    "ctsi_slice_metrics_workspace_doubles": (_sz, [_i, _i, _i, _i, _i], False),
    "ctsi_slice_metrics": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], True),
    "ctsi_comm_unique_id": (_i, [_vp], True),
    "ctsi_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i], True),
    "ctsi_comm_destroy": (None, [_vp], False),
    "ctsi_comm_rank": (_i, [_vp], False),
    "ctsi_comm_world": (_i, [_vp], False),
    "ctsi_halo_exchange": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp], True),
    "ctsi_halo_exchange_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _ll, _vp], True),
    "ctsi_gn_allreduce": (_i, [_vp, _vp, _i, _vp, _ll, _vp], True),
    "ctsi_comm_allgather": (_i, [_vp, _vp, _vp, _sz, _vp], True),
    "ctsi_memset_async": (_i, [_vp, _i, _sz, _vp], True),
    "ctsi_adamw_chunk_elems": (_i, [], False),
    "ctsi_adamw_multi": (_i, [_vp, _vp, _vp, _i, _vp], True),
    "ctsi_copy_scale_multi": (_i, [_vp, _vp, _i, _vp], True),
    "ctsi_device_error_status": (_i, [C.POINTER(C.c_uint), C.POINTER(C.c_uint), _i], True),
    "ctsi_graph_begin_capture": (_i, [_vp], True),
    "ctsi_graph_end_capture": (_i, [_vp, C.POINTER(_vp)], True),
    "ctsi_graph_launch": (_i, [_vp, _vp], True),
    "ctsi_graph_destroy": (None, [_vp], False),
    "ctsi_event_create": (_i, [C.POINTER(_vp)], True),
    "ctsi_event_record": (_i, [_vp, _vp], True),
    "ctsi_stream_wait_event": (_i, [_vp, _vp], True),
    "ctsi_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)], True),
    "ctsi_event_destroy": (None, [_vp], False),
}


def build(force: bool = False) -> Path:
    """Compile every HIP source for gfx950 into libctsi.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", str(CSRC_DIR), "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", str(CSRC_DIR), "-j4"], capture_output=True, text=True)
    if proc.returncode != 0 or not LIB_PATH.exists():
        raise CtsiError("building libctsi.so failed:\n" + proc.stdout[-4000:] + proc.stderr[-4000:])
    return LIB_PATH


class _Lib:
    def __init__(self, path: Path):
        if not path.exists():
            raise CtsiError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {CSRC_DIR}`.  There is no CPU fallback for the HIP engine.")
        try:
            self._dll = C.CDLL(str(path))
        except OSError as exc:  # missing ROCm runtime etc.
            raise CtsiError(f"cannot load {path}: {exc}") from exc
        self.raw = {}
        for name, (restype, argtypes, status) in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as exc:
                raise CtsiError(f"{path} does not export {name}") from exc
            fn.restype = restype
            fn.argtypes = argtypes
            self.raw[name] = fn
            setattr(self, name[len("ctsi_"):], self._checked(name, fn) if status else fn)

    def _checked(self, name, fn):
        last_error = self._dll.ctsi_last_error
        last_error.restype = C.c_char_p

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                msg = last_error()
                raise CtsiError(f"{name} failed (rc={rc}): {msg.decode() if msg else '?'}")
            return rc

        call.__name__ = name
        return call


_LIB = None


def get_lib() -> _Lib:
    """Load libctsi.so once; raises CtsiError when it is missing (never falls back)."""
    global _LIB
    if _LIB is None:
        _LIB = _Lib(Path(os.environ.get("CTSI_LIB", LIB_PATH)))
    return _LIB
