"""ctypes binding of include/higsfa.h (the drop-in boundary).  No fallback: if the native
library is missing this module raises at first use."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HG_U8, HG_F32, HG_F64 = 0, 1, 2
HG_OK = 0
HG_ERR_ARG, HG_ERR_FORMAT, HG_ERR_DIM, HG_ERR_DEVICE, HG_ERR_NOMEM, HG_ERR_STATE = -1, -2, -3, -4, -5, -6
HG_PLAN_GENERIC, HG_PLAN_FUSED = 0, 1

# HIGSFA_LIB: another build of the same library (same-box A/B of two commits, tools/build_ref_lib.sh) — never a different backend
_LIB_PATH = os.environ.get("HIGSFA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhigsfa.so")
_lib = None


class HgInfo(C.Structure):
    _fields_ = [("input_dim", C.c_int64), ("output_dim", C.c_int64), ("n_top_nodes", C.c_int32),
                ("plan_kind", C.c_int32), ("n_stages", C.c_int32), ("device", C.c_int32),
                ("weight_bytes", C.c_int64), ("flops_per_row", C.c_int64),
                ("padded_flops_per_row", C.c_int64), ("workspace_bytes", C.c_int64)]


class HgCascadeConsts(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "regression_width", "regression_height", "desired_sampling", "tolerance_posxy_deviation", "tolerance_scale_deviation",
        "tolerance_angle_deviation", "max_scale_radio", "min_scale_radio", "net_Dang", "cut_off_face")]


class HgCascadeLevel(C.Structure):      # hg_cascade_level: one pyramid level of the first-stage grid (face_analysis.py:630-652)
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_double) for n in ("x_stop", "y_stop", "patch_w", "patch_h", "max_dx", "max_dy", "base_side")]


class HgCascadeStage(C.Structure):
    _fields_ = [("type", C.c_int32), ("serial", C.c_int32), ("flow", C.c_void_p), ("classifier", C.c_void_p)]


HG_STAGE = {"Disc": 0, "PosX": 1, "PosY": 2, "PAng": 3, "Scale": 4}


class NativeLibraryMissing(ImportError):
    pass


def lib():
    """Load libhigsfa.so (built by ``python -m pyfaceanalysis_amd.build``).

    Load order with PyTorch: the torch wheel bundles its own ROCm runtime (torch/lib/libamdhip64.so, librocblas.so,
    librocsolver.so); loaded first, it is the one runtime everything binds to.  The other way round the system
    runtime owns the GPU and a later ``import torch`` reports "No HIP GPUs are available".  So when torch is installed
    it is imported here, before the library (HIGSFA_NO_TORCH=1 skips this for torch-free processes)."""
    global _lib
    if _lib is not None:
        return _lib
    import sys
    if "torch" not in sys.modules and not os.environ.get("HIGSFA_NO_TORCH"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise NativeLibraryMissing(
            "%s not found: the HIP library is the only execution path of this package; build it "
            "with `python -m pyfaceanalysis_amd.build` (or __graft_entry__.build())" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
    sigs = {
        "hg_version": (C.c_int, []),
        "hg_last_error": (C.c_char_p, []),
        "hg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "hg_flow_load": (C.c_int, [vp, sz, i32, C.POINTER(vp)]),
        "hg_flow_free": (None, [vp]),
        "hg_flow_info": (C.c_int, [vp, C.POINTER(HgInfo)]),
        "hg_flow_describe": (C.c_int, [vp, C.c_char_p, sz, C.POINTER(sz)]),
        "hg_flow_to_device": (C.c_int, [vp, i32]),
        "hg_flow_reserve": (C.c_int, [vp, i64]),
        "hg_flow_execute": (C.c_int, [vp, vp, i32, i64, i64, vp, i32, i64, i64]),
        "hg_flow_execute_sharded": (C.c_int, [vp, vp, i32, i64, i64, vp, i32, i64, i64, C.POINTER(C.c_int), i32]),
        "hg_flow_execute_device": (C.c_int, [vp, vp, i32, i64, i64, vp, i32, i64, i64, vp]),
        "hg_event_create": (C.c_int, [C.POINTER(vp)]),
        "hg_event_create_on": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int]),
        "hg_event_destroy": (None, [vp]),
        "hg_event_record": (C.c_int, [vp, vp]),
        "hg_stream_wait_event": (C.c_int, [vp, vp]),
        "hg_event_query": (C.c_int, [vp]),
        "hg_flow_host_transport": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "hg_host_pack_probe": (C.c_int, [vp, i32, i64, i64, i64, i32, C.POINTER(C.c_double)]),
        "hg_host_store_probe": (C.c_int, [i32, sz, i32, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "hg_host_dma_probe": (C.c_int, [i32, sz, i32, C.POINTER(C.c_double)]),
        "hg_flow_set_profiling": (C.c_int, [vp, i32]),
        "hg_flow_stage_times": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64), i32, C.POINTER(i32)]),
        "hg_flow_stage_name": (C.c_int, [vp, i32, C.c_char_p, sz]),
        "hg_flow_reset_profile": (C.c_int, [vp]),
        "hg_gauss_create": (C.c_int, [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, i32, C.POINTER(vp)]),
        "hg_gauss_free": (None, [vp]),
        "hg_gauss_regression_device": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, vp]),
        "hg_gauss_regression": (C.c_int, [vp, vp, i32, i64, i64, vp, vp]),
        "hg_gauss_regression_multi_device": (C.c_int, [C.POINTER(vp), i32, vp, i32, i64, i64, vp, i64, vp]),
        "hg_patcher_create": (C.c_int, [i32, C.POINTER(vp)]),
        "hg_patcher_free": (None, [vp]),
        "hg_patcher_extract_device": (C.c_int, [vp, vp, i32, i32, i32, i64, vp, i64, i32, i32, vp, i32, i64, vp]),
        "hg_patcher_extract_keyed_device": (C.c_int, [vp, C.c_uint64, vp, i32, i32, i32, i64, vp, i64, i32, i32, vp, i32, i64, vp]),
        "hg_patcher_extract": (C.c_int, [vp, vp, i32, i32, i32, i64, vp, i64, i32, i32, vp, i32, i64]),
        "hg_patcher_extract_rotate_device": (C.c_int, [vp, vp, i32, i32, i32, i64, vp, vp, i64, i32, i32, vp, i32, i64, vp]),
        "hg_patcher_extract_rotate": (C.c_int, [vp, vp, i32, i32, i32, i64, vp, vp, i64, i32, i32, vp, i32, i64]),
        "hg_cascade_update_device": (C.c_int, [i32, i32, C.POINTER(HgCascadeConsts), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "hg_cascade_compact_device": (C.c_int, [i32, vp, i64, vp, vp, vp]),
        "hg_gather_rows_device": (C.c_int, [i32, vp, vp, i64, vp, vp, i64, vp]),
        "hg_cascade_create": (C.c_int, [C.POINTER(HgCascadeStage), i32, i32, i32, i32, C.POINTER(HgCascadeConsts), vp, i32, i32, C.POINTER(vp)]),
        "hg_cascade_free": (None, [vp]),
        "hg_cascade_detect_device": (C.c_int, [vp, vp, i32, i32, i64, vp, vp, i64, vp, vp, vp, vp, i64, C.POINTER(i64), vp, C.POINTER(i64), vp]),
        "hg_cascade_detect_levels_device": (C.c_int, [vp, vp, i32, i32, i64, C.POINTER(HgCascadeLevel), i32, vp, vp, vp, vp, i64, C.POINTER(i64), vp, C.POINTER(i64), vp]),
        "hg_cascade_detect_frame_device": (C.c_int, [vp, vp, i32, i32, i64, i32, i32, C.POINTER(HgCascadeLevel), i32, vp, vp, vp, vp, i64, C.POINTER(i64), vp, C.POINTER(i64), vp]),
        "hg_cascade_grid_device": (C.c_int, [i32, C.POINTER(HgCascadeLevel), i32, vp, vp, i64, C.POINTER(i64), vp]),
        "hg_sfa_train_layer": (C.c_int, [vp, i32, i32, i64, i64, vp, C.c_int32, C.c_int32, i32, vp, vp, vp, vp]),
        "hg_pca_train_layer": (C.c_int, [vp, i32, i32, i64, i64, vp, C.c_int32, C.c_int32, i32, vp, vp, vp, vp]),
        "hg_train_apply_device": (C.c_int, [vp, i32, i64, i64, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, i64, i32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "hg_version", "hg_last_error", "hg_device_count", "hg_flow_load", "hg_flow_free", "hg_flow_info",
    "hg_flow_describe", "hg_flow_to_device", "hg_flow_reserve", "hg_flow_execute", "hg_flow_execute_sharded",
    "hg_flow_execute_device", "hg_event_create", "hg_event_create_on", "hg_event_destroy", "hg_event_record", "hg_stream_wait_event", "hg_event_query", "hg_flow_host_transport", "hg_host_pack_probe", "hg_host_store_probe", "hg_host_dma_probe", "hg_flow_set_profiling", "hg_flow_stage_times", "hg_flow_stage_name",
    "hg_flow_reset_profile", "hg_gauss_create", "hg_gauss_free", "hg_gauss_regression_device",
    "hg_gauss_regression", "hg_patcher_create", "hg_patcher_free", "hg_patcher_extract_device",
    "hg_patcher_extract_keyed_device", "hg_patcher_extract", "hg_patcher_extract_rotate_device", "hg_patcher_extract_rotate", "hg_cascade_update_device",
    "hg_cascade_compact_device", "hg_gather_rows_device", "hg_cascade_create", "hg_cascade_free", "hg_cascade_detect_device",
    "hg_cascade_detect_levels_device", "hg_cascade_detect_frame_device", "hg_cascade_grid_device", "hg_gauss_regression_multi_device",
    "hg_sfa_train_layer", "hg_pca_train_layer", "hg_train_apply_device",
)

_EXC = {HG_ERR_ARG: ValueError, HG_ERR_FORMAT: ValueError, HG_ERR_DIM: ValueError,
        HG_ERR_DEVICE: RuntimeError, HG_ERR_NOMEM: MemoryError, HG_ERR_STATE: RuntimeError}


class NodeException(ValueError):
    """Counterpart of mdp.NodeException (dimension mismatches raise this in MDP)."""


def check(rc):
    if rc == HG_OK:
        return
    msg = lib().hg_last_error().decode("utf-8", "replace")
    if rc == HG_ERR_DIM:
        raise NodeException(msg)
    raise _EXC.get(rc, RuntimeError)("higsfa: " + msg)


def np_dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.uint8:
        return HG_U8
    if dt == np.float32:
        return HG_F32
    if dt == np.float64:
        return HG_F64
    return None
