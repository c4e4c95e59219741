"""ctypes binding of libfrcnn_hip.so (the C ABI declared in include/frcnn_hip.h).

The library is the product: there is no CPU or eager fallback.  Importing this
module without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRCNN_HIP_LIB selects another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("FRCNN_HIP_LIB") or os.path.join(_HERE, "lib", "libfrcnn_hip.so")

OK = 0
E_BADARG, E_WORKSPACE, E_LAUNCH, E_UNSUPPORTED, E_STATIC_LIMIT = -1, -2, -3, -4, -5
NMS_STATIC_BYTES = 16 << 20
F_VALID, F_POSITIVE, F_NEGATIVE = 1, 2, 4
S_OK, S_SHORT_BATCH = 0, 1


class FrcnnError(RuntimeError):
    def __init__(self, fn, status, msg):
        super().__init__(f"{fn} failed: {msg} (status {status})")
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C objectdetection_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    return C.CDLL(LIB_PATH)


lib = _load()

_p, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_i64, _u64, _sz = C.c_int64, C.c_uint64, C.c_size_t

_SIGS = {
    "frcnn_abi_version": (C.c_int, []),
    "frcnn_status_string": (C.c_char_p, [_i]),
    "frcnn_set_tuning": (_i, [C.c_char_p, C.c_longlong]),
    "frcnn_get_tuning": (_i, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "frcnn_anchors": (_i, [_p, _i, _i, C.POINTER(_d), _i, C.POINTER(_d), _i, _d, _d, _p]),
    "frcnn_iou_assign": (_i, [_p, _i, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p]),
    "frcnn_rpn_sample": (_i, [_p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _u64, _i, _i, _i,
                              _p, _p, _p, _p, _p, _p, _p, _p]),
    "frcnn_encode": (_i, [_p, _p, _i64, _i, _p, _p]),
    "frcnn_decode_clip": (_i, [_p, _p, _i, _i64, _i, _p, _p]),
    "frcnn_topk": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "frcnn_topk_sort_workspace_bytes": (_sz, [_i, _i]),
    "frcnn_topk_sort_plan": (_i, [_i, _i, _sz]),
    "frcnn_topk_sort": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "frcnn_nms_workspace_bytes": (_sz, [_i, _i, _i]),
    "frcnn_nms": (_i, [_p, _i, _i, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "frcnn_nms_static": (_i, [_p, _i, _i, _f, _i, _p, _p, _p, _p]),
    "frcnn_nms_fused_workspace_bytes": (_sz, [_i, _i, _i]),
    "frcnn_nms_fused": (_i, [_p, _i, _i, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "frcnn_roi_crop_pool_fwd": (_i, [_p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _p, _i, _i, _i,
                                     _i, _p, _p, _p]),
    "frcnn_roi_fwd_workspace_bytes": (_sz, [_i, _i]),
    "frcnn_roi_fwd_plan": (_i, [_i] * 9),
    "frcnn_roi_crop_pool_fwd_ws": (_i, [_p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _p, _i, _i, _i,
                                        _i, _p, _p, _p, _sz, _p]),
    "frcnn_roi_fwd_prepare": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "frcnn_roi_crop_pool_fwd_prepared": (_i, [_p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _p, _i,
                                              _i, _i, _i, _p, _p, _p, _sz, _p]),
    "frcnn_nchw_to_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "frcnn_roi_crop_pool_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "frcnn_label_proposals": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p,
                                   _p]),
    "frcnn_sample_proposals": (_i, [_p, _i, _i, _i, _p, _p, _p, _u64, _i, _i, _i, _p, _p, _p]),
    "frcnn_detect_postprocess": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p,
                                      _p, _p, _p]),
    "frcnn_rpn_loss": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "frcnn_head_loss": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "frcnn_pad_gt": (_i, [_p, _p, _p, _p, _i, _u64, _i, _i, _i, _p, _p, _p, _p]),
    "frcnn_flip_images": (_i, [_p, _p, _u64, _i, _i, _i, _i, _i, _p, _p]),
    "frcnn_fc1_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTS = tuple(_SIGS)


TUNING_UNSET = -(1 << 63)


def set_tuning(key: str, value=None) -> None:
    """frcnn_set_tuning: value None restores the built-in choice."""
    check("frcnn_set_tuning", lib.frcnn_set_tuning(key.encode(), TUNING_UNSET if value is None
                                                   else int(value)))


def get_tuning(key: str):
    v = C.c_longlong(0)
    check("frcnn_get_tuning", lib.frcnn_get_tuning(key.encode(), C.byref(v)))
    return None if v.value == TUNING_UNSET else v.value


def status_string(status: int) -> str:
    return lib.frcnn_status_string(int(status)).decode()


def check(fn: str, status: int) -> None:
    if status != OK:
        raise FrcnnError(fn, status, status_string(status))
