"""ctypes front-end of oracle/frcnn_oracle_c.c (TEST INFRASTRUCTURE ONLY: used by tests/ and
by bench.py's cpu_baseline leg; never by the product package)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfrcnn_oracle_c.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.oc_nms.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def iou_argmax(lib, anchors, gt):
    a = np.ascontiguousarray(anchors, np.float32).reshape(-1, 4)
    g = np.ascontiguousarray(gt, np.float32).reshape(-1, 4)
    mi = np.empty(a.shape[0], np.float32)
    best = np.empty(a.shape[0], np.int32)
    lib.oc_iou_argmax(_p(a), a.shape[0], _p(g), g.shape[0], _p(mi), _p(best))
    return mi, best


def topk(lib, scores, k):
    s = np.ascontiguousarray(scores, np.float32).reshape(-1)
    v = np.empty(k, np.float32)
    i = np.empty(k, np.int32)
    lib.oc_topk(_p(s), s.shape[0], k, _p(v), _p(i))
    return v, i


def decode_gather(lib, anchors, deltas, idx):
    a = np.ascontiguousarray(anchors, np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(deltas, np.float32).reshape(-1, 4)
    ix = np.ascontiguousarray(idx, np.int32)
    out = np.empty((ix.shape[0], 4), np.float32)
    lib.oc_decode_gather(_p(a), _p(d), _p(ix), ix.shape[0], _p(out))
    return out


def nms(lib, boxes, thr, post_n):
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    prop = np.empty((post_n, 4), np.float32)
    keep = np.empty(post_n, np.int32)
    n = lib.oc_nms(_p(b), b.shape[0], C.c_float(thr), post_n, _p(prop), _p(keep))
    return prop, keep, n


def roi_crop_pool(lib, fmap, rois, crop=(14, 14), pool=2, want_winner=True):
    f = np.ascontiguousarray(fmap, np.float32)
    H, W, Cc = f.shape
    r = np.ascontiguousarray(rois, np.float32).reshape(-1, 4)
    ph, pw = crop[0] // pool, crop[1] // pool
    out = np.empty((r.shape[0], ph, pw, Cc), np.float32)
    win = np.empty((r.shape[0], ph, pw, Cc), np.uint8) if want_winner else None
    lib.oc_roi_crop_pool(_p(f), H, W, Cc, _p(r), r.shape[0], crop[0], crop[1], pool, _p(out),
                         _p(win) if win is not None else None)
    return out, win


def roi_crop_pool_bwd(lib, dY, winner, rois, fmap_hwc, crop=(14, 14), pool=2):
    """One image: dY / winner [R,ph,pw,C], rois [R,4] -> dF [H,W,C]."""
    H, W, Cc = fmap_hwc
    g = np.ascontiguousarray(dY, np.float32)
    r = np.ascontiguousarray(rois, np.float32).reshape(-1, 4)
    w = None if winner is None else np.ascontiguousarray(winner, np.uint8)
    out = np.empty((H, W, Cc), np.float32)
    lib.oc_roi_crop_pool_bwd(_p(g), _p(w) if w is not None else None, _p(r), r.shape[0], H, W, Cc,
                             crop[0], crop[1], pool, _p(out))
    return out


def fc1(lib, x, w, bias=None, relu=True):
    """x [M,K], w [K,N], bias [N] | None -> act(x w + bias) as the k-ordered fmaf chain."""
    xx = np.ascontiguousarray(x, np.float32)
    xx = xx.reshape(-1, xx.shape[-1])
    ww = np.ascontiguousarray(w, np.float32)
    M, K = xx.shape
    N = ww.shape[1]
    assert ww.shape[0] == K
    b = None if bias is None else np.ascontiguousarray(bias, np.float32).reshape(N)
    out = np.empty((M, N), np.float32)
    lib.oc_fc1(_p(xx), _p(ww), _p(b) if b is not None else None, C.c_long(M), K, N, int(bool(relu)),
               _p(out))
    return out
