"""ctypes binding of oracle/libsvo_oracle.so -- the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class Pyramid(C.Structure):
    _fields_ = [("nlevels", C.c_int), ("pad", C.c_int), ("w", C.c_int * 8), ("h", C.c_int * 8),
                ("pitch", C.c_int * 8), ("data", C.POINTER(C.c_uint8) * 8)]


class PnPResult(C.Structure):
    _fields_ = [("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("R", C.c_double * 9),
                ("n_inliers", C.c_int), ("ransac_iters", C.c_int), ("best_iter", C.c_int),
                ("lm_iters", C.c_int), ("ok", C.c_int)]


class StepResult(C.Structure):
    _fields_ = [("n_prev_kps", C.c_int), ("n_cur_kps", C.c_int), ("n_tracked", C.c_int),
                ("n_inliers", C.c_int), ("ok", C.c_int), ("fail_stage", C.c_int),
                ("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("R", C.c_double * 9),
                ("T_rel_inv", C.c_double * 16)]


class TrackParams(C.Structure):
    _fields_ = [("P1", C.c_double * 12), ("P2", C.c_double * 12),
                ("feature_match_error", C.c_double), ("num_features_tracking", C.c_int),
                ("inlier_rate", C.c_double), ("iterations", C.c_int), ("reproj_err", C.c_float),
                ("confidence", C.c_float), ("fast_thr", C.c_int),
                ("min_t2", C.c_double), ("max_t2", C.c_double)]


def build(force=False):
    """Compile oracle/libsvo_oracle.so with gcc (a no-op when it is up to date)."""
    so = os.path.join(_HERE, "libsvo_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsvo_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_rng_next.restype = C.c_uint32
    return _LIB


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def fast(img, thr=20, nms=True, cap=1 << 17):
    img, p = _u8(img)
    h, w = img.shape
    out = np.zeros(cap, dtype=KP_DTYPE)
    n = lib().orc_fast9_16(p, w, h, w, int(thr), int(bool(nms)), out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n].copy()


def pyr_down(img):
    img, p = _u8(img)
    h, w = img.shape
    dst = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(p, w, h, w, dst.ctypes.data_as(C.POINTER(C.c_uint8)), dst.shape[1])
    return dst


class PyramidHandle:
    def __init__(self, img, win=21, max_level=3):
        img, p = _u8(img)
        self.s = Pyramid()
        h, w = img.shape
        lib().orc_pyramid_build(p, w, h, w, win, max_level, C.byref(self.s))

    def level(self, l, padded=False):
        s = self.s
        w, h, pad, pitch = s.w[l], s.h[l], s.pad, s.pitch[l]
        buf = np.ctypeslib.as_array(s.data[l], shape=((h + 2 * pad), pitch)).copy()
        return buf if padded else buf[pad:pad + h, pad:pad + w].copy()

    @property
    def nlevels(self):
        return self.s.nlevels

    def __del__(self):
        try:
            lib().orc_pyramid_free(C.byref(self.s))
        except Exception:
            pass


def lk_track(prev, nxt, pts, win=21, max_level=3, max_iter=30, eps=0.01, min_eig=1e-3, threads=1):
    """prev/nxt: u8 images or PyramidHandle; pts: (n,2) float32 -> (next_pts, status)."""
    pp = prev if isinstance(prev, PyramidHandle) else PyramidHandle(prev, win, max_level)
    pn = nxt if isinstance(nxt, PyramidHandle) else PyramidHandle(nxt, win, max_level)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    rc = lib().orc_lk_track(C.byref(pp.s), C.byref(pn.s), pts.ctypes.data_as(C.c_void_p), n,
                            out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p),
                            win, max_iter, C.c_double(eps), C.c_float(min_eig), threads)
    assert rc == 0
    return out, st


LK_ACCUM_EXACT, LK_ACCUM_FLOAT_RASTER, LK_ACCUM_FLOAT_SSE, LK_ACCUM_LEGACY_SSE2, LK_ACCUM_SIMD128 = 0, 1, 2, 3, 4
COMPAT_TRIANGULATE, COMPAT_PNP_REFIT, COMPAT_PNP_MINIMAL, COMPAT_LK_LANES = 0, 1, 2, 3


def set_opencv_compat(knob, value):
    """Version forks of the restated OpenCV callees (oracle/geom.c; process-wide, tests only).
    Returns the previous value; 0 is the canonical choice the HIP path is compared with."""
    old = lib().orc_get_opencv_compat(int(knob))
    lib().orc_set_opencv_compat(int(knob), int(value))
    return old



def set_lk_accum(mode):
    """Sensitivity switch of the LK sums A11, A12, A22, b1, b2 (process-wide; tests only): 0 = exact int64
    (canonical choice C0, the parity target), 1 = upstream's float accumulation in raster order, 2 = in the
    lane order of upstream's SSE2 block.  Returns the previous mode."""
    old = lib().orc_lk_get_accum()
    lib().orc_lk_set_accum(int(mode))
    return old


def circular_keep(p0, p1, p2, p3, p0r, s0, s1, s2, s3, match_err=3.0):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (p0, p1, p2, p3, p0r)]
    sts = [np.ascontiguousarray(a, np.uint8) for a in (s0, s1, s2, s3)]
    n = arrs[0].shape[0]
    keep = np.zeros(n, np.uint8)
    m = lib().orc_circular_keep(*[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                *[a.ctypes.data_as(C.c_void_p) for a in sts], n,
                                C.c_double(match_err), keep.ctypes.data_as(C.c_void_p))
    return keep, m


def triangulate(P1, P2, x1, x2, want4=False):
    P1 = np.ascontiguousarray(P1, np.float64).reshape(12)
    P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
    x1 = np.ascontiguousarray(x1, np.float32).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
    n = x1.shape[0]
    out = np.zeros((n, 3), np.float32)
    out4 = np.zeros((4, n), np.float32)
    lib().orc_triangulate(P1.ctypes.data_as(C.c_void_p), P2.ctypes.data_as(C.c_void_p),
                          x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), n,
                          out.ctypes.data_as(C.c_void_p), out4.ctypes.data_as(C.c_void_p))
    return (out, out4) if want4 else out


def pnp_ransac(obj, img, K, iterations=500, reproj_err=0.5, confidence=0.99):
    obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    n = obj.shape[0]
    res = PnPResult()
    mask = np.zeros(max(n, 1), np.uint8)
    # the reference passes confidence through a float (src/tracking.cpp:481)
    conf = float(np.float32(confidence))
    lib().orc_pnp_ransac(obj.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p), n,
                         K.ctypes.data_as(C.c_void_p), int(iterations), C.c_float(reproj_err),
                         C.c_double(conf), C.byref(res), mask.ctypes.data_as(C.c_void_p))
    return dict(ok=res.ok, rvec=np.array(res.rvec), tvec=np.array(res.tvec),
                R=np.array(res.R).reshape(3, 3), n_inliers=res.n_inliers,
                ransac_iters=res.ransac_iters, best_iter=res.best_iter, lm_iters=res.lm_iters,
                mask=mask[:n].copy())


def jacobi_svd(A):
    """SVD of an m x n (m >= n) double matrix the way cv::SVD::compute does: (W, U, Vt)."""
    A = np.asarray(A, np.float64)
    m, n = A.shape
    At = np.ascontiguousarray(A.T.copy())
    W = np.zeros(n)
    Vt = np.zeros((n, n))
    lib().orc_jacobi_svd(At.ctypes.data_as(C.c_void_p), m, n, W.ctypes.data_as(C.c_void_p),
                         Vt.ctypes.data_as(C.c_void_p))
    return W, At.T.copy(), Vt


def epnp(pws, us, fu, fv, uc, vc):
    pws = np.ascontiguousarray(pws, np.float64).reshape(-1, 3)
    us = np.ascontiguousarray(us, np.float64).reshape(-1, 2)
    R = np.zeros(9)
    t = np.zeros(3)
    lib().orc_epnp(pws.ctypes.data_as(C.c_void_p), us.ctypes.data_as(C.c_void_p), pws.shape[0],
                   C.c_double(fu), C.c_double(fv), C.c_double(uc), C.c_double(vc),
                   R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    return R.reshape(3, 3), t


def rodrigues_vec2mat(r, jac=False):
    r = np.ascontiguousarray(r, np.float64).reshape(3)
    R = np.zeros(9)
    J = np.zeros(27)
    lib().orc_rodrigues_vec2mat(r.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                                J.ctypes.data_as(C.c_void_p) if jac else None)
    return (R.reshape(3, 3), J.reshape(3, 9)) if jac else R.reshape(3, 3)


def rodrigues_mat2vec(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    r = np.zeros(3)
    lib().orc_rodrigues_mat2vec(R.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
    return r


def rng_sequence(n, seed=0xFFFFFFFFFFFFFFFF):
    st = C.c_uint64(seed)
    return [lib().orc_rng_next(C.byref(st)) for _ in range(n)]


def gate_and_accumulate(R, t, pose, min_t2=0.0005 ** 2, max_t2=100.0):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    pose = np.ascontiguousarray(pose, np.float64).reshape(16).copy()
    Ti = np.zeros(16)
    rc = lib().orc_gate_and_accumulate(R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                       C.c_double(min_t2), C.c_double(max_t2),
                                       pose.ctypes.data_as(C.c_void_p), Ti.ctypes.data_as(C.c_void_p))
    return rc, pose.reshape(4, 4), Ti.reshape(4, 4)


def make_params(P1, P2, feature_match_error=3.0, num_features_tracking=5, inlier_rate=0.01,
                iterations=500, reproj_err=0.5, confidence=0.99, fast_thr=20,
                min_t2=0.0005 * 0.0005, max_t2=100.0):
    p = TrackParams()
    for i, v in enumerate(np.asarray(P1, np.float64).reshape(12)):
        p.P1[i] = v
    for i, v in enumerate(np.asarray(P2, np.float64).reshape(12)):
        p.P2[i] = v
    p.feature_match_error = feature_match_error
    p.num_features_tracking = num_features_tracking
    p.inlier_rate = inlier_rate
    p.iterations = iterations
    p.reproj_err = reproj_err
    p.confidence = confidence
    p.fast_thr = fast_thr
    p.min_t2 = min_t2
    p.max_t2 = max_t2
    return p


def lk_track_step(params, prevL, prevR, curL, curR, prev_kps, pose, cur_cap=1 << 16, threads=1,
                  want_tracks=False):
    """One Tracking::LK_StereoF2F_PnP_Track step.  Returns (result dict, cur_kps, new pose)."""
    imgs = [_u8(a) for a in (prevL, prevR, curL, curR)]
    h, w = imgs[0][0].shape
    prev_kps = np.ascontiguousarray(prev_kps, dtype=KP_DTYPE)
    n = prev_kps.shape[0]
    cur = np.zeros(cur_cap, dtype=KP_DTYPE)
    pose = np.ascontiguousarray(pose, np.float64).reshape(16).copy()
    res = StepResult()
    tracks = np.zeros((4, max(n, 1), 2), np.float32)
    lib().orc_lk_track_step(C.byref(params), imgs[0][1], imgs[1][1], imgs[2][1], imgs[3][1], w, h, w,
                            prev_kps.ctypes.data_as(C.c_void_p), n, cur.ctypes.data_as(C.c_void_p),
                            cur_cap, pose.ctypes.data_as(C.c_void_p), C.byref(res),
                            tracks.ctypes.data_as(C.c_void_p) if want_tracks else None, threads)
    d = dict(ok=res.ok, fail_stage=res.fail_stage, n_prev_kps=res.n_prev_kps,
             n_cur_kps=res.n_cur_kps, n_tracked=res.n_tracked, n_inliers=res.n_inliers,
             rvec=np.array(res.rvec), tvec=np.array(res.tvec), R=np.array(res.R).reshape(3, 3),
             T_rel_inv=np.array(res.T_rel_inv).reshape(4, 4))
    if want_tracks:
        d["tracks"] = tracks[:, :res.n_tracked].copy()
    return d, cur[:min(res.n_cur_kps, cur_cap)].copy(), pose.reshape(4, 4)


# ---- ORB path -------------------------------------------------------------------------------
def orb_setup(nfeatures=2000, scale_factor=1.2, nlevels=8):
    sc = np.zeros(nlevels, np.float32)
    inv = np.zeros(nlevels, np.float32)
    quota = np.zeros(nlevels, np.int32)
    umax = np.zeros(16, np.int32)
    lib().orc_orb_setup(int(nfeatures), C.c_float(scale_factor), int(nlevels), sc.ctypes.data_as(C.c_void_p),
                        inv.ctypes.data_as(C.c_void_p), quota.ctypes.data_as(C.c_void_p), umax.ctypes.data_as(C.c_void_p))
    return sc, inv, quota, umax


def resize_linear(img, dw, dh):
    img, p = _u8(img)
    h, w = img.shape
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(p, w, h, w, out.ctypes.data_as(C.c_void_p), dw, dh, dw)
    return out


def orb_pyramid_level(img, level, scale_factor=1.2, nlevels=8):
    img, p = _u8(img)
    h, w = img.shape
    ow, oh = C.c_int(0), C.c_int(0)
    lib().orc_orb_pyramid_level(p, w, h, w, C.c_float(scale_factor), nlevels, level, None, C.byref(ow), C.byref(oh))
    out = np.zeros((oh.value, ow.value), np.uint8)
    lib().orc_orb_pyramid_level(p, w, h, w, C.c_float(scale_factor), nlevels, level, out.ctypes.data_as(C.c_void_p),
                                C.byref(ow), C.byref(oh))
    return out


def fast_atan2(y, x):
    lib().orc_fast_atan2.restype = C.c_float
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def gauss7_kernel():
    k = (C.c_int * 7)()
    lib().orc_gauss7_kernel(k)
    return list(k)


def gauss_blur7(img):
    img, p = _u8(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_gauss_blur7(p, w, h, w, out.ctypes.data_as(C.c_void_p), w)
    return out


def orb_extract(img, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, cap=8192):
    """ORBextractor::operator(): (keypoints, descriptors (n,32) uint8, per-level counts)."""
    img, p = _u8(img)
    h, w = img.shape
    kps = np.zeros(cap, dtype=KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    per = np.zeros(8, np.int32)
    n = lib().orc_orb_extract(p, w, h, w, int(nfeatures), C.c_float(scale_factor), int(nlevels), int(ini_th),
                              int(min_th), kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                              per.ctypes.data_as(C.c_void_p))
    return kps[:n].copy(), desc[:n].copy(), per


def match_hamming(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros(max(len(q), 1), np.int32)
    dist = np.zeros(max(len(q), 1), np.float32)
    lib().orc_match_hamming(q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                            idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
    return idx[:len(q)], dist[:len(q)]


def orb_robust_match(lastL, dLastL, lastR, dLastR, curL, dCurL, match_err=3.0):
    ks = [np.ascontiguousarray(k, dtype=KP_DTYPE) for k in (lastL, lastR, curL)]
    ds = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in (dLastL, dLastR, dCurL)]
    cap = max(len(ks[0]), 1)
    outs = [np.zeros((cap, 2), np.float32) for _ in range(3)]
    m = lib().orc_orb_robust_match(ks[0].ctypes.data_as(C.c_void_p), ds[0].ctypes.data_as(C.c_void_p), len(ks[0]),
                                   ks[1].ctypes.data_as(C.c_void_p), ds[1].ctypes.data_as(C.c_void_p), len(ks[1]),
                                   ks[2].ctypes.data_as(C.c_void_p), ds[2].ctypes.data_as(C.c_void_p), len(ks[2]),
                                   C.c_double(match_err), *[o.ctypes.data_as(C.c_void_p) for o in outs])
    return [o[:m].copy() for o in outs]          # t2_left, t1_left, t1_right


def orb_track_step(params, lastL, dLastL, lastR, dLastR, curL, dCurL, pose):
    ks = [np.ascontiguousarray(k, dtype=KP_DTYPE) for k in (lastL, lastR, curL)]
    ds = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in (dLastL, dLastR, dCurL)]
    pose = np.ascontiguousarray(pose, np.float64).reshape(16).copy()
    res = StepResult()
    lib().orc_orb_track_step(C.byref(params), ks[0].ctypes.data_as(C.c_void_p), ds[0].ctypes.data_as(C.c_void_p), len(ks[0]),
                             ks[1].ctypes.data_as(C.c_void_p), ds[1].ctypes.data_as(C.c_void_p), len(ks[1]),
                             ks[2].ctypes.data_as(C.c_void_p), ds[2].ctypes.data_as(C.c_void_p), len(ks[2]),
                             pose.ctypes.data_as(C.c_void_p), C.byref(res))
    d = dict(ok=res.ok, fail_stage=res.fail_stage, n_prev_kps=res.n_prev_kps, n_cur_kps=res.n_cur_kps,
             n_tracked=res.n_tracked, n_inliers=res.n_inliers, rvec=np.array(res.rvec), tvec=np.array(res.tvec),
             R=np.array(res.R).reshape(3, 3), T_rel_inv=np.array(res.T_rel_inv).reshape(4, 4))
    return d, pose.reshape(4, 4)


def orb_distribute(xyr, min_x, max_x, min_y, max_y, n_features):
    """DistributeOctTree on (n, 3) float32 candidates (x, y, response): indices of the kept ones, list order."""
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    sel = np.zeros(max(len(xyr), 1), np.int32)
    m = lib().orc_orb_distribute(xyr.ctypes.data_as(C.c_void_p), len(xyr), int(min_x), int(max_x), int(min_y), int(max_y),
                                 int(n_features), sel.ctypes.data_as(C.c_void_p))
    return sel[:m].copy()


def orb_candidates(img, level, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, cap=1 << 16):
    img, p = _u8(img)
    h, w = img.shape
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_orb_candidates(p, w, h, w, C.c_float(scale_factor), nlevels, level, ini_th, min_th,
                                 out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()
