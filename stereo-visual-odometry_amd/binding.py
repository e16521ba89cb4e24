"""ctypes binding of libsvo_hip.so (include/svo_abi.h).  No compute happens in Python."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MEM_HOST, MEM_DEVICE = 0, 1

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class SvoError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_keypoints", C.c_int32),
                ("max_batch", C.c_int32), ("num_slots", C.c_int32), ("fast_threshold", C.c_int32),
                ("num_features_tracking", C.c_int32), ("iterations", C.c_int32),
                ("reproj_err", C.c_float), ("confidence", C.c_float),
                ("feature_match_error", C.c_double), ("inlier_rate", C.c_double),
                ("min_move2", C.c_double), ("max_move2", C.c_double),
                ("P1", C.c_double * 12), ("P2", C.c_double * 12),
                ("track_mode", C.c_int32), ("orb_nfeatures", C.c_int32), ("orb_scale_factor", C.c_float),
                ("orb_nlevels", C.c_int32), ("orb_ini_th", C.c_int32), ("orb_min_th", C.c_int32),
                ("lk_accum", C.c_int32), ("fast_keep_strongest", C.c_int32)]


MODE_LK, MODE_ORB = 0, 1
LK_ACCUM_EXACT, LK_ACCUM_SSE2, LK_ACCUM_SIMD128, LK_ACCUM_SSE2_LEGACY = 0, 1, 2, 3          # svo_config.lk_accum


class PnPResult(C.Structure):
    _fields_ = [("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("R", C.c_double * 9),
                ("n_inliers", C.c_int32), ("ransac_iters", C.c_int32), ("best_iter", C.c_int32),
                ("lm_iters", C.c_int32), ("ok", C.c_int32), ("_pad", C.c_int32)]


class StepResult(C.Structure):
    _fields_ = [("ok", C.c_int32), ("fail_stage", C.c_int32), ("n_prev_kps", C.c_int32),
                ("n_cur_kps", C.c_int32), ("n_tracked", C.c_int32), ("n_inliers", C.c_int32),
                ("ransac_iters", C.c_int32), ("lm_iters", C.c_int32),
                ("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("R", C.c_double * 9),
                ("T_rel_inv", C.c_double * 16), ("pose", C.c_double * 16)]


STEP_DTYPE = np.dtype([("ok", "<i4"), ("fail_stage", "<i4"), ("n_prev_kps", "<i4"),
                       ("n_cur_kps", "<i4"), ("n_tracked", "<i4"), ("n_inliers", "<i4"),
                       ("ransac_iters", "<i4"), ("lm_iters", "<i4"), ("rvec", "<f8", 3),
                       ("tvec", "<f8", 3), ("R", "<f8", 9), ("T_rel_inv", "<f8", 16),
                       ("pose", "<f8", 16)])
assert STEP_DTYPE.itemsize == C.sizeof(StepResult)


def library_path():
    return os.path.join(_HERE, "libsvo_hip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libsvo_hip.so (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


def load_library():
    """dlopen libsvo_hip.so; raises SvoError when it is missing (there is no CPU fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise SvoError(f"{path} is missing: run __graft_entry__.build() (hipcc) first; "
                       "this package has no CPU fallback")
    lib = C.CDLL(path)
    if lib.svo_config_bytes() != C.sizeof(Config):
        raise SvoError(f"{path}: svo_config is {lib.svo_config_bytes()} bytes, this binding's Config {C.sizeof(Config)} "
                       "(stale build? run __graft_entry__.build())")
    lib.svo_last_error.restype = C.c_char_p
    lib.svo_last_error.argtypes = [C.c_void_p]
    lib.svo_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p)]
    lib.svo_destroy.argtypes = [C.c_void_p]
    lib.svo_destroy.restype = None
    lib.svo_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
    lib.svo_default_config.restype = None
    lib.svo_track_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int]
    lib.svo_get_frame_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.svo_get_last_tracks.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(C.c_int)]
    lib.svo_get_batch_tracks.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(C.c_int)]
    lib.svo_chain_relative.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.svo_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.svo_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.svo_upload_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]
    lib.svo_upload_frames_at.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]
    lib.svo_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.svo_signal_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.svo_signal_stream_inputs.argtypes = [C.c_void_p, C.c_void_p]
    lib.svo_wait_upload.argtypes = [C.c_void_p, C.c_int]
    lib.svo_track_uploaded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.svo_track_uploaded_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.svo_collect_results.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.svo_results_ready.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.svo_set_pose.argtypes = [C.c_void_p, C.c_void_p]
    _LIB = lib
    return lib


def device_count():
    """svo_device_count: HIP devices visible to this process (0 when there is none)."""
    n = C.c_int(0)
    load_library().svo_device_count(C.byref(n))
    return n.value


def default_config(width, height, **overrides):
    cfg = Config()
    load_library().svo_default_config(C.byref(cfg), int(width), int(height))
    for k, v in overrides.items():
        if k in ("P1", "P2"):
            arr = getattr(cfg, k)
            for i, x in enumerate(np.asarray(v, np.float64).reshape(12)):
                arr[i] = float(x)
        else:
            setattr(cfg, k, v)
    return cfg


def _ptr(a):
    """Raw pointer + memory kind of a numpy array (host) or a torch tensor (host or cuda)."""
    if a is None:
        return None, MEM_HOST
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data), MEM_HOST
    # torch tensor
    assert a.is_contiguous()
    return C.c_void_p(a.data_ptr()), (MEM_DEVICE if a.is_cuda else MEM_HOST)


class Context:
    """One svo_ctx: owns every device buffer of the hot path on one GPU."""

    def __init__(self, width, height, device=0, **cfg_overrides):
        self.lib = load_library()
        self.cfg = default_config(width, height, **cfg_overrides)
        h = C.c_void_p()
        rc = self.lib.svo_create(C.byref(self.cfg), int(device), C.byref(h))
        if rc != 0:
            raise SvoError(f"svo_create failed with {rc} (no usable HIP device?) -- there is no CPU fallback")
        self.h = h
        self.width, self.height = int(width), int(height)

    def close(self):
        if getattr(self, "h", None):
            self.lib.svo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow_soft=False):
        if rc < 0 or (rc > 0 and not allow_soft):
            raise SvoError(f"svo call failed ({rc}): {self.lib.svo_last_error(self.h).decode()}")
        return rc

    # ---- misc -------------------------------------------------------------------------------
    def sync(self):
        self._check(self.lib.svo_sync(self.h))

    def set_stream(self, stream_handle):
        self._check(self.lib.svo_set_stream(self.h, C.c_void_p(stream_handle)))
        self._stream_handle = int(stream_handle or 0)

    def wait_stream(self, stream_handle):
        """The context's stream waits ON THE DEVICE for everything queued on `stream_handle` so far (svo_wait_stream, ABI v7):
        call it before handing the library a device buffer another stream is still filling."""
        self._check(self.lib.svo_wait_stream(self.h, C.c_void_p(stream_handle)))

    def signal_stream(self, stream_handle):
        """`stream_handle` waits ON THE DEVICE for everything the context has queued so far, the side-stream pose stage of
        an overlap-mode batch included (svo_signal_stream): what a consumer of device-resident results on another stream
        calls instead of svo_sync()."""
        self._check(self.lib.svo_signal_stream(self.h, C.c_void_p(stream_handle)))

    def signal_stream_inputs(self, stream_handle):
        """`stream_handle` waits ON THE DEVICE until the kernels that read the caller's frames have run (the front end);
        the side-stream pose stage is NOT waited for (svo_signal_stream_inputs, ABI v9)."""
        self._check(self.lib.svo_signal_stream_inputs(self.h, C.c_void_p(stream_handle)))

    def _order_in(self, t):
        """Tensors torch has just produced (an output's zero fill is a kernel on TORCH's current stream; an input may still
        be being written there) are ordered before the library's kernels on the device -- no host synchronisation."""
        import torch
        other = torch.cuda.current_stream(t.device).cuda_stream
        if other and other == getattr(self, "_stream_handle", 0):
            return False                   # the context was handed this very stream (set_stream): already in order
        self.wait_stream(other)
        return True

    def _order_out(self, t):
        """... and what the library wrote into them before whatever torch's current stream does next."""
        import torch
        other = torch.cuda.current_stream(t.device).cuda_stream
        if other and other == getattr(self, "_stream_handle", 0):
            return                         # (and a signal would make the next front end wait for this batch's pose stage)
        self.signal_stream(other)

    @property
    def num_levels(self):
        return self.lib.svo_num_levels(self.h)

    def set_overlap(self, on=True):
        """Pose stage of batch k on a side stream, overlapped with batch k+1's front end."""
        self._check(self.lib.svo_set_overlap(self.h, int(on)))

    def wait_results(self):
        self._check(self.lib.svo_wait_results(self.h))

    def enable_timing(self, on=True):
        self._check(self.lib.svo_enable_timing(self.h, int(on)))

    def get_timing(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = self.lib.svo_get_timing(self.h, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    # ---- stage API --------------------------------------------------------------------------
    def _img(self, img):
        """(pointer, row pitch in bytes, memory kind) of a u8 image; rows may be padded (pitch >= width)."""
        assert tuple(img.shape) == (self.height, self.width), (img.shape, self.height, self.width)
        if isinstance(img, np.ndarray):
            assert img.dtype == np.uint8 and img.strides[1] == 1
            return C.c_void_p(img.ctypes.data), int(img.strides[0]), MEM_HOST
        assert img.element_size() == 1 and img.stride(1) == 1
        return C.c_void_p(img.data_ptr()), int(img.stride(0)), (MEM_DEVICE if img.is_cuda else MEM_HOST)

    def fast_detect(self, img, threshold=20, nonmax=True, cap=None):
        cap = cap or self.cfg.max_keypoints
        p, pitch, mem = self._img(img)
        out = np.zeros(cap, dtype=KP_DTYPE)
        n = C.c_int(0)
        self._check(self.lib.svo_fast_detect(self.h, p, pitch, mem, int(threshold), int(bool(nonmax)),
                                             C.c_void_p(out.ctypes.data), cap, C.byref(n)))
        return out[:n.value].copy()

    def build_pyramid(self, slot, img):
        p, pitch, mem = self._img(img)
        self._check(self.lib.svo_build_pyramid(self.h, int(slot), p, pitch, mem))

    def read_pyramid_level(self, slot, level):
        w, h = C.c_int(0), C.c_int(0)
        self._check(self.lib.svo_read_pyramid_level(self.h, slot, level, None, 0, MEM_HOST,
                                                    C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._check(self.lib.svo_read_pyramid_level(self.h, slot, level, C.c_void_p(out.ctypes.data),
                                                    w.value, MEM_HOST, C.byref(w), C.byref(h)))
        return out

    def lk_track(self, slot_prev, slot_next, pts):
        """pts: (n,2) float32 numpy (host) or torch cuda tensor -> (next_pts, status), same kind."""
        if isinstance(pts, np.ndarray):
            pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
            n = pts.shape[0]
            out = np.zeros((n, 2), np.float32)
            st = np.zeros(n, np.uint8)
        else:
            import torch
            n = pts.shape[0]
            out = torch.zeros((n, 2), dtype=torch.float32, device=pts.device)
            st = torch.zeros(n, dtype=torch.uint8, device=pts.device)
            self._order_in(st)
        pi, mem = _ptr(pts)
        po, _ = _ptr(out)
        ps, _ = _ptr(st)
        self._check(self.lib.svo_lk_track(self.h, slot_prev, slot_next, pi, n, po, ps, mem))
        if mem == MEM_DEVICE:
            self._order_out(st)
        return out, st

    def circular_match(self, slots, t1_left):
        """slots = (prevL, prevR, curL, curR); returns the four compacted (M,2) arrays."""
        if isinstance(t1_left, np.ndarray):
            t1_left = np.ascontiguousarray(t1_left, np.float32).reshape(-1, 2)
            n = t1_left.shape[0]
            outs = [np.zeros((max(n, 1), 2), np.float32) for _ in range(4)]
        else:
            import torch
            n = t1_left.shape[0]
            outs = [torch.zeros((max(n, 1), 2), dtype=torch.float32, device=t1_left.device) for _ in range(4)]
            self._order_in(outs[0])
        pi, mem = _ptr(t1_left)
        m = C.c_int(0)
        self._check(self.lib.svo_circular_match(self.h, *[int(s) for s in slots], pi, n,
                                                *[_ptr(o)[0] for o in outs], C.byref(m), mem))
        return [o[:m.value] for o in outs]

    def triangulate(self, P1, P2, x1, x2):
        P1 = np.ascontiguousarray(P1, np.float64).reshape(12)
        P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
        if isinstance(x1, np.ndarray):
            x1 = np.ascontiguousarray(x1, np.float32).reshape(-1, 2)
            x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
            out = np.zeros((x1.shape[0], 3), np.float32)
        else:
            import torch
            out = torch.zeros((x1.shape[0], 3), dtype=torch.float32, device=x1.device)
            self._order_in(out)
        p1, mem = _ptr(x1)
        p2, _ = _ptr(x2)
        self._check(self.lib.svo_triangulate(self.h, C.c_void_p(P1.ctypes.data), C.c_void_p(P2.ctypes.data),
                                             p1, p2, x1.shape[0], _ptr(out)[0], mem))
        if mem == MEM_DEVICE:
            self._order_out(out)
        return out

    def pnp_ransac(self, obj, img, K, iterations=500, reproj_err=0.5, confidence=0.99):
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        if isinstance(obj, np.ndarray):
            obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
            img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
            mask = np.zeros(max(obj.shape[0], 1), np.uint8)
        else:
            import torch
            mask = torch.zeros(max(obj.shape[0], 1), dtype=torch.uint8, device=obj.device)
            self._order_in(mask)
        n = obj.shape[0]
        res = PnPResult()
        po, mem = _ptr(obj)
        conf = float(np.float32(confidence))      # a float at reference src/tracking.cpp:481
        self._check(self.lib.svo_pnp_ransac(self.h, po, _ptr(img)[0], n, C.c_void_p(K.ctypes.data),
                                            int(iterations), C.c_float(reproj_err), C.c_double(conf),
                                            C.byref(res), _ptr(mask)[0], mem))
        if mem == MEM_DEVICE:
            self._order_out(mask)
            mask = mask.cpu().numpy()
        return dict(ok=res.ok, rvec=np.array(res.rvec), tvec=np.array(res.tvec),
                    R=np.array(res.R).reshape(3, 3), n_inliers=res.n_inliers,
                    ransac_iters=res.ransac_iters, best_iter=res.best_iter, lm_iters=res.lm_iters,
                    mask=np.asarray(mask[:n]).copy())

    # ---- ORB path ---------------------------------------------------------------------------
    def orb_extract(self, img, cap=None):
        """ORBextractor::operator(): (keypoints, descriptors (n,32) uint8, per-level counts)."""
        cap = cap or self.cfg.max_keypoints
        p, pitch, mem = self._img(img)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        per = np.zeros(8, np.int32)
        n = C.c_int(0)
        self._check(self.lib.svo_orb_extract(self.h, p, pitch, mem, C.c_void_p(kps.ctypes.data),
                                             C.c_void_p(desc.ctypes.data), cap, C.byref(n), C.c_void_p(per.ctypes.data)))
        return kps[:n.value].copy(), desc[:n.value].copy(), per

    def orb_read_level(self, level):
        w, h = C.c_int(0), C.c_int(0)
        self._check(self.lib.svo_orb_read_level(self.h, level, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._check(self.lib.svo_orb_read_level(self.h, level, C.c_void_p(out.ctypes.data), C.byref(w), C.byref(h)))
        return out

    def orb_read_candidates(self, level, cap=8192):
        out = np.zeros((cap, 4), np.float32)
        n = C.c_int(0)
        self._check(self.lib.svo_orb_read_candidates(self.h, level, C.c_void_p(out.ctypes.data), cap, C.byref(n)))
        return out[:min(n.value, cap), :3].copy()

    def match_hamming(self, query, train):
        if isinstance(query, np.ndarray):
            query = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
            train = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
            idx = np.zeros(max(len(query), 1), np.int32)
            dist = np.zeros(max(len(query), 1), np.float32)
        else:
            import torch
            idx = torch.zeros(max(len(query), 1), dtype=torch.int32, device=query.device)
            dist = torch.zeros(max(len(query), 1), dtype=torch.float32, device=query.device)
            self._order_in(dist)
        pq, mem = _ptr(query)
        self._check(self.lib.svo_match_hamming(self.h, pq, len(query), _ptr(train)[0], len(train), _ptr(idx)[0],
                                               _ptr(dist)[0], mem))
        if mem == MEM_DEVICE:
            self._order_out(dist)
        return idx[:len(query)], dist[:len(query)]

    # ---- fused API --------------------------------------------------------------------------
    def add_frame(self, left, right):
        pl, pitch, mem = self._img(left)
        pr, pitch_r, mem_r = self._img(right)
        assert pitch == pitch_r and mem == mem_r
        res = StepResult()
        if mem == MEM_DEVICE:
            # ORB mode reads level 0 IN PLACE until the end of the front end, and the frames may still be being written on
            # torch's stream: order them before the library's kernels.  svo_add_frame returns with the record on the host,
            # i.e. after everything that reads the frames, so nothing is left to order afterwards.
            self._order_in(left)
        rc = self._check(self.lib.svo_add_frame(self.h, pl, pr, pitch, mem, C.byref(res)), allow_soft=True)
        return rc, np.frombuffer(bytes(res), dtype=STEP_DTYPE)[0].copy()

    def reset(self):
        self._check(self.lib.svo_reset(self.h))

    def get_pose(self):
        pose = np.zeros(16)
        self._check(self.lib.svo_get_pose(self.h, C.c_void_p(pose.ctypes.data)))
        return pose.reshape(4, 4)

    def track_batch(self, left_frames, right_frames, pose0=None, results=None):
        """left/right_frames: torch cuda uint8 tensors (F, h, pitch>=w) viewed as (F, h, w).
        Returns a numpy structured array of F-1 step results (or fills the given cuda tensor)."""
        F = left_frames.shape[0]
        assert left_frames.is_cuda and right_frames.is_cuda
        assert left_frames.stride(2) == 1 and left_frames.stride() == right_frames.stride()
        pitch, fstride = left_frames.stride(1), left_frames.stride(0)
        p0 = None
        if pose0 is not None:
            pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
            p0 = C.c_void_p(pose0.ctypes.data)
        if results is None:
            out = np.zeros(F - 1, dtype=STEP_DTYPE)
            rp, rmem = C.c_void_p(out.ctypes.data), MEM_HOST
        else:
            out = results
            rp, rmem = C.c_void_p(results.data_ptr()), MEM_DEVICE
        # The frames (read in place until the end of the front end in ORB mode) and a device result buffer (its zero fill) may
        # still be in flight on torch's current stream, and with device results the call returns while the kernels run:
        # torch's stream is ordered before the launch, and the kernels that READ THE FRAMES before whatever torch's stream
        # does next (freeing or overwriting them) -- svo_signal_stream_inputs, not svo_signal_stream: making torch's stream
        # wait for the side-stream pose stage after every batch would chain batch k + 1's front end behind batch k's pose
        # stage through that stream and end their overlap.  The RECORDS of a device result buffer are complete after
        # signal_stream(consumer) / sync(), as before.  Two event operations each, no host synchronisation; skipped when
        # the context runs ON torch's current stream (set_stream).
        ordered = self._order_in(left_frames)
        self._check(self.lib.svo_track_batch(self.h, C.c_void_p(left_frames.data_ptr()),
                                             C.c_void_p(right_frames.data_ptr()), int(pitch), int(fstride),
                                             int(F), p0, rp, rmem))
        if results is not None and ordered:
            import torch
            self.signal_stream_inputs(torch.cuda.current_stream(left_frames.device).cuda_stream)
        return out

    # ---- host-resident frame batches (svo_upload_frames / svo_track_uploaded) -----------------
    def host_frames(self, n_frames, pitch=None):
        """(n_frames, height, pitch) uint8 numpy view over page-locked host memory (svo_host_alloc);
        release it with host_free(view)."""
        pitch = pitch or self.width
        nbytes = int(n_frames) * self.height * int(pitch)
        p = C.c_void_p()
        self._check(self.lib.svo_host_alloc(self.h, nbytes, C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        view = np.frombuffer(buf, dtype=np.uint8).reshape(int(n_frames), self.height, int(pitch))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[view.ctypes.data] = p
        return view

    def host_free(self, view):
        p = self._pinned.pop(view.ctypes.data)
        self._check(self.lib.svo_host_free(self.h, p))

    def upload_frames(self, buf, left, right, first_slot=0):
        """left/right: (F, height, pitch) uint8 host arrays (ideally from host_frames); asynchronous.  first_slot: the frame
        slot of the device buffer the first of them goes to (svo_upload_frames_at: a stream whose halo frame is carried on
        the device uploads its new frames into slots 1..)."""
        assert left.shape == right.shape and left.strides == right.strides and left.strides[2] == 1
        self._check(self.lib.svo_upload_frames_at(self.h, int(buf), int(first_slot), C.c_void_p(left.ctypes.data),
                                                  C.c_void_p(right.ctypes.data), int(left.strides[1]), int(left.strides[0]),
                                                  int(left.shape[0])))

    def wait_upload(self, buf):
        self._check(self.lib.svo_wait_upload(self.h, int(buf)))

    def track_uploaded(self, buf, n_frames, pose0=None):
        p0 = None
        if pose0 is not None:
            pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
            p0 = C.c_void_p(pose0.ctypes.data)
        out = np.zeros(int(n_frames) - 1, dtype=STEP_DTYPE)
        self._check(self.lib.svo_track_uploaded(self.h, int(buf), int(n_frames), p0, C.c_void_p(out.ctypes.data), MEM_HOST))
        return out

    def track_uploaded_async(self, buf, n_frames, pose0=None, continue_chain=False, carry_frame=False):
        """svo_track_uploaded without waiting for the GPU; the records are fetched by collect_results()
        (up to two batches outstanding, collected in launch order).  carry_frame (with continue_chain): frame 0 of this
        batch is the previous batch's last frame -- its features are carried over on the device, not computed again."""
        p0 = None
        if pose0 is not None:
            pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
            p0 = C.c_void_p(pose0.ctypes.data)
        flags = (1 if continue_chain else 0) | (2 if continue_chain and carry_frame else 0)
        self._check(self.lib.svo_track_uploaded_async(self.h, int(buf), int(n_frames), p0, flags))

    def collect_results(self, n_pairs):
        out = np.zeros(int(n_pairs), dtype=STEP_DTYPE)
        self._check(self.lib.svo_collect_results(self.h, C.c_void_p(out.ctypes.data), int(n_pairs)))
        return out

    def results_ready(self):
        """Pairs of the oldest outstanding async batch when its records are complete, else 0 (svo_results_ready)."""
        n = C.c_int(0)
        self._check(self.lib.svo_results_ready(self.h, C.byref(n)))
        return n.value

    def set_pose(self, pose):
        pose = np.ascontiguousarray(pose, np.float64).reshape(16)
        self._check(self.lib.svo_set_pose(self.h, C.c_void_p(pose.ctypes.data)))

    def chain_relative(self, T_rel_inv, ok, pose0=None):
        """poses[p] = pose0 * prod_{q<=p, ok[q]} T[q] (svo_chain_relative).  numpy arrays or torch cuda
        tensors: T (n, 16) float64, ok (n,) int32.  Returns (n, 16) of the same kind."""
        p0 = None
        if pose0 is not None:
            pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
            p0 = C.c_void_p(pose0.ctypes.data)
        if isinstance(T_rel_inv, np.ndarray):
            T = np.ascontiguousarray(T_rel_inv, np.float64).reshape(-1, 16)
            okc = np.ascontiguousarray(ok, np.int32)
            out = np.zeros_like(T)
            tp, op_, up, mem = C.c_void_p(T.ctypes.data), C.c_void_p(okc.ctypes.data), C.c_void_p(out.ctypes.data), MEM_HOST
        else:
            import torch
            T = T_rel_inv.reshape(-1, 16).contiguous()
            okc = ok.to(torch.int32).contiguous()
            assert T.is_cuda and okc.is_cuda and T.dtype == torch.float64
            out = torch.zeros_like(T)
            self._order_in(out)
            tp, op_, up, mem = C.c_void_p(T.data_ptr()), C.c_void_p(okc.data_ptr()), C.c_void_p(out.data_ptr()), MEM_DEVICE
        self._check(self.lib.svo_chain_relative(self.h, tp, op_, int(T.shape[0]), p0, up, mem))
        if mem == MEM_DEVICE:
            self._order_out(out)        # device tensors: complete in stream order, for torch's current stream too
        return out

    def frame_keypoints(self, side=0, with_descriptors=False, cap=65536):
        """Keypoints detected on the frame last given to add_frame (svo_get_frame_keypoints)."""
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8) if with_descriptors else None
        n = C.c_int(0)
        self._check(self.lib.svo_get_frame_keypoints(self.h, int(side), C.c_void_p(kps.ctypes.data),
                                                     C.c_void_p(desc.ctypes.data) if with_descriptors else None, cap, C.byref(n)))
        return (kps[:n.value].copy(), desc[:n.value].copy()) if with_descriptors else kps[:n.value].copy()

    def batch_tracks(self, pair, cap=65536):
        """(t1_left, t1_right, t2_right, t2_left, inlier) of pair `pair` of the last track_batch launch."""
        pts = [np.zeros((cap, 2), np.float32) for _ in range(4)]
        inl = np.zeros(cap, np.uint8)
        n = C.c_int(0)
        self._check(self.lib.svo_get_batch_tracks(self.h, int(pair), *[C.c_void_p(p.ctypes.data) for p in pts],
                                                  C.c_void_p(inl.ctypes.data), cap, C.byref(n)))
        return [p[:n.value].copy() for p in pts] + [inl[:n.value].copy()]

    def last_tracks(self, cap=65536):
        """(t1_left, t1_right, t2_right, t2_left, inlier) of the pair last tracked by add_frame."""
        pts = [np.zeros((cap, 2), np.float32) for _ in range(4)]
        inl = np.zeros(cap, np.uint8)
        n = C.c_int(0)
        self._check(self.lib.svo_get_last_tracks(self.h, *[C.c_void_p(p.ctypes.data) for p in pts], C.c_void_p(inl.ctypes.data),
                                                 cap, C.byref(n)))
        return [p[:n.value].copy() for p in pts] + [inl[:n.value].copy()]
