"""Synthetic KITTI-like stereo sequences with ground-truth poses (test/bench data plumbing).

There is no KITTI data in the build or GPU environment, so BASELINE
config #1/#2 run on "S0": a procedurally textured corridor (ground, two walls, a ceiling) rendered
by ray casting through the reference rig of config/default.yaml:33-57 (fx = fy = 718.856,
cx = 607.193, cy = 185.216, baseline 0.537 m) along a forward trajectory with a slowly oscillating
yaw.  Left and right views are rendered from the two camera centres, so stereo disparity, temporal
flow and the PnP geometry are all consistent with the known SE(3) motion.

Written with torch so the same code renders small CPU cases for tests and full-size frames on
cuda for bench.py.  Only integer hashing and float32 arithmetic that is independent of reduction
order is used, but CPU and GPU renderings are NOT required to match bit for bit: every consumer
(HIP path, oracle) is handed the SAME rendered uint8 frames.
"""
import math

import torch

KITTI_K = dict(fx=718.856, fy=718.856, cx=607.193, cy=185.216)
KITTI_BASELINE = 0.537


def proj_matrices(fx=KITTI_K["fx"], fy=KITTI_K["fy"], cx=KITTI_K["cx"], cy=KITTI_K["cy"],
                  baseline=KITTI_BASELINE):
    """P1 = K[I|0], P2 = K[I|t], t = (-baseline, 0, 0)  (reference src/parameter.cpp:42-45)."""
    P1 = [fx, 0.0, cx, 0.0, 0.0, fy, cy, 0.0, 0.0, 0.0, 1.0, 0.0]
    P2 = [fx, 0.0, cx, fx * (-baseline), 0.0, fy, cy, 0.0, 0.0, 0.0, 1.0, 0.0]
    return P1, P2


def _hash01(u, v, k):
    """Integer lattice hash -> float in [0,1).  u, v int64 tensors, k python int."""
    h = (u * 73856093) ^ (v * 19349663) ^ (k * 83492791)
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return (h & 0xFFFF).to(torch.float32) / 65536.0


class StereoSequence:
    """Procedural corridor seen by a forward-moving rectified stereo rig."""

    def __init__(self, width=1241, height=376, n_frames=101, seed=20200710, device="cpu",
                 fx=None, fy=None, cx=None, cy=None, baseline=KITTI_BASELINE,
                 step=1.0, yaw_amp=0.02, yaw_period=48.0, scales=(0.35, 1.4, 5.6),
                 weights=(0.55, 0.3, 0.15), supersample=2, half_width=7.0, cam_height=1.65,
                 ceil_height=6.0, fog=90.0):
        # default intrinsics: KITTI at full size, scaled with the image width otherwise
        s = width / 1241.0
        self.w, self.h = int(width), int(height)
        self.fx = fx if fx is not None else KITTI_K["fx"] * s
        self.fy = fy if fy is not None else KITTI_K["fy"] * s
        self.cx = cx if cx is not None else KITTI_K["cx"] * s
        if cy is None:
            cy = KITTI_K["cy"] * s if abs(height - 376.0 * s) < 2.0 else (height - 1) * 0.5
        self.cy = cy
        self.baseline = baseline
        self.n_frames = n_frames
        self.seed = int(seed)
        self.device = torch.device(device)
        self.step, self.yaw_amp, self.yaw_period = step, yaw_amp, yaw_period
        self.scales, self.weights, self.ss = scales, weights, int(supersample)
        self.half_width, self.cam_height, self.ceil_height, self.fog = half_width, cam_height, ceil_height, fog
        self._poses = self._make_poses()

    # ---- trajectory ------------------------------------------------------------------------
    def _make_poses(self):
        """T_wc (camera -> world) per frame, float64, world = KITTI convention (x right, y down,
        z forward).  Step length varies in [0.8, 1.2]*step, yaw oscillates."""
        g = torch.Generator().manual_seed(self.seed)
        steps = (0.8 + 0.4 * torch.rand(self.n_frames, generator=g, dtype=torch.float64)) * self.step
        poses = []
        x = z = 0.0
        for t in range(self.n_frames):
            yaw = self.yaw_amp * math.sin(2.0 * math.pi * t / self.yaw_period)
            c, s = math.cos(yaw), math.sin(yaw)
            T = torch.tensor([[c, 0.0, s, x], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, z],
                              [0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)
            poses.append(T)
            x += s * float(steps[t])
            z += c * float(steps[t])
        return torch.stack(poses)

    def poses_wc(self):
        return self._poses.clone()

    def relative_gt(self, t):
        """T mapping camera t-1 coordinates to camera t coordinates (what PnP estimates)."""
        return torch.linalg.inv(self._poses[t]) @ self._poses[t - 1]

    def proj(self):
        return proj_matrices(self.fx, self.fy, self.cx, self.cy, self.baseline)

    # ---- rendering -------------------------------------------------------------------------
    def _texture(self, u, v, plane):
        val = torch.zeros_like(u)
        for k, (sc, wt) in enumerate(zip(self.scales, self.weights)):
            iu = torch.floor(u / sc).to(torch.int64)
            iv = torch.floor(v / sc).to(torch.int64)
            val = val + wt * _hash01(iu, iv, self.seed % 65521 + 131 * plane + 17 * k)
        return val

    def _shade(self, ox, oy, oz, dx, dy, dz):
        """Nearest hit among ground (y = cam_height), ceiling (y = -ceil_height), walls
        (x = +-half_width); returns intensity in [0,1]."""
        big = 1e9
        eps = 1e-9
        sg = torch.where(dy > eps, (self.cam_height - oy) / dy, torch.full_like(dy, big))
        sc = torch.where(dy < -eps, (-self.ceil_height - oy) / dy, torch.full_like(dy, big))
        sl = torch.where(dx < -eps, (-self.half_width - ox) / dx, torch.full_like(dx, big))
        sr = torch.where(dx > eps, (self.half_width - ox) / dx, torch.full_like(dx, big))
        s = torch.minimum(torch.minimum(sg, sc), torch.minimum(sl, sr))
        X, Y, Z = ox + s * dx, oy + s * dy, oz + s * dz
        tex = torch.where(s == sg, self._texture(X, Z, 0),
                          torch.where(s == sc, self._texture(X, Z, 1),
                                      torch.where(s == sl, self._texture(Z, Y, 2),
                                                  self._texture(Z, Y, 3))))
        fogw = torch.exp(-s / self.fog)
        return tex * fogw + 0.5 * (1.0 - fogw)

    def render(self, t):
        """Returns (left, right) uint8 tensors of shape (h, w) on self.device."""
        dev = self.device
        T = self._poses[t].to(torch.float32)
        R = T[:3, :3].to(dev)
        p = T[:3, 3]
        ss = self.ss
        offs = [(i + 0.5) / ss - 0.5 for i in range(ss)]
        vs = torch.arange(self.h, device=dev, dtype=torch.float32)
        us = torch.arange(self.w, device=dev, dtype=torch.float32)
        out = []
        for cam in range(2):
            # camera centre in world: left at p, right at p + R * (baseline, 0, 0)
            o = p + T[:3, 0] * (self.baseline * cam)
            acc = torch.zeros(self.h, self.w, device=dev, dtype=torch.float32)
            for oy_ in offs:
                for ox_ in offs:
                    xc = ((us + ox_) - self.cx) / self.fx
                    yc = ((vs + oy_) - self.cy) / self.fy
                    xcg, ycg = torch.meshgrid(xc, yc, indexing="xy")
                    dx = R[0, 0] * xcg + R[0, 1] * ycg + R[0, 2]
                    dy = R[1, 0] * xcg + R[1, 1] * ycg + R[1, 2]
                    dz = R[2, 0] * xcg + R[2, 1] * ycg + R[2, 2]
                    acc += self._shade(float(o[0]), float(o[1]), float(o[2]), dx, dy, dz)
            img = (acc / (ss * ss) * 255.0).clamp(0, 255).round().to(torch.uint8)
            out.append(img)
        return out[0], out[1]

    def render_range(self, t0, t1):
        """Stacked (left, right) of frames [t0, t1): two uint8 tensors (n, h, w)."""
        Ls, Rs = [], []
        for t in range(t0, t1):
            L, R = self.render(t)
            Ls.append(L)
            Rs.append(R)
        return torch.stack(Ls), torch.stack(Rs)
