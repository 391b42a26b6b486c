"""Minimal stand-in for the `pypose` LieTensor API that VSLAM/CameraTracker.py and VSLAM/mast3r_slam/geometry.py
use (pypose is a pip dependency of the reference, README.md:72, unpinned, NOT in /root/reference and not
installed here).  Test infrastructure only: tests/golden/make_golden_tracker.py puts it in sys.modules as `pypose`
so that the REFERENCE's own tracker code can be executed on CPU to produce golden vectors.

Semantics restated from pypose's published definitions:
  Sim3 data  [..., 8] = (t xyz, q xyzw, s);   Act(p) = s R(q) p + t;   matrix() = [[s R, t], [0, 1]]
  A.mul(B) = A o B;   Inv();   quat2unit = quaternion normalised;   X.add(a) = Exp(a) o X
  sim3 data  [..., 7] = (tau, phi, sigma);    Exp(): q = exp(phi), s = e^sigma, t = W tau with
      W = C I + A [phi]x + B [phi]x^2   (the A, B, C of Sim(3), with their small-angle / small-sigma limits)
"""
import torch


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:4]
    qv, p = torch.broadcast_tensors(qv, p)
    uv = 2.0 * torch.linalg.cross(qv, p)
    return p + w * uv + torch.linalg.cross(qv, uv)


class Sim3:
    def __init__(self, data):
        self.data = torch.as_tensor(data)

    # plumbing the reference touches
    def tensor(self):
        return self.data

    def to(self, *a, **k):
        return Sim3(self.data.to(*a, **k))

    def clone(self):
        return Sim3(self.data.clone())

    @property
    def shape(self):
        return self.data.shape

    # group operations
    def _parts(self):
        return self.data[..., 0:3], self.data[..., 3:7], self.data[..., 7:8]

    def Act(self, p):
        t, q, s = self._parts()
        return s * _qrot(q, p) + t

    def Inv(self):
        t, q, s = self._parts()
        qi = q * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=q.dtype)
        return Sim3(torch.cat([-_qrot(qi, t) / s, qi, 1.0 / s], -1))

    def mul(self, other):
        t1, q1, s1 = self._parts()
        t2, q2, s2 = other._parts()
        return Sim3(torch.cat([s1 * _qrot(q1, t2) + t1, _qmul(q1, q2), s1 * s2], -1))

    __mul__ = mul
    __matmul__ = mul

    def add(self, tau):
        """pypose.add on a Lie group: y = Exp(a) * x (left-multiplicative retraction), a in the tangent space."""
        return sim3(torch.as_tensor(tau)[..., :7]).Exp().mul(self)

    def matrix(self):
        t, q, s = self._parts()
        eye = torch.eye(3, dtype=q.dtype).expand(*q.shape[:-1], 3, 3)
        R = torch.stack([_qrot(q, eye[..., :, c]) for c in range(3)], -1)
        M = torch.zeros(*q.shape[:-1], 4, 4, dtype=q.dtype)
        M[..., :3, :3] = s[..., None] * R
        M[..., :3, 3] = t
        M[..., 3, 3] = 1.0
        return M


class sim3:
    def __init__(self, data):
        self.data = torch.as_tensor(data)

    def Exp(self):
        tau, phi, sigma = self.data[..., 0:3], self.data[..., 3:6], self.data[..., 6:7]
        scale = torch.exp(sigma)
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small_t, small_s = th < 1e-6, sigma.abs() < 1e-6
        th_s = torch.where(small_t, torch.ones_like(th), th)
        th2_s = th_s * th_s
        sg_s = torch.where(small_s, torch.ones_like(sigma), sigma)
        imag = torch.where(small_t, 0.5 - th2 / 48.0, torch.sin(0.5 * th_s) / th_s)
        real = torch.where(small_t, 1.0 - th2 / 8.0, torch.cos(0.5 * th_s))
        q = torch.cat([imag * phi, real], -1)
        C = torch.where(small_s, torch.ones_like(sigma), (scale - 1.0) / sg_s)
        a, b, c = scale * torch.sin(th_s), scale * torch.cos(th_s), th2_s + sg_s * sg_s
        A = torch.where(small_s,
                        torch.where(small_t, torch.full_like(th, 0.5), (1.0 - torch.cos(th_s)) / th2_s),
                        torch.where(small_t, ((sg_s - 1.0) * scale + 1.0) / (sg_s * sg_s),
                                    (a * sg_s + (1.0 - b) * th_s) / (th_s * c)))
        B = torch.where(small_s,
                        torch.where(small_t, torch.full_like(th, 1.0 / 6.0), (th_s - torch.sin(th_s)) / (th2_s * th_s)),
                        torch.where(small_t, (scale * 0.5 * sg_s * sg_s + scale - 1.0 - sg_s * scale) / (sg_s ** 3),
                                    (C - ((b - 1.0) * sg_s + a * th_s) / c) / th2_s))
        pt = torch.linalg.cross(phi, tau)
        ppt = torch.linalg.cross(phi, pt)
        return Sim3(torch.cat([C * tau + A * pt + B * ppt, q, scale], -1))


def quat2unit(T):
    d = T.data
    q = d[..., 3:7]
    return Sim3(torch.cat([d[..., 0:3], q / q.norm(dim=-1, keepdim=True), d[..., 7:8]], -1))


def identity_Sim3(*size):
    d = torch.zeros(*size, 8)
    d[..., 6] = 1.0
    d[..., 7] = 1.0
    return Sim3(d)
