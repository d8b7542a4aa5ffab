"""Camera + matrix helpers that produce the 144-byte BasicDataUBO and the 464-byte AtmosphericDataUBO blobs.

Host-side mirror (harness glue) of
  Camera            /root/reference/OpenTK-PathTracer/src/Camera.cs:7-83
  BasicDataUBO      /root/reference/OpenTK-PathTracer/src/MainWindow.cs:131-132,195-197,278-279
  atmosphere UBO    /root/reference/OpenTK-PathTracer/src/Render/AtmosphericScatterer.cs:72-89
and of the OpenTK 3.3.2 Matrix4 helpers those call (NuGet package `OpenTK` 3.3.2, pinned in
OpenTK-PathTracer.csproj:42; NOT vendored under /root/reference, so the formulas below are a restatement of the
library's published algorithm — LookAt, CreatePerspectiveFieldOfView, Invert).  Parity is unpinned at this
boundary, which is harmless: the oracle, the llvmpipe run of the reference GLSL and the HIP path all consume the
same blobs, whoever made them.

Byte convention: OpenTK matrices are row-major with row vectors (v' = v * M); the 64 bytes are Row0..Row3.  GLSL
reads the same bytes column-major, i.e. sees M^T and computes M^T * v — the same transform.
"""
from __future__ import annotations

import numpy as np

F = np.float32
FOV_DEGREES = F(103.0)  # MainWindow.cs:18
NEAR_PLANE, FAR_PLANE = F(0.005), F(1000.0)  # MainWindow.cs:18,32


def degrees_to_radians(deg) -> np.float32:
    """OpenTK MathHelper.DegreesToRadians(float): degrees * (float)(PI/180)."""
    return F(F(deg) * F(np.pi / 180.0))


def _normalize(v):
    v = np.asarray(v, dtype=F)
    return (v / F(np.sqrt(F(np.dot(v, v))))).astype(F)


def look_at(eye, target, up) -> np.ndarray:
    """OpenTK Matrix4.LookAt (row-vector convention). Returns 4x4 float32, rows = OpenTK rows."""
    eye, target, up = (np.asarray(a, dtype=F) for a in (eye, target, up))
    z = _normalize(eye - target)
    x = _normalize(np.cross(up, z).astype(F))
    y = _normalize(np.cross(z, x).astype(F))
    m = np.zeros((4, 4), dtype=F)
    m[0] = [x[0], y[0], z[0], 0]
    m[1] = [x[1], y[1], z[1], 0]
    m[2] = [x[2], y[2], z[2], 0]
    m[3] = [-F(np.dot(x, eye)), -F(np.dot(y, eye)), -F(np.dot(z, eye)), 1]
    return m


def perspective_fov(fovy, aspect, z_near, z_far) -> np.ndarray:
    """OpenTK Matrix4.CreatePerspectiveFieldOfView -> CreatePerspectiveOffCenter."""
    fovy, aspect, n, f = F(fovy), F(aspect), F(z_near), F(z_far)
    y_max = F(n * F(np.tan(F(0.5) * fovy)))
    y_min = -y_max
    x_min, x_max = F(y_min * aspect), F(y_max * aspect)
    m = np.zeros((4, 4), dtype=F)
    m[0, 0] = F(2.0) * n / (x_max - x_min)
    m[1, 1] = F(2.0) * n / (y_max - y_min)
    m[2, 0] = (x_max + x_min) / (x_max - x_min)
    m[2, 1] = (y_max + y_min) / (y_max - y_min)
    m[2, 2] = -(f + n) / (f - n)
    m[2, 3] = F(-1.0)
    m[3, 2] = -(F(2.0) * f * n) / (f - n)
    return m


def inverted(m: np.ndarray) -> np.ndarray:
    """OpenTK Matrix4.Inverted(); computed in float64 and rounded once (OpenTK uses float Gauss-Jordan)."""
    return np.linalg.inv(m.astype(np.float64)).astype(F)


class Camera:
    """Camera.cs:16-30 — yaw/pitch fly camera; only its pose matters for the integrator."""

    def __init__(self, position=(-17.14, 3.53, -8.62), up=(0.0, 1.0, 0.0), look_x=-32.2, look_y=0.8):
        # defaults = MainWindow.cs:36
        self.position = np.asarray(position, dtype=F)
        self.up = np.asarray(up, dtype=F)
        self.look_x, self.look_y = F(look_x), F(look_y)
        rx, ry = degrees_to_radians(self.look_x), degrees_to_radians(self.look_y)
        self.view_dir = np.array([F(np.cos(rx)) * F(np.cos(ry)), F(np.sin(ry)), F(np.sin(rx)) * F(np.cos(ry))], dtype=F)
        self.view = look_at(self.position, self.position + self.view_dir, self.up)  # Camera.cs:79-82


def basic_data_ubo(camera: Camera, width: int, height: int, fov_degrees=FOV_DEGREES) -> bytes:
    """144 bytes: InvProjection@0 (MainWindow.cs:278-279), InvView@64 (:131), ViewPos@128 (:132; 16 B written)."""
    inv_proj = inverted(perspective_fov(degrees_to_radians(fov_degrees), F(width) / F(height), NEAR_PLANE, FAR_PLANE))
    inv_view = inverted(camera.view)
    blob = np.zeros(36, dtype=F)
    blob[0:16] = inv_proj.reshape(-1)
    blob[16:32] = inv_view.reshape(-1)
    blob[32:35] = camera.position
    return blob.tobytes()


def atmospheric_data_ubo() -> bytes:
    """464 bytes: InvProjection (90 deg, aspect 1, near .1, far 10) + 6 InvView (AtmosphericScatterer.cs:72-89)."""
    inv_proj = inverted(perspective_fov(degrees_to_radians(90.0), 1.0, 0.1, 10.0))
    dirs_ups = [((1, 0, 0), (0, -1, 0)), ((-1, 0, 0), (0, -1, 0)), ((0, 1, 0), (0, 0, 1)),
                ((0, -1, 0), (0, 0, -1)), ((0, 0, 1), (0, -1, 0)), ((0, 0, -1), (0, -1, 0))]
    blob = np.zeros(16 * 7 + 4, dtype=F)
    blob[0:16] = inv_proj.reshape(-1)
    zero = np.zeros(3, dtype=F)
    for i, (d, u) in enumerate(dirs_ups):
        v = look_at(zero, zero + np.asarray(d, dtype=F), np.asarray(u, dtype=F))  # Camera.GenerateMatrix
        blob[16 + 16 * i:32 + 16 * i] = inverted(v).reshape(-1)
    return blob.tobytes()


def atmosphere_light_pos(time=0.5) -> np.ndarray:
    """AtmosphericScatterer.cs:41 — (0, sin(2*pi*t), cos(2*pi*t)) * 149600000e3f, float math."""
    a = degrees_to_radians(F(time) * F(360.0))
    return (np.array([0.0, F(np.sin(a)), F(np.cos(a))], dtype=F) * F(149600000e3)).astype(F)
