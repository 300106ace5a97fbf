"""Software rasteriser for one env's state (the role of CarEnv.__render_frame, car_env.py:766-807, and
of log_video's frames, train.py:23-50, without pygame / cv2): walls, reward gates (the next one highlighted), the car as
a heading arrow and its rays out to the distances the observation reports.  numpy only; PNG written with zlib."""
import struct
import zlib

import numpy as np

WIDTH, HEIGHT = 1280, 720
COLORS = {"background": (24, 24, 28), "wall": (235, 235, 235), "gate": (60, 110, 60), "next_gate": (90, 230, 90),
          "ray": (70, 110, 200), "car": (240, 80, 60)}


def _line(img, x0, y0, x1, y1, color):
    h, w, _ = img.shape
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.rint(np.linspace(x0, x1, n)).astype(np.int64)
    ys = np.rint(np.linspace(y0, y1, n)).astype(np.int64)
    ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
    img[ys[ok], xs[ok]] = color


def rasterise(walls, gates, px, py, rot_deg, ray_obs, next_gate=0, num_rays_nominal=None, size=(640, 360)):
    """walls / gates: [n, 4] segments in the 1280 x 720 frame (Track.geometry()); ray_obs: the observation's ray part
    (distance / 1000, car_env.py:593); rot_deg: heading in degrees.  Returns uint8 [H, W, 3]."""
    w, h = size
    sx, sy = w / WIDTH, h / HEIGHT
    img = np.empty((h, w, 3), np.uint8)
    img[:] = COLORS["background"]
    for i, g in enumerate(np.asarray(gates)):
        _line(img, g[0] * sx, g[1] * sy, g[2] * sx, g[3] * sy, COLORS["next_gate" if i == next_gate else "gate"])
    R = len(ray_obs)
    n = num_rays_nominal or R
    step = 360 // n                                            # car_env.py:269
    for i, d in enumerate(np.asarray(ray_obs, np.float64) * 1000.0):
        a = np.radians(rot_deg + i * step)
        _line(img, px * sx, py * sy, (px + d * np.cos(a)) * sx, (py + d * np.sin(a)) * sy, COLORS["ray"])
    for s in np.asarray(walls):
        _line(img, s[0] * sx, s[1] * sy, s[2] * sx, s[3] * sy, COLORS["wall"])
    a = np.radians(rot_deg)
    c, s_ = np.cos(a), np.sin(a)
    nose = (px + 14 * c, py + 14 * s_)
    for lx, ly in ((-8, -6), (-8, 6)):                         # a small arrow: two tail points to the nose, and the base
        _line(img, (px + lx * c - ly * s_) * sx, (py + lx * s_ + ly * c) * sy, nose[0] * sx, nose[1] * sy, COLORS["car"])
    _line(img, (px - 8 * c + 6 * s_) * sx, (py - 8 * s_ - 6 * c) * sy, (px - 8 * c - 6 * s_) * sx, (py - 8 * s_ + 6 * c) * sy,
          COLORS["car"])
    return img


def write_png(path, img):
    """Minimal PNG encoder (8-bit RGB, one IDAT)."""
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
