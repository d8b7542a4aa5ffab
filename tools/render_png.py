"""Render the reference's default scene on the GPU and write a tone-mapped PNG (ACES + gamma through pt_present_rgba8).
    python tools/render_png.py out.png [W H frames]
The reference flips vertically when saving (Framebuffer.cs:79: GL rows are bottom-up); so does this."""
import os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
out = sys.argv[1]
W, H, frames = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (960, 540, 256)
cam = pkg.camera.Camera()
pt = pkg.PathTracer(None, W, H, 13, 1, 20.0, 0.14)          # MainWindow.cs:189 defaults
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(pkg.scene.default_scene())
pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
for _ in range(frames):
    pt.Render()
img = pt.Present()[::-1, :, :3]
raw = b"".join(b"\x00" + img[y].tobytes() for y in range(H))
chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
open(out, "wb").write(png)
print(f"wrote {out}: {W}x{H}, {pt.Samples} samples/pixel")
