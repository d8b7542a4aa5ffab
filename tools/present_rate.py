"""Interactive-mode rates: what the reference's loop does when it shows every frame (MainWindow.cs:43-56: Render() ->
PostProcesser.Render(Result) -> blit -> SwapBuffers).
  blocking:      Render() + pt_read_result (RGBA32F) / pt_present_rgba8 (tone-mapped RGBA8) per displayed frame — nothing
                 overlaps, the host waits for render + copy;
  non-blocking:  Render(); pt_present_rgba8_async(f % 2); pt_present_wait((f - 1) % 2) — frame f+1 renders while frame f's
                 8.3 MB cross PCIe into the library's pinned slot.
Usage: python tools/present_rate.py [--devices 0,0] [--json gpurun_out/present_rate.json]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--devices", default=None, help="comma-separated HIP device ids: use a group handle (pt_create_multi)")
ap.add_argument("--json", default=None)
ap.add_argument("--frames", type=int, default=400)
a = ap.parse_args()
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
extra = {"devices": [int(x) for x in a.devices.split(",")]} if a.devices else {}
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14, **extra)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
t = time.perf_counter()
while time.perf_counter() - t < 0.15:   # clock warm-up
    for _ in range(64): pt.Render()
    pt.Synchronize()
results = {}
def rate(name, fn, n=a.frames, drain=None):
    for i in range(8): fn(i)
    if drain: drain()
    pt.Synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    if drain: drain()
    ms = (time.perf_counter() - t) * 1e3 / n
    results[name] = {"ms_per_displayed_frame": round(ms, 4), "msamples_per_s": round(W * H / ms / 1e3, 1)}
    print(f"{name:64s} {ms:7.3f} ms per displayed frame  {W * H / ms / 1e3:8.1f} Msamples/s", flush=True)
pageable_f, pageable_b = np.empty((H, W, 4), np.float32), np.empty((H, W, 4), np.uint8)
def blocking(read, dst):
    def f(i):
        pt.Render(); read(dst)
    return f
rate("render + pt_read_result (RGBA32F, blocking)", blocking(pt.ReadInto, pageable_f), n=100)
rate("render + pt_present_rgba8 (ACES + gamma, RGBA8, blocking)", blocking(pt.PresentInto, pageable_b), n=100)
checksum = [0]
def nonblocking(i):
    pt.Render()
    pt.PresentAsync(i & 1)
    if i > 0:
        img, idx = pt.PresentWait((i - 1) & 1)
        checksum[0] += int(img[0, 0, 0])       # touch the image like a consumer would
def drain_nb():
    pt.PresentWait((a.frames - 1) & 1)
rate("render + async present, two slots (wait for the other slot's frame)", nonblocking, drain=lambda: None)
seen = [False, False]
def nonblocking_reuse(i):
    # the recommended double-buffered order: wait for a slot only when it is about to be reused (it holds frame i - 2:
    # long landed), so the host never waits for a copy that is still in flight and stays two frames ahead of the display
    pt.Render()
    if seen[i & 1]:
        img, idx = pt.PresentWait(i & 1)
        checksum[0] += int(img[0, 0, 0])
    pt.PresentAsync(i & 1)
    seen[i & 1] = True
rate("render + async present, two slots (wait for a slot before reusing it)", nonblocking_reuse)
def nonblocking3(i):
    pt.Render()
    pt.PresentAsync(i % 3)
    if i > 1:
        pt.PresentWait((i - 2) % 3)
rate("render + async present, three slots (two frames in flight)", nonblocking3)
def every4(i):
    for _ in range(4): pt.Render()
    pt.PresentAsync(i & 1)
    if i > 0: pt.PresentWait((i - 1) & 1)
rate("4 x render + async present (a display slower than the renderer)", every4, n=a.frames // 4)
results["4 x render + async present (a display slower than the renderer)"]["ms_per_rendered_frame"] = round(
    results["4 x render + async present (a display slower than the renderer)"]["ms_per_displayed_frame"] / 4, 4)
if not a.devices:
    # interop-style present: the slots tone-map into caller-owned DEVICE memory (pt_present_bind_device_image), nothing crosses PCIe —
    # what a host pays whose GL context lives on the same GPU and displays a HIP-registered buffer
    import torch
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s_, b_ in enumerate(bufs): pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
    seen2 = [False, False]
    def bound(i):
        pt.Render()
        if seen2[i & 1]: pt.PresentWait(i & 1)
        pt.PresentAsync(i & 1)
        seen2[i & 1] = True
    rate("render + async present into BOUND DEVICE images (no host copy; interop-style)", bound)
    for s_ in range(2):
        pt.PresentWait(s_); pt.BindPresentImage(s_, None)
if a.json:
    os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
    json.dump({"image": [W, H], "devices": extra.get("devices", [0]), "csrc_hash": pkg.native.csrc_hash(), "results": results}, open(a.json, "w"), indent=1)
