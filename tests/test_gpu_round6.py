"""GPU tests added in round 6.
  * The reference's REAL frame loop: MainWindow.OnUpdateFrame re-uploads InvView and ViewPos on every focused update, moved or not
    (/root/reference/OpenTK-PathTracer/src/MainWindow.cs:131-132), between every pair of PathTracer.Render() calls (:40-69).  Unchanged
    bytes must be no input change for the library: frames keep pipelining (launches per step < 1), the cached tile masks become valid;
    a changed byte still flushes, invalidates, and the image stays the oracle's bit for bit across a camera move mid-run.
  * The same rule for pt_upload_game_objects (Gui.cs:212-216 re-uploads the picked object on every slider event) and pt_set_params
    (PathTracer.cs:11-83: a setter per GUI touch).
Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest

import configs
from test_gpu_abi_round2 import make_tracer
from test_gpu_parity import assert_bit_exact

pytestmark = pytest.mark.gpu


def _reupload_camera(pt, basic):
    """OnUpdateFrame's two SubData calls (MainWindow.cs:131-132)."""
    pt.BasicDataUBO.SubData(64, 64, basic[64:128])
    pt.BasicDataUBO.SubData(128, 16, basic[128:144])


def test_reference_frame_loop_pipelines_and_gets_cached_masks(pkg, native_lib):
    """1,000 iterations of {re-upload InvView + ViewPos (unchanged), Render()} at 1080p with the automatic batch size: the uploads must
    neither flush nor invalidate — far fewer launches than frames, cached tile masks valid, no input-change flush counted."""
    w = configs.Workload("refloop", "default", 1920, 1080, 8, "sky_f32_32")
    _, basic, _, _, _ = configs.inputs(w)
    pt = make_tracer(pkg, w)
    pt.SetFrameBatch(0)
    for _ in range(8):  # (an idle GPU launches its first frames at once)
        _reupload_camera(pt, basic)
        pt.Render()
    pt.Synchronize()
    s0 = pkg.native.debug_launch_stats(pt._h)
    n = 1000
    for _ in range(n):
        _reupload_camera(pt, basic)
        pt.Render()
    pt.Synchronize()
    s1 = pkg.native.debug_launch_stats(pt._h)
    img = pt.Result
    pt.Dispose()
    launches = s1["launches"] - s0["launches"]
    print(f"reference loop: {n} frames in {launches} launches ({launches / n:.4f} per step), masks valid {s1['tile_masks_valid']}, "
          f"mask builds {s1['mask_builds']}, input-change flushes {s1['input_change_flushes'] - s0['input_change_flushes']}")
    assert s1["input_change_flushes"] == s0["input_change_flushes"], "an unchanged upload was treated as an input change"
    assert launches < n / 8, f"{launches} launches for {n} frames: the redundant uploads broke the pipelining"
    assert s1["tile_masks_valid"] and s1["mask_builds"] >= 1
    assert np.isfinite(img).all() and (img[..., 3] == 1).all()


def test_camera_move_mid_run_flushes_and_stays_bit_exact(pkg, native_lib, oracle):
    """Frames 0..4 with camera A (each preceded by the redundant re-upload), then camera B's bytes arrive between two Render() calls
    WITHOUT a reset (legal over the C ABI: the running mean simply continues), frames 5..9 with camera B re-uploaded every frame.
    The pending frames of camera A must have been rendered with camera A: image == oracle's two-stage accumulation, bit for bit."""
    w = configs.Workload("cammove", "default", 200, 117, 8, "sky_f32_32")
    sc, basic_a, objs, env, kw = configs.inputs(w)
    cam_b = pkg.camera.Camera(position=(-15.0, 4.0, -7.5), look_x=-40.0, look_y=-3.0)
    basic_b = pkg.camera.basic_data_ubo(cam_b, w.width, w.height)
    assert bytes(basic_a) != bytes(basic_b)
    pt = make_tracer(pkg, w)
    for _ in range(5):
        _reupload_camera(pt, basic_a)
        pt.Render()
    s0 = pkg.native.debug_launch_stats(pt._h)
    for _ in range(5):
        _reupload_camera(pt, basic_b)
        pt.Render()
    s1 = pkg.native.debug_launch_stats(pt._h)
    got = pt.Result
    pt.Dispose()
    assert s1["input_change_flushes"] > s0["input_change_flushes"], "changed camera bytes did not count as an input change"
    want = oracle.render(w.width, w.height, basic_a, objs, env, num_frames=5, **kw)
    want = oracle.render(w.width, w.height, basic_b, objs, env, frame_start=5, num_frames=5, image=want, **kw)
    assert_bit_exact(got, want, "camera moved between frames 4 and 5 (no reset)")


def test_changed_camera_invalidates_cached_masks(pkg, native_lib, oracle):
    """Masks cached for camera A must not survive camera B: after the move the image still equals the oracle (a stale mask would cull
    objects camera B sees), and the masks are rebuilt once B has been left alone."""
    w = configs.Workload("maskmove", "default", 1920, 1080, 4, "sky_f32_32")
    sc, basic_a, objs, env, kw = configs.inputs(w)
    cam_b = pkg.camera.Camera(position=(5.0, 2.0, 6.0), look_x=140.0, look_y=-5.0)
    basic_b = pkg.camera.basic_data_ubo(cam_b, w.width, w.height)
    pt = make_tracer(pkg, w)
    pt.SetFrameBatch(4)
    for _ in range(40):
        _reupload_camera(pt, basic_a)
        pt.Render()
    pt.Synchronize()
    assert pkg.native.debug_launch_stats(pt._h)["tile_masks_valid"]
    pt.UploadBasicData(basic_b)
    assert not pkg.native.debug_launch_stats(pt._h)["tile_masks_valid"]
    pt.ResetRenderer()  # MainWindow.cs:128-129: a moved camera restarts the accumulation
    for _ in range(2):
        _reupload_camera(pt, basic_b)
        pt.Render()
    got = pt.Result
    rows = slice(400, 432)  # (the oracle on 32 rows of the 1080p frame: seconds)
    want = oracle.render(w.width, w.height, basic_b, objs, env, num_frames=2, y0=rows.start, rows=32, **kw)
    assert_bit_exact(got[rows], want, "two frames after a camera move at 1080p (rows 400..431)")
    for _ in range(40):
        _reupload_camera(pt, basic_b)
        pt.Render()
    pt.Synchronize()
    st = pkg.native.debug_launch_stats(pt._h)
    pt.Dispose()
    assert st["tile_masks_valid"] and st["mask_builds"] >= 2


def test_unchanged_object_and_param_uploads_are_no_input_change(pkg, native_lib, oracle):
    """Gui.cs:212-216 re-uploads the picked object on every slider event, PathTracer.cs:11-83 pushes a uniform per setter call: the same
    bytes / values again neither join nor flush; different ones do, and the image follows the oracle."""
    w = configs.Workload("objs", "default", 160, 96, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = make_tracer(pkg, w)
    base = pkg.native.debug_launch_stats(pt._h)["input_change_flushes"]  # (the scene's first upload: every object is a change)
    objs_np = np.frombuffer(bytes(objs), dtype=np.uint8)
    for _ in range(4):
        pt.GameObjectsUBO.SubData(80 * 3, 80, objs_np[80 * 3:80 * 4])          # sphere 3, unchanged
        pt.GameObjectsUBO.SubData(20480 + 96 * 2, 96, objs_np[20480 + 96 * 2:20480 + 96 * 3])  # cuboid 2, unchanged
        pt.RayDepth = w.ray_depth                                               # same value again
        pt.Render()
    s0 = pkg.native.debug_launch_stats(pt._h)
    assert s0["input_change_flushes"] == base
    # now really edit sphere 3 (albedo) between frames 3 and 4
    edited = objs_np.copy()
    edited[80 * 3 + 16:80 * 3 + 28].view(np.float32)[:] = (0.9, 0.1, 0.1)
    pt.GameObjectsUBO.SubData(80 * 3, 80, edited[80 * 3:80 * 4])
    for _ in range(3):
        pt.Render()
    s1 = pkg.native.debug_launch_stats(pt._h)
    got = pt.Result
    pt.Dispose()
    assert s1["input_change_flushes"] == base + 1
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=4, **kw)
    want = oracle.render(w.width, w.height, basic, edited.tobytes(), env, frame_start=4, num_frames=3, image=want, **kw)
    assert_bit_exact(got, want, "object edit between frames 3 and 4, redundant uploads around it")


def test_multi_gpu_check_script_dry_run(pkg, native_lib, tmp_path):
    """tools/multi_gpu_check.sh is the one-command acceptance run for an N-GPU box, and no such box has existed in any round: its JSON
    checks are kept alive by running them on bench.py --gpus 2 --share-gpu (two gloo ranks on this GPU)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "multi")
    p = subprocess.run(["bash", os.path.join(root, "tools", "multi_gpu_check.sh"), "--dry-run", out], capture_output=True, text=True, timeout=1200)
    print(p.stdout[-1500:], p.stderr[-1500:])
    assert p.returncode == 0, "dry run of tools/multi_gpu_check.sh failed"
    rep = json.load(open(os.path.join(out, "multi_gpu_check.json")))
    assert rep["ok"] and rep["runs"] and rep["runs"][0]["gpus"] == 2 and not rep["runs"][0]["errors"]


def test_group_handle_reports_its_gather_path_and_defaults_to_tile_row_bands(pkg, native_lib, oracle):
    """pt_multi_gather_is_direct (a same-device group is always direct), and the default partition since round 6: 8-row bands — of 1080p's
    135 tile rows 8 parts own 16 or 17 (136 image rows at most; 16-row bands left the largest share 144)."""
    w = configs.Workload("g8", "default", 1920, 1080, 8, "sky_f32_32")
    pt = make_tracer(pkg, w, devices=[0] * 8)
    assert pt.GatherIsDirect
    rows = []
    for part in range(8):
        from opentk_pathtracer_amd.distributed import interleaved_rows
        rows.append(len(interleaved_rows(1080, part, 8, 8)))
    assert max(rows) == 136 and sum(rows) == 1080
    pt.Render()
    got = pt.Result
    pt.Dispose()
    sc, basic, objs, env, kw = configs.inputs(w)
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=1, y0=128, rows=24, **kw)
    assert_bit_exact(got[128:152], want, "default-partition group handle, rows 128..151 of 1080p")


# ------------------------------------------------------------------------------------------------ frame-fed launches
class _Tune:
    """Set library knobs for one test (they are process-global) and restore the defaults afterwards."""
    DEFAULTS = {"feed": 1, "feed_min_tiles": 12000, "feed_idle_us": 150, "feed_display": 0}

    def __init__(self, pkg, **knobs):
        self.pkg, self.knobs = pkg, knobs

    def __enter__(self):
        for k, v in self.knobs.items():
            self.pkg.native.debug_set(k, v)

    def __exit__(self, *exc):
        for k in self.knobs:
            self.pkg.native.debug_set(k, self.DEFAULTS[k])


def _open_feed(pt, frames_first=3):
    """Bring the handle into the state in which single frames go out as tagged launches (the host has pipelined frames before), so that
    the next Render() of a batch-1 host opens a frame-fed launch."""
    from opentk_pathtracer_amd import native
    assert frames_first == 3
    pt.Render()
    pt.Synchronize()
    # two frames issued back to back while the GPU is kept busy by nothing: issue them as ONE pending pair through a frame-batch limit
    pt.SetFrameBatch(2)
    for attempt in range(50):
        pt.ResetRenderer()
        pt.Render()  # frame 0 again (a reset only rewinds the counter: frame 0 weights the old contents by 0)
        pt.Render()
        pt.Render()
        pt.Synchronize()
        if native.debug_launch_stats(pt._h)["saw_batch"]:
            break
    assert native.debug_launch_stats(pt._h)["saw_batch"], "could not get two frames into one launch"
    pt.SetFrameBatch(1)


@pytest.mark.parametrize("size,frames", [((200, 117), 12), ((64, 40), 150), ((8, 8), 70), ((333, 211), 40)], ids=lambda v: str(v))
def test_frame_fed_launch_is_bit_exact(pkg, native_lib, oracle, size, frames):
    """Render() of a batch-1 host: the first call opens a frame-fed launch (room for 64 frames), the calls that follow PUBLISH their frame
    into it — the wavefronts stay resident between frames; past 64 frames the next fed launch chains on it.  The image equals the
    oracle's frame-by-frame accumulation bit for bit (small images: consecutive frames of a tile meet in one wavefront all the time)."""
    w = configs.Workload("fed", "default", size[0], size[1], 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    with _Tune(pkg, feed_min_tiles=0, feed_idle_us=20000):
        pt = make_tracer(pkg, w)
        _open_feed(pt)
        for _ in range(frames):
            pt.Render()
        st = pkg.native.debug_launch_stats(pt._h)
        got = pt.Result
        pt.Dispose()
    print(f"{size}: {frames} frames: {st['feed_opens']} fed launch(es), {st['published']} frames published, {st['launches']} launches in all")
    assert st["feed_opens"] >= 1 and st["published"] >= min(frames, 64) - 4, "the frames did not go through a frame-fed launch"
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=3 + frames, **kw)
    assert_bit_exact(got, want, f"{frames} frames through frame-fed launches at {size}")
    assert pkg.native.debug_handover_stats  # (kept for symmetry with the hand-over tests)


@pytest.mark.parametrize("size", [(24, 16), (40, 24), (136, 72)], ids=lambda v: str(v))
def test_frame_fed_launches_that_close_inside_a_ticket_keep_the_ticket_count(pkg, native_lib, oracle, size):
    """Tickets hand out tiles in chunks (8, or 4 for small batched launches) and a fed launch's frames end wherever their tiles end: an
    image whose tiles per frame are no multiple of the chunk (6, 15, 153 here; 1440 x 900 has 20,340) leaves the workgroup that holds the
    straddling ticket with an unpublished REST when the launch closes.  That rest is not the workgroup's failing ticket — the ticket
    counts as handed out in the host's accounting — so the workgroup still draws one.  (It did not, up to round 6: every such close left
    the device's counter one short of the host's base, and after as many closes as a launch of a tiny image has workgroups, launches drew
    nothing but failing tickets and rendered nothing.  Found by tools/handover_stress --tune feed_min_tiles=0.)  60 launches that close
    after 1, 3 or 5 frames, then a frame through a classic launch: bit-exact, no repair pass needed."""
    w = configs.Workload("fedtickets", "default", size[0], size[1], 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    with _Tune(pkg, feed_min_tiles=0, feed_idle_us=200000):
        pt = make_tracer(pkg, w)
        _open_feed(pt)
        total = 3
        for i in range(60):
            for _ in range((1, 3, 5)[i % 3]):
                pt.Render()
            total += (1, 3, 5)[i % 3]
            pt.Synchronize()  # (a join closes the open launch)
        st = pkg.native.debug_launch_stats(pt._h)
        pt.SetFrameBatch(64)
        pt.Render()
        total += 1
        got = pt.Result
        ho = pkg.native.debug_handover_stats(pt._h)
        pt.Dispose()
    print(f"{size}: {st['feed_opens']} fed launches closed by joins, {st['published']} frames published; repair: {ho}")
    assert st["feed_opens"] >= 40, "the frames did not go through frame-fed launches"
    assert ho["pairs_repaired"] == 0 and ho["inconsistent"] == 0, "a launch had to be repaired"
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=total, **kw)
    assert_bit_exact(got, want, f"{total} frames through 60 frame-fed launches at {size}")


def test_frame_fed_launch_ends_itself_when_the_host_stops_and_nothing_is_lost(pkg, native_lib, oracle):
    """A host that stops rendering must not keep the GPU: the launch's wavefronts wait feed_idle_us for the next frame, then the launch
    abandons itself (reason "idle"); the host's next call repairs whatever a racing publish left undone and launches anew.  Frames
    rendered before and after the pauses: bit-exact."""
    import time
    w = configs.Workload("fedidle", "default", 160, 96, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    with _Tune(pkg, feed_min_tiles=0, feed_idle_us=100):
        pt = make_tracer(pkg, w)
        _open_feed(pt)
        total = 3
        for burst in (5, 1, 9, 2):
            for _ in range(burst):
                pt.Render()
            total += burst
            time.sleep(0.01)  # 10 ms >> 100 us: the open launch ends idle
        st = pkg.native.debug_launch_stats(pt._h)
        got = pt.Result
        ho = pkg.native.debug_handover_stats(pt._h)
        pt.Dispose()
    print(f"idle: {st['feed_opens']} fed launches, {st['feed_idle']} ended idle, {st['published']} published; repair: {ho}")
    assert st["feed_idle"] >= 1, "no fed launch ended idle although the host paused for 10 ms"
    assert ho["inconsistent"] == 0
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=total, **kw)
    assert_bit_exact(got, want, "frame-fed launches with host pauses")


def test_frame_fed_present_every_frame_shows_the_right_frames(pkg, native_lib, oracle):
    """The reference's loop (MainWindow.cs:40-69) over a frame-fed launch with the FUSED DISPLAY: Render(); PresentAsync(f % 2) into a slot
    bound to device memory (the interop-style present); the slot of two frames ago is waited for.  The frame shown is tone-mapped by the
    NEXT frame's tile passes (they read every pixel of it anyway), a one-wavefront gate tells when its image is complete.  Every
    presented image == oracle's post-process of the oracle's accumulation after exactly that many frames; a last frame without a
    successor is tone-mapped the classic way when it is waited for.  (The fused display is OFF by default — knob feed_display: with a host
    that runs at most two frames ahead it measures slower than the per-frame launches of round 3, DESIGN.md section 3.1 — this test keeps
    the mechanism exact.)"""
    torch = pytest.importorskip("torch")
    # (small image only: the test reads the bound images back with torch between frames, and at 1080p the runtime's copy kernel finds
    # no room beside a resident launch at six workgroups per CU until that launch ends — the reason the mode is not the default)
    for size in ((224, 126),):
        w = configs.Workload("fedpresent", "default", size[0], size[1], 4, "sky_f32_32")
        sc, basic, objs, env, kw = configs.inputs(w)
        frames = 40 if size[0] < 1000 else 14
        acc = oracle.render(w.width, w.height, basic, objs, env, num_frames=frames, dump_each=True, **kw)
        with _Tune(pkg, feed_min_tiles=0, feed_idle_us=20000, feed_display=1):
            pt = make_tracer(pkg, w)
            bufs = [torch.zeros((w.height, w.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            for s_, b_ in enumerate(bufs):
                pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
            shown = {}
            for f in range(frames):
                pt.Render()
                if f >= 2:
                    _, idx = pt.PresentWait(f % 2)
                    shown[idx] = bufs[f % 2].cpu().numpy().copy()
                pt.PresentAsync(f % 2)
            for s_ in ((frames - 2) % 2, (frames - 1) % 2):
                _, idx = pt.PresentWait(s_)
                shown[idx] = bufs[s_].cpu().numpy().copy()
            st = pkg.native.debug_launch_stats(pt._h)
            final = pt.Result
            for s_ in range(2):
                pt.BindPresentImage(s_, None)
            pt.Dispose()
        print(f"{size}: present loop: {st['feed_opens']} fed launches, {st['published']} frames published, {st['feed_idle']} ended idle")
        assert sorted(shown) == list(range(1, frames + 1))
        for idx, img in shown.items():
            assert np.array_equal(img, oracle.postprocess(acc[idx - 1])[1]), f"{size}: presented frame {idx}"
        assert_bit_exact(final, acc[-1], "accumulation image after the present loop")
        assert st["published"] >= frames // 2, "the present loop did not go through a frame-fed launch"


def test_frame_fed_launch_closes_on_input_changes_and_observations(pkg, native_lib, oracle):
    """Everything that changes an input, or lets the host observe the image, closes the open fed launch first: a camera move mid-run, a
    read, a reset — frames before the change keep the old inputs (bit-exact), alpha is 1 whenever the host looks."""
    w = configs.Workload("fedclose", "default", 200, 117, 6, "sky_f32_32")
    sc, basic_a, objs, env, kw = configs.inputs(w)
    cam_b = pkg.camera.Camera(position=(-15.0, 4.0, -7.5), look_x=-40.0, look_y=-3.0)
    basic_b = pkg.camera.basic_data_ubo(cam_b, w.width, w.height)
    with _Tune(pkg, feed_min_tiles=0, feed_idle_us=20000):
        pt = make_tracer(pkg, w)
        _open_feed(pt)
        for _ in range(6):
            pt.Render()
        assert pkg.native.debug_launch_stats(pt._h)["feed_open"]
        mid = pt.Result                       # a read closes the launch, joins, restores alpha = 1
        assert not pkg.native.debug_launch_stats(pt._h)["feed_open"] and (mid[..., 3] == 1).all()
        for _ in range(4):
            pt.Render()
        pt.UploadBasicData(basic_b)           # new camera bytes: frames 9..12 were rendered with camera A
        for _ in range(5):
            pt.Render()
        got = pt.Result
        pt.ResetRenderer()
        for _ in range(3):
            pt.Render()
        again = pt.Result
        pt.Dispose()
    want_mid = oracle.render(w.width, w.height, basic_a, objs, env, num_frames=9, **kw)
    assert_bit_exact(mid, want_mid, "read in the middle of a fed launch")
    want = oracle.render(w.width, w.height, basic_a, objs, env, frame_start=9, num_frames=4, image=want_mid.copy(), **kw)
    want = oracle.render(w.width, w.height, basic_b, objs, env, frame_start=13, num_frames=5, image=want, **kw)
    assert_bit_exact(got, want, "camera move between fed frames")
    # (after a reset frame 0 weights the old contents by 0, PathTracer.cs:139)
    want2 = oracle.render(w.width, w.height, basic_b, objs, env, num_frames=3, image=want.copy(), **kw)
    assert_bit_exact(again, want2, "reset after fed launches")
