"""Host-side mirror of the reference's renderer classes over the C ABI (harness glue for tests and bench).

  PathTracer            /root/reference/OpenTK-PathTracer/src/Render/PathTracer.cs:9-141
  AtmosphericScatterer  /root/reference/OpenTK-PathTracer/src/Render/AtmosphericScatterer.cs:9-119
  BufferObject.SubData  /root/reference/OpenTK-PathTracer/src/Render/Objects/BufferObject.cs:37-48

Member names and argument meaning follow the C# classes (NumSpheres, RayDepth, SPP, Render(), SetSize(),
ResetRenderer(), Samples, Result ...), so the call sequence of the reference's MainWindow can be replayed as is.
Every method is a thin call into libmi355pt.so; there is no Python or CPU implementation of the integrator here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native
from .native import check


class EnvironmentMap:
    """A cube texture as the host would hand it to GL: 6 faces (+X,-X,+Y,-Y,+Z,-Z), RGBA32F or SRGB8_A8
    (MainWindow.cs:177-187)."""

    def __init__(self, faces: np.ndarray):
        faces = np.ascontiguousarray(faces)
        if faces.ndim != 4 or faces.shape[0] != 6 or faces.shape[1] != faces.shape[2] or faces.shape[3] != 4:
            raise ValueError("faces must have shape (6, S, S, 4)")
        if faces.dtype == np.float32:
            self.format = native.PT_ENV_RGBA32F
        elif faces.dtype == np.uint8:
            self.format = native.PT_ENV_SRGB8_A8
        else:
            raise ValueError("faces must be float32 (RGBA32F) or uint8 (SRGB8_A8)")
        self.faces = faces
        self.size = faces.shape[1]


class UniformBuffer:
    """BufferObject bound as UBO 0 (BasicDataUBO) or UBO 1 (GameObjectsUBO): SubData(offset, size, data)."""

    def __init__(self, tracer: "PathTracer", which: str):
        self._t, self._which = tracer, which

    def SubData(self, offset: int, size: int, data) -> None:
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if isinstance(data, (bytes, bytearray))
                                   else np.asarray(data))
        if buf.nbytes < size:
            raise ValueError("data shorter than size")
        fn = self._t._lib.pt_upload_basic_data if self._which == "basic" else self._t._lib.pt_upload_game_objects
        check(fn(self._t._h, offset, size, buf.ctypes.data_as(C.c_void_p)), self._t._h)


class PathTracer:
    def __init__(self, environmentMap, width: int, height: int, rayDepth: int, spp: int, focalLength: float,
                 apertureDiamater: float, device: int = 0, devices=None):
        """`devices` = list of HIP device ids: ONE renderer row-tiled over those GPUs inside the library
        (pt_create_multi; the gather over xGMI happens inside Result / Present).  Otherwise one GPU (`device`)."""
        self._lib = native.load()
        h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            check(self._lib.pt_create_multi(ids, len(devices), width, height, C.byref(h)))
        else:
            check(self._lib.pt_create(device, width, height, C.byref(h)))
        self._h = h
        self.devices = list(devices) if devices is not None else [device]
        self.band_rows, self.band_world, self.band_rank = 0, 1, 0
        self.Width, self.Height = width, height
        self._numSpheres = self._numCuboids = 0
        self._rayDepth, self._spp = rayDepth, spp
        self._focalLength, self._apertureDiameter = focalLength, apertureDiamater
        self._push_params()
        self._env = None
        self.BasicDataUBO = UniformBuffer(self, "basic")       # MainWindow.cs:195-197
        self.GameObjectsUBO = UniformBuffer(self, "objects")   # MainWindow.cs:199-201
        self.y0, self.rows = 0, height
        if environmentMap is not None:
            self.EnvironmentMap = environmentMap

    # -- the six uniform-setting properties, PathTracer.cs:11-83
    def _push_params(self):
        check(self._lib.pt_set_params(self._h, self._numSpheres, self._numCuboids, self._rayDepth, self._spp,
                                      self._focalLength, self._apertureDiameter), self._h)

    def _prop(name):  # noqa: N805
        def get(self):
            return getattr(self, name)

        def set_(self, v):
            setattr(self, name, v)
            self._push_params()
        return property(get, set_)

    NumSpheres = _prop("_numSpheres")
    NumCuboids = _prop("_numCuboids")
    RayDepth = _prop("_rayDepth")
    SPP = _prop("_spp")
    FocalLength = _prop("_focalLength")
    ApertureDiameter = _prop("_apertureDiameter")
    del _prop

    @property
    def EnvironmentMap(self):  # PathTracer.cs:85
        return self._env

    @EnvironmentMap.setter
    def EnvironmentMap(self, env):
        if isinstance(env, AtmosphericScatterer):
            env._select(self)
        else:
            if not isinstance(env, EnvironmentMap):
                env = EnvironmentMap(env)
            ptrs = (C.c_void_p * 6)(*[env.faces[f].ctypes.data_as(C.c_void_p) for f in range(6)])
            check(self._lib.pt_set_environment(self._h, env.size, env.format, ptrs), self._h)
        self._env = env

    @property
    def Samples(self) -> int:  # PathTracer.cs:112
        return self.FrameIndex * self._spp

    @property
    def FrameIndex(self) -> int:
        v = C.c_int()
        check(self._lib.pt_get_frame_index(self._h, C.byref(v)), self._h)
        return v.value

    def Render(self) -> int:  # PathTracer.cs:114-129
        total = C.c_int()
        check(self._lib.pt_render(self._h, C.byref(total)), self._h)
        return total.value

    def SetSize(self, width: int, height: int) -> None:  # PathTracer.cs:131-135
        check(self._lib.pt_set_size(self._h, width, height), self._h)
        self.Width, self.Height = width, height
        self.y0, self.rows = 0, height
        self.band_rows, self.band_world, self.band_rank = 0, 1, 0

    def ResetRenderer(self) -> None:  # PathTracer.cs:137-140
        check(self._lib.pt_reset(self._h), self._h)

    # -- multi-GPU tiling + plumbing (no reference counterpart)
    def SetTile(self, y0: int, rows: int) -> None:
        check(self._lib.pt_set_tile(self._h, y0, rows), self._h)
        self.y0, self.rows = y0, rows
        self.band_rows, self.band_world, self.band_rank = 0, 1, 0

    def SetInterleavedTile(self, rank: int, world: int, band_rows: int) -> None:
        check(self._lib.pt_set_interleaved_tile(self._h, rank, world, band_rows), self._h)
        from .distributed import interleaved_rows
        self.y0, self.rows = 0, len(interleaved_rows(self.Height, rank, world, band_rows))
        self.band_rows, self.band_world, self.band_rank = band_rows, world, rank

    # -- persistence (the reference only has the screenshot button, Gui.cs:28-33)
    def SaveCheckpoint(self, path) -> None:
        from . import checkpoint
        checkpoint.save_checkpoint(path, self)

    def LoadCheckpoint(self, path, strict: bool = True) -> dict:
        from . import checkpoint
        return checkpoint.load_checkpoint(path, self, strict)

    def SaveScreenshot(self, path) -> None:  # Gui.cs:28-33 -> Framebuffer.cs:67-82
        from . import checkpoint
        checkpoint.save_screenshot(path, self)

    @property
    def Result(self) -> np.ndarray:
        """The RGBA32F `Result` image of this tile, read back to the host: (rows, Width, 4), row 0 = image row y0."""
        out = np.empty((self.rows, self.Width, 4), dtype=np.float32)
        check(self._lib.pt_read_result(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), 0), self._h)
        return out

    def ReadInto(self, out: np.ndarray) -> np.ndarray:
        assert out.dtype == np.float32 and out.shape == (self.rows, self.Width, 4) and out.flags.c_contiguous
        check(self._lib.pt_read_result(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), 0), self._h)
        return out

    def PresentInto(self, out: np.ndarray) -> np.ndarray:
        assert out.dtype == np.uint8 and out.shape == (self.rows, self.Width, 4) and out.flags.c_contiguous
        check(self._lib.pt_present_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), 0), self._h)
        return out

    def Present(self) -> np.ndarray:
        """ScreenEffect.Render(PathTracer.Result) (ScreenEffect.cs:29-37, PostProcessing/fragment.glsl): the tone-mapped
        RGBA8 image of this tile, (rows, Width, 4) uint8, row 0 = image row y0."""
        out = np.empty((self.rows, self.Width, 4), dtype=np.uint8)
        check(self._lib.pt_present_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), 0), self._h)
        return out

    def PresentAsync(self, slot: int) -> None:
        """Non-blocking present (pt_present_rgba8_async): tone map + device-to-host copy of the image as it is now into
        the library's pinned slot; Render() calls that follow overlap the copy."""
        check(self._lib.pt_present_rgba8_async(self._h, slot), self._h)

    def PresentWait(self, slot: int):
        """-> (image, frame_index): (rows, Width, 4) uint8 VIEW of the slot's pinned host image (valid until the next
        PresentAsync on that slot) and the number of frames it shows."""
        st = self.__dict__.setdefault("_present_state", {})
        if "args" not in st:
            st["args"] = (C.POINTER(C.c_uint8)(), C.c_size_t(), C.c_int())
            st["views"] = {}
        ptr, pitch, frame = st["args"]
        rc = self._lib.pt_present_wait(self._h, slot, C.byref(ptr), C.byref(pitch), C.byref(frame))
        if rc:
            check(rc, self._h)
        if not ptr:  # a slot bound to caller-owned device memory (BindPresentImage): the image is there, nothing came to the host
            return None, frame.value
        key = (slot, C.addressof(ptr.contents), self.rows, self.Width)
        img = st["views"].get(key)
        if img is None:  # one numpy view per pinned slot image (building it costs more than the call itself)
            assert pitch.value == self.Width * 4
            img = st["views"][key] = np.ctypeslib.as_array(ptr, shape=(self.rows, self.Width, 4))
        return img, frame.value

    def BindPresentImage(self, slot: int, device_ptr, nbytes: int = 0) -> None:
        """pt_present_bind_device_image: present slot `slot` tone-maps into caller-owned device memory (interop-style); None restores."""
        check(self._lib.pt_present_bind_device_image(self._h, slot, C.c_void_p(device_ptr) if device_ptr else None, nbytes), self._h)

    def SetPartition(self, band_rows: int) -> None:
        """Group handles: block-cyclic bands of `band_rows` image rows per device (0 = contiguous row blocks)."""
        check(self._lib.pt_multi_set_partition(self._h, band_rows), self._h)

    @property
    def GatherIsDirect(self) -> bool:
        """Group handles: True when every gather copy goes over a direct peer link (pt_multi_gather_is_direct); a staged group must not
        be the source of a scaling number."""
        v = C.c_int()
        check(self._lib.pt_multi_gather_is_direct(self._h, C.byref(v)), self._h)
        return bool(v.value)

    def PostProcessDevice(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._lib.pt_postprocess_device(self._h, C.byref(p), C.byref(n)), self._h)
        return p.value, n.value

    def WriteResult(self, image: np.ndarray, frame_index: int) -> None:
        img = np.ascontiguousarray(image, dtype=np.float32)
        assert img.shape == (self.rows, self.Width, 4)
        check(self._lib.pt_write_result(self._h, img.ctypes.data_as(C.POINTER(C.c_float)), 0, frame_index), self._h)

    def Synchronize(self) -> None:
        check(self._lib.pt_synchronize(self._h), self._h)

    def ResultDevicePtr(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._lib.pt_result_device_ptr(self._h, C.byref(p), C.byref(n)), self._h)
        return p.value, n.value

    def BindResultBuffer(self, device_ptr: int | None, nbytes: int = 0) -> None:
        check(self._lib.pt_bind_result_buffer(self._h, C.c_void_p(device_ptr), nbytes), self._h)

    def SetStream(self, hip_stream: int | None) -> None:
        check(self._lib.pt_set_stream(self._h, C.c_void_p(hip_stream)), self._h)

    def SetVariant(self, variant: int) -> None:
        check(self._lib.pt_set_variant(self._h, variant), self._h)

    def SetFrameBatch(self, max_frames: int) -> None:
        """Largest number of consecutive Render() calls one launch pipelines (1 = launch every frame at once)."""
        check(self._lib.pt_set_frame_batch(self._h, max_frames), self._h)

    def TimerBegin(self) -> None:
        check(self._lib.pt_timer_begin(self._h), self._h)

    def TimerEnd(self) -> float:
        ms = C.c_float()
        check(self._lib.pt_timer_end(self._h, C.byref(ms)), self._h)
        return ms.value

    def ReadEnvironment(self) -> np.ndarray:
        s = C.c_int()
        check(self._lib.pt_read_environment(self._h, None, C.byref(s)), self._h)
        out = np.empty((6, s.value, s.value, 4), dtype=np.float32)
        check(self._lib.pt_read_environment(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(s)), self._h)
        return out

    # -- convenience: replay the host's upload sequence for a whole scene + camera
    def UploadScene(self, scene) -> None:
        """LoadScene()'s upload loop (MainWindow.cs:265-266): one SubData per object, then the counts."""
        for obj in scene.objects():
            d = obj.gpu_data()
            self.GameObjectsUBO.SubData(obj.buffer_offset, d.nbytes, d)
        self._numSpheres, self._numCuboids = scene.num_spheres, scene.num_cuboids
        self._push_params()

    def UploadBasicData(self, blob: bytes) -> None:
        """The three SubData calls of MainWindow.cs:131-132,279 (InvProjection, InvView, ViewPos)."""
        self.BasicDataUBO.SubData(0, 64, blob[0:64])
        self.BasicDataUBO.SubData(64, 64, blob[64:128])
        self.BasicDataUBO.SubData(128, 16, blob[128:144])

    def Dispose(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class AtmosphericScatterer:
    """AtmosphericScatterer.cs:9-119 — precomputes the sky cube on the GPU of the PathTracer it is attached to."""

    def __init__(self, size: int, atmo_ubo: bytes, light_pos, tracer: PathTracer | None = None):
        self.Size = size
        self.ISteps, self.JSteps = 50, 15          # AtmosphericScatterer.cs:92-93
        self.LightIntensity = 15.0                 # :94
        self.LightPos = np.asarray(light_pos, dtype=np.float32)  # from Time, :41
        self._ubo = bytes(atmo_ubo)
        self._tracer = tracer

    def SetSize(self, size: int) -> None:  # :115-118
        self.Size = size

    def _select(self, tracer: PathTracer) -> None:
        self._tracer = tracer
        self.Render()

    def Render(self) -> None:  # :102-113
        t = self._tracer
        if t is None:
            raise RuntimeError("AtmosphericScatterer is not attached to a PathTracer")
        buf = np.frombuffer(self._ubo, dtype=np.uint8)
        check(t._lib.pt_atmosphere_upload_data(t._h, 0, buf.nbytes, buf.ctypes.data_as(C.c_void_p)), t._h)
        lp = np.ascontiguousarray(self.LightPos, dtype=np.float32)
        check(t._lib.pt_atmosphere_render(t._h, self.Size, self.ISteps, self.JSteps,
                                          lp.ctypes.data_as(C.POINTER(C.c_float)), max(self.LightIntensity, 0.0)), t._h)

    @property
    def Result(self) -> np.ndarray:
        return self._tracer.ReadEnvironment()
