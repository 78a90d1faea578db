"""Render driver: what Game1.LoadContent / Update do around the trace loop (Game1.cs:151-249), with the
worker threads replaced by calls into libsrt.so."""
import ctypes as C

import numpy as np

from . import _native as N
from .raytracing import Camera, LightList, SurfaceList


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Renderer:
    """Owns one libsrt context: one GPU (`device`), or a list of them (`devices`: HIP ordinals, or "all") over which the
    library splits every frame, as Game1 splits it over its worker threads (Game1.cs:161-167).  Not thread-safe, like
    the single UI thread of Game1."""

    def __init__(self, device=0, library=None, strict=True, devices=None, gather=N.GATHER_RCCL):
        self._lib = N.load(library, strict=strict)
        self._ctx = N._CTX()
        cfg = N.Config(device=device)
        if devices == "all":
            cfg.n_devices = N.ALL_DEVICES
        elif devices is not None:
            devices = [int(d) for d in devices]
            if not 1 <= len(devices) <= N.MAX_DEVICES:
                raise ValueError("1..%d devices" % N.MAX_DEVICES)
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = d
        cfg.gather = int(gather)
        rc = self._lib.srt_create(C.byref(cfg), C.byref(self._ctx))
        if rc != N.OK:
            raise N.SrtError(rc, self._lib.srt_last_error(None).decode())
        self.device = device if devices is None else (devices[0] if devices != "all" else 0)

    def device_info(self):
        """-> {"devices": [HIP ordinals], "gather": GATHER_*, "rccl_ranks": ranks of the RCCL communicator or 0}."""
        info = N.DeviceInfo()
        self._check(self._lib.srt_get_device_info(self._ctx, C.byref(info)))
        return {"devices": [info.devices[i] for i in range(info.n_devices)], "gather": info.gather, "rccl_ranks": info.rccl_ranks}

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.srt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != N.OK:
            raise N.SrtError(rc, self._lib.srt_last_error(self._ctx).decode())

    # -- scene ------------------------------------------------------------------------------------
    def set_camera(self, camera: Camera):
        arrs = [np.ascontiguousarray(v, dtype=np.float32) for v in
                (camera.position, camera.lowerLeftCorner, camera.horizontal, camera.vertical)]
        self._check(self._lib.srt_set_camera(self._ctx, *[_fp(a) for a in arrs]))

    def set_scene(self, world: SurfaceList, lights: LightList, camera: Camera = None, world_kind=N.WORLD_BVHNODE):
        """world.Init(); nodeWorld = new BVHNode(world); lights.init()  (Game1.cs:136-144)."""
        L = self._lib
        flat = world.flatten()
        mats = flat["materials"]
        marr = (N.Material * max(len(mats), 1))()
        for i, m in enumerate(mats):
            marr[i] = N.Material((C.c_float * 3)(*m.albedo), float(m.metalness), float(m.smoothness), (C.c_float * 3)(*m.emission))
        self._check(L.srt_set_materials(self._ctx, marr, len(mats)))
        s = flat["spheres"]
        cols = [np.ascontiguousarray(s[:, i]) for i in range(4)]
        self._check(L.srt_set_spheres(self._ctx, *[_fp(c) for c in cols], _up(flat["sphere_mat"]), len(s)))
        r = flat["rects"]
        cols = [np.ascontiguousarray(r[:, i]) for i in range(5)]
        self._check(L.srt_set_rects(self._ctx, *[_fp(c) for c in cols], _up(flat["rect_mat"]), len(r)))
        tp, tn = flat["tri_pos"], flat["tri_nrm"]
        self._check(L.srt_set_triangles(self._ctx, _fp(tp), _fp(tn), _up(flat["tri_mat"]),
                                        flat["tri_flags"].ctypes.data_as(C.POINTER(C.c_uint8)), len(tp)))
        # world.surfaces as one ordered list; triangles, spheres, rects is the library's default and needs no call
        ot, oi = flat["order_type"], flat["order_index"]
        first = np.array([0, len(tp), len(tp) + len(s)], dtype=np.int64)    # global index: all triangles, then all spheres, then all rects
        if not np.array_equal(first[ot] + oi, np.arange(len(ot), dtype=np.int64)):
            self._check(L.srt_set_surface_order(self._ctx, ot.ctypes.data_as(C.POINTER(C.c_uint8)), _up(oi), len(ot)))
        else:
            self._check(L.srt_set_surface_order(self._ctx, None, None, 0))
        ll = lights.lights
        larr = (N.PointLight * max(len(ll), 1))()
        for i, l in enumerate(ll):
            larr[i] = N.PointLight((C.c_float * 3)(*l.position), float(l.brightness), (C.c_float * 3)(*l.lightColor))
        self._check(L.srt_set_lights(self._ctx, larr, len(ll)))
        if camera is not None:
            self.set_camera(camera)
        self._check(L.srt_commit_scene(self._ctx, int(world_kind)))

    # -- rendering --------------------------------------------------------------------------------
    @staticmethod
    def params(width, height, samples=500, max_depth=4, color_clamp=1.0, seed=1, mode=N.MODE_COLOR,
               layout=N.LAYOUT_FRAME, tile_w=0, tile_h=0, rank=0, world_size=1):
        """Defaults are Game1's hard-coded fields (Game1.cs:41-43)."""
        return N.RenderParams(width=width, height=height, samples=samples, max_depth=max_depth, color_clamp=color_clamp,
                              seed=seed, mode=mode, layout=layout, tile_w=tile_w, tile_h=tile_h, rank=rank, world_size=world_size)

    def local_pixels(self, p):
        n = self._lib.srt_local_pixels(C.byref(p))
        if n < 0:
            raise N.SrtError(int(n), "bad render parameters")
        return int(n)

    def render(self, width, height, counted=False, **kw):
        """Blocking render of the full frame -> (height, width) uint32, packed R | G<<8 | B<<16 | A<<24."""
        p = self.params(width, height, **kw)
        n = self.local_pixels(p)
        out = np.zeros(n, dtype=np.uint32)
        fn = self._lib.srt_render_counted if counted else self._lib.srt_render
        self._check(fn(self._ctx, C.byref(p), out.ctypes.data_as(C.c_void_p)))
        if p.layout == N.LAYOUT_FRAME:
            return out.reshape(height, width)
        return out

    def render_mix_buffers(self, width, height, n_buffers, **kw):
        """The reference's mixBufferTechnique (Game1.cs:174-175, :209-220, :320-388): n_buffers one-sample renders of the
        whole frame, each divided by `samples`, mixed pairwise in 8-bit colour -> (height, width) uint32."""
        p = self.params(width, height, **kw)
        out = np.zeros(width * height, dtype=np.uint32)
        self._check(self._lib.srt_render_mix_buffers(self._ctx, C.byref(p), int(n_buffers), out.ctypes.data_as(C.c_void_p)))
        return out.reshape(height, width)

    def render_device(self, p, device_ptr, capacity_pixels, stream=None):
        """Launches the render into device memory on `stream` (a hipStream_t as int; 0/None = HIP's default stream)
        without synchronising."""
        self._check(self._lib.srt_render_device(self._ctx, C.byref(p), C.c_void_p(device_ptr), capacity_pixels,
                                                C.c_void_p(stream or 0)))

    def render_device_static(self, p, device_ptr, capacity_pixels, stream=None):
        """render_device for callers that may not allocate: works in the library's static per-GPU workspace."""
        self._check(self._lib.srt_render_device_static(self._ctx, C.byref(p), C.c_void_p(device_ptr), capacity_pixels,
                                                       C.c_void_p(stream or 0)))

    def render_async(self, p, out):
        self._check(self._lib.srt_render_async(self._ctx, C.byref(p), out.ctypes.data_as(C.c_void_p)))

    def render_progressive(self, p, out, passes):
        """Like render_async, in at least `passes` passes with a preview after each (Game1.Update's live picture)."""
        self._check(self._lib.srt_render_progressive(self._ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), int(passes)))

    def progress(self):
        done, total = C.c_int(0), C.c_int(0)
        self._check(self._lib.srt_progress(self._ctx, C.byref(done), C.byref(total)))
        return done.value, total.value

    def read_preview(self, out):
        self._check(self._lib.srt_read_preview(self._ctx, out.ctypes.data_as(C.c_void_p)))

    # -- post-process filters of the finished picture (Sorting.*, MiscFunctions.cs:122-416) --------
    def _post(self, fn, array, *args):
        a = np.ascontiguousarray(array, dtype=np.uint32).copy()
        if a.ndim != 2:
            raise ValueError("array must be a (height, width) picture")
        h, w = a.shape
        self._check(fn(self._ctx, a.ctypes.data_as(C.c_void_p), w, h, *args))
        return a

    def blur_array(self, array, weight, passes=1):
        """Sorting.BlurArray (MiscFunctions.cs:124), `passes` times, with its in-place semantics -> new picture."""
        return self._post(self._lib.srt_blur_array, array, C.c_float(weight), int(passes))

    def contrast_blur_array(self, array, weight, contrast, passes=1):
        """Sorting.ContrastBlurArray (MiscFunctions.cs:193)."""
        return self._post(self._lib.srt_contrast_blur_array, array, C.c_float(weight), C.c_float(contrast), int(passes))

    def despeckle(self, array, weight, threshold, passes=1):
        """Sorting.Despeckle (MiscFunctions.cs:337), what Game1.Update runs blurPasses times (Game1.cs:226-232)."""
        return self._post(self._lib.srt_despeckle, array, C.c_float(weight), C.c_float(threshold), int(passes))

    def poll(self):
        done = C.c_int(0)
        self._check(self._lib.srt_poll(self._ctx, C.byref(done)))
        return bool(done.value)

    def wait(self):
        self._check(self._lib.srt_wait(self._ctx))

    def assemble_tiles(self, p, gathered_ptr, stride_pixels, frame_ptr, stream=None):
        self._check(self._lib.srt_assemble_tiles(self._ctx, C.byref(p), C.c_void_p(gathered_ptr), stride_pixels,
                                                 C.c_void_p(frame_ptr), C.c_void_p(stream or 0)))

    @staticmethod
    def _stats_dict(st):
        out = {k: getattr(st, k) for k, _ in N.Stats._fields_ if k != "reserved"}
        out["section_cycles"] = list(st.section_cycles)
        out["section_lane_cycles"] = list(st.section_lane_cycles)
        out["shadow_unasked"] = int(st.shadow_unasked)
        out["gather_ms"], out["assemble_ms"] = float(st.gather_ms), float(st.assemble_ms)
        return out

    def stats(self):
        st = N.Stats()
        self._check(self._lib.srt_get_stats(self._ctx, C.byref(st)))
        return self._stats_dict(st)

    def device_stats(self, index):
        """One device's share (index = position in the device list) of what stats() last collected."""
        st = N.Stats()
        self._check(self._lib.srt_get_device_stats(self._ctx, int(index), C.byref(st)))
        return self._stats_dict(st)


def rgba_to_rgb8(img):
    """(H, W) packed uint32 -> (H, W, 3) uint8."""
    return np.stack([(img >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8)
