"""MI355X-native path tracer: Python plumbing over the C ABI of include/rtk.h.

The product is ``librtk_hip.so`` (hand-written HIP for gfx950, csrc/) driven through
the C ABI; this module only loads it with ctypes, mirrors the ABI structs and wraps
handles in small classes so that tests and ``bench.py`` can pass torch device
pointers and streams.  Scenes come from ``librtk_host.so`` -- the reference's scene
API re-implemented in C++ (host/), which flattens a ``hittable`` graph into the
``rtk_scene_desc`` the ABI takes.

There is no CPU rendering path here.  If ``librtk_hip.so`` is missing or no gfx950
device is usable, ``Renderer`` raises; nothing falls back to the oracle.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
from typing import Optional

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_PKG_DIR)
DEFAULT_HIP_LIB_PATH = os.path.join(_PKG_DIR, "librtk_hip.so")
# RTK_HIP_LIB substitutes a diagnostic build (tools/: A/B libraries, the -DRTK_PROFILE build) and is honoured only together
# with RTK_DEV_TOOLS=1: a stray RTK_HIP_LIB in the environment of a test or bench run is an error, never a silent swap.
if os.environ.get("RTK_HIP_LIB") and os.environ.get("RTK_DEV_TOOLS") != "1":
    raise ImportError("RTK_HIP_LIB is set without RTK_DEV_TOOLS=1: refusing to load a substitute kernel library "
                      f"({os.environ['RTK_HIP_LIB']}) in place of {DEFAULT_HIP_LIB_PATH}")
HIP_LIB_PATH = os.environ.get("RTK_HIP_LIB") or DEFAULT_HIP_LIB_PATH
HOST_LIB_PATH = os.path.join(_PKG_DIR, "librtk_host.so")

RTK_ABI_VERSION = 2
RTK_REAL_F64 = 0
RTK_REAL_F32 = 1
TILE_PIXELS = 64

SCENE_SEED = 0x5EED2025  # construction RNG seed of the BASELINE scenes
RENDER_SEED = 1

# BASELINE.json configs -> scene_library.h names
CONFIG_SCENES = {
    "c1": "three_spheres",
    "c2": "book1_final",
    "c3": "cornell_box",
    "c4": "mesh",
    "c5": "book2_final",
}


class RtkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rtk error {code}: {message}")
        self.code = code


# ----------------------------------------------------------------------------- ABI structs
class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class Camera(C.Structure):
    """rtk_camera (include/rtk.h): the values camera::initialize() derives."""

    _fields_ = [
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
        ("background", Vec3), ("center", Vec3), ("pixel00_loc", Vec3),
        ("pixel_delta_u", Vec3), ("pixel_delta_v", Vec3),
        ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3),
        ("defocus_angle", C.c_double), ("pixel_samples_scale", C.c_double),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32), ("real_mode", C.c_int32), ("rank", C.c_int32), ("n_ranks", C.c_int32),
        ("count_work", C.c_int32), ("variant", C.c_int32), ("stream", C.c_void_p),
    ]


class OptimizeOpts(C.Structure):
    """rtk_optimize_opts (include/rtk.h)."""

    _fields_ = [("has_eye", C.c_int32), ("max_leaf", C.c_int32), ("eye", Vec3), ("prim_cost_scale", C.c_double),
                ("free_media_order", C.c_int32), ("_pad", C.c_int32)]


class OptimizeInfo(C.Structure):
    _fields_ = [("exact", C.c_int32), ("has_media", C.c_int32), ("has_triangles", C.c_int32),
                ("n_bvh_nodes_in", C.c_int32), ("n_bvh_nodes_out", C.c_int32), ("n_ordered_items", C.c_int32),
                ("expected_cost", C.c_double), ("box_margin", C.c_double)]


def _optimize_opts(eye, max_leaf, prim_cost_scale, free_media_order) -> OptimizeOpts:
    return OptimizeOpts(1 if eye is not None else 0, max_leaf, eye if eye is not None else Vec3(0, 0, 0), prim_cost_scale, 1 if free_media_order else 0, 0)


def _optimize_info(info: OptimizeInfo) -> dict:
    # exact: rtk_optimize_info.exact != 0 (bit-identical to the reference order); proven: == 2 (no triangles -- with them it is
    # identical in every measurement but not provable, include/rtk.h)
    return {"exact": bool(info.exact), "proven": info.exact == 2, "exactness": int(info.exact),
            "has_media": bool(info.has_media), "has_triangles": bool(info.has_triangles),
            "n_bvh_nodes_in": info.n_bvh_nodes_in, "n_bvh_nodes_out": info.n_bvh_nodes_out, "n_ordered_items": info.n_ordered_items,
            "expected_cost": info.expected_cost, "box_margin": info.box_margin}


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)  # rtk_progress_fn
GATHER_AUTO, GATHER_PEER, GATHER_RCCL = 0, 1, 2                     # rtk_gather_mode

COUNTER_FIELDS = (
    "samples", "segments", "box_tests", "sphere_tests", "quad_tests", "triangle_tests",
    "xform_enters", "medium_tests", "surface_hits", "noise_calls", "texel_fetches", "rng_draws",
)


class WorkCounters(C.Structure):
    _fields_ = [(name, C.c_uint64) for name in COUNTER_FIELDS]

    def as_dict(self) -> dict:
        return {name: int(getattr(self, name)) for name in COUNTER_FIELDS}


def algorithmic_bytes_per_sample(counters: dict, spp: int, real_mode: int, f32_boxes: bool = False) -> float:
    """SURVEY.md 8(d) byte model: bytes the sample loop must touch per sample.

    fp32 record sizes: BVH node 32, sphere 32, quad 68, triangle 76, instance
    transform 24, medium 12, material 32, perlin::noise 120, texel 4, framebuffer
    12 B/pixel.  In f64 mode the real-valued fields double (the 4-byte indices and
    the u8 texel do not).  ``f32_boxes``: the launched kernel reads the MIXED program's
    32-byte f32 culling-box records (F_F32_BOX) whatever the arithmetic type of the primitives.
    """
    n = max(1, counters["samples"])
    f64 = real_mode == RTK_REAL_F64
    node = 56 if (f64 and not f32_boxes) else 32        # 6 reals + 2 u32
    sphere = 60 if f64 else 32      # 7 reals + material
    quad = 132 if f64 else 68       # 16 reals + material
    tri = 124 if f64 else 76        # 12 reals + 6 float uv + material
    xform = 44 if f64 else 24       # 5 reals + child
    medium = 16 if f64 else 12
    material = 48 if f64 else 32
    noise = 216 if f64 else 120     # 8 gradients of 3 reals + 6 perm ints
    fb = 24 if f64 else 12
    total = (node * counters["box_tests"] + sphere * counters["sphere_tests"] + quad * counters["quad_tests"]
             + tri * counters["triangle_tests"] + xform * counters["xform_enters"] + medium * counters["medium_tests"]
             + material * counters["surface_hits"] + noise * counters["noise_calls"] + 4 * counters["texel_fetches"])
    return total / n + fb / max(1, spp)


# ----------------------------------------------------------------------------- libraries
_host_lib = None
_hip_lib = None


def host_lib() -> C.CDLL:
    """librtk_host.so: scene construction + flattening (no GPU needed)."""
    global _host_lib
    if _host_lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_LIB_PATH)
        lib.rtkh_scene_build.restype = C.c_void_p
        lib.rtkh_scene_build.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        lib.rtkh_scene_load.restype = C.c_void_p
        lib.rtkh_scene_load.argtypes = [C.c_char_p]
        lib.rtkh_scene_free.argtypes = [C.c_void_p]
        lib.rtkh_scene_desc.restype = C.c_void_p
        lib.rtkh_scene_desc.argtypes = [C.c_void_p]
        lib.rtkh_scene_save.argtypes = [C.c_void_p, C.c_char_p]
        lib.rtkh_scene_rng_draws.restype = C.c_uint64
        lib.rtkh_scene_rng_draws.argtypes = [C.c_void_p]
        lib.rtkh_scene_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Camera)]
        lib.rtkh_camera_derive.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(Camera)]
        lib.rtkh_image_texels.restype = C.c_int64
        lib.rtkh_image_texels.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int64]
        _host_lib = lib
    return _host_lib


def _one_hip_runtime() -> None:
    """One HIP runtime per process.  librtk_hip.so links /opt/rocm's libamdhip64; a PyTorch-ROCm wheel carries its own copy, and
    whichever runtime touches the devices FIRST owns them -- a process that rendered before it imported torch finds torch.cuda
    without devices ("No HIP GPUs are available").  With torch loaded first the library's HIP calls bind to torch's runtime (its
    libraries sit in the global symbol scope), which is also what makes torch streams and data_ptr()s valid arguments of the C
    ABI.  So where torch is installed it is imported before a HIP library of this package is loaded."""
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:   # a broken torch install must not take the renderer down with it
            pass


def hip_lib() -> C.CDLL:
    """librtk_hip.so: the kernels + C ABI.  Raises if it was not built."""
    global _hip_lib
    if _hip_lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} not built: the HIP extension is required, there is no CPU path")
        _one_hip_runtime()
        lib = C.CDLL(HIP_LIB_PATH)
        lib.rtk_abi_version.restype = C.c_int
        lib.rtk_last_error.restype = C.c_char_p
        lib.rtk_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.rtk_destroy.argtypes = [C.c_void_p]
        lib.rtk_scene_upload.argtypes = [C.c_void_p, C.c_void_p]
        lib.rtk_tiles_per_rank.restype = C.c_int64
        lib.rtk_tiles_per_rank.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.rtk_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rtk_tiles_unpermute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rtk_render_host.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rtk_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.rtk_debug_closest_hit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        try:
            lib.rtk_frame_launches.argtypes = [C.POINTER(Camera), C.POINTER(RenderOpts)]
            lib.rtk_debug_scatter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.rtk_debug_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.rtk_debug_get_ray.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.rtk_render_multi_enqueue.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]
            lib.rtk_multi_wait.argtypes = [C.c_void_p]
            lib.rtk_multi_frame_plan.argtypes = [C.c_int64, C.POINTER(C.c_int32)]
        except AttributeError:
            if HIP_LIB_PATH == DEFAULT_HIP_LIB_PATH:   # an A/B library of an older round (tools/, RTK_DEV_TOOLS=1) may lack the newer entry points
                raise
        lib.rtk_scene_optimize.argtypes = [C.c_void_p, C.POINTER(OptimizeOpts), C.POINTER(C.c_void_p), C.POINTER(OptimizeInfo)]
        lib.rtk_scene_upload_fast.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OptimizeOpts), C.POINTER(OptimizeInfo)]
        lib.rtk_scene_optimized_free.restype = None
        lib.rtk_scene_optimized_free.argtypes = [C.c_void_p]
        lib.rtk_kernel_name.restype = C.c_char_p
        lib.rtk_kernel_name.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.rtk_scene_upload_optimized.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OptimizeOpts)]
        lib.rtk_init_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        lib.rtk_multi_destroy.argtypes = [C.c_void_p]
        lib.rtk_multi_device_count.argtypes = [C.c_void_p]
        lib.rtk_multi_uses_rccl.argtypes = [C.c_void_p]
        lib.rtk_multi_ctx.restype = C.c_void_p
        lib.rtk_multi_ctx.argtypes = [C.c_void_p, C.c_int]
        lib.rtk_multi_scene_upload.argtypes = [C.c_void_p, C.c_void_p]
        lib.rtk_multi_scene_upload_fast.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OptimizeOpts), C.POINTER(OptimizeInfo)]
        lib.rtk_render_multi_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]
        lib.rtk_render_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p]
        lib.rtk_set_progress_callback.argtypes = [C.c_void_p, PROGRESS_FN, C.c_void_p, C.c_int]
        if lib.rtk_abi_version() != RTK_ABI_VERSION and HIP_LIB_PATH == DEFAULT_HIP_LIB_PATH:
            raise RuntimeError("librtk_hip.so ABI version mismatch")
        _hip_lib = lib
    return _hip_lib


MICROBENCH_LIB_PATH = os.path.join(_PKG_DIR, "librtk_microbench.so")


def microbench(device: int = 0) -> dict:
    """Measured ceilings of the box (csrc/rtk_microbench.hip): HBM stream copy, LDS read rates, VALU issue rates per class.
    A measurement tool for bench.py's roofline object; raises when the library or a device is missing."""
    if not os.path.exists(MICROBENCH_LIB_PATH):
        raise RuntimeError(f"{MICROBENCH_LIB_PATH} not built (run __graft_entry__.build())")
    _one_hip_runtime()
    lib = C.CDLL(MICROBENCH_LIB_PATH)
    lib.rtk_microbench_names.restype = C.c_char_p
    lib.rtk_microbench_last_error.restype = C.c_char_p
    names = lib.rtk_microbench_names().decode().split(",")
    out = (C.c_double * len(names))()
    rc = lib.rtk_microbench_run(device, out, len(names))
    if rc != 0:
        raise RuntimeError(f"rtk_microbench_run failed ({rc}): {lib.rtk_microbench_last_error().decode()}")
    return dict(zip(names, [float(v) for v in out]))


def tiles_per_rank(width: int, height: int, n_ranks: int) -> int:
    """Same arithmetic as rtk_tiles_per_rank (usable without the HIP library)."""
    tiles = ((width + 7) // 8) * ((height + 7) // 8)
    return (tiles + n_ranks - 1) // n_ranks


# ----------------------------------------------------------------------------- scenes
class Scene:
    """A flattened scene (rtk_scene_desc) owned by librtk_host.so."""

    def __init__(self, handle: int, name: str):
        if not handle:
            raise ValueError(f"could not build/load scene {name!r}")
        self._h = handle
        self.name = name

    @classmethod
    def build(cls, name: str, scene_seed: int = SCENE_SEED, image_file: Optional[str] = None) -> "Scene":
        h = host_lib().rtkh_scene_build(name.encode(), scene_seed, (image_file or "").encode())
        return cls(h, name)

    @classmethod
    def load(cls, path: str) -> "Scene":
        return cls(host_lib().rtkh_scene_load(path.encode()), os.path.basename(path))

    @property
    def desc_ptr(self) -> int:
        return host_lib().rtkh_scene_desc(self._h)

    def save(self, path: str) -> None:
        if host_lib().rtkh_scene_save(self._h, path.encode()) != 0:
            raise IOError(f"cannot write {path}")

    def camera(self, width: int = 0, height: int = 0, spp: int = 0, depth: int = 0) -> Camera:
        cam = Camera()
        rc = host_lib().rtkh_scene_camera(self._h, width, height, spp, depth, C.byref(cam))
        if rc != 0:
            raise ValueError(f"cannot derive a {width}x{height} camera for {self.name} (rc={rc})")
        return cam

    def fast_order(self, eye: Optional[Vec3] = None, max_leaf: int = 0, prim_cost_scale: float = 0.0, free_media_order: bool = False) -> "FastOrderScene":
        """The same primitives re-grouped by rtk_scene_optimize (host-only pass of librtk_hip.so)."""
        return FastOrderScene(self, eye, max_leaf, prim_cost_scale, free_media_order)

    def close(self) -> None:
        if self._h:
            host_lib().rtkh_scene_free(self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastOrderScene:
    """rtk_scene_optimize output: a description that borrows the tables of `base` (kept alive here).

    ``exact`` says whether rendering it gives bit-identical images to the reference order (always, unless media were
    re-grouped too: ``free_media_order``); ``proven`` whether that is a proof (scenes without triangles) or a
    measurement (``rtk_optimize_info.exact`` 2 vs 1).
    """

    def __init__(self, base: Scene, eye: Optional[Vec3] = None, max_leaf: int = 0, prim_cost_scale: float = 0.0, free_media_order: bool = False):
        self.base = base
        self.name = base.name + "+fast_order"
        self.opts = opts = _optimize_opts(eye, max_leaf, prim_cost_scale, free_media_order)
        out, info = C.c_void_p(), OptimizeInfo()
        rc = hip_lib().rtk_scene_optimize(base.desc_ptr, C.byref(opts), C.byref(out), C.byref(info))
        if rc != 0 or not out.value:
            raise RtkError(rc, "rtk_scene_optimize failed")
        self._h = out.value
        self.exact = bool(info.exact)
        self.proven = info.exact == 2
        self.info = _optimize_info(info)

    @property
    def desc_ptr(self) -> int:
        return self._h

    def camera(self, *args, **kwargs) -> Camera:
        return self.base.camera(*args, **kwargs)

    def close(self) -> None:
        if getattr(self, "_h", None):
            hip_lib().rtk_scene_optimized_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def derive_camera(image_width: int, aspect_ratio: float, *, spp: int = 10, max_depth: int = 10, vfov: float = 90.0, lookfrom=(0.0, 0.0, 0.0),
                  lookat=(0.0, 0.0, -1.0), vup=(0.0, 1.0, 0.0), defocus_angle: float = 0.0, focus_dist: float = 10.0) -> Camera:
    """camera::derive() of the drop-in camera (host/rtk_camera.h = camera::initialize, Camera.txt:136-175) for raw public fields."""
    cam = Camera()
    v3 = lambda v: (C.c_double * 3)(*v)  # noqa: E731
    rc = host_lib().rtkh_camera_derive(image_width, aspect_ratio, spp, max_depth, vfov, v3(lookfrom), v3(lookat), v3(vup), defocus_angle, focus_dist, C.byref(cam))
    if rc != 0:
        raise ValueError("rtkh_camera_derive failed")
    return cam


def load_image_texels(path: str):
    """The RGB8 texels image_texture reads for an image file (PPM or baseline JPEG), as rtw_image holds them
    (stbi_loadf's gamma-2.2 mapping and float_to_byte applied).  Returns an (H, W, 3) uint8 array or None."""
    import numpy as np

    w, h = C.c_int(), C.c_int()
    n = host_lib().rtkh_image_texels(path.encode(), C.byref(w), C.byref(h), None, 0)
    if n < 0:
        return None
    out = np.zeros((h.value, w.value, 3), np.uint8)
    host_lib().rtkh_image_texels(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, n)
    return out


def write_synthetic_earth(path: str, width: int = 1024, height: int = 512) -> str:
    """Procedural RGB8 texture standing in for earthmap.jpg (binary PPM)."""
    import numpy as np

    y, x = np.mgrid[0:height, 0:width]
    lon = x / width * 2 * np.pi
    lat = (y / height - 0.5) * np.pi
    land = np.sin(3 * lon) * np.cos(2 * lat) + 0.5 * np.sin(7 * lon + 1.3) * np.sin(5 * lat) + 0.25 * np.cos(13 * lon * np.cos(lat))
    r = np.where(land > 0.15, 60 + 120 * np.clip(land, 0, 1), 20)
    g = np.where(land > 0.15, 110 + 90 * np.clip(1 - land, 0, 1), 60 + 40 * np.cos(lat))
    b = np.where(land > 0.15, 40, 150 + 80 * np.cos(lat))
    img = np.stack([r, g, b], -1).clip(0, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (width, height))
        f.write(img.tobytes())
    return path


# ----------------------------------------------------------------------------- renderer
class Renderer:
    """One rtk_ctx bound to a HIP device.  Raises RtkError when no gfx950 device is usable."""

    def __init__(self, device: int = 0):
        self._lib = hip_lib()
        ctx = C.c_void_p()
        self._check(self._lib.rtk_init(device, C.byref(ctx)))
        self._ctx = ctx
        self.device = device

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtkError(rc, self._lib.rtk_last_error().decode())

    def upload(self, scene) -> None:
        self._check(self._lib.rtk_scene_upload(self._ctx, scene.desc_ptr))

    def upload_fast(self, scene, eye: Optional[Vec3] = None, max_leaf: int = 0, prim_cost_scale: float = 0.0, free_media_order: bool = False) -> dict:
        """rtk_scene_upload_fast: optimise the visiting order and upload, with the fast-order kernels.
        Returns the rtk_optimize_info fields."""
        opts = _optimize_opts(eye, max_leaf, prim_cost_scale, free_media_order)
        info = OptimizeInfo()
        self._check(self._lib.rtk_scene_upload_fast(self._ctx, scene.desc_ptr, C.byref(opts), C.byref(info)))
        return _optimize_info(info)

    def upload_optimized(self, scene, eye: Optional[Vec3] = None, free_media_order: bool = False) -> None:
        """rtk_scene_upload_optimized: a description that already IS a re-grouped hierarchy (rtk_scene_optimize output, or a
        hand-built one whose primitive nodes carry reference ranks in rtk_node.c), with the fast-order kernels."""
        opts = _optimize_opts(eye, 0, 0.0, free_media_order)
        self._check(self._lib.rtk_scene_upload_optimized(self._ctx, scene.desc_ptr, C.byref(opts)))

    def scene_info(self) -> dict:
        n, b64, b32 = C.c_int32(), C.c_int64(), C.c_int64()
        self._check(self._lib.rtk_scene_info(self._ctx, C.byref(n), C.byref(b64), C.byref(b32)))
        return {"program_ops": n.value, "bytes_f64": b64.value, "bytes_f32": b32.value}

    def kernel_name(self, real_mode: int = RTK_REAL_F64, variant: int = 0) -> str:
        return self._lib.rtk_kernel_name(self._ctx, real_mode, variant).decode()

    def render_device(self, cam: Camera, d_linear: int, d_rgb8: int = 0, *, seed: int = RENDER_SEED, real_mode: int = RTK_REAL_F64,
                      rank: int = 0, n_ranks: int = 1, d_counters: int = 0, variant: int = 0, stream: int = 0) -> None:
        """Enqueue the render kernel; all pointers are raw device addresses, nothing synchronises."""
        opts = RenderOpts(seed, real_mode, rank, n_ranks, 1 if d_counters else 0, variant, stream or None)
        self._check(self._lib.rtk_render_device(self._ctx, C.byref(cam), C.byref(opts), d_linear or None, d_rgb8 or None, d_counters or None))

    def unpermute(self, width: int, height: int, n_ranks: int, real_mode: int, d_gathered: int, d_linear: int, d_rgb8: int = 0, stream: int = 0) -> None:
        self._check(self._lib.rtk_tiles_unpermute(self._ctx, width, height, n_ranks, real_mode, d_gathered, d_linear or None, d_rgb8 or None, stream or None))

    def render_host(self, cam: Camera, *, seed: int = RENDER_SEED, real_mode: int = RTK_REAL_F64, count: bool = False, variant: int = 0):
        """Render the whole image and return (linear float64 HxWx3, rgb8 HxWx3, counters dict | None)."""
        import numpy as np

        h, w = cam.image_height, cam.image_width
        linear = np.zeros((h, w, 3), np.float64)
        rgb8 = np.zeros((h, w, 3), np.uint8)
        counters = WorkCounters()
        opts = RenderOpts(seed, real_mode, 0, 1, 1 if count else 0, variant, None)
        self._check(self._lib.rtk_render_host(self._ctx, C.byref(cam), C.byref(opts), linear.ctypes.data, rgb8.ctypes.data,
                                              C.byref(counters) if count else None))
        return linear, rgb8, (counters.as_dict() if count else None)

    def closest_hit(self, rays, keys, real_mode: int = RTK_REAL_F64):
        """Known-answer helper: hittable::hit of the scene root for rays [n,9] (o, d, time, tmin, tmax) with RNG keys
        [n,3] (seed, pixel, sample).  Returns (records [n,12], draws [n])."""
        import numpy as np

        rays = np.ascontiguousarray(rays, np.float64)
        keys = np.ascontiguousarray(keys, np.uint32)
        n = rays.shape[0]
        out = np.zeros((n, 12), np.float64)
        draws = np.zeros(n, np.uint64)
        self._check(self._lib.rtk_debug_closest_hit(self._ctx, real_mode, n, rays.ctypes.data, keys.ctypes.data, out.ctypes.data, draws.ctypes.data))
        return out, draws

    def debug_scatter(self, materials, rays, records, keys, real_mode: int = RTK_REAL_F64):
        """rtk_debug_scatter: (out [n][14] = scattered, scattered ray o(3) d(3), attenuation(3), time, emitted(3); draws [n])."""
        import numpy as np

        materials = np.ascontiguousarray(materials, np.int32)
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        records = np.ascontiguousarray(records, np.float64).reshape(-1, 11)
        keys = np.ascontiguousarray(keys, np.uint32).reshape(-1, 3)
        n = materials.shape[0]
        out, draws = np.zeros((n, 14), np.float64), np.zeros(n, np.uint64)
        self._check(self._lib.rtk_debug_scatter(self._ctx, real_mode, n, materials.ctypes.data, rays.ctypes.data, records.ctypes.data, keys.ctypes.data,
                                                out.ctypes.data, draws.ctypes.data))
        return out, draws

    def debug_texture(self, textures, uvp, real_mode: int = RTK_REAL_F64):
        """rtk_debug_texture: (colour [n][3], work [n][2] = perlin::noise calls, texel fetches)."""
        import numpy as np

        textures = np.ascontiguousarray(textures, np.int32)
        uvp = np.ascontiguousarray(uvp, np.float64).reshape(-1, 5)
        n = textures.shape[0]
        out, work = np.zeros((n, 3), np.float64), np.zeros((n, 2), np.uint64)
        self._check(self._lib.rtk_debug_texture(self._ctx, real_mode, n, textures.ctypes.data, uvp.ctypes.data, out.ctypes.data, work.ctypes.data))
        return out, work

    def debug_get_ray(self, cam: Camera, seed: int, pixel_sample, real_mode: int = RTK_REAL_F64):
        """rtk_debug_get_ray: (ray [n][7] = origin, direction, time; draws [n]) for pixel_sample [n][3] = i, j, sample."""
        import numpy as np

        ijs = np.ascontiguousarray(pixel_sample, np.int32).reshape(-1, 3)
        n = ijs.shape[0]
        out, draws = np.zeros((n, 7), np.float64), np.zeros(n, np.uint64)
        self._check(self._lib.rtk_debug_get_ray(self._ctx, real_mode, C.byref(cam), seed, n, ijs.ctypes.data, out.ctypes.data, draws.ctypes.data))
        return out, draws

    def set_progress(self, fn=None, interval_ms: int = 100) -> None:
        """rtk_set_progress_callback: ``fn(done, total)`` is called from the thread that runs a blocking render
        (render_host), at most every ``interval_ms``; None switches it off."""
        self._progress = PROGRESS_FN(lambda done, total, _user: fn(done, total)) if fn else C.cast(None, PROGRESS_FN)
        self._check(self._lib.rtk_set_progress_callback(self._ctx, self._progress, None, interval_ms))

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.rtk_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """rtk_multi: several GPUs of one node behind one call (one host thread, replicated scene, interleaved tiles, one
    gather to the first device).  ``devices`` are HIP ordinals; an ordinal may repeat (ranks then share a GPU)."""

    def __init__(self, devices, gather: int = GATHER_AUTO):
        self._lib = hip_lib()
        devs = (C.c_int * len(devices))(*devices)
        handle = C.c_void_p()
        self._check(self._lib.rtk_init_multi(len(devices), devs, gather, C.byref(handle)))
        self._m = handle
        self.devices = list(devices)

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtkError(rc, self._lib.rtk_last_error().decode())

    @property
    def uses_rccl(self) -> bool:
        return bool(self._lib.rtk_multi_uses_rccl(self._m))

    def upload(self, scene) -> None:
        self._check(self._lib.rtk_multi_scene_upload(self._m, scene.desc_ptr))

    def upload_fast(self, scene, eye: Optional[Vec3] = None) -> dict:
        opts = _optimize_opts(eye, 0, 0.0, False)
        info = OptimizeInfo()
        self._check(self._lib.rtk_multi_scene_upload_fast(self._m, scene.desc_ptr, C.byref(opts), C.byref(info)))
        return _optimize_info(info)

    def kernel_name(self, real_mode: int = RTK_REAL_F64, variant: int = 0) -> str:
        return self._lib.rtk_kernel_name(self._lib.rtk_multi_ctx(self._m, 0), real_mode, variant).decode()

    def set_progress(self, fn=None, interval_ms: int = 100) -> None:
        self._progress = PROGRESS_FN(lambda done, total, _user: fn(done, total)) if fn else C.cast(None, PROGRESS_FN)
        self._check(self._lib.rtk_set_progress_callback(self._lib.rtk_multi_ctx(self._m, 0), self._progress, None, interval_ms))

    def render_host(self, cam: Camera, *, seed: int = RENDER_SEED, real_mode: int = RTK_REAL_F64, variant: int = 0):
        """rtk_render_multi: (linear float64 HxWx3, rgb8 HxWx3)."""
        import numpy as np

        h, w = cam.image_height, cam.image_width
        linear = np.zeros((h, w, 3), np.float64)
        rgb8 = np.zeros((h, w, 3), np.uint8)
        opts = RenderOpts(seed, real_mode, 0, 1, 0, variant, None)
        self._check(self._lib.rtk_render_multi(self._m, C.byref(cam), C.byref(opts), linear.ctypes.data, rgb8.ctypes.data))
        return linear, rgb8

    def render_device(self, cam: Camera, d_linear: int, d_rgb8: int = 0, *, seed: int = RENDER_SEED, real_mode: int = RTK_REAL_F64, variant: int = 0) -> None:
        """rtk_render_multi_device: blocking; the image is resident on devices[0] on return."""
        opts = RenderOpts(seed, real_mode, 0, 1, 0, variant, None)
        self._check(self._lib.rtk_render_multi_device(self._m, C.byref(cam), C.byref(opts), d_linear or None, d_rgb8 or None))

    def enqueue_device(self, cam: Camera, d_linear: int, d_rgb8: int = 0, *, seed: int = RENDER_SEED, real_mode: int = RTK_REAL_F64, variant: int = 0) -> None:
        """rtk_render_multi_enqueue: returns once the frame is enqueued (two frames in flight: this frame's gather and
        un-permute overlap the next frame's renders); the buffers must stay valid until wait()."""
        opts = RenderOpts(seed, real_mode, 0, 1, 0, variant, None)
        self._check(self._lib.rtk_render_multi_enqueue(self._m, C.byref(cam), C.byref(opts), d_linear or None, d_rgb8 or None))

    def wait(self) -> None:
        """rtk_multi_wait: every enqueued frame is complete on devices[0]."""
        self._check(self._lib.rtk_multi_wait(self._m))

    def close(self) -> None:
        if getattr(self, "_m", None):
            self._lib.rtk_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
