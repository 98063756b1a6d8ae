"""ctypes binding of the CPU oracle (oracle/libakaze_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path (akaze-rust_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libakaze_ref.so")


class RefConfig(C.Structure):
    """Mirror of types::evolution::Config (akaze/src/types/evolution.rs:8-38)."""
    _fields_ = [
        ("num_sublevels", C.c_uint32),
        ("max_octave_evolution", C.c_uint32),
        ("base_scale_offset", C.c_double),
        ("initial_contrast", C.c_double),
        ("contrast_percentile", C.c_double),
        ("contrast_factor_num_bins", C.c_uint64),
        ("derivative_factor", C.c_double),
        ("detector_threshold", C.c_double),
        ("descriptor_channels", C.c_uint64),
        ("descriptor_pattern_size", C.c_uint64),
    ]


KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"),
     ("octave", "<u8"), ("class_id", "<u8"), ("angle", "<f4"), ("_pad", "<u4")])
MATCH_DTYPE = np.dtype([("index_0", "<u8"), ("index_1", "<u8"), ("distance", "<f8")])

PLANES = ["Lt", "Lsmooth", "Lx", "Ly", "Lxx", "Lyy", "Lxy", "Lflow", "Lstep", "Ldet"]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "akaze_ref.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    u8p = C.POINTER(C.c_uint8)
    L.ref_config_default.argtypes = [C.POINTER(RefConfig)]
    L.ref_gaussian_kernel.argtypes = [C.c_float, C.c_uint64, fp]
    L.ref_scharr_kernels.argtypes = [C.c_uint32, fp, fp]
    for name in ("ref_horizontal_filter", "ref_vertical_filter"):
        getattr(L, name).argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, fp]
    L.ref_gaussian_blur.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp]
    L.ref_half_size.argtypes = [fp, C.c_int, C.c_int, fp]
    L.ref_scharr.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, fp]
    L.ref_pm_g2.argtypes = [fp, fp, C.c_int, C.c_int, C.c_double, fp]
    L.ref_contrast_factor.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64]
    L.ref_contrast_factor.restype = C.c_double
    L.ref_fed_tau.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, dp, C.c_uint64]
    L.ref_fed_tau.restype = C.c_int64
    L.ref_fed_step.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_double]
    L.ref_extract_f32.argtypes = [fp, C.c_int, C.c_int, C.POINTER(RefConfig), C.c_uint32]
    L.ref_extract_f32.restype = C.c_void_p
    L.ref_extract_u8.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(RefConfig), C.c_uint32]
    L.ref_extract_u8.restype = C.c_void_p
    L.ref_result_free.argtypes = [C.c_void_p]
    for name in ("ref_result_num_levels", "ref_result_num_keypoints", "ref_result_num_extrema",
                 "ref_result_desc_bytes"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_uint64
    L.ref_result_contrast.argtypes = [C.c_void_p]
    L.ref_result_contrast.restype = C.c_double
    L.ref_result_keypoints.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_result_descriptors.argtypes = [C.c_void_p, u8p]
    L.ref_result_level_info.argtypes = [C.c_void_p, C.c_uint64, dp, dp] + [C.POINTER(C.c_uint32)] * 5 + [
        C.POINTER(C.c_uint64), dp, C.c_uint64]
    L.ref_result_plane.argtypes = [C.c_void_p, C.c_uint64, C.c_int, fp]
    L.ref_result_plane.restype = C.c_int64
    L.ref_descriptor_match.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double,
                                       C.c_void_p]
    L.ref_descriptor_match.restype = C.c_uint64
    L.ref_remove_outliers.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_uint64, C.c_float, C.c_float, C.c_void_p]
    L.ref_remove_outliers.restype = C.c_uint64
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def default_config(**overrides):
    c = RefConfig()
    lib().ref_config_default(C.byref(c))
    for k, v in overrides.items():
        setattr(c, k, v)
    return c


def gaussian_kernel(r, size):
    out = np.zeros(size, np.float32)
    lib().ref_gaussian_kernel(float(r), size, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def scharr_kernels(scale):
    n = 2 * scale + 1
    m, o = np.zeros(n, np.float32), np.zeros(n, np.float32)
    fp = C.POINTER(C.c_float)
    lib().ref_scharr_kernels(scale, m.ctypes.data_as(fp), o.ctypes.data_as(fp))
    return m, o


def _filter(fn, img, kern):
    img, pi = _f(img)
    kern, pk = _f(kern)
    h, w = img.shape
    out = np.empty_like(img)
    fn(pi, w, h, pk, len(kern), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def horizontal_filter(img, kern):
    return _filter(lib().ref_horizontal_filter, img, kern)


def vertical_filter(img, kern):
    return _filter(lib().ref_vertical_filter, img, kern)


def gaussian_blur(img, r):
    img, pi = _f(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().ref_gaussian_blur(pi, w, h, float(r), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def half_size(img):
    img, pi = _f(img)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().ref_half_size(pi, w, h, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def scharr(img, x_order, y_order, sigma):
    img, pi = _f(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().ref_scharr(pi, w, h, int(x_order), int(y_order), int(sigma), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def pm_g2(lx, ly, k):
    lx, px = _f(lx)
    ly, py = _f(ly)
    h, w = lx.shape
    out = np.empty_like(lx)
    lib().ref_pm_g2(px, py, w, h, float(k), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def contrast_factor(img, percentile=0.7, gscale=1.0, nbins=300):
    img, pi = _f(img)
    h, w = img.shape
    return lib().ref_contrast_factor(pi, w, h, percentile, gscale, nbins)


def fed_tau(T, M=1, tau_max=0.25, reordering=True):
    out = np.zeros(4096, np.float64)
    n = lib().ref_fed_tau(T, M, tau_max, int(reordering), out.ctypes.data_as(C.POINTER(C.c_double)), len(out))
    if n < 0:
        raise ValueError("reference does not terminate for this T (n == 1 with reordering)")
    return out[:n].copy()


def fed_step(lt, lflow, tau, lstep=None):
    """Returns (Lt', Lstep) after one calculate_step."""
    lt = np.array(lt, dtype=np.float32, order="C", copy=True)
    lflow, pf = _f(lflow)
    h, w = lt.shape
    lstep = np.zeros_like(lt) if lstep is None else np.array(lstep, dtype=np.float32, order="C", copy=True)
    fp = C.POINTER(C.c_float)
    lib().ref_fed_step(lt.ctypes.data_as(fp), pf, lstep.ctypes.data_as(fp), w, h, float(tau))
    return lt, lstep


class RefResult:
    """Owns an oracle extract_features result."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle extract failed")
        self._h = handle
        L = lib()
        self.num_levels = L.ref_result_num_levels(handle)
        self.num_keypoints = L.ref_result_num_keypoints(handle)
        self.num_extrema = L.ref_result_num_extrema(handle)
        self.desc_bytes = L.ref_result_desc_bytes(handle)
        self.contrast = L.ref_result_contrast(handle)

    def keypoints(self):
        out = np.zeros(self.num_keypoints, KEYPOINT_DTYPE)
        lib().ref_result_keypoints(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def descriptors(self):
        out = np.zeros((self.num_keypoints, self.desc_bytes), np.uint8)
        if self.num_keypoints:
            lib().ref_result_descriptors(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def level_info(self, lvl):
        et, es = C.c_double(), C.c_double()
        o, s, ss, w, h = (C.c_uint32() for _ in range(5))
        nt = C.c_uint64()
        tau = np.zeros(4096, np.float64)
        rc = lib().ref_result_level_info(self._h, lvl, C.byref(et), C.byref(es), C.byref(o), C.byref(s), C.byref(ss),
                                         C.byref(w), C.byref(h), C.byref(nt),
                                         tau.ctypes.data_as(C.POINTER(C.c_double)), len(tau))
        if rc != 0:
            raise IndexError(lvl)
        return dict(etime=et.value, esigma=es.value, octave=o.value, sublevel=s.value, sigma_size=ss.value,
                    w=w.value, h=h.value, tau=tau[:nt.value].copy())

    def plane(self, lvl, name):
        pid = PLANES.index(name) if isinstance(name, str) else int(name)
        n = lib().ref_result_plane(self._h, lvl, pid, None)
        if n < 0:
            raise IndexError((lvl, name))
        info = self.level_info(lvl)
        if n == 0:
            return np.zeros((0, 0), np.float32)
        out = np.empty((info["h"], info["w"]), np.float32)
        assert out.size == n
        lib().ref_result_plane(self._h, lvl, pid, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def describe(self, keypoints, compute_orientation=True):
        """compute_main_orientation + extract_descriptors for caller-supplied keypoints on this pyramid."""
        kp = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE).copy()
        desc = np.zeros((len(kp), self.desc_bytes), np.uint8)
        fn = lib().ref_result_describe
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        fn.restype = None
        fn(self._h, kp.ctypes.data_as(C.c_void_p), len(kp), int(bool(compute_orientation)), desc.ctypes.data_as(C.c_void_p))
        return kp, desc

    def close(self):
        if self._h:
            lib().ref_result_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def extract(img, cfg=None, threads=1):
    """img: 2-D uint8 (converted v*1/255 like create_unit_float_image) or float32."""
    cfg = cfg or default_config()
    img = np.ascontiguousarray(img)
    h, w = img.shape
    if img.dtype == np.uint8:
        hdl = lib().ref_extract_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, C.byref(cfg), threads)
    else:
        img = img.astype(np.float32, copy=False)
        hdl = lib().ref_extract_f32(img.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.byref(cfg), threads)
    return RefResult(hdl)


def descriptor_match(d0, d1, distance_threshold=10000, lowes_ratio=0.86):
    d0 = np.ascontiguousarray(d0, np.uint8)
    d1 = np.ascontiguousarray(d1, np.uint8)
    n0, nb = d0.shape if d0.ndim == 2 else (0, d1.shape[1])
    n1 = d1.shape[0]
    out = np.zeros(max(n0, 1), MATCH_DTYPE)
    u8p = C.POINTER(C.c_uint8)
    n = lib().ref_descriptor_match(d0.ctypes.data_as(u8p), n0, d1.ctypes.data_as(u8p), n1, nb, distance_threshold,
                                   lowes_ratio, out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def random_seed(s0=42, s1=69):
    """random::default().seed([s0, s1]) for the calling thread ([42, 69] is what a fresh thread starts with)."""
    lib().ref_random_seed.argtypes = [C.c_uint64, C.c_uint64]
    lib().ref_random_seed.restype = None
    lib().ref_random_seed(s0, s1)


def remove_outliers(keypoints_0, keypoints_1, matches, num_trials=1000, eps_model=0.05, eps_inlier=3.0):
    k0 = np.ascontiguousarray(keypoints_0, KEYPOINT_DTYPE)
    k1 = np.ascontiguousarray(keypoints_1, KEYPOINT_DTYPE)
    m = np.ascontiguousarray(matches, MATCH_DTYPE)
    out = np.zeros(max(1, len(m)), MATCH_DTYPE)
    n = lib().ref_remove_outliers(k0.ctypes.data_as(C.c_void_p), len(k0), k1.ctypes.data_as(C.c_void_p), len(k1),
                                  m.ctypes.data_as(C.c_void_p), len(m), num_trials, eps_model, eps_inlier,
                                  out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()
