"""Python host binding of libakaze_hip.so (the C ABI in include/akaze_hip.h).

Mirrors the reference crate's public interface for the hot path — same names, argument
meaning and error behaviour (the reference panics; here an AkazeError is raised):

    akaze::extract_features   (akaze/src/lib.rs:167-194)   -> extract_features()
    akaze::match_features     (akaze/src/lib.rs:252-275)   -> match_features()  (descriptor part)
    types::evolution::Config  (akaze/src/types/evolution.rs:8-55) -> Config
    pub mod ops / types::image                              -> ops.* on torch CUDA tensors

torch is used only as the owner of device memory and streams; all compute is in the HIP
library.  There is no CPU fallback: if libakaze_hip.so is missing or no gfx950 GPU is
visible, constructing a Context raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("AKAZE_HIP_LIB") or os.path.join(_PKG, "libakaze_hip.so")  # override: kernel tuning A/B builds

AKZ_KEEP_ALL_PLANES = 1
AKZ_NO_HOST_DESCRIPTORS = 2
AKZ_NO_DETECT = 4
AKZ_INPUT_READY = 8

PLANES = ["Lt", "Lsmooth", "Lx", "Ly", "Lxx", "Lyy", "Lxy", "Lflow", "Lstep", "Ldet"]


class AkazeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"akaze_hip status {status}: {msg}")
        self.status = status


class Config(C.Structure):
    """types::evolution::Config (akaze/src/types/evolution.rs:8-38); Config() == Config::default()."""
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

    def __init__(self, **kw):
        super().__init__()
        lib().akz_config_default(C.byref(self))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


STAGES = ["blur0", "contrast", "prep", "fed", "detector", "nms", "host_kp", "orient", "mldb", "total"]


class KernelRow(C.Structure):
    """akz_kernel_row (akaze_hip_debug.h)"""
    _fields_ = [("stage", C.c_uint32), ("kind", C.c_uint32), ("param", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("n", C.c_uint32), ("launches", C.c_uint64), ("px", C.c_uint64), ("px_steps", C.c_uint64), ("ms", C.c_double)]


KERNEL_ROW_KINDS = {1: "k_level_march", 2: "k_fed_own", 3: "k_octave_resident", 4: "k_detector_tiled", 5: "k_detector_march"}


class Gate(C.Structure):
    """akz_gate (akaze_hip_debug.h)"""
    _fields_ = [("name", C.c_char_p), ("value", C.c_double), ("unit", C.c_char_p), ("meaning", C.c_char_p)]


def gates():
    """akz_debug_gates: the table of every size / host-thread gate (csrc/akz_gates.hpp) as a list of dicts; needs no GPU."""
    p, n = C.c_void_p(), C.c_uint64()
    _check(lib().akz_debug_gates(C.byref(p), C.byref(n)))
    rows = C.cast(p, C.POINTER(Gate))
    return [dict(name=rows[i].name.decode(), value=rows[i].value, unit=rows[i].unit.decode(), meaning=rows[i].meaning.decode())
            for i in range(n.value)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 10), ("fed_launches", C.c_uint64), ("fed_px_steps", C.c_uint64),
                ("calls", C.c_uint64), ("pixels", C.c_uint64), ("det_launches", C.c_uint64), ("det_px", C.c_uint64), ("fused_px", C.c_uint64),
                ("placement_probed", C.c_uint32), ("placement_early_stages", C.c_uint32), ("placement_replaced", C.c_uint32),
                ("placement_shared", C.c_uint32), ("placement_retries", C.c_uint32), ("placement_reserved", C.c_uint32)]

    def as_dict(self):
        d = {k: self.ms[i] for i, k in enumerate(STAGES)}
        d.update(fed_launches=self.fed_launches, fed_px_steps=self.fed_px_steps, calls=self.calls,
                 pixels=self.pixels, det_launches=self.det_launches, det_px=self.det_px, fused_px=self.fused_px,
                 placement={"probed": bool(self.placement_probed), "early_stages_on": "copy stream" if self.placement_early_stages == 2 else "caller's stream",
                            "streams_replaced": self.placement_replaced, "streams_still_shared": self.placement_shared,
                            "measurements_repeated": self.placement_retries})
        return d


KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"),
     ("octave", "<u8"), ("class_id", "<u8"), ("angle", "<f4"), ("_pad", "<u4")])
MATCH_DTYPE = np.dtype([("index_0", "<u8"), ("index_1", "<u8"), ("distance", "<f8")])

_lib = None


def lib():
    """Load libakaze_hip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C akaze-rust_amd). There is no CPU fallback.")
    # torch bundles its own HIP runtime (libamdhip64.so.7).  Import it first so that this library's
    # NEEDED libamdhip64.so.7 binds to the copy torch already loaded: two HIP/HSA runtimes in one
    # process do not both see the GPU.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    pu32, pu64, pf64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    sig = {
        "akz_abi_version": ([], i32),
        "akz_last_error": ([], C.c_char_p),
        "akz_config_default": ([C.POINTER(Config)], None),
        "akz_ctx_create": ([i32, vp, C.POINTER(vp)], i32),
        "akz_ctx_destroy": ([vp], i32),
        "akz_stream_create": ([i32, C.POINTER(vp)], i32),
        "akz_stream_destroy": ([i32, vp], i32),
        "akz_ctx_synchronize": ([vp], i32),
        "akz_ctx_stream": ([vp], vp),
        "akz_device_malloc": ([vp, C.c_size_t, C.POINTER(vp)], i32),
        "akz_device_free": ([vp, vp], i32),
        "akz_memcpy_h2d": ([vp, vp, vp, C.c_size_t], i32),
        "akz_memcpy_d2h": ([vp, vp, vp, C.c_size_t], i32),
        "akz_fed_tau_by_process_time": ([f64, i32, f64, i32, pf64, u64, pu64], i32),
        "akz_gaussian_kernel": ([C.c_float, u64, fp], i32),
        "akz_scharr_kernels": ([u32, fp, fp], i32),
        "akz_plan_num_levels": ([u32, u32, C.POINTER(Config), pu64], i32),
        "akz_plan_level_info": ([u32, u32, C.POINTER(Config), u64, pf64, pf64, pu32, pu32, pu32, pu32, pu32, pu32,
                                 pu64, pf64, u64], i32),
        "akz_op_horizontal_filter": ([vp, vp, vp, u32, u32, u32, fp, u32], i32),
        "akz_op_vertical_filter": ([vp, vp, vp, u32, u32, u32, fp, u32], i32),
        "akz_op_gaussian_blur": ([vp, vp, vp, u32, u32, u32, C.c_float], i32),
        "akz_op_gaussian_blur_u8": ([vp, vp, vp, u32, u32, u32, C.c_float], i32),
        "akz_op_half_size": ([vp, vp, vp, u32, u32, u32], i32),
        "akz_op_scharr": ([vp, vp, vp, u32, u32, u32, i32, i32, u32], i32),
        "akz_debug_rcp_f64_to_f32": ([vp, vp, vp, u64], i32),
        "akz_debug_kernel_rows": ([vp, vp, u32, vp, i32], i32),
        "akz_debug_gates": ([C.POINTER(vp), pu64], i32),
        "akz_debug_device_libm": ([vp, C.POINTER(i32), C.POINTER(i32)], i32),
        "akz_debug_set_device_libm": ([vp, i32], i32),
        "akz_debug_libm_eval": ([vp, vp, vp, vp, u64, i32], i32),
        "akz_ctx_calibrate_gates": ([vp, pu64, pu64, C.POINTER(C.c_double)], i32),
        "akz_op_pm_g2": ([vp, vp, vp, vp, u32, u32, u32, vp], i32),
        "akz_op_contrast_factor": ([vp, vp, u32, u32, u32, f64, f64, u64, vp], i32),
        "akz_op_flow": ([vp, vp, vp, u32, u32, u32, vp, u32], i32),
        "akz_op_fed_steps": ([vp, vp, vp, vp, u32, u32, u32, pf64, u32], i32),
        "akz_op_detector_response": ([vp, vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, u32], i32),
        "akz_extract_gray_u8": ([vp, vp, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_gray_f32": ([vp, vp, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_device_u8": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_device_f32": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_begin_device_u8": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_begin_device_f32": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_begin_host_u8": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_begin_host_f32": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_extract_finish": ([vp, C.POINTER(vp)], i32),
        "akz_extract_from_planes": ([vp, u32, u32, C.POINTER(Config), C.POINTER(vp), u64, u32, C.POINTER(vp)], i32),
        "akz_random_color": ([vp], i32),
        "akz_draw_circle": ([vp, u32, u32, C.c_float, C.c_float, vp, C.c_float], i32),
        "akz_draw_line": ([vp, u32, u32, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_float], i32),
        "akz_job_abandon": ([vp], i32),
        "akz_result_free": ([vp], i32),
        "akz_result_num_images": ([vp, pu64], i32),
        "akz_result_counts": ([vp, u64, pu64, pu64, pu64], i32),
        "akz_result_keypoints": ([vp, u64, vp], i32),
        "akz_result_descriptors": ([vp, u64, vp], i32),
        "akz_result_device_descriptors": ([vp, u64, C.POINTER(vp), pu64], i32),
        "akz_result_contrast": ([vp, u64, pf64], i32),
        "akz_result_copy_device_descriptors": ([vp, vp, u64, pu64], i32),
        "akz_result_level_info": ([vp, u64, pf64, pf64, pu32, pu32, pu32, pu32, pu32, pu64, pf64, u64], i32),
        "akz_fetch_plane": ([vp, u64, u64, i32, vp, pu64], i32),
        "akz_result_device_plane": ([vp, u64, u64, i32, C.POINTER(vp)], i32),
        "akz_descriptor_match": ([vp, vp, u64, vp, u64, u64, u64, f64, vp, pu64], i32),
        "akz_descriptor_match_device": ([vp, vp, u64, vp, u64, u64, f64, vp, vp], i32),
        "akz_ctx_graph_probe": ([vp, vp, u32, u32, u32, C.POINTER(Config), u32, u32, pf64, pf64, pu64], i32),
        "akz_debug_march_bands": ([i32, u32, u32, u32, i32, C.POINTER(i32), u32, pu32], i32),
        "akz_ctx_set_lanes": ([vp, u32], i32),
        "akz_ctx_set_eager_finish": ([vp, i32], i32),
        "akz_fed_kernel_name": ([], C.c_char_p),
        "akz_ctx_set_host_threads": ([vp, u32], i32),
        "akz_debug_set_match_chunks": ([vp, u32, u32], i32),
        "akz_debug_set_host_sort": ([vp, i32], i32),
        "akz_debug_set_select": ([vp, i32], i32),
        "akz_debug_select_info": ([vp, C.POINTER(C.c_int)], i32),
        "akz_debug_set_schedule": ([vp, i32, i32], i32),
        "akz_debug_stream_placement": ([vp, C.POINTER(i32)], i32),
        "akz_detector_kernel_name": ([], C.c_char_p),
        "akz_remove_outliers": ([vp, u64, vp, u64, vp, u64, u64, C.c_float, C.c_float, vp, pu64], i32),
        "akz_estimate_fundamental_matrix": ([vp, u64, vp, u64, vp, C.c_float, fp, C.POINTER(i32)], i32),
        "akz_match_features": ([vp, vp, u64, vp, u64, vp, u64, vp, u64, u64, f64, u64, C.c_float, vp, pu64], i32),
        "akz_write_features": ([C.c_char_p, vp, u64, vp, u64], i32),
        "akz_read_features": ([C.c_char_p, vp, vp, u64, u64, pu64, pu64, pu64], i32),
        "akz_write_matches": ([C.c_char_p, vp, u64], i32),
        "akz_read_matches": ([C.c_char_p, vp, u64, pu64], i32),
        "akz_host_select_keypoints": ([u32, u32, C.POINTER(Config), vp, u64, vp, u64, pu64, pu64], i32),
        "akz_result_describe_keypoints": ([vp, u64, vp, u64, i32, vp], i32),
        "akz_random_seed": ([u64, u64], i32),
        "akz_image_load": ([C.c_char_p, pu32, pu32, pu32, C.POINTER(vp)], i32),
        "akz_image_load_luma": ([C.c_char_p, pu32, pu32, C.POINTER(vp)], i32),
        "akz_image_load_rgb": ([C.c_char_p, pu32, pu32, C.POINTER(vp)], i32),
        "akz_image_free": ([vp], None),
        "akz_extract_features_file": ([vp, C.c_char_p, C.POINTER(Config), u32, C.POINTER(vp)], i32),
        "akz_config_to_json": ([C.POINTER(Config), C.c_char_p, u64, pu64], i32),
        "akz_config_from_json": ([C.c_char_p, C.POINTER(Config)], i32),
        "akz_image_save_png": ([C.c_char_p, vp, u32, u32, u32], i32),
        "akz_image_save_plane_png": ([C.c_char_p, vp, u32, u32], i32),
        "akz_write_evolutions": ([vp, u64, C.c_char_p], i32),
        "akz_draw_keypoints": ([vp, u32, u32, vp, u64], i32),
        "akz_draw_matches": ([vp, u32, u32, vp, u32, u32, vp, u64, vp, u64, vp, u64, pu32, pu32, C.POINTER(vp)], i32),
        "akz_ctx_set_profiling": ([vp, i32], i32),
        "akz_ctx_set_fed_mode": ([vp, i32], i32),
        "akz_ctx_set_match_mode": ([vp, i32], i32),
        "akz_ctx_set_candidate_hint": ([vp, u32], i32),
        "akz_descriptor_match_sets_device": ([vp, vp, u64, vp, C.POINTER(u64), u64, u64, f64, vp, vp], i32),
        "akz_descriptor_match_sets_mutual_device": ([vp, vp, u64, vp, C.POINTER(u64), u64, u64, f64, vp, vp, vp, vp], i32),
        "akz_ctx_set_detector_mode": ([vp, i32], i32),
        "akz_ctx_set_prep_mode": ([vp, i32], i32),
        "akz_ctx_get_profile": ([vp, C.POINTER(Profile), i32], i32),
        "akz_ctx_get_profile2": ([vp, C.POINTER(Profile), u64, i32], i32),
        "akz_ctx_warmup": ([vp], i32),
        "akz_debug_placement_verdict": ([C.c_float, C.c_float, C.c_float, C.c_float], i32),
        "akz_synth_frame_u8": ([vp, u32, u32, u64, C.c_int32, C.c_int32], i32),
        "akz_comm_unique_id": ([vp], i32),
        "akz_comm_create": ([i32, vp, i32, i32, C.POINTER(vp)], i32),
        "akz_comm_create_external": ([i32, i32, i32, C.POINTER(vp)], i32),
        "akz_gather_blocks": ([vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)], i32),
        "akz_gather_deliver": ([vp, vp], i32),
        "akz_comm_destroy": ([vp], i32),
        "akz_comm_set_timeout": ([vp, f64], i32),
        "akz_comm_place_streams": ([vp, vp], i32),
        "akz_comm_info": ([vp, C.POINTER(i32), C.POINTER(i32)], i32),
        "akz_gather_descriptors": ([vp, vp, u64, C.POINTER(vp), pu64], i32),
        "akz_gather_begin": ([vp, C.POINTER(vp), u64, u64, C.POINTER(vp)], i32),
        "akz_gather_begin_rows": ([vp, vp, u64, u64, vp, C.POINTER(vp)], i32),
        "akz_gather_begin_image_rows": ([vp, vp, pu64, u64, u64, vp, C.POINTER(vp)], i32),
        "akz_pairs_plan": ([vp, C.POINTER(vp)], i32),
        "akz_pairs_lead_sets": ([vp, u64, pu64, u64, pu64], i32),
        "akz_gather_image_rows": ([vp, i32, pu64, u64, pu64], i32),
        "akz_match_all_pairs": ([vp, vp, u64, C.c_double, C.POINTER(vp)], i32),
        "akz_pairs_info": ([vp, pu64, pu64, pu64], i32),
        "akz_pairs_image_rows": ([vp, u64, pu64, C.POINTER(i32)], i32),
        "akz_pairs_holder": ([vp, u64, u64, C.POINTER(i32)], i32),
        "akz_pairs_matches": ([vp, u64, u64, vp, u64, pu64], i32),
        "akz_pairs_free": ([vp], i32),
        "akz_pairs_totals": ([vp, pu64, pu64, pu64], i32),
        "akz_gather_stream_wait": ([vp, vp], i32),
        "akz_gather_finish": ([vp, C.POINTER(vp), pu64, pu64, pu64], i32),
        "akz_gather_free": ([vp], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)  # AttributeError here == a symbol of include/akaze_hip.h is missing
        fn.argtypes = args
        fn.restype = res
    L._declared = sorted(sig)
    _lib = L
    return L


def _check(status):
    if status != 0:
        raise AkazeError(status, lib().akz_last_error().decode("utf-8", "replace"))


# ------------------------------------------------------------------------------------------
# host-side planning helpers (no GPU needed)
# ------------------------------------------------------------------------------------------
def fed_tau_by_process_time(T, M=1, tau_max=0.25, reordering=True):
    """ops::fed_tau::fed_tau_by_process_time (akaze/src/ops/fed_tau.rs:27-30)."""
    n = C.c_uint64()
    _check(lib().akz_fed_tau_by_process_time(T, M, tau_max, int(reordering), None, 0, C.byref(n)))
    out = np.zeros(n.value, np.float64)
    _check(lib().akz_fed_tau_by_process_time(T, M, tau_max, int(reordering),
                                             out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
    return out


def gaussian_kernel(sigma, kernel_size):
    out = np.zeros(kernel_size, np.float32)
    _check(lib().akz_gaussian_kernel(sigma, kernel_size, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def scharr_kernels(scale):
    n = 2 * scale + 1
    m, o = np.zeros(n, np.float32), np.zeros(n, np.float32)
    fp = C.POINTER(C.c_float)
    _check(lib().akz_scharr_kernels(scale, m.ctypes.data_as(fp), o.ctypes.data_as(fp)))
    return m, o


def plan_levels(w, h, cfg=None):
    """types::evolution::allocate_evolutions (akaze/src/types/evolution.rs:135-161) + level sizes."""
    cfg = cfg or Config()
    n = C.c_uint64()
    _check(lib().akz_plan_num_levels(w, h, C.byref(cfg), C.byref(n)))
    out = []
    for lvl in range(n.value):
        et, es = C.c_double(), C.c_double()
        o, s, ss, lw, lh, ds = (C.c_uint32() for _ in range(6))
        nt = C.c_uint64()
        tau = np.zeros(8192, np.float64)
        _check(lib().akz_plan_level_info(w, h, C.byref(cfg), lvl, C.byref(et), C.byref(es), C.byref(o), C.byref(s),
                                         C.byref(ss), C.byref(lw), C.byref(lh), C.byref(ds), C.byref(nt),
                                         tau.ctypes.data_as(C.POINTER(C.c_double)), len(tau)))
        out.append(dict(etime=et.value, esigma=es.value, octave=o.value, sublevel=s.value, sigma_size=ss.value,
                        w=lw.value, h=lh.value, det_sigma=ds.value, tau=tau[:nt.value].copy()))
    return out


CANDIDATE_DTYPE = np.dtype([("level", "<u4"), ("idx", "<u4"), ("v", "<f4"), ("xp", "<f4"), ("xm", "<f4"),
                            ("yp", "<f4"), ("ym", "<f4"), ("_pad", "<u4")])


def host_select_keypoints(w, h, cfg, cands):
    """Host part of detect_keypoints (scale_space_extrema.rs:43-178) on NMS candidates; no GPU needed.
    Returns (keypoints without angle, number of extrema before the sub-pixel step)."""
    cands = np.ascontiguousarray(cands, CANDIDATE_DTYPE)
    out = np.zeros(max(1, len(cands)), KEYPOINT_DTYPE)
    n, ne = C.c_uint64(), C.c_uint64()
    _check(lib().akz_host_select_keypoints(w, h, C.byref(cfg), cands.ctypes.data_as(C.c_void_p), len(cands),
                                           out.ctypes.data_as(C.c_void_p), len(out), C.byref(n), C.byref(ne)))
    return out[:n.value].copy(), ne.value


def synth_frame(w, h, frame_index=0, shift=(0, 0)):
    """Deterministic synthetic 8-bit luma frame; host utility."""
    out = np.empty((h, w), np.uint8)
    _check(lib().akz_synth_frame_u8(out.ctypes.data_as(C.c_void_p), w, h, frame_index, shift[0], shift[1]))
    return out


# ------------------------------------------------------------------------------------------
# GPU context
# ------------------------------------------------------------------------------------------
class Context:
    """One GPU + one HIP stream.  Not thread-safe; one per host thread / per rank."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().akz_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            lib().akz_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().akz_ctx_synchronize(self._h))

    @property
    def stream(self):
        return lib().akz_ctx_stream(self._h)

    def set_fed_mode(self, mode):
        """2 = k_fed_own, temporally fused (default), 0 = k_fed_step, one launch per step."""
        _check(lib().akz_ctx_set_fed_mode(self._h, int(mode)))

    def set_candidate_hint(self, per_image):
        """Room for extrema candidates per image in the next extraction (a list that overflows is enlarged and the
        extrema pass repeated by finish: same results)."""
        _check(lib().akz_ctx_set_candidate_hint(self._h, int(per_image)))

    def set_lanes(self, lanes):
        """Deal jobs below 2.4 Mpx to `lanes` child contexts in turn (1 = off): the launch chains of consecutive single
        frames then overlap on the chip."""
        _check(lib().akz_ctx_set_lanes(self._h, int(lanes)))

    def set_eager_finish(self, on=True):
        """With lanes: the finish half of every job dealt to a lane starts on that lane's own thread as soon as the job
        has been begun; `finish()` then only collects the result (same results)."""
        _check(lib().akz_ctx_set_eager_finish(self._h, 1 if on else 0))

    def debug_set_select(self, mode):
        """akz_debug_set_select: keypoint selection on the device (2), on the host from the device's neighbour lists (1 / True),
        from the host's grids (0 / False), automatic (None)."""
        _check(lib().akz_debug_set_select(self._h, -1 if mode is None else (2 if mode == 2 and mode is not True else (1 if mode else 0))))

    def debug_set_schedule(self, key, value):
        """akz_debug_set_schedule: schedule variants of a large batch (measurement hook, identical results)."""
        _check(lib().akz_debug_set_schedule(self._h, int(key), int(value)))

    def debug_stream_placement(self):
        """akz_debug_stream_placement: what the context's stream-placement probe found."""
        info = (C.c_int32 * 4)()
        _check(lib().akz_debug_stream_placement(self._h, info))
        return {"probed": bool(info[0]), "early_stages": ("context stream", "own stream", "copy stream")[info[1]],
                "streams_replaced": int(info[2]), "streams_sharing_a_queue": int(info[3])}

    def set_match_mode(self, mode):
        """2 = automatic (default), 1 = matrix-core matcher, 0 = popcount matcher."""
        _check(lib().akz_ctx_set_match_mode(self._h, int(mode)))

    def set_detector_mode(self, mode):
        """2 = automatic (default: column march for large launches, one LDS-tiled kernel for small ones), 5 = column
        march, 4 = one LDS-tiled kernel, 0 = LDS-tiled kernel pair."""
        _check(lib().akz_ctx_set_detector_mode(self._h, int(mode)))

    def set_prep_mode(self, mode):
        """Level preparation: 2 = automatic (default: fused with the first diffusion steps for large launches), 3 = fused
        wherever supported, 1 = streaming kernel, 0 = LDS-tiled kernel."""
        _check(lib().akz_ctx_set_prep_mode(self._h, int(mode)))

    def warmup(self):
        """akz_ctx_warmup: the stream-placement probe now (idle streams, no capture) instead of inside the first large batch."""
        _check(lib().akz_ctx_warmup(self._h))

    def set_profiling(self, on=True):
        """0/False off, 1/True every stage, 2 light (FED spans + host-clock stages only)."""
        _check(lib().akz_ctx_set_profiling(self._h, int(on)))

    def get_profile(self, reset=True):
        p = Profile()
        _check(lib().akz_ctx_get_profile2(self._h, C.byref(p), C.sizeof(Profile), int(reset)))
        return p.as_dict()

    def debug_device_libm(self):
        """akz_debug_device_libm -> (available, last_job): 0 = the host's libm, 1 / 2 = the device's FMA / SSE2 forms of glibc's
        atan2f / cosf / sinf (csrc/akz_libm.hpp), proven equal to this process's libm by a self-test."""
        a, b = C.c_int32(), C.c_int32()
        _check(lib().akz_debug_device_libm(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_set_device_libm(self, on):
        """False: angles and their cosines / sines always from the host's libm; True / None: automatic."""
        _check(lib().akz_debug_set_device_libm(self._h, 0 if on is False else -1))

    def debug_libm_eval(self, a, b, fma=True):
        """(atan2f(a, b), cosf(a), sinf(a)) per element as the device forms them: torch CUDA float32 tensors -> [n, 3]"""
        import torch
        out = torch.empty((a.numel(), 3), dtype=torch.float32, device=a.device)
        _check(lib().akz_debug_libm_eval(self._h, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), int(bool(fma))))
        return out

    def calibrate_gates(self):
        """akz_ctx_calibrate_gates: the two job gates from timings on this machine -> (sync_px, async_px, ms[5][4])"""
        a, b = C.c_uint64(), C.c_uint64()
        ms = (C.c_double * 20)()
        _check(lib().akz_ctx_calibrate_gates(self._h, C.byref(a), C.byref(b), ms))
        return a.value, b.value, [[ms[i * 4 + k] for k in range(4)] for i in range(5)]

    def kernel_rows(self, reset=True):
        """akz_debug_kernel_rows: the FED / detector spans by kernel variant and launch shape (profiling must be on)."""
        n = C.c_uint32()
        rows = (KernelRow * 256)()
        _check(lib().akz_debug_kernel_rows(self._h, C.cast(rows, C.c_void_p), 256, C.cast(C.pointer(n), C.c_void_p), int(reset)))
        out = []
        for r in rows[:min(n.value, 256)]:
            if r.launches == 0:
                continue
            name = KERNEL_ROW_KINDS.get(r.kind, str(r.kind))
            if r.kind == 1:
                name += f"<{r.param & 15}{', Lstep' if r.param & 32 else ''}{', HALF' if r.param & 16 else ''}>"
            elif r.kind in (4, 5):
                name += f"<{r.param}>"
            out.append(dict(kernel=name, kind=r.kind, param=r.param, w=r.w, h=r.h, n=r.n, launches=int(r.launches), px=int(r.px),
                            px_steps=int(r.px_steps), ms=float(r.ms)))
        return out

    def set_host_threads(self, threads):
        """akz_ctx_set_host_threads: host threads of the finish half (0 = automatic)."""
        _check(lib().akz_ctx_set_host_threads(self._h, int(threads)))

    def debug_select_info(self):
        """akz_debug_select_info: (where the last job's selection ran: 0 grids / 1 lists / 2 device, most rounds of an image,
        images that fell back to the host, candidates)."""
        info = (C.c_int * 8)()
        _check(lib().akz_debug_select_info(self._h, info))
        return tuple(info)

    def debug_set_host_sort(self, on):
        """akz_debug_set_host_sort: candidates bucketed and sorted on the host instead of the device sort."""
        _check(lib().akz_debug_set_host_sort(self._h, -1 if on is None else (1 if on else 0)))

    def debug_set_match_chunks(self, pair_chunks=0, set_chunks=0):
        """akz_debug_set_match_chunks (include/akaze_hip_debug.h): force the matcher's chunk counts; 0 = automatic."""
        _check(lib().akz_debug_set_match_chunks(self._h, int(pair_chunks), int(set_chunks)))

    def graph_probe(self, frames, options=None, keep_all_planes=True, reps=50):
        """akz_ctx_graph_probe: (ms per hipGraph launch of the begin phase, ms per plain enqueue of it, graph nodes)."""
        options = options or Config()
        t = frames if frames.dim() == 3 else frames.unsqueeze(0)
        n, h, w = t.shape
        g, p, nodes = C.c_double(), C.c_double(), C.c_uint64()
        _check(lib().akz_ctx_graph_probe(self._h, C.c_void_p(t.data_ptr()), w, h, n, C.byref(options),
                                         AKZ_KEEP_ALL_PLANES if keep_all_planes else 0, reps, C.byref(g), C.byref(p),
                                         C.byref(nodes)))
        return g.value, p.value, nodes.value

    # ---- the hot path --------------------------------------------------------------------
    def extract_features(self, image, options=None, keep_all_planes=True, host_descriptors=True):
        """akaze::extract_features on an in-memory luma image (2-D uint8 or float32 numpy array, or a
        torch CUDA tensor of shape [H, W] or [N, H, W]).  Returns an ExtractResult."""
        options = options or Config()
        flags = (AKZ_KEEP_ALL_PLANES if keep_all_planes else 0) | (0 if host_descriptors else AKZ_NO_HOST_DESCRIPTORS)
        res = C.c_void_p()
        L = lib()
        if isinstance(image, np.ndarray):
            img = np.ascontiguousarray(image)
            if img.ndim != 2:
                raise ValueError("host images must be 2-D (one luma plane)")
            h, w = img.shape
            if img.dtype == np.uint8:
                _check(L.akz_extract_gray_u8(self._h, img.ctypes.data_as(C.c_void_p), w, h, C.byref(options), flags,
                                             C.byref(res)))
            else:
                img = img.astype(np.float32, copy=False)
                _check(L.akz_extract_gray_f32(self._h, img.ctypes.data_as(C.c_void_p), w, h, C.byref(options), flags,
                                              C.byref(res)))
        else:  # torch CUDA tensor
            import torch
            t = image
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
                raise ValueError("device images must be contiguous torch CUDA tensors")
            if t.dim() == 2:
                t = t.unsqueeze(0)
            n, h, w = t.shape
            if t.dtype == torch.uint8:
                fn = L.akz_extract_device_u8
            elif t.dtype == torch.float32:
                fn = L.akz_extract_device_f32
            else:
                raise ValueError("uint8 or float32 images only")
            _check(fn(self._h, C.c_void_p(t.data_ptr()), w, h, n, C.byref(options), flags, C.byref(res)))
        return ExtractResult(self, res)

    def extract_features_file(self, path, options=None, keep_all_planes=True):
        """akaze::extract_features(input_image_path, options) — akaze/src/lib.rs:167-194."""
        cfg = options or Config()
        res = C.c_void_p()
        _check(lib().akz_extract_features_file(self._h, os.fsencode(path), C.byref(cfg),
                                               AKZ_KEEP_ALL_PLANES if keep_all_planes else 0, C.byref(res)))
        return ExtractResult(self, res)

    def extract_from_planes(self, w, h, planes, options=None, detect=True):
        """ops::scale_space_extrema::detect_keypoints / ops::descriptors::extract_descriptors on evolutions the caller
        holds: planes[level][name] -> 2-D float32 array (names of PLANES; missing ones are skipped; Lt, Lx, Ly and —
        for detection — Ldet are required).  Returns an ExtractResult."""
        options = options or Config()
        n = len(planes)
        keep = []
        tab = (C.c_void_p * (n * 10))()
        for l, lv in enumerate(planes):
            for p, name in enumerate(PLANES):
                a = lv.get(name)
                if a is None or a.size == 0:
                    continue
                a = np.ascontiguousarray(a, np.float32)
                keep.append(a)
                tab[l * 10 + p] = a.ctypes.data
        res = C.c_void_p()
        _check(lib().akz_extract_from_planes(self._h, w, h, C.byref(options), tab, n, 0 if detect else AKZ_NO_DETECT,
                                             C.byref(res)))
        return ExtractResult(self, res)

    def extract_begin(self, frames, options=None, keep_all_planes=True, host_descriptors=True, input_ready=False):
        """First half of extract_features on a torch CUDA tensor [N, H, W] (uint8 or float32): enqueue the
        GPU work up to the extrema candidates and return a Job without synchronising.  input_ready=True (AKZ_INPUT_READY):
        the frames are complete in device memory now (nothing pending on any stream writes them): the first stages of a
        large batch then run ahead, under the kernels of the batch begun before."""
        import torch
        options = options or Config()
        flags = (AKZ_KEEP_ALL_PLANES if keep_all_planes else 0) | (0 if host_descriptors else AKZ_NO_HOST_DESCRIPTORS) | \
                (AKZ_INPUT_READY if input_ready else 0)
        t = frames
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
            raise ValueError("device images must be contiguous torch CUDA tensors")
        if t.dim() == 2:
            t = t.unsqueeze(0)
        n, h, w = t.shape
        fn = lib().akz_extract_begin_device_u8 if t.dtype == torch.uint8 else lib().akz_extract_begin_device_f32
        job = C.c_void_p()
        _check(fn(self._h, C.c_void_p(t.data_ptr()), w, h, n, C.byref(options), flags, C.byref(job)))
        return Job(self, job, t)

    def extract_begin_host(self, frames, options=None, keep_all_planes=True, host_descriptors=True):
        """extract_begin for frames in HOST memory (akz_extract_begin_host_*): a numpy array or a torch CPU tensor
        [N, H, W] (uint8 or float32), ideally pinned (torch .pin_memory()): the upload runs on the context's copy
        stream, under the kernels of the batch begun before.  The Job keeps the frames alive until it is finished."""
        import torch
        options = options or Config()
        flags = (AKZ_KEEP_ALL_PLANES if keep_all_planes else 0) | (0 if host_descriptors else AKZ_NO_HOST_DESCRIPTORS)
        t = frames
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        if not (isinstance(t, torch.Tensor) and not t.is_cuda and t.is_contiguous()):
            raise ValueError("host images must be contiguous numpy arrays or torch CPU tensors")
        if t.dtype not in (torch.uint8, torch.float32):
            raise ValueError("host images must be uint8 or float32")
        if t.dim() == 2:
            t = t.unsqueeze(0)
        n, h, w = t.shape
        fn = lib().akz_extract_begin_host_u8 if t.dtype == torch.uint8 else lib().akz_extract_begin_host_f32
        job = C.c_void_p()
        _check(fn(self._h, C.c_void_p(t.data_ptr()), w, h, n, C.byref(options), flags, C.byref(job)))
        return Job(self, job, t)

    def descriptor_match(self, d0, d1, distance_threshold=10000, lowes_ratio=0.86):
        """ops::feature_matching::descriptor_match (akaze/src/ops/feature_matching.rs:23-94)."""
        d0 = np.ascontiguousarray(d0, np.uint8)
        d1 = np.ascontiguousarray(d1, np.uint8)
        nb = d0.shape[1] if d0.ndim == 2 and d0.shape[0] else (d1.shape[1] if d1.ndim == 2 else 61)
        n0 = d0.shape[0] if d0.ndim == 2 else 0
        n1 = d1.shape[0] if d1.ndim == 2 else 0
        out = np.zeros(max(n0, 1), MATCH_DTYPE)
        n = C.c_uint64()
        _check(lib().akz_descriptor_match(self._h, d0.ctypes.data_as(C.c_void_p), n0, d1.ctypes.data_as(C.c_void_p),
                                          n1, nb, distance_threshold, lowes_ratio, out.ctypes.data_as(C.c_void_p),
                                          C.byref(n)))
        return out[:n.value].copy()

    def descriptor_match_sets_device(self, q, train, rows, distance_threshold=10000, lowes_ratio=0.86):
        """One query set against several train sets in one launch: q [n0, 64] uint8 CUDA tensor, train the sets'
        64-byte rows one after the other ([sum(rows), 64]), rows the row count of every set.  Returns (matches
        [n_sets, n0, 24] uint8 — akz_match records, set k's first counts[k] are valid, index_1 relative to the
        set — and counts [n_sets] int64), the same as one descriptor_match_device call per set."""
        import torch
        n0, ns = int(q.shape[0]), len(rows)
        out = torch.empty((ns, max(n0, 1), 24), dtype=torch.uint8, device=q.device)
        if n0 == 0:
            out = out[:, :0]
        cnt = torch.zeros(max(ns, 1), dtype=torch.int64, device=q.device)
        arr = (C.c_uint64 * max(ns, 1))(*[int(r) for r in rows])
        _check(lib().akz_descriptor_match_sets_device(self._h, C.c_void_p(q.data_ptr()) if n0 else None, n0,
                                                      C.c_void_p(train.data_ptr()) if sum(rows) else None, arr, ns,
                                                      distance_threshold, lowes_ratio, C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(cnt.data_ptr())))
        return out, cnt[:ns]

    def descriptor_match_sets_mutual_device(self, q, train, rows, distance_threshold=10000, lowes_ratio=0.86):
        """descriptor_match_sets_device plus the opposite direction of every block from the same pass: returns (matches,
        counts, col_matches [sum(rows), 24] uint8 -- set k's list starts at row sum(rows[:k]) --, col_counts [n_sets])."""
        import torch
        n0, ns, tot = int(q.shape[0]), len(rows), int(sum(rows))
        out = torch.empty((ns, max(n0, 1), 24), dtype=torch.uint8, device=q.device)
        if n0 == 0:
            out = out[:, :0]
        cnt = torch.zeros(max(ns, 1), dtype=torch.int64, device=q.device)
        cout = torch.empty((max(tot, 1), 24), dtype=torch.uint8, device=q.device)
        ccnt = torch.zeros(max(ns, 1), dtype=torch.int64, device=q.device)
        arr = (C.c_uint64 * max(ns, 1))(*[int(r) for r in rows])
        _check(lib().akz_descriptor_match_sets_mutual_device(
            self._h, C.c_void_p(q.data_ptr()) if n0 else None, n0, C.c_void_p(train.data_ptr()) if tot else None, arr, ns,
            distance_threshold, lowes_ratio, C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(cout.data_ptr()),
            C.c_void_p(ccnt.data_ptr())))
        return out, cnt[:ns], cout[:tot], ccnt[:ns]

    def descriptor_match_device(self, d0, d1, distance_threshold=10000, lowes_ratio=0.86):
        """Same on torch CUDA uint8 tensors of 64-byte descriptor rows; returns (matches tensor view, count)."""
        import torch
        n0, n1 = d0.shape[0], d1.shape[0]
        out = torch.empty((max(n0, 1), 24), dtype=torch.uint8, device=d0.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=d0.device)
        _check(lib().akz_descriptor_match_device(self._h, C.c_void_p(d0.data_ptr()), n0, C.c_void_p(d1.data_ptr()),
                                                 n1, distance_threshold, lowes_ratio, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(cnt.data_ptr())))
        return out, cnt

    # ---- per-op entry points on torch CUDA float32 tensors [N, H, W] or [H, W] -----------
    def _nhw(self, t):
        if t.dim() == 2:
            return 1, t.shape[0], t.shape[1]
        return t.shape[0], t.shape[1], t.shape[2]

    def _taps(self, taps):
        taps = np.ascontiguousarray(taps, np.float32)
        return taps, taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps)

    def horizontal_filter(self, img, taps):
        import torch
        n, h, w = self._nhw(img)
        out = torch.empty_like(img)
        k, pk, nk = self._taps(taps)
        _check(lib().akz_op_horizontal_filter(self._h, img.data_ptr(), out.data_ptr(), w, h, n, pk, nk))
        return out

    def vertical_filter(self, img, taps):
        import torch
        n, h, w = self._nhw(img)
        out = torch.empty_like(img)
        k, pk, nk = self._taps(taps)
        _check(lib().akz_op_vertical_filter(self._h, img.data_ptr(), out.data_ptr(), w, h, n, pk, nk))
        return out

    def gaussian_blur(self, img, sigma):
        import torch
        n, h, w = self._nhw(img)
        out = torch.empty(img.shape, dtype=torch.float32, device=img.device)
        fn = lib().akz_op_gaussian_blur_u8 if img.dtype == torch.uint8 else lib().akz_op_gaussian_blur
        _check(fn(self._h, img.data_ptr(), out.data_ptr(), w, h, n, sigma))
        return out

    def half_size(self, img):
        import torch
        n, h, w = self._nhw(img)
        shape = (h // 2, w // 2) if img.dim() == 2 else (n, h // 2, w // 2)
        out = torch.empty(shape, dtype=torch.float32, device=img.device)
        _check(lib().akz_op_half_size(self._h, img.data_ptr(), out.data_ptr(), w, h, n))
        return out

    def scharr(self, img, x_order, y_order, sigma_size):
        import torch
        n, h, w = self._nhw(img)
        out = torch.empty_like(img)
        _check(lib().akz_op_scharr(self._h, img.data_ptr(), out.data_ptr(), w, h, n, int(x_order), int(y_order),
                                   sigma_size))
        return out

    def debug_rcp_f64_to_f32(self, x):
        """(1.0 / x) as f32 for a CUDA float64 tensor, the way pm_g2 forms it inside the level kernels"""
        import torch
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _check(lib().akz_debug_rcp_f64_to_f32(self._h, x.data_ptr(), out.data_ptr(), x.numel()))
        return out

    def pm_g2(self, lx, ly, k):
        """k: python float or a torch CUDA float64 tensor with one value per image."""
        import torch
        n, h, w = self._nhw(lx)
        out = torch.empty_like(lx)
        kt = k if isinstance(k, torch.Tensor) else torch.full((n,), float(k), dtype=torch.float64, device=lx.device)
        _check(lib().akz_op_pm_g2(self._h, lx.data_ptr(), ly.data_ptr(), out.data_ptr(), w, h, n, kt.data_ptr()))
        return out

    def contrast_factor(self, img, percentile=0.7, gscale=1.0, nbins=300):
        import torch
        n, h, w = self._nhw(img)
        out = torch.zeros(n, dtype=torch.float64, device=img.device)
        _check(lib().akz_op_contrast_factor(self._h, img.data_ptr(), w, h, n, percentile, gscale, nbins,
                                            out.data_ptr()))
        return out

    def flow(self, lsmooth, k, k_scale_pow=0):
        import torch
        n, h, w = self._nhw(lsmooth)
        out = torch.empty_like(lsmooth)
        kt = k if isinstance(k, torch.Tensor) else torch.full((n,), float(k), dtype=torch.float64,
                                                              device=lsmooth.device)
        _check(lib().akz_op_flow(self._h, lsmooth.data_ptr(), out.data_ptr(), w, h, n, kt.data_ptr(), k_scale_pow))
        return out

    def fed_steps(self, lt, lflow, taus, want_lstep=False):
        """calculate_step for each tau, in place on `lt` (as the reference mutates evolution.Lt)."""
        import torch
        n, h, w = self._nhw(lt)
        taus = np.ascontiguousarray(taus, np.float64)
        lstep = torch.zeros_like(lt) if want_lstep else None
        _check(lib().akz_op_fed_steps(self._h, lt.data_ptr(), lflow.data_ptr(), lstep.data_ptr() if want_lstep else None,
                                      w, h, n, taus.ctypes.data_as(C.POINTER(C.c_double)), len(taus)))
        return lstep

    def detector_response(self, lsmooth, sigma_size, keep_second=True):
        import torch
        n, h, w = self._nhw(lsmooth)
        names = ["Lx", "Ly", "Lxx", "Lyy", "Lxy", "Ldet"]
        outs = {k: torch.empty_like(lsmooth) for k in names if keep_second or k in ("Lx", "Ly", "Ldet")}
        p = lambda k: outs[k].data_ptr() if k in outs else None
        _check(lib().akz_op_detector_response(self._h, lsmooth.data_ptr(), sigma_size, p("Lx"), p("Ly"), p("Lxx"),
                                              p("Lyy"), p("Lxy"), p("Ldet"), w, h, n))
        return outs


class Job:
    """An extraction in flight (akz_extract_begin_*); finish() exactly once."""

    def __init__(self, ctx, handle, frames):
        self._ctx, self._h, self._frames = ctx, handle, frames  # keep the frames alive

    def finish(self):
        res = C.c_void_p()
        h, self._h = self._h, None
        _check(lib().akz_extract_finish(h, C.byref(res)))
        self._frames = None
        return ExtractResult(self._ctx, res)

    def abandon(self):
        """Drop the extraction without a result (akz_job_abandon)."""
        if getattr(self, "_h", None):
            lib().akz_job_abandon(self._h)
            self._h = None
        self._frames = None

    def __del__(self):
        self.abandon()


class ExtractResult:
    """(Vec<EvolutionStep>, Vec<Keypoint>, Vec<Descriptor>) of akaze::extract_features, per image of the
    batch; EvolutionStep images stay on the GPU and are fetched lazily."""

    def __init__(self, ctx, handle):
        self._ctx = ctx
        self._h = handle
        n = C.c_uint64()
        _check(lib().akz_result_num_images(handle, C.byref(n)))
        self.num_images = n.value

    def close(self):
        if self._h:
            lib().akz_result_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self, img=0):
        nl, nk, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().akz_result_counts(self._h, img, C.byref(nl), C.byref(nk), C.byref(nb)))
        return nl.value, nk.value, nb.value

    def keypoints(self, img=0):
        _, nk, _ = self.counts(img)
        out = np.zeros(nk, KEYPOINT_DTYPE)
        _check(lib().akz_result_keypoints(self._h, img, out.ctypes.data_as(C.c_void_p)))
        return out

    def descriptors(self, img=0):
        _, nk, nb = self.counts(img)
        out = np.zeros((nk, nb), np.uint8)
        _check(lib().akz_result_descriptors(self._h, img, out.ctypes.data_as(C.c_void_p)))
        return out

    def device_descriptors(self, img=0):
        """(device address, rows) of the 64-byte descriptor rows of image `img`."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().akz_result_device_descriptors(self._h, img, C.byref(p), C.byref(n)))
        return p.value, n.value

    def copy_device_descriptors(self, dst):
        """D2D copy of every image's 64-byte descriptor rows into the torch CUDA uint8 tensor dst [rows, 64].  The copy runs on a
        stream of the context's own and is complete on return; what torch itself still has queued for `dst` (the fill of a
        torch.zeros) is waited for first, or it could land after the copy."""
        import torch
        torch.cuda.current_stream(dst.device).synchronize()
        n = C.c_uint64()
        _check(lib().akz_result_copy_device_descriptors(self._h, C.c_void_p(dst.data_ptr()), dst.shape[0],
                                                        C.byref(n)))
        return n.value

    def describe_keypoints(self, keypoints, img=0, compute_orientation=True):
        """compute_main_orientation + extract_descriptors for caller-supplied keypoints on the retained pyramid.
        Returns (keypoints with the angle filled in, descriptors[n, desc_bytes])."""
        kp = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE).copy()
        nb = self.counts(img)[2] or ((6 + 36 + 120) * 3 + 7) // 8
        desc = np.zeros((len(kp), nb), np.uint8)
        _check(lib().akz_result_describe_keypoints(self._h, img, kp.ctypes.data, len(kp), int(bool(compute_orientation)),
                                                   desc.ctypes.data))
        return kp, desc

    def write_evolutions(self, directory, img=0):
        """types::evolution::write_evolutions: one normalised PNG per plane and level."""
        os.makedirs(directory, exist_ok=True)
        _check(lib().akz_write_evolutions(self._h, img, os.fsencode(directory)))

    def contrast(self, img=0):
        k = C.c_double()
        _check(lib().akz_result_contrast(self._h, img, C.byref(k)))
        return k.value

    def level_info(self, lvl):
        et, es = C.c_double(), C.c_double()
        o, s, ss, w, h = (C.c_uint32() for _ in range(5))
        nt = C.c_uint64()
        tau = np.zeros(8192, np.float64)
        _check(lib().akz_result_level_info(self._h, lvl, C.byref(et), C.byref(es), C.byref(o), C.byref(s),
                                           C.byref(ss), C.byref(w), C.byref(h), C.byref(nt),
                                           tau.ctypes.data_as(C.POINTER(C.c_double)), len(tau)))
        return dict(etime=et.value, esigma=es.value, octave=o.value, sublevel=s.value, sigma_size=ss.value,
                    w=w.value, h=h.value, tau=tau[:nt.value].copy())

    def plane(self, lvl, name, img=0):
        pid = PLANES.index(name) if isinstance(name, str) else int(name)
        n = C.c_uint64()
        _check(lib().akz_fetch_plane(self._h, img, lvl, pid, None, C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 0), np.float32)
        info = self.level_info(lvl)
        out = np.empty((info["h"], info["w"]), np.float32)
        _check(lib().akz_fetch_plane(self._h, img, lvl, pid, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out


# ------------------------------------------------------------------------------------------
# module-level mirror of the reference's free functions
# ------------------------------------------------------------------------------------------
_default_ctx = None


def _take(ptr, n, dtype=np.uint8):
    """Copy n bytes out of a buffer the library allocated, then release it."""
    try:
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy().view(dtype)
    finally:
        lib().akz_image_free(ptr)


def load_image(path):
    """image::open(path): (h, w) uint8 for luma files, (h, w, 3) for colour ones."""
    w, h, ch, px = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p()
    _check(lib().akz_image_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(ch), C.byref(px)))
    a = _take(px, w.value * h.value * ch.value)
    return a.reshape(h.value, w.value) if ch.value == 1 else a.reshape(h.value, w.value, 3)


def load_image_luma(path):
    """image::open(path).to_luma(): (h, w) uint8 — the input of create_unit_float_image."""
    w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
    _check(lib().akz_image_load_luma(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)))
    return _take(px, w.value * h.value).reshape(h.value, w.value)


def load_image_rgb(path):
    w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
    _check(lib().akz_image_load_rgb(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)))
    return _take(px, w.value * h.value * 3).reshape(h.value, w.value, 3)


def save_png(path, pixels):
    a = np.ascontiguousarray(pixels, np.uint8)
    ch = 1 if a.ndim == 2 else a.shape[2]
    _check(lib().akz_image_save_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], ch))


def save_plane_png(path, plane):
    """types::image::save: min/max normalise, scale to 8 bit, write."""
    a = np.ascontiguousarray(plane, np.float32)
    _check(lib().akz_image_save_plane_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]))


def config_to_json(cfg=None):
    cfg = cfg or Config()
    n = C.c_uint64()
    buf = C.create_string_buffer(1024)
    _check(lib().akz_config_to_json(C.byref(cfg), buf, 1024, C.byref(n)))
    return buf.value.decode()


def config_from_json(text, base=None):
    cfg = base or Config()
    _check(lib().akz_config_from_json(text.encode(), C.byref(cfg)))
    return cfg


def random_seed(s0=42, s1=69):
    """random::default().seed([s0, s1]) for the calling thread: restarts the stream that RANSAC sampling and the
    drawing colours consume ([42, 69] is the state a fresh thread starts with)."""
    _check(lib().akz_random_seed(s0, s1))


def draw_keypoints(rgb, keypoints):
    """types::keypoint::draw_keypoints: returns a new RGB image with the keypoints blended in."""
    out = np.ascontiguousarray(rgb, np.uint8).copy()
    kp = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE)
    _check(lib().akz_draw_keypoints(out.ctypes.data, out.shape[1], out.shape[0], kp.ctypes.data, len(kp)))
    return out


def draw_matches(rgb0, rgb1, keypoints_0, keypoints_1, matches):
    a, b = np.ascontiguousarray(rgb0, np.uint8), np.ascontiguousarray(rgb1, np.uint8)
    k0, k1 = np.ascontiguousarray(keypoints_0, KEYPOINT_DTYPE), np.ascontiguousarray(keypoints_1, KEYPOINT_DTYPE)
    m = np.ascontiguousarray(matches, MATCH_DTYPE)
    w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
    _check(lib().akz_draw_matches(a.ctypes.data, a.shape[1], a.shape[0], b.ctypes.data, b.shape[1], b.shape[0],
                                  k0.ctypes.data, len(k0), k1.ctypes.data, len(k1), m.ctypes.data, len(m),
                                  C.byref(w), C.byref(h), C.byref(px)))
    return _take(px, w.value * h.value * 3).reshape(h.value, w.value, 3)


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def extract_features(image, options=None, ctx=None):
    """akaze::extract_features(image, options) -> (evolutions, keypoints, descriptors).
    `evolutions` is the ExtractResult (planes fetched lazily), keypoints a structured array,
    descriptors an [n, 61] uint8 array."""
    r = (ctx or default_context()).extract_features(image, options)
    return r, r.keypoints(0), r.descriptors(0)


def serialize_features_to_file(keypoints, descriptors, path):
    """akaze_util::serialize_features_to_file (akaze-util/src/lib.rs:17-30): ".json" -> serde_json, else bincode."""
    k = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE)
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(len(k), -1) if len(k) else np.zeros((0, 61), np.uint8)
    _check(lib().akz_write_features(os.fsencode(path), k.ctypes.data_as(C.c_void_p), len(k),
                                    d.ctypes.data_as(C.c_void_p), d.shape[1]))


def deserialize_features_from_file(path):
    """akaze_util::deserialize_features_from_file (lib.rs:33-42) -> (keypoints, descriptors[n, bytes])."""
    nk, nd, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _check(lib().akz_read_features(os.fsencode(path), None, None, 0, 0, C.byref(nk), C.byref(nd), C.byref(nb)))
    k = np.zeros(nk.value, KEYPOINT_DTYPE)
    d = np.zeros((nd.value, nb.value), np.uint8)
    _check(lib().akz_read_features(os.fsencode(path), k.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                   max(1, len(k)), max(1, d.size), C.byref(nk), C.byref(nd), C.byref(nb)))
    return k, d


def serialize_matches_to_file(matches, path):
    """akaze_util::serialize_matches_to_file (lib.rs:44-53)."""
    m = np.ascontiguousarray(matches, MATCH_DTYPE)
    _check(lib().akz_write_matches(os.fsencode(path), m.ctypes.data_as(C.c_void_p), len(m)))


def deserialize_matches_from_file(path):
    """akaze_util::deserialize_matches_from_file (lib.rs:56-67)."""
    n = C.c_uint64()
    _check(lib().akz_read_matches(os.fsencode(path), None, 0, C.byref(n)))
    m = np.zeros(n.value, MATCH_DTYPE)
    _check(lib().akz_read_matches(os.fsencode(path), m.ctypes.data_as(C.c_void_p), max(1, len(m)), C.byref(n)))
    return m


def estimate_fundamental_matrix(keypoints_0, keypoints_1, matches8, epsilon):
    """ops::estimate_fundamental_matrix::estimate_fundamental_matrix (:17-69): 3x3 float32 matrix or None."""
    k0 = np.ascontiguousarray(keypoints_0, KEYPOINT_DTYPE)
    k1 = np.ascontiguousarray(keypoints_1, KEYPOINT_DTYPE)
    m = np.ascontiguousarray(matches8, MATCH_DTYPE)
    if len(m) != 8:
        raise ValueError("exactly 8 matches")
    f = np.zeros(9, np.float32)
    found = C.c_int()
    _check(lib().akz_estimate_fundamental_matrix(k0.ctypes.data, len(k0), k1.ctypes.data, len(k1), m.ctypes.data, epsilon,
                                                 f.ctypes.data_as(C.POINTER(C.c_float)), C.byref(found)))
    return f.reshape(3, 3) if found.value else None


def remove_outliers(keypoints_0, keypoints_1, matches, num_trials, epsilon_model, epsilon_inlier):
    """ops::estimate_fundamental_matrix::remove_outliers (estimate_fundamental_matrix.rs:99-165); host only."""
    k0 = np.ascontiguousarray(keypoints_0, KEYPOINT_DTYPE)
    k1 = np.ascontiguousarray(keypoints_1, KEYPOINT_DTYPE)
    m = np.ascontiguousarray(matches, MATCH_DTYPE)
    out = np.zeros(max(1, len(m)), MATCH_DTYPE)
    n = C.c_uint64()
    _check(lib().akz_remove_outliers(k0.ctypes.data_as(C.c_void_p), len(k0), k1.ctypes.data_as(C.c_void_p), len(k1),
                                     m.ctypes.data_as(C.c_void_p), len(m), num_trials, epsilon_model, epsilon_inlier,
                                     out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[:n.value].copy()


def match_features(keypoints_0, descriptors_0, keypoints_1, descriptors_1, lowes_ratio, ransac_trials=None,
                   ransac_epsilon_inliers=None, ctx=None):
    """akaze::match_features (lib.rs:252-275): descriptor_match(d0, d1, 10000, ratio) on the GPU, then — when
    ransac_trials / ransac_epsilon_inliers are given, as in the reference signature — the RANSAC
    fundamental-matrix filter on the host.  Without them only the (bit-exact) descriptor stage runs."""
    c = ctx or default_context()
    if ransac_trials is None:
        return c.descriptor_match(descriptors_0, descriptors_1, 10000, lowes_ratio)
    k0 = np.ascontiguousarray(keypoints_0, KEYPOINT_DTYPE)
    k1 = np.ascontiguousarray(keypoints_1, KEYPOINT_DTYPE)
    d0 = np.ascontiguousarray(descriptors_0, np.uint8)
    d1 = np.ascontiguousarray(descriptors_1, np.uint8)
    out = np.zeros(max(1, len(d0)), MATCH_DTYPE)
    n = C.c_uint64()
    nb0 = d0.shape[1] if d0.ndim == 2 and len(d0) else None
    nb1 = d1.shape[1] if d1.ndim == 2 and len(d1) else None
    if nb0 is not None and nb1 is not None and nb0 != nb1:
        raise ValueError(f"descriptor lengths differ: {nb0} and {nb1} bytes")
    _check(lib().akz_match_features(c._h, k0.ctypes.data_as(C.c_void_p), len(k0), d0.ctypes.data_as(C.c_void_p), len(d0),
                                    k1.ctypes.data_as(C.c_void_p), len(k1), d1.ctypes.data_as(C.c_void_p), len(d1),
                                    nb0 or nb1 or 61, lowes_ratio, ransac_trials,
                                    ransac_epsilon_inliers, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[:n.value].copy()


# ------------------------------------------------------------------------------------------
# multi-GPU: per-image sharding and the one exchange step of the path
# ------------------------------------------------------------------------------------------
def shard_frames(num_frames, rank, world_size):
    """Frame indices owned by `rank`: image i -> GPU i mod G; extraction needs no
    collective."""
    return list(range(rank, num_frames, world_size))


COMM_ID_BYTES = 128


def comm_unique_id():
    """akz_comm_unique_id: rank 0 creates it, every rank receives the same 128 bytes (any transport)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().akz_comm_unique_id(buf))
    return bytes(buf)


class Gather:
    """One exchange in flight (akz_gather_begin*): finish() waits on the host, stream_wait() makes a stream wait."""

    def __init__(self, comm, handle):
        self._comm, self._h = comm, handle
        comm._gathers.add(self)  # akz_comm_destroy deletes its gather objects: Comm.close() invalidates the handles

    def stream_wait(self, stream):
        _check(lib().akz_gather_stream_wait(self._h, C.c_void_p(stream)))

    def blocks(self):
        """akz_gather_blocks (external transport): (device address of this rank's block, device address where the blocks of
        all ranks go, bytes per block)."""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(lib().akz_gather_blocks(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def deliver(self, stream=None):
        """akz_gather_deliver: every rank's block has been written where blocks() said, in the order of `stream`."""
        _check(lib().akz_gather_deliver(self._h, C.c_void_p(stream) if stream else None))

    def plan_all_pairs(self):
        """akz_pairs_plan: who matches what (akz_match_all_pairs' holder map and lead sets) without matching -- host arithmetic."""
        p = C.c_void_p()
        _check(lib().akz_pairs_plan(self._h, C.byref(p)))
        return Pairs(None, p)

    def exchange_over(self, group=None):
        """External transport over torch.distributed (any backend; gloo in the one-GPU rehearsals): this rank's block D2H,
        one all-gather of the fixed-size blocks, H2D, deliver.  Synchronous.  On a HOST communicator the blocks ARE host
        memory: the all-gather runs on them in place."""
        import torch
        import torch.distributed as dist
        send, recv, nbytes = self.blocks()
        world = self._comm.nranks
        if self._comm.host:
            h_send = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
            h_recv = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(world * nbytes,)))
            if world == 1 and not dist.is_initialized():
                h_recv.copy_(h_send)
            else:
                dist.all_gather_into_tensor(h_recv, h_send, group=group)
            self.deliver()
            return
        if world == 1 and not dist.is_initialized():  # a single rank without a process group: its block is the job
            copy_raw(recv, send, nbytes, 3)
            self.deliver()
            return
        stage = self._comm.__dict__.setdefault("_stage", {})   # pinned staging, kept per block size (pinning costs milliseconds)
        if stage.get("nbytes") != nbytes:
            stage.update(nbytes=nbytes, send=torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                         recv=torch.empty(world * nbytes, dtype=torch.uint8).pin_memory())
        h_send, h_recv = stage["send"], stage["recv"]
        copy_raw(h_send.data_ptr(), send, nbytes, 2)       # D2H
        dist.all_gather_into_tensor(h_recv, h_send, group=group)
        copy_raw(recv, h_recv.data_ptr(), world * nbytes, 1)  # H2D
        self.deliver()

    def finish(self, want_counts=True):
        """-> (device address of the blocks, rows per block, counts per rank, images per rank); rank r's descriptor
        rows start one 64-byte row into block r."""
        p, br = C.c_void_p(), C.c_uint64()
        n = self._comm.nranks
        cnt, img = (C.c_uint64 * n)(), (C.c_uint64 * n)()
        _check(lib().akz_gather_finish(self._h, C.byref(p), C.byref(br), cnt if want_counts else None,
                                       img if want_counts else None))
        return p.value, br.value, list(cnt), list(img)

    def image_rows(self, rank):
        """akz_gather_image_rows: rows of every image of `rank`'s shard (after finish())."""
        n = C.c_uint64()
        _check(lib().akz_gather_image_rows(self._h, int(rank), None, 0, C.byref(n)))
        out = (C.c_uint64 * max(1, n.value))()
        _check(lib().akz_gather_image_rows(self._h, int(rank), out, n.value, C.byref(n)))
        return [int(v) for v in out[:n.value]]

    def match_all_pairs(self, ctx, distance_threshold=10000, lowes_ratio=0.86):
        """akz_match_all_pairs (BASELINE configs[4]): every unordered image pair once, both directions, on the rank that owns
        the pair's lead image (Pairs.holder)."""
        p = C.c_void_p()
        _check(lib().akz_match_all_pairs(ctx._h, self._h, int(distance_threshold), float(lowes_ratio), C.byref(p)))
        return Pairs(ctx, p)

    def free(self):
        if self._h:
            lib().akz_gather_free(self._h)
            self._h = None
        self._comm._gathers.discard(self)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pairs:
    """Result of akz_match_all_pairs: both match lists of every image pair this rank holds (Pairs.holder / held)."""

    def __init__(self, ctx, handle):
        self._ctx, self._h = ctx, handle  # (the context must outlive the pairs object)
        n, f, o = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().akz_pairs_info(self._h, C.byref(n), C.byref(f), C.byref(o)))
        self.n_images, self.first_owned, self.n_owned = n.value, f.value, o.value

    def image_rows(self, image):
        r, o = C.c_uint64(), C.c_int32()
        _check(lib().akz_pairs_image_rows(self._h, int(image), C.byref(r), C.byref(o)))
        return r.value, o.value

    def count(self, query, image):
        n = C.c_uint64()
        _check(lib().akz_pairs_matches(self._h, int(query), int(image), None, 0, C.byref(n)))
        return n.value

    def matches(self, query, image):
        n = self.count(query, image)
        out = np.zeros(max(n, 1), MATCH_DTYPE)
        m = C.c_uint64()
        _check(lib().akz_pairs_matches(self._h, int(query), int(image), out.ctypes.data_as(C.c_void_p), n, C.byref(m)))
        return out[:m.value].copy()

    def holder(self, a, b):
        """akz_pairs_holder: the rank whose Pairs object holds both lists of the pair {a, b}."""
        r = C.c_int32()
        _check(lib().akz_pairs_holder(self._h, int(a), int(b), C.byref(r)))
        return r.value

    def held(self, rank):
        """ordered pairs (query, image) whose lists this rank holds"""
        return [(a, b) for a in range(self.n_images) for b in range(self.n_images) if a != b and self.holder(a, b) == rank]

    def lead_sets(self, lead_image):
        """akz_pairs_lead_sets: the images this rank matches `lead_image` (one it owns) against"""
        n = C.c_uint64()
        _check(lib().akz_pairs_lead_sets(self._h, int(lead_image), None, 0, C.byref(n)))
        out = (C.c_uint64 * max(1, n.value))()
        _check(lib().akz_pairs_lead_sets(self._h, int(lead_image), out, n.value, C.byref(n)))
        return [int(v) for v in out[:n.value]]

    def total_matches(self, rank=0):
        return sum(self.count(a, b) for a, b in self.held(rank))

    def totals(self):
        """akz_pairs_totals: (match lists held, records in them, descriptor pairs whose distance was formed); waits for the
        launches."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().akz_pairs_totals(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def free(self):
        if self._h:
            lib().akz_pairs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Comm:
    """Communicator of the descriptor exchange behind the C ABI (akz_comm_*): one per rank.  unique_id = the bytes of
    comm_unique_id() from rank 0: RCCL carries the blocks (akz_comm_create); unique_id = None: the caller does
    (akz_comm_create_external; Gather.blocks / Gather.deliver)."""

    HOST = -1  # AKZ_COMM_HOST: an external-transport communicator whose blocks and rows are HOST memory (no GPU call)

    def __init__(self, device, unique_id, rank, nranks):
        self._h = C.c_void_p()
        self.external = unique_id is None
        self.host = self.external and int(device) == Comm.HOST
        if self.external:
            _check(lib().akz_comm_create_external(int(device), int(rank), int(nranks), C.byref(self._h)))
        else:
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
            _check(lib().akz_comm_create(int(device), buf, int(rank), int(nranks), C.byref(self._h)))
        self.rank, self.nranks, self.device = int(rank), int(nranks), int(device)

    def set_timeout(self, seconds):
        """akz_comm_set_timeout: how long finish() waits for an exchange before AKZ_ERR_TIMEOUT (0: without limit)."""
        _check(lib().akz_comm_set_timeout(self._h, float(seconds)))

    def place_streams(self, ctx):
        """akz_comm_place_streams: the communicator's streams onto queues / pipes that the context's busy streams do not use
        (once, before the first step)."""
        _check(lib().akz_comm_place_streams(self._h, ctx._h))

    @property
    def _gathers(self):
        import weakref
        if "_gs" not in self.__dict__:
            self.__dict__["_gs"] = weakref.WeakSet()
        return self.__dict__["_gs"]

    def close(self):
        if self._h:
            for g in list(self._gathers):  # exchanges still held by the caller: retire them while their objects exist
                g.free()
            lib().akz_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_begin(self, results, cap_rows):
        """Enqueue the all-gather of the descriptor rows of `results` (ExtractResults of this rank, in order); returns
        once the local rows are copied, never waits for the collective."""
        arr = (C.c_void_p * max(1, len(results)))(*[r._h for r in results])
        g = C.c_void_p()
        _check(lib().akz_gather_begin(self._h, arr, len(results), int(cap_rows), C.byref(g)))
        return Gather(self, g)

    def gather_begin_rows(self, rows, cap_rows, producer_stream=None):
        """The same for a torch CUDA uint8 tensor [n, 64] that is complete in the order of producer_stream."""
        g = C.c_void_p()
        n = int(rows.shape[0])
        _check(lib().akz_gather_begin_rows(self._h, C.c_void_p(rows.data_ptr()) if n else None, n, int(cap_rows),
                                           C.c_void_p(producer_stream) if producer_stream else None, C.byref(g)))
        gg = Gather(self, g)
        gg._keep = rows
        return gg

    def gather_begin_image_rows(self, rows, rows_per_image, cap_rows, producer_stream=None):
        """akz_gather_begin_image_rows: the rows of several images back to back -- a torch CUDA uint8 tensor [n, 64], or, on a
        HOST communicator, a C-contiguous numpy uint8 array [n, 64]."""
        g = C.c_void_p()
        per = (C.c_uint64 * max(1, len(rows_per_image)))(*[int(v) for v in rows_per_image])
        n = int(rows.shape[0])
        assert n == sum(int(v) for v in rows_per_image)
        ptr = (rows.ctypes.data if self.host else rows.data_ptr()) if n else None
        _check(lib().akz_gather_begin_image_rows(self._h, C.c_void_p(ptr) if n else None, per, len(rows_per_image), int(cap_rows),
                                                 C.c_void_p(producer_stream) if producer_stream else None, C.byref(g)))
        gg = Gather(self, g)
        gg._keep = rows
        return gg

    def gather_descriptors(self, rows):
        """akz_gather_descriptors: synchronous; -> (torch uint8 [sum, 64] copy, counts)."""
        import torch
        n = int(rows.shape[0])
        p = C.c_void_p()
        cnt = (C.c_uint64 * self.nranks)()
        torch.cuda.current_stream(rows.device).synchronize()
        _check(lib().akz_gather_descriptors(self._h, C.c_void_p(rows.data_ptr()) if n else None, n, C.byref(p), cnt))
        counts = [int(v) for v in cnt]
        total = sum(counts)
        out = torch.empty((total, 64), dtype=torch.uint8, device=rows.device)
        if total:
            copy_d2d(out.data_ptr(), p.value, total * 64)
        return out, counts


_hip_memcpy = None
_hip_stream_sync = None


def copy_raw(dst, src, nbytes, kind):
    """hipMemcpy through the HIP runtime torch already loaded, COMPLETE on return (binding helper); kind 1 H2D, 2 D2H,
    3 D2D.  (A device-to-device hipMemcpy may return before the copy has run -- it is enqueued on the null stream, which work
    on a non-blocking stream does not wait for -- so that kind is followed by a synchronisation of the null stream.)"""
    global _hip_memcpy, _hip_stream_sync
    if _hip_memcpy is None:
        import torch
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        fn = hip.hipMemcpy
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        fn.restype = C.c_int
        sy = hip.hipStreamSynchronize
        sy.argtypes = [C.c_void_p]
        sy.restype = C.c_int
        _hip_memcpy, _hip_stream_sync = fn, sy
    if _hip_memcpy(C.c_void_p(dst), C.c_void_p(src), nbytes, int(kind)) != 0:
        raise AkazeError(-2, "hipMemcpy failed")
    if int(kind) == 3 and _hip_stream_sync(None) != 0:
        raise AkazeError(-2, "hipStreamSynchronize failed")


def copy_d2d(dst, src, nbytes):
    """Synchronous device-to-device copy (binding helper)."""
    copy_raw(dst, src, nbytes, 3)


def gather_descriptor_rows(local_rows, group=None, cap_rows=None):
    """All-gather of 64-byte descriptor rows before a cross-image brute-force match: every rank
    contributes a [n_r, 64] uint8 tensor (any n_r >= 0).  Two collectives: the row counts, then the rows
    padded to a common capacity.  With the "nccl" backend this is RCCL over xGMI; tensors stay on the GPU.

    cap_rows=None: capacity = the largest shard (needs one host sync to read the counts); returns
    ([sum n_r, 64] rows in rank order, counts per rank as a list).
    cap_rows=int:  fixed capacity, NO host synchronisation — returns the padded [world, cap_rows, 64]
    tensor and the device tensor of counts; rows beyond a rank's count are zero.  A shard that does not fit
    contributes no rows: its count (> cap_rows, seen by every rank) says so and the caller repeats the
    exchange with a larger capacity on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_rows.device
    n_local = int(local_rows.shape[0])
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    if cap_rows is not None:
        # a shard that does not fit still takes part (a rank that skipped the collective would leave its peers waiting
        # in it): it contributes no rows, and every rank sees the overflow in the gathered counts
        padded = torch.zeros((cap_rows, 64), dtype=torch.uint8, device=dev)
        if 0 < n_local <= cap_rows:
            padded[:n_local] = local_rows
        gathered = torch.empty((world * cap_rows, 64), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, padded, group=group)
        return gathered.view(world, cap_rows, 64), cnts
    counts = [int(v) for v in cnts.tolist()]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
    if n_local:
        padded[:n_local] = local_rows
    gathered = torch.empty((world * cap, 64), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    rows = torch.cat([gathered[r * cap:r * cap + counts[r]] for r in range(world)], dim=0)
    return rows, counts


def gather_descriptor_sets(local_sets, group=None):
    """The exchange step of a cross-GPU all-pairs match (BASELINE configs[4]): every rank
    contributes the descriptor rows of ITS images ([n_i, 64] uint8 tensors, in its shard order); afterwards every
    rank holds every image's rows.  Three all-gathers (images per rank, rows per image, the rows); with the "nccl"
    backend they are RCCL over xGMI and the tensors stay on the GPU.

    Returns (sets, owners): one rows tensor per image of the whole job, rank-major, and the rank that owns it."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_sets[0].device if local_sets else torch.device("cpu")
    n_img = torch.tensor([len(local_sets)], dtype=torch.int64, device=dev)
    imgs = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(imgs, n_img, group=group)
    imgs = [int(v) for v in imgs.tolist()]
    cap = max(max(imgs), 1)
    mine = torch.zeros(cap, dtype=torch.int64, device=dev)
    if local_sets:
        mine[:len(local_sets)] = torch.tensor([int(t.shape[0]) for t in local_sets], dtype=torch.int64, device=dev)
    per_img = torch.zeros(world * cap, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(per_img, mine, group=group)
    per_img = per_img.view(world, cap).tolist()
    local = torch.cat(local_sets, dim=0) if local_sets else torch.zeros((0, 64), dtype=torch.uint8, device=dev)
    rows, _ = gather_descriptor_rows(local, group)
    sets, owners, pos = [], [], 0
    for r in range(world):
        for i in range(imgs[r]):
            n = int(per_img[r][i])
            sets.append(rows[pos:pos + n])
            owners.append(r)
            pos += n
    return sets, owners


def pairs_lead(a, b):
    """The image of the unordered pair {a, b} that serves as the query set of its block (akz_match_all_pairs' rule: the
    lower one if the indices differ by an odd number, else the higher one -- every image leads about half of its pairs)."""
    lo, hi = min(a, b), max(a, b)
    return lo if (hi - lo) & 1 else hi


def all_pairs_match(local_sets, match_fn, group=None, match_sets_fn=None):
    """Cross-GPU all-pairs Hamming match: after gather_descriptor_sets every UNORDERED image pair is matched once, in both
    directions, by the rank that owns the pair's lead image (pairs_lead); nothing else is exchanged.  match_fn(rows_i,
    rows_j) is Context.descriptor_match_device on the GPUs.  With match_sets_fn(rows_i, train_rows, rows_per_set) ->
    ([matches of rows_i against set 0, ...], [matches of set 0 against rows_i, ...]) (Context.
    descriptor_match_sets_mutual_device: one launch per lead image, both directions from one pass) match_fn is not called.

    Returns {(i, j): matches} for the ordered pairs this rank holds, i and j being global (rank-major) image indices."""
    import torch
    import torch.distributed as dist
    sets, owners = gather_descriptor_sets(local_sets, group)
    rank = dist.get_rank(group)
    out = {}
    for i, owner in enumerate(owners):
        if owner != rank:
            continue
        led = [j for j in range(len(sets)) if j != i and pairs_lead(i, j) == i]
        if not led:
            continue
        if match_sets_fn is not None:
            fwd, rev = match_sets_fn(sets[i], torch.cat([sets[j] for j in led], dim=0), [int(sets[j].shape[0]) for j in led])
            for k, j in enumerate(led):
                out[(i, j)] = fwd[k]
                out[(j, i)] = rev[k]
            continue
        for j in led:
            out[(i, j)] = match_fn(sets[i], sets[j])
            out[(j, i)] = match_fn(sets[j], sets[i])
    return out

