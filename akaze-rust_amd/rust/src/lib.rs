//! Drop-in replacement for the `akaze` crate (indianajohn/akaze-rust) over the MI355X HIP library.
//!
//! Same crate name, module tree and public signatures as the reference:
//! `akaze::{extract_features, match_features}`, `akaze::types::{image, evolution, keypoint, feature_match}` and
//! `akaze::ops::*`.  Every numeric routine goes through the C ABI of `include/akaze_hip.h` (`libakaze_hip.so`); the
//! Rust side only owns host data (`Vec<f32>` images, the evolution pyramid, keypoint / descriptor / match vectors)
//! and turns non-zero status codes into panics — the reference's own error behaviour (`unwrap()` at
//! akaze/src/lib.rs:171).
//!
//! Every `pub` item names the reference item it stands for (paths relative to the reference repository).
//!
//! UNVERIFIED: the build image has no Rust toolchain, so this file has never been compiled; `tools/check_shim.py`
//! checks its item names and argument lists against the reference sources and its `extern "C"` block against
//! `include/akaze_hip.h`.
#![allow(non_snake_case)]
#![allow(clippy::too_many_arguments)]

use std::os::raw::{c_char, c_int, c_void};
use std::path::PathBuf;

// ------------------------------------------------------------------------------------------------
// FFI: the C ABI of include/akaze_hip.h
// ------------------------------------------------------------------------------------------------
#[repr(C)]
#[derive(Copy, Clone, Default)]
pub(crate) struct AkzKeypoint {
    x: f32,
    y: f32,
    response: f32,
    size: f32,
    octave: u64,
    class_id: u64,
    angle: f32,
    _pad: u32,
}

pub(crate) mod ffi {
    use super::AkzKeypoint;
    use crate::types::evolution::Config;
    use crate::types::feature_match::Match;
    use std::os::raw::{c_char, c_int, c_void};
    extern "C" {
        pub fn akz_last_error() -> *const c_char;
        pub fn akz_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut c_void) -> c_int;
        pub fn akz_device_malloc(ctx: *mut c_void, bytes: usize, d_out: *mut *mut c_void) -> c_int;
        pub fn akz_device_free(ctx: *mut c_void, d_ptr: *mut c_void) -> c_int;
        pub fn akz_memcpy_h2d(ctx: *mut c_void, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
        pub fn akz_memcpy_d2h(ctx: *mut c_void, dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
        pub fn akz_fed_tau_by_process_time(T: f64, M: c_int, tau_max: f64, reordering: c_int, out: *mut f64, cap: u64,
                                           n: *mut u64) -> c_int;
        pub fn akz_plan_num_levels(w: u32, h: u32, cfg: *const Config, n_levels: *mut u64) -> c_int;
        pub fn akz_plan_level_info(w: u32, h: u32, cfg: *const Config, level: u64, etime: *mut f64, esigma: *mut f64,
                                   octave: *mut u32, sublevel: *mut u32, sigma_size: *mut u32, level_w: *mut u32,
                                   level_h: *mut u32, detector_sigma: *mut u32, n_tau: *mut u64, tau: *mut f64,
                                   tau_cap: u64) -> c_int;
        pub fn akz_op_horizontal_filter(ctx: *mut c_void, d_in: *const f32, d_out: *mut f32, w: u32, h: u32, n: u32,
                                        taps: *const f32, ntaps: u32) -> c_int;
        pub fn akz_op_vertical_filter(ctx: *mut c_void, d_in: *const f32, d_out: *mut f32, w: u32, h: u32, n: u32,
                                      taps: *const f32, ntaps: u32) -> c_int;
        pub fn akz_op_gaussian_blur(ctx: *mut c_void, d_in: *const f32, d_out: *mut f32, w: u32, h: u32, n: u32,
                                    sigma: f32) -> c_int;
        pub fn akz_op_half_size(ctx: *mut c_void, d_in: *const f32, d_out: *mut f32, w: u32, h: u32, n: u32) -> c_int;
        pub fn akz_op_scharr(ctx: *mut c_void, d_in: *const f32, d_out: *mut f32, w: u32, h: u32, n: u32, x_order: c_int,
                             y_order: c_int, sigma_size: u32) -> c_int;
        pub fn akz_op_contrast_factor(ctx: *mut c_void, d_in: *const f32, w: u32, h: u32, n: u32, percentile: f64,
                                      gradient_histogram_scale: f64, num_bins: u64, d_k_out: *mut f64) -> c_int;
        pub fn akz_op_fed_steps(ctx: *mut c_void, d_lt: *mut f32, d_lflow: *const f32, d_lstep: *mut f32, w: u32, h: u32,
                                n: u32, taus: *const f64, n_tau: u32) -> c_int;
        pub fn akz_op_detector_response(ctx: *mut c_void, d_lsmooth: *const f32, sigma_size: u32, d_lx: *mut f32,
                                        d_ly: *mut f32, d_lxx: *mut f32, d_lyy: *mut f32, d_lxy: *mut f32,
                                        d_ldet: *mut f32, w: u32, h: u32, n: u32) -> c_int;
        pub fn akz_extract_gray_u8(ctx: *mut c_void, img: *const u8, w: u32, h: u32, cfg: *const Config, flags: u32,
                                   out: *mut *mut c_void) -> c_int;
        pub fn akz_extract_from_planes(ctx: *mut c_void, w: u32, h: u32, cfg: *const Config, planes: *const *const f32,
                                       n_levels: u64, flags: u32, out: *mut *mut c_void) -> c_int;
        pub fn akz_result_free(res: *mut c_void) -> c_int;
        pub fn akz_result_counts(res: *const c_void, img: u64, n_levels: *mut u64, n_keypoints: *mut u64,
                                 desc_bytes: *mut u64) -> c_int;
        pub fn akz_result_keypoints(res: *const c_void, img: u64, out: *mut AkzKeypoint) -> c_int;
        pub fn akz_result_descriptors(res: *const c_void, img: u64, out: *mut u8) -> c_int;
        pub fn akz_result_describe_keypoints(res: *const c_void, img: u64, kps: *mut AkzKeypoint, n_keypoints: u64,
                                             compute_orientation: c_int, descriptors: *mut u8) -> c_int;
        pub fn akz_result_level_info(res: *const c_void, level: u64, etime: *mut f64, esigma: *mut f64, octave: *mut u32,
                                     sublevel: *mut u32, sigma_size: *mut u32, w: *mut u32, h: *mut u32, n_tau: *mut u64,
                                     tau: *mut f64, tau_cap: u64) -> c_int;
        pub fn akz_fetch_plane(res: *const c_void, img: u64, level: u64, plane: c_int, out: *mut f32, n_px: *mut u64) -> c_int;
        pub fn akz_descriptor_match(ctx: *mut c_void, d0: *const u8, n0: u64, d1: *const u8, n1: u64, desc_bytes: u64,
                                    distance_threshold: u64, lowes_ratio: f64, out: *mut Match, n_out: *mut u64) -> c_int;
        pub fn akz_remove_outliers(keypoints_0: *const AkzKeypoint, n0: u64, keypoints_1: *const AkzKeypoint, n1: u64,
                                   matches: *const Match, n_matches: u64, num_trials: u64, epsilon_model: f32,
                                   epsilon_inlier: f32, out: *mut Match, n_out: *mut u64) -> c_int;
        pub fn akz_estimate_fundamental_matrix(keypoints_0: *const AkzKeypoint, n0: u64, keypoints_1: *const AkzKeypoint,
                                               n1: u64, matches8: *const Match, epsilon: f32, f: *mut f32,
                                               found: *mut c_int) -> c_int;
        pub fn akz_match_features(ctx: *mut c_void, keypoints_0: *const AkzKeypoint, n_keypoints_0: u64,
                                  descriptors_0: *const u8, n_descriptors_0: u64, keypoints_1: *const AkzKeypoint,
                                  n_keypoints_1: u64, descriptors_1: *const u8, n_descriptors_1: u64, desc_bytes: u64,
                                  lowes_ratio: f64, ransac_trials: u64, ransac_epsilon_inliers: f32, out: *mut Match,
                                  n_out: *mut u64) -> c_int;
        pub fn akz_image_save_plane_png(path: *const c_char, plane: *const f32, width: u32, height: u32) -> c_int;
        pub fn akz_image_free(pixels: *mut c_void);
        pub fn akz_random_color(rgb: *mut u8) -> c_int;
        pub fn akz_draw_circle(rgb: *mut u8, width: u32, height: u32, x: f32, y: f32, color: *const u8, radius: f32) -> c_int;
        pub fn akz_draw_line(rgb: *mut u8, width: u32, height: u32, x0: f32, y0: f32, x1: f32, y1: f32, color: *const u8,
                             radius: f32) -> c_int;
        pub fn akz_draw_keypoints(rgb: *mut u8, width: u32, height: u32, kps: *const AkzKeypoint, n: u64) -> c_int;
        pub fn akz_draw_matches(rgb0: *const u8, w0: u32, h0: u32, rgb1: *const u8, w1: u32, h1: u32,
                                kp0: *const AkzKeypoint, n0: u64, kp1: *const AkzKeypoint, n1: u64, matches: *const Match,
                                n_matches: u64, out_w: *mut u32, out_h: *mut u32, out_rgb: *mut *mut u8) -> c_int;
    }
}

pub(crate) fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(ffi::akz_last_error()) }.to_string_lossy().into_owned();
        panic!("akaze_hip status {}: {}", status, msg);
    }
}

thread_local! {
    // one context per host thread (the C ABI's threading rule); device from AKAZE_HIP_DEVICE
    static CTX: *mut c_void = {
        let dev = std::env::var("AKAZE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx: *mut c_void = std::ptr::null_mut();
        check(unsafe { ffi::akz_ctx_create(dev, std::ptr::null_mut(), &mut ctx) });
        ctx
    };
}
pub(crate) fn ctx() -> *mut c_void {
    CTX.with(|c| *c)
}

const AKZ_KEEP_ALL_PLANES: u32 = 1;
const AKZ_NO_DETECT: u32 = 4;

pub(crate) fn raw_keypoints(ks: &[types::keypoint::Keypoint]) -> Vec<AkzKeypoint> {
    ks.iter()
        .map(|k| AkzKeypoint {
            x: k.point.0,
            y: k.point.1,
            response: k.response,
            size: k.size,
            octave: k.octave as u64,
            class_id: k.class_id as u64,
            angle: k.angle,
            _pad: 0,
        })
        .collect()
}
pub(crate) fn to_keypoints(raw: &[AkzKeypoint]) -> Vec<types::keypoint::Keypoint> {
    raw.iter()
        .map(|k| types::keypoint::Keypoint {
            point: (k.x, k.y),
            response: k.response,
            size: k.size,
            octave: k.octave as usize,
            class_id: k.class_id as usize,
            angle: k.angle,
        })
        .collect()
}
pub(crate) fn flat_descriptors(ds: &[types::keypoint::Descriptor]) -> (Vec<u8>, usize) {
    let nb = ds.first().map(|d| d.vector.len()).unwrap_or(0);
    for d in ds {
        assert!(d.vector.len() == nb, "descriptors of different lengths");
    }
    (ds.iter().flat_map(|d| d.vector.iter().cloned()).collect(), nb)
}

/// A device plane for the duration of one op: upload, run, download.
pub(crate) struct DevPlane {
    ptr: *mut c_void,
    len: usize,
}
impl DevPlane {
    pub(crate) fn zeros(len: usize) -> DevPlane {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { ffi::akz_device_malloc(ctx(), len.max(1) * 4, &mut p) });
        DevPlane { ptr: p, len }
    }
    pub(crate) fn upload(data: &[f32]) -> DevPlane {
        let d = DevPlane::zeros(data.len());
        if !data.is_empty() {
            check(unsafe { ffi::akz_memcpy_h2d(ctx(), d.ptr, data.as_ptr() as *const c_void, data.len() * 4) });
        }
        d
    }
    pub(crate) fn download(&self) -> Vec<f32> {
        let mut out = vec![0f32; self.len];
        if self.len > 0 {
            check(unsafe { ffi::akz_memcpy_d2h(ctx(), out.as_mut_ptr() as *mut c_void, self.ptr, self.len * 4) });
        }
        out
    }
    pub(crate) fn f32(&self) -> *mut f32 {
        self.ptr as *mut f32
    }
}
impl Drop for DevPlane {
    fn drop(&mut self) {
        unsafe { ffi::akz_device_free(ctx(), self.ptr) };
    }
}

// ------------------------------------------------------------------------------------------------
// types
// ------------------------------------------------------------------------------------------------
pub mod types {
    /// akaze/src/types/image.rs
    pub mod image {
        use crate::{check, ctx, ffi, DevPlane};
        use image::{DynamicImage, GrayImage, RgbImage};
        use std::path::PathBuf;

        /// akaze/src/types/image.rs:32-36: row-major `Vec<f32>`, index `w*y + x`.
        #[derive(Debug, Clone)]
        pub struct GrayFloatImage {
            pub buffer: Vec<f32>,
            width: usize,
            height: usize,
        }
        /// akaze/src/types/image.rs:37-76
        pub trait ImageFunctions {
            fn width(&self) -> usize;
            fn height(&self) -> usize;
            fn new(width: usize, height: usize) -> Self;
            fn half_size(&self) -> Self;
            fn get(&self, x: usize, y: usize) -> f32;
            fn put(&mut self, x: usize, y: usize, pixel_value: f32);
        }
        impl ImageFunctions for GrayFloatImage {
            fn width(&self) -> usize {
                self.width
            }
            fn height(&self) -> usize {
                self.height
            }
            fn new(width: usize, height: usize) -> Self {
                GrayFloatImage { buffer: vec![0f32; width * height], width, height }
            }
            fn get(&self, x: usize, y: usize) -> f32 {
                self.buffer[self.width * y + x]
            }
            fn put(&mut self, x: usize, y: usize, pixel_value: f32) {
                self.buffer[self.width * y + x] = pixel_value;
            }
            /// akaze/src/types/image.rs:102-118 — `akz_op_half_size`
            fn half_size(&self) -> Self {
                let (w, h) = (self.width / 2, self.height / 2);
                let src = DevPlane::upload(&self.buffer);
                let dst = DevPlane::zeros(w * h);
                check(unsafe { ffi::akz_op_half_size(ctx(), src.f32(), dst.f32(), self.width as u32, self.height as u32, 1) });
                GrayFloatImage::from_buffer(dst.download(), w, h)
            }
        }
        impl GrayFloatImage {
            pub(crate) fn from_buffer(buffer: Vec<f32>, width: usize, height: usize) -> GrayFloatImage {
                debug_assert!(buffer.len() == width * height);
                GrayFloatImage { buffer, width, height }
            }
        }

        /// akaze/src/types/image.rs:127-140 (`f32::from(v) * 1f32 / 255f32` per luma byte)
        pub fn create_unit_float_image(input_image: &DynamicImage) -> GrayFloatImage {
            let gray: GrayImage = input_image.to_luma();
            let (w, h) = (gray.width() as usize, gray.height() as usize);
            let buffer = gray.into_raw().into_iter().map(|v| f32::from(v) * 1f32 / 255f32).collect();
            GrayFloatImage::from_buffer(buffer, w, h)
        }
        /// akaze/src/types/image.rs:148-160
        pub fn create_dynamic_image(input_image: &GrayFloatImage) -> DynamicImage {
            let bytes: Vec<u8> = input_image.buffer.iter().map(|v| (*v * 255f32) as u8).collect();
            DynamicImage::ImageLuma8(
                GrayImage::from_raw(input_image.width() as u32, input_image.height() as u32, bytes).unwrap(),
            )
        }
        /// akaze/src/types/image.rs:168-197 (min / max to [0, 1])
        pub fn normalize(input_image: &GrayFloatImage) -> GrayFloatImage {
            let mut lo = std::f32::MAX;
            let mut hi = std::f32::MIN;
            for v in &input_image.buffer {
                lo = lo.min(*v);
                hi = hi.max(*v);
            }
            let range = hi - lo;
            let buffer = input_image.buffer.iter().map(|v| (*v - lo) / range).collect();
            GrayFloatImage::from_buffer(buffer, input_image.width(), input_image.height())
        }
        /// akaze/src/types/image.rs:204-210 — `akz_image_save_plane_png` (normalise, scale to 8 bit, write)
        pub fn save(input_image: &GrayFloatImage, path: PathBuf) {
            if input_image.buffer.is_empty() {
                return;
            }
            let c = std::ffi::CString::new(path.to_string_lossy().into_owned()).unwrap();
            check(unsafe {
                ffi::akz_image_save_plane_png(c.as_ptr(), input_image.buffer.as_ptr(), input_image.width() as u32,
                                              input_image.height() as u32)
            });
        }
        /// akaze/src/types/image.rs:218-231 (despite its name the reference ADDS the second image)
        pub fn sqrt_squared(image_1: &mut GrayFloatImage, image_2: &GrayFloatImage) {
            debug_assert!(image_1.width() == image_2.width() && image_1.height() == image_2.height());
            for (a, b) in image_1.buffer.iter_mut().zip(image_2.buffer.iter()) {
                *a += *b;
            }
        }
        /// akaze/src/types/image.rs:239-260: every pixel takes the value at its coordinates clamped to the interior
        pub fn fill_border(output: &mut GrayFloatImage, half_width: usize) {
            let (w, h) = (output.width(), output.height());
            if w <= 2 * half_width || h <= 2 * half_width {
                return;
            }
            for y in 0..h {
                let cy = y.max(half_width).min(h - 1 - half_width);
                for x in 0..w {
                    let cx = x.max(half_width).min(w - 1 - half_width);
                    if cx != x || cy != y {
                        let v = output.get(cx, cy);
                        output.put(x, y, v);
                    }
                }
            }
        }
        fn filter(image: &GrayFloatImage, kernel: &[f32], horizontal: bool) -> GrayFloatImage {
            let (w, h) = (image.width(), image.height());
            let src = DevPlane::upload(&image.buffer);
            let dst = DevPlane::zeros(w * h);
            let f = if horizontal { ffi::akz_op_horizontal_filter } else { ffi::akz_op_vertical_filter };
            check(unsafe { f(ctx(), src.f32(), dst.f32(), w as u32, h as u32, 1, kernel.as_ptr(), kernel.len() as u32) });
            GrayFloatImage::from_buffer(dst.download(), w, h)
        }
        /// akaze/src/types/image.rs:270-295 — `akz_op_horizontal_filter` (incl. fill_border)
        pub fn horizontal_filter(image: &GrayFloatImage, kernel: &[f32]) -> GrayFloatImage {
            filter(image, kernel, true)
        }
        /// akaze/src/types/image.rs:305-332 — `akz_op_vertical_filter`
        pub fn vertical_filter(image: &GrayFloatImage, kernel: &[f32]) -> GrayFloatImage {
            filter(image, kernel, false)
        }
        /// akaze/src/types/image.rs:374-380 — `akz_op_gaussian_blur`
        pub fn gaussian_blur(image: &GrayFloatImage, r: f32) -> GrayFloatImage {
            let (w, h) = (image.width(), image.height());
            let src = DevPlane::upload(&image.buffer);
            let dst = DevPlane::zeros(w * h);
            check(unsafe { ffi::akz_op_gaussian_blur(ctx(), src.f32(), dst.f32(), w as u32, h as u32, 1, r) });
            GrayFloatImage::from_buffer(dst.download(), w, h)
        }
        /// akaze/src/types/image.rs:385-392 — the calling thread's default random source, as in the `random` crate
        pub fn random_color() -> (u8, u8, u8) {
            let mut c = [0u8; 3];
            check(unsafe { ffi::akz_random_color(c.as_mut_ptr()) });
            (c[0], c[1], c[2])
        }
        /// akaze/src/types/image.rs:418-443
        pub fn draw_circle(input_image: &mut RgbImage, point: (f32, f32), rgb: (u8, u8, u8), radius: f32) {
            let (w, h) = (input_image.width(), input_image.height());
            let c = [rgb.0, rgb.1, rgb.2];
            check(unsafe { ffi::akz_draw_circle(input_image.as_mut_ptr(), w, h, point.0, point.1, c.as_ptr(), radius) });
        }
        /// akaze/src/types/image.rs:453-480
        pub fn draw_line(input_image: &mut RgbImage, point_0: (f32, f32), point_1: (f32, f32), rgb: (u8, u8, u8), radius: f32) {
            let (w, h) = (input_image.width(), input_image.height());
            let c = [rgb.0, rgb.1, rgb.2];
            check(unsafe {
                ffi::akz_draw_line(input_image.as_mut_ptr(), w, h, point_0.0, point_0.1, point_1.0, point_1.1, c.as_ptr(), radius)
            });
        }
    }

    /// akaze/src/types/evolution.rs
    pub mod evolution {
        use crate::types::image::{save, GrayFloatImage, ImageFunctions};
        use crate::{check, ffi};
        use serde::{Deserialize, Serialize};
        use std::path::PathBuf;

        /// Same fields, order and defaults as the reference (akaze/src/types/evolution.rs:8-55); `#[repr(C)]` so that
        /// it is also `akz_config`.
        #[repr(C)]
        #[derive(Debug, Copy, Clone, Serialize, Deserialize)]
        pub struct Config {
            pub num_sublevels: u32,
            pub max_octave_evolution: u32,
            pub base_scale_offset: f64,
            pub initial_contrast: f64,
            pub contrast_percentile: f64,
            pub contrast_factor_num_bins: usize,
            pub derivative_factor: f64,
            pub detector_threshold: f64,
            pub descriptor_channels: usize,
            pub descriptor_pattern_size: usize,
        }
        impl Default for Config {
            fn default() -> Config {
                Config {
                    num_sublevels: 4,
                    max_octave_evolution: 4,
                    base_scale_offset: 1.6,
                    initial_contrast: 0.001,
                    contrast_percentile: 0.7,
                    contrast_factor_num_bins: 300,
                    derivative_factor: 1.5,
                    detector_threshold: 0.001,
                    descriptor_channels: 3,
                    descriptor_pattern_size: 10,
                }
            }
        }

        /// akaze/src/types/evolution.rs:59-92: five scalars, ten images, the FED step sizes — all public, as there.
        #[derive(Debug)]
        pub struct EvolutionStep {
            pub etime: f64,
            pub esigma: f64,
            pub octave: u32,
            pub sublevel: u32,
            pub sigma_size: u32,
            pub Lt: GrayFloatImage,
            pub Lsmooth: GrayFloatImage,
            pub Lx: GrayFloatImage,
            pub Ly: GrayFloatImage,
            pub Lxx: GrayFloatImage,
            pub Lyy: GrayFloatImage,
            pub Lxy: GrayFloatImage,
            pub Lflow: GrayFloatImage,
            pub Lstep: GrayFloatImage,
            pub Ldet: GrayFloatImage,
            pub fed_tau_steps: Vec<f64>,
        }
        impl EvolutionStep {
            pub(crate) fn empty(etime: f64, esigma: f64, octave: u32, sublevel: u32, sigma_size: u32, tau: Vec<f64>) -> EvolutionStep {
                let z = || GrayFloatImage::new(0, 0);
                EvolutionStep {
                    etime, esigma, octave, sublevel, sigma_size,
                    Lt: z(), Lsmooth: z(), Lx: z(), Ly: z(), Lxx: z(), Lyy: z(), Lxy: z(), Lflow: z(), Lstep: z(), Ldet: z(),
                    fed_tau_steps: tau,
                }
            }
            /// the ten images in the order of `akz_plane` (= the reference's field order)
            pub(crate) fn planes(&self) -> [&GrayFloatImage; 10] {
                [&self.Lt, &self.Lsmooth, &self.Lx, &self.Ly, &self.Lxx, &self.Lyy, &self.Lxy, &self.Lflow, &self.Lstep, &self.Ldet]
            }
            pub(crate) fn planes_mut(&mut self) -> [&mut GrayFloatImage; 10] {
                [&mut self.Lt, &mut self.Lsmooth, &mut self.Lx, &mut self.Ly, &mut self.Lxx, &mut self.Lyy, &mut self.Lxy,
                 &mut self.Lflow, &mut self.Lstep, &mut self.Ldet]
            }
        }

        /// akaze/src/types/evolution.rs:135-161 — `akz_plan_num_levels` / `akz_plan_level_info` (0x0 images, as there)
        pub fn allocate_evolutions(width: u32, height: u32, options: Config) -> Vec<EvolutionStep> {
            let mut n = 0u64;
            check(unsafe { ffi::akz_plan_num_levels(width, height, &options, &mut n) });
            let mut out_vec = Vec::with_capacity(n as usize);
            for level in 0..n {
                let (mut etime, mut esigma) = (0f64, 0f64);
                let (mut octave, mut sublevel, mut sigma_size, mut lw, mut lh, mut ds) = (0u32, 0u32, 0u32, 0u32, 0u32, 0u32);
                let mut n_tau = 0u64;
                let mut tau = vec![0f64; 8192];
                check(unsafe {
                    ffi::akz_plan_level_info(width, height, &options, level, &mut etime, &mut esigma, &mut octave, &mut sublevel,
                                             &mut sigma_size, &mut lw, &mut lh, &mut ds, &mut n_tau, tau.as_mut_ptr(), 8192)
                });
                tau.truncate(n_tau as usize);
                out_vec.push(EvolutionStep::empty(etime, esigma, octave, sublevel, sigma_size, tau));
            }
            out_vec
        }

        /// akaze/src/types/evolution.rs:163-168.  `set_extension(".png")` keeps the dot of its argument, so the
        /// reference's files are called `Lt_00000..png`; reproduced through the same std call.
        fn build_path(mut destination_dir: PathBuf, path_label: &str, idx: usize) -> PathBuf {
            destination_dir.push(format!("{}{:05}.png", path_label, idx));
            destination_dir.set_extension(".png");
            destination_dir
        }
        /// akaze/src/types/evolution.rs:175-218: one normalised PNG per image and level (0x0 images are skipped)
        pub fn write_evolutions(evolutions: &[EvolutionStep], destination_dir: PathBuf) {
            const LABELS: [&str; 10] = ["Lt_", "Lsmooth_", "Lx_", "Ly_", "Lxx_", "Lyy_", "Lxy_", "Lflow_", "Lstep_", "Ldet_"];
            for (i, evolution) in evolutions.iter().enumerate() {
                for (image, label) in evolution.planes().iter().zip(LABELS.iter()) {
                    save(image, build_path(destination_dir.clone(), label, i));
                }
            }
        }
    }

    /// akaze/src/types/keypoint.rs
    pub mod keypoint {
        use crate::{check, ffi, raw_keypoints};
        use image::{DynamicImage, RgbImage};
        use serde::{Deserialize, Serialize};

        /// akaze/src/types/keypoint.rs:8-30
        #[derive(Debug, Clone, Copy, Serialize, Deserialize)]
        pub struct Keypoint {
            pub point: (f32, f32),
            pub response: f32,
            pub size: f32,
            pub octave: usize,
            pub class_id: usize,
            pub angle: f32,
        }
        /// akaze/src/types/keypoint.rs:34-36
        #[derive(Debug, Clone, Serialize, Deserialize)]
        pub struct Descriptor {
            pub vector: Vec<u8>,
        }
        /// akaze/src/types/keypoint.rs:39-42
        #[derive(Debug, Clone, Serialize, Deserialize)]
        pub struct Results {
            pub keypoints: Vec<Keypoint>,
            pub descriptors: Vec<Descriptor>,
        }
        /// akaze/src/types/keypoint.rs:52-56 — `akz_draw_keypoints`
        pub fn draw_keypoints_to_image(input_image: &mut RgbImage, keypoints: &[Keypoint]) {
            let raw = raw_keypoints(keypoints);
            let (w, h) = (input_image.width(), input_image.height());
            check(unsafe { ffi::akz_draw_keypoints(input_image.as_mut_ptr(), w, h, raw.as_ptr(), raw.len() as u64) });
        }
        /// akaze/src/types/keypoint.rs:68-72
        pub fn draw_keypoints(input_image: &DynamicImage, keypoints: &[Keypoint]) -> RgbImage {
            let mut rgb_image = input_image.to_rgb();
            draw_keypoints_to_image(&mut rgb_image, keypoints);
            rgb_image
        }
    }

    /// akaze/src/types/feature_match.rs
    pub mod feature_match {
        use crate::types::keypoint::Keypoint;
        use crate::{check, ffi, raw_keypoints};
        use image::RgbImage;
        use serde::{Deserialize, Serialize};

        /// akaze/src/types/feature_match.rs:9-16; identical layout to `akz_match`
        #[repr(C)]
        #[derive(Debug, Clone, Copy, Serialize, Deserialize)]
        pub struct Match {
            pub index_0: usize,
            pub index_1: usize,
            pub distance: f64,
        }
        /// akaze/src/types/feature_match.rs:32-82 — `akz_draw_matches`
        pub fn draw_matches(input_image_0: &RgbImage, input_image_1: &RgbImage, keypoints_0: &[Keypoint],
                            keypoints_1: &[Keypoint], matches: &[Match]) -> RgbImage {
            let (k0, k1) = (raw_keypoints(keypoints_0), raw_keypoints(keypoints_1));
            let (mut w, mut h) = (0u32, 0u32);
            let mut px: *mut u8 = std::ptr::null_mut();
            check(unsafe {
                ffi::akz_draw_matches(input_image_0.as_ptr(), input_image_0.width(), input_image_0.height(), input_image_1.as_ptr(),
                                      input_image_1.width(), input_image_1.height(), k0.as_ptr(), k0.len() as u64, k1.as_ptr(),
                                      k1.len() as u64, matches.as_ptr(), matches.len() as u64, &mut w, &mut h, &mut px)
            });
            let bytes = unsafe { std::slice::from_raw_parts(px, (w as usize) * (h as usize) * 3) }.to_vec();
            unsafe { ffi::akz_image_free(px as *mut std::os::raw::c_void) };
            RgbImage::from_raw(w, h, bytes).unwrap()
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ops
// ------------------------------------------------------------------------------------------------
pub mod ops {
    /// akaze/src/ops/contrast_factor.rs
    pub mod contrast_factor {
        use crate::types::image::{GrayFloatImage, ImageFunctions};
        use crate::{check, ctx, ffi, DevPlane};
        use std::os::raw::c_void;
        /// akaze/src/ops/contrast_factor.rs:18-71 — `akz_op_contrast_factor`
        pub fn compute_contrast_factor(image: &GrayFloatImage, percentile: f64, gradient_histogram_scale: f64, num_bins: usize) -> f64 {
            let src = DevPlane::upload(&image.buffer);
            let k = DevPlane::zeros(2);  // one f64
            check(unsafe {
                ffi::akz_op_contrast_factor(ctx(), src.f32(), image.width() as u32, image.height() as u32, 1, percentile,
                                            gradient_histogram_scale, num_bins as u64, k.f32() as *mut f64)
            });
            let mut out = 0f64;
            check(unsafe { ffi::akz_memcpy_d2h(ctx(), &mut out as *mut f64 as *mut c_void, k.f32() as *const c_void, 8) });
            out
        }
    }
    /// akaze/src/ops/derivatives.rs
    pub mod derivatives {
        use crate::types::image::{GrayFloatImage, ImageFunctions};
        use crate::{check, ctx, ffi, DevPlane};
        /// akaze/src/ops/derivatives.rs:112-130 — `akz_op_scharr`, which covers all four order combinations as the
        /// reference does: x only / y only; both = the HORIZONTAL derivative added to itself (:118-122, through
        /// the misnamed `sqrt_squared`, types/image.rs:218-231); neither = a new zero image (:127-128).
        pub fn scharr(image: &GrayFloatImage, x_order: bool, y_order: bool, sigma_size: u32) -> GrayFloatImage {
            let (w, h) = (image.width(), image.height());
            let src = DevPlane::upload(&image.buffer);
            let dst = DevPlane::zeros(w * h);
            check(unsafe {
                ffi::akz_op_scharr(ctx(), src.f32(), dst.f32(), w as u32, h as u32, 1, x_order as i32, y_order as i32, sigma_size)
            });
            GrayFloatImage::from_buffer(dst.download(), w, h)
        }
    }
    /// akaze/src/ops/descriptors.rs
    pub mod descriptors {
        use crate::types::evolution::{Config, EvolutionStep};
        use crate::types::keypoint::{Descriptor, Keypoint};
        use crate::{check, ffi, raw_keypoints};
        /// akaze/src/ops/descriptors.rs:14-27 — the caller's `Lt`, `Lx`, `Ly` planes are uploaded
        /// (`akz_extract_from_planes` with AKZ_NO_DETECT) and `akz_result_describe_keypoints` runs the M-LDB kernel
        /// with the keypoints' own angles.
        pub fn extract_descriptors(evolutions: &[EvolutionStep], keypoints: &[Keypoint], options: Config) -> Vec<Descriptor> {
            if keypoints.is_empty() {
                return vec![];
            }
            let res = crate::upload_pyramid(evolutions, options, crate::AKZ_NO_DETECT);
            let mut raw = raw_keypoints(keypoints);
            let nb = ((6 + 36 + 120) * options.descriptor_channels + 7) / 8;
            let mut bytes = vec![0u8; raw.len() * nb];
            check(unsafe { ffi::akz_result_describe_keypoints(res.0, 0, raw.as_mut_ptr(), raw.len() as u64, 0, bytes.as_mut_ptr()) });
            bytes.chunks(nb).map(|c| Descriptor { vector: c.to_vec() }).collect()
        }
    }
    /// akaze/src/ops/detector_response.rs
    pub mod detector_response {
        use crate::types::evolution::{Config, EvolutionStep};
        use crate::types::image::{GrayFloatImage, ImageFunctions};
        use crate::{check, ctx, ffi, DevPlane};
        /// akaze/src/ops/detector_response.rs:38-55 — `akz_op_detector_response` per evolution
        /// (sigma_size = round(esigma * derivative_factor / 2^octave), :42)
        pub fn detector_response(evolutions: &mut Vec<EvolutionStep>, options: Config) {
            for ev in evolutions.iter_mut() {
                let (w, h) = (ev.Lsmooth.width(), ev.Lsmooth.height());
                let ratio = f64::powf(2.0f64, f64::from(ev.octave));
                let sigma_size = f64::round(ev.esigma * options.derivative_factor / ratio) as u32;
                let src = DevPlane::upload(&ev.Lsmooth.buffer);
                let out: Vec<DevPlane> = (0..6).map(|_| DevPlane::zeros(w * h)).collect();
                check(unsafe {
                    ffi::akz_op_detector_response(ctx(), src.f32(), sigma_size, out[0].f32(), out[1].f32(), out[2].f32(),
                                                  out[3].f32(), out[4].f32(), out[5].f32(), w as u32, h as u32, 1)
                });
                ev.Lx = GrayFloatImage::from_buffer(out[0].download(), w, h);
                ev.Ly = GrayFloatImage::from_buffer(out[1].download(), w, h);
                ev.Lxx = GrayFloatImage::from_buffer(out[2].download(), w, h);
                ev.Lyy = GrayFloatImage::from_buffer(out[3].download(), w, h);
                ev.Lxy = GrayFloatImage::from_buffer(out[4].download(), w, h);
                ev.Ldet = GrayFloatImage::from_buffer(out[5].download(), w, h);
            }
        }
    }
    /// akaze/src/ops/estimate_fundamental_matrix.rs (host code inside the library)
    pub mod estimate_fundamental_matrix {
        use crate::types::feature_match::Match;
        use crate::types::keypoint::Keypoint;
        use crate::{check, ffi, raw_keypoints};
        use nalgebra::Matrix3;
        /// akaze/src/ops/estimate_fundamental_matrix.rs:17-69 — `akz_estimate_fundamental_matrix`
        pub fn estimate_fundamental_matrix(keypoints_0: &[Keypoint], keypoints_1: &[Keypoint], matches: &mut [Match],
                                           epsilon: f32) -> Option<Matrix3<f32>> {
            debug_assert!(matches.len() == 8);
            let (k0, k1) = (raw_keypoints(keypoints_0), raw_keypoints(keypoints_1));
            let mut f = [0f32; 9];
            let mut found = 0;
            check(unsafe {
                ffi::akz_estimate_fundamental_matrix(k0.as_ptr(), k0.len() as u64, k1.as_ptr(), k1.len() as u64, matches.as_ptr(),
                                                     epsilon, f.as_mut_ptr(), &mut found)
            });
            if found != 0 {
                Some(Matrix3::new(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8]))
            } else {
                None
            }
        }
        /// akaze/src/ops/estimate_fundamental_matrix.rs:99-165 — `akz_remove_outliers`
        pub fn remove_outliers(keypoints_0: &[Keypoint], keypoints_1: &[Keypoint], matches: &[Match], num_trials: usize,
                               epsilon_model: f32, epsilon_inlier: f32) -> Vec<Match> {
            let (k0, k1) = (raw_keypoints(keypoints_0), raw_keypoints(keypoints_1));
            let mut out = vec![Match { index_0: 0, index_1: 0, distance: 0.0 }; matches.len().max(1)];
            let mut n = 0u64;
            check(unsafe {
                ffi::akz_remove_outliers(k0.as_ptr(), k0.len() as u64, k1.as_ptr(), k1.len() as u64, matches.as_ptr(),
                                         matches.len() as u64, num_trials as u64, epsilon_model, epsilon_inlier, out.as_mut_ptr(),
                                         &mut n)
            });
            out.truncate(n as usize);
            out
        }
    }
    /// akaze/src/ops/feature_matching.rs
    pub mod feature_matching {
        use crate::types::feature_match::Match;
        use crate::types::keypoint::Descriptor;
        use crate::{check, ctx, ffi, flat_descriptors};
        /// akaze/src/ops/feature_matching.rs:23-94 — `akz_descriptor_match`
        pub fn descriptor_match(descriptors_0: &[Descriptor], descriptors_1: &[Descriptor], distance_threshold: usize,
                                lowes_ratio: f64) -> Vec<Match> {
            let ((d0, nb0), (d1, nb1)) = (flat_descriptors(descriptors_0), flat_descriptors(descriptors_1));
            assert!(nb0 == nb1 || descriptors_0.is_empty() || descriptors_1.is_empty(), "descriptor lengths differ");
            let mut out = vec![Match { index_0: 0, index_1: 0, distance: 0.0 }; descriptors_0.len().max(1)];
            let mut n = 0u64;
            check(unsafe {
                ffi::akz_descriptor_match(ctx(), d0.as_ptr(), descriptors_0.len() as u64, d1.as_ptr(), descriptors_1.len() as u64,
                                          nb0.max(nb1).max(1) as u64, distance_threshold as u64, lowes_ratio, out.as_mut_ptr(), &mut n)
            });
            out.truncate(n as usize);
            out
        }
    }
    /// akaze/src/ops/fed_tau.rs
    pub mod fed_tau {
        use crate::{check, ffi};
        /// akaze/src/ops/fed_tau.rs:27-30 — `akz_fed_tau_by_process_time` (the n == 1 case, where the reference never
        /// terminates, panics here)
        pub fn fed_tau_by_process_time(T: f64, M: i32, tau_max: f64, reordering: bool) -> Vec<f64> {
            let mut n = 0u64;
            check(unsafe { ffi::akz_fed_tau_by_process_time(T, M, tau_max, reordering as i32, std::ptr::null_mut(), 0, &mut n) });
            let mut out = vec![0f64; n as usize];
            check(unsafe { ffi::akz_fed_tau_by_process_time(T, M, tau_max, reordering as i32, out.as_mut_ptr(), n, &mut n) });
            out
        }
    }
    /// akaze/src/ops/nonlinear_diffusion.rs
    pub mod nonlinear_diffusion {
        use crate::types::evolution::EvolutionStep;
        use crate::types::image::{GrayFloatImage, ImageFunctions};
        use crate::{check, ctx, ffi, DevPlane};
        use nalgebra::Vector4;
        /// akaze/src/ops/nonlinear_diffusion.rs:15-144 — one `akz_op_fed_steps` step: `Lt += Lstep`, `Lstep` stored
        pub fn calculate_step(evolution_step: &mut EvolutionStep, step_size: f64) {
            let (w, h) = (evolution_step.Lt.width(), evolution_step.Lt.height());
            let lt = DevPlane::upload(&evolution_step.Lt.buffer);
            let lflow = DevPlane::upload(&evolution_step.Lflow.buffer);
            let lstep = DevPlane::zeros(w * h);
            check(unsafe { ffi::akz_op_fed_steps(ctx(), lt.f32(), lflow.f32(), lstep.f32(), w as u32, h as u32, 1, &step_size, 1) });
            evolution_step.Lt = GrayFloatImage::from_buffer(lt.download(), w, h);
            evolution_step.Lstep = GrayFloatImage::from_buffer(lstep.download(), w, h);
        }
        /// akaze/src/ops/nonlinear_diffusion.rs:149-173: one flux term,
        /// `(c(x+px0, y+py0) + c(x+px1, y+py1)) * (Ld(x+px2, y+py2) - Ld(x+px3, y+py3))` (host arithmetic, as there)
        pub fn eval(c: &GrayFloatImage, Ld: &GrayFloatImage, x: usize, y: usize, plus_x: impl Into<Vector4<i32>>,
                    plus_y: impl Into<Vector4<i32>>) -> f32 {
            let (px, py) = (plus_x.into(), plus_y.into());
            let at = |img: &GrayFloatImage, i: usize| img.get((x as i32 + px[i]) as usize, (y as i32 + py[i]) as usize);
            (at(c, 0) + at(c, 1)) * (at(Ld, 2) - at(Ld, 3))
        }
    }
    /// akaze/src/ops/scale_space_extrema.rs
    pub mod scale_space_extrema {
        use crate::types::evolution::{Config, EvolutionStep};
        use crate::types::keypoint::Keypoint;
        use crate::{check, ffi, to_keypoints, AkzKeypoint};
        /// akaze/src/ops/scale_space_extrema.rs:199-203 — the caller's `Ldet`, `Lx`, `Ly` (and `Lt`) planes are
        /// uploaded (`akz_extract_from_planes`); extrema test on the GPU, cache logic / refinement on the host inside
        /// the library, orientation sums on the GPU.
        pub fn detect_keypoints(evolutions: &mut Vec<EvolutionStep>, options: Config) -> Vec<Keypoint> {
            let res = crate::upload_pyramid(evolutions, options, 0);
            let (mut nl, mut nk, mut nb) = (0u64, 0u64, 0u64);
            check(unsafe { ffi::akz_result_counts(res.0, 0, &mut nl, &mut nk, &mut nb) });
            let mut raw = vec![AkzKeypoint::default(); nk as usize];
            check(unsafe { ffi::akz_result_keypoints(res.0, 0, raw.as_mut_ptr()) });
            to_keypoints(&raw)
        }
    }
}

pub(crate) struct ResultHandle(pub(crate) *mut c_void);
impl Drop for ResultHandle {
    fn drop(&mut self) {
        unsafe { ffi::akz_result_free(self.0) };
    }
}

/// `akz_extract_from_planes` on the images a caller's evolutions hold (level size from `Lt`)
pub(crate) fn upload_pyramid(evolutions: &[types::evolution::EvolutionStep], options: types::evolution::Config, flags: u32) -> ResultHandle {
    use types::image::ImageFunctions;
    assert!(!evolutions.is_empty(), "no evolutions");
    let (w, h) = (evolutions[0].Lt.width() as u32, evolutions[0].Lt.height() as u32);
    let mut table: Vec<*const f32> = Vec::with_capacity(evolutions.len() * 10);
    for ev in evolutions {
        for image in ev.planes().iter() {
            table.push(if image.buffer.is_empty() { std::ptr::null() } else { image.buffer.as_ptr() });
        }
    }
    let mut res: *mut c_void = std::ptr::null_mut();
    check(unsafe { ffi::akz_extract_from_planes(ctx(), w, h, &options, table.as_ptr(), evolutions.len() as u64, flags, &mut res) });
    ResultHandle(res)
}

// ------------------------------------------------------------------------------------------------
// the crate's two entry points
// ------------------------------------------------------------------------------------------------
/// akaze::extract_features (akaze/src/lib.rs:167-194): decode + `to_luma` stay on the host via the `image` crate
/// exactly as in the reference; everything after `create_unit_float_image` runs on the GPU.  The returned evolutions
/// carry all ten images of every level (downloaded once, 0.44 GB for a 1080p frame) — what the reference returns.
pub fn extract_features(input_image_path: PathBuf, options: types::evolution::Config)
                        -> (Vec<types::evolution::EvolutionStep>, Vec<types::keypoint::Keypoint>, Vec<types::keypoint::Descriptor>) {
    use types::evolution::EvolutionStep;
    use types::image::GrayFloatImage;
    use types::keypoint::Descriptor;
    let luma = image::open(input_image_path).unwrap().to_luma();
    let (w, h) = (luma.width(), luma.height());
    let mut res: *mut c_void = std::ptr::null_mut();
    check(unsafe { ffi::akz_extract_gray_u8(ctx(), luma.as_ptr(), w, h, &options, AKZ_KEEP_ALL_PLANES, &mut res) });
    let handle = ResultHandle(res);
    let (mut nl, mut nk, mut nb) = (0u64, 0u64, 0u64);
    check(unsafe { ffi::akz_result_counts(res, 0, &mut nl, &mut nk, &mut nb) });
    let mut raw = vec![AkzKeypoint::default(); nk as usize];
    check(unsafe { ffi::akz_result_keypoints(res, 0, raw.as_mut_ptr()) });
    let mut bytes = vec![0u8; (nk * nb) as usize];
    check(unsafe { ffi::akz_result_descriptors(res, 0, bytes.as_mut_ptr()) });
    let keypoints = to_keypoints(&raw);
    let descriptors = if nb == 0 { vec![] } else { bytes.chunks(nb as usize).map(|c| Descriptor { vector: c.to_vec() }).collect() };
    let mut evolutions = Vec::with_capacity(nl as usize);
    for level in 0..nl {
        let (mut etime, mut esigma) = (0f64, 0f64);
        let (mut octave, mut sublevel, mut sigma_size, mut lw, mut lh) = (0u32, 0u32, 0u32, 0u32, 0u32);
        let mut n_tau = 0u64;
        let mut tau = vec![0f64; 8192];
        check(unsafe {
            ffi::akz_result_level_info(res, level, &mut etime, &mut esigma, &mut octave, &mut sublevel, &mut sigma_size, &mut lw,
                                       &mut lh, &mut n_tau, tau.as_mut_ptr(), 8192)
        });
        tau.truncate(n_tau as usize);
        let mut ev = EvolutionStep::empty(etime, esigma, octave, sublevel, sigma_size, tau);
        for (plane, image) in ev.planes_mut().iter_mut().enumerate() {
            let mut n_px = 0u64;
            check(unsafe { ffi::akz_fetch_plane(res, 0, level, plane as c_int, std::ptr::null_mut(), &mut n_px) });
            if n_px == 0 {
                continue;  // level 0 has no Lflow / Lstep: 0x0, as in the reference
            }
            let mut buffer = vec![0f32; n_px as usize];
            check(unsafe { ffi::akz_fetch_plane(res, 0, level, plane as c_int, buffer.as_mut_ptr(), &mut n_px) });
            **image = GrayFloatImage::from_buffer(buffer, lw as usize, lh as usize);
        }
        evolutions.push(ev);
    }
    drop(handle);
    (evolutions, keypoints, descriptors)
}

/// akaze::match_features (akaze/src/lib.rs:252-275): the Hamming stage (descriptor_match with distance threshold
/// 10000) runs on the GPU, the RANSAC filter (remove_outliers, epsilon_model 0.05) on the host inside the library.
pub fn match_features(keypoints_0: &[types::keypoint::Keypoint], descriptors_0: &[types::keypoint::Descriptor],
                      keypoints_1: &[types::keypoint::Keypoint], descriptors_1: &[types::keypoint::Descriptor],
                      lowes_ratio: f64, ransac_trials: usize, ransac_epsilon_inliers: f32) -> Vec<types::feature_match::Match> {
    use types::feature_match::Match;
    let ((d0, nb0), (d1, nb1)) = (flat_descriptors(descriptors_0), flat_descriptors(descriptors_1));
    assert!(nb0 == nb1 || descriptors_0.is_empty() || descriptors_1.is_empty(), "descriptor lengths differ");
    let (k0, k1) = (raw_keypoints(keypoints_0), raw_keypoints(keypoints_1));
    let mut out = vec![Match { index_0: 0, index_1: 0, distance: 0.0 }; descriptors_0.len().max(1)];
    let mut n = 0u64;
    check(unsafe {
        ffi::akz_match_features(ctx(), k0.as_ptr(), k0.len() as u64, d0.as_ptr(), descriptors_0.len() as u64, k1.as_ptr(),
                                k1.len() as u64, d1.as_ptr(), descriptors_1.len() as u64, nb0.max(nb1).max(1) as u64, lowes_ratio,
                                ransac_trials as u64, ransac_epsilon_inliers, out.as_mut_ptr(), &mut n)
    });
    out.truncate(n as usize);
    out
}

#[allow(dead_code)]
fn _unused(_: *const c_char) {}
