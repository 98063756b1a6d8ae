//! Drop-in replacement for the hot path of the `akaze` crate over the MI355X HIP library.
//!
//! Keeps `akaze::extract_features`, `akaze::match_features` and `types::evolution::Config`
//! with the reference's signatures; every call goes through the C ABI of
//! `include/akaze_hip.h`.  Non-zero status codes become panics, which is the reference's own
//! error behaviour (`unwrap()` at akaze/src/lib.rs:171).
//!
//! UNVERIFIED: this image has no Rust toolchain, so this file has never been compiled.
#![allow(non_snake_case)]
use std::os::raw::{c_char, c_int, c_void};
use std::path::PathBuf;

pub mod types {
    pub mod evolution {
        use serde::{Deserialize, Serialize};
        /// Same fields, order and defaults as the reference (akaze/src/types/evolution.rs:8-55);
        /// `#[repr(C)]` so that it is also `akz_config`.
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
        /// Scalars of the reference's EvolutionStep plus lazily fetched images
        /// (akaze/src/types/evolution.rs:59-92).  Images stay in HBM until `image()` is called.
        pub struct EvolutionStep {
            pub etime: f64,
            pub esigma: f64,
            pub octave: u32,
            pub sublevel: u32,
            pub sigma_size: u32,
            pub width: u32,
            pub height: u32,
            pub fed_tau_steps: Vec<f64>,
            pub(crate) result: std::rc::Rc<crate::ResultHandle>,
            pub(crate) level: u64,
        }
        /// Field order of the reference struct: Lt, Lsmooth, Lx, Ly, Lxx, Lyy, Lxy, Lflow, Lstep, Ldet.
        #[repr(C)]
        #[derive(Copy, Clone)]
        pub enum Plane { Lt = 0, Lsmooth, Lx, Ly, Lxx, Lyy, Lxy, Lflow, Lstep, Ldet }
        impl EvolutionStep {
            /// Row-major f32 pixels of one plane (empty for the 0x0 planes of level 0).
            pub fn image(&self, plane: Plane) -> Vec<f32> {
                let mut n: u64 = 0;
                crate::check(unsafe {
                    crate::akz_fetch_plane(self.result.0, 0, self.level, plane as i32, std::ptr::null_mut(), &mut n)
                });
                let mut out = vec![0f32; n as usize];
                if n > 0 {
                    crate::check(unsafe {
                        crate::akz_fetch_plane(self.result.0, 0, self.level, plane as i32, out.as_mut_ptr(), &mut n)
                    });
                }
                out
            }
        }
    }
    pub mod keypoint {
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
    }
    pub mod feature_match {
        use serde::{Deserialize, Serialize};
        /// akaze/src/types/feature_match.rs:9-16 ; identical layout to akz_match
        #[repr(C)]
        #[derive(Debug, Clone, Copy, Serialize, Deserialize)]
        pub struct Match {
            pub index_0: usize,
            pub index_1: usize,
            pub distance: f64,
        }
    }
}
use types::evolution::{Config, EvolutionStep};
use types::feature_match::Match;
use types::keypoint::{Descriptor, Keypoint};

#[repr(C)]
#[derive(Copy, Clone, Default)]
struct AkzKeypoint { x: f32, y: f32, response: f32, size: f32, octave: u64, class_id: u64, angle: f32, _pad: u32 }

extern "C" {
    fn akz_last_error() -> *const c_char;
    fn akz_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut c_void) -> c_int;
    fn akz_extract_gray_u8(ctx: *mut c_void, img: *const u8, w: u32, h: u32, cfg: *const Config, flags: u32,
                           out: *mut *mut c_void) -> c_int;
    fn akz_result_free(res: *mut c_void) -> c_int;
    fn akz_result_counts(res: *const c_void, img: u64, n_levels: *mut u64, n_kp: *mut u64, desc_bytes: *mut u64) -> c_int;
    fn akz_result_keypoints(res: *const c_void, img: u64, out: *mut AkzKeypoint) -> c_int;
    fn akz_result_descriptors(res: *const c_void, img: u64, out: *mut u8) -> c_int;
    fn akz_result_level_info(res: *const c_void, level: u64, etime: *mut f64, esigma: *mut f64, octave: *mut u32,
                             sublevel: *mut u32, sigma_size: *mut u32, w: *mut u32, h: *mut u32, n_tau: *mut u64,
                             tau: *mut f64, tau_cap: u64) -> c_int;
    pub(crate) fn akz_fetch_plane(res: *const c_void, img: u64, level: u64, plane: c_int, out: *mut f32,
                                  n_px: *mut u64) -> c_int;
    fn akz_match_features(ctx: *mut c_void, kp0: *const AkzKeypoint, d0: *const u8, n0: u64, kp1: *const AkzKeypoint,
                          d1: *const u8, n1: u64, desc_bytes: u64, lowes_ratio: f64, ransac_trials: u64,
                          ransac_epsilon_inliers: f32, out: *mut Match, n_out: *mut u64) -> c_int;
}

pub(crate) fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(akz_last_error()) }.to_string_lossy().into_owned();
        panic!("akaze_hip status {}: {}", status, msg);
    }
}

pub struct ResultHandle(pub(crate) *mut c_void);
impl Drop for ResultHandle {
    fn drop(&mut self) { unsafe { akz_result_free(self.0); } }
}

thread_local! {
    // one context per host thread (the C ABI's threading rule); device from AKAZE_HIP_DEVICE
    static CTX: *mut c_void = {
        let dev = std::env::var("AKAZE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx: *mut c_void = std::ptr::null_mut();
        check(unsafe { akz_ctx_create(dev, std::ptr::null_mut(), &mut ctx) });
        ctx
    };
}

const AKZ_KEEP_ALL_PLANES: u32 = 1;

/// akaze::extract_features (akaze/src/lib.rs:167-194): decode + to_luma stay on the host via the
/// `image` crate exactly as in the reference; everything after `create_unit_float_image` runs on the GPU.
pub fn extract_features(input_image_path: PathBuf, options: Config) -> (Vec<EvolutionStep>, Vec<Keypoint>, Vec<Descriptor>) {
    let luma = image::open(input_image_path).unwrap().to_luma();
    let (w, h) = (luma.width(), luma.height());
    let mut res: *mut c_void = std::ptr::null_mut();
    CTX.with(|ctx| check(unsafe {
        akz_extract_gray_u8(*ctx, luma.as_ptr(), w, h, &options, AKZ_KEEP_ALL_PLANES, &mut res)
    }));
    let handle = std::rc::Rc::new(ResultHandle(res));
    let (mut nl, mut nk, mut nb) = (0u64, 0u64, 0u64);
    check(unsafe { akz_result_counts(res, 0, &mut nl, &mut nk, &mut nb) });
    let mut raw = vec![AkzKeypoint::default(); nk as usize];
    check(unsafe { akz_result_keypoints(res, 0, raw.as_mut_ptr()) });
    let mut bytes = vec![0u8; (nk * nb) as usize];
    check(unsafe { akz_result_descriptors(res, 0, bytes.as_mut_ptr()) });
    let keypoints = raw.iter().map(|k| Keypoint {
        point: (k.x, k.y), response: k.response, size: k.size,
        octave: k.octave as usize, class_id: k.class_id as usize, angle: k.angle,
    }).collect();
    let descriptors = bytes.chunks(nb as usize).map(|c| Descriptor { vector: c.to_vec() }).collect();
    let mut evolutions = Vec::with_capacity(nl as usize);
    for level in 0..nl {
        let (mut etime, mut esigma) = (0f64, 0f64);
        let (mut octave, mut sublevel, mut sigma_size, mut lw, mut lh) = (0u32, 0u32, 0u32, 0u32, 0u32);
        let mut n_tau = 0u64;
        let mut tau = vec![0f64; 4096];
        check(unsafe { akz_result_level_info(res, level, &mut etime, &mut esigma, &mut octave, &mut sublevel,
                                             &mut sigma_size, &mut lw, &mut lh, &mut n_tau, tau.as_mut_ptr(), 4096) });
        tau.truncate(n_tau as usize);
        evolutions.push(EvolutionStep { etime, esigma, octave, sublevel, sigma_size, width: lw, height: lh,
                                        fed_tau_steps: tau, result: handle.clone(), level });
    }
    (evolutions, keypoints, descriptors)
}

/// akaze::match_features (akaze/src/lib.rs:252-275): the Hamming stage (descriptor_match with distance
/// threshold 10000) runs on the GPU, the RANSAC filter (remove_outliers, epsilon_model 0.05) on the host
/// inside the library.
pub fn match_features(keypoints_0: &[Keypoint], descriptors_0: &[Descriptor], keypoints_1: &[Keypoint],
                      descriptors_1: &[Descriptor], lowes_ratio: f64, ransac_trials: usize,
                      ransac_epsilon_inliers: f32) -> Vec<Match> {
    let nb = descriptors_0.first().or(descriptors_1.first()).map(|d| d.vector.len()).unwrap_or(61);
    let flat = |ds: &[Descriptor]| ds.iter().flat_map(|d| d.vector.iter().cloned()).collect::<Vec<u8>>();
    let raw = |ks: &[Keypoint]| ks.iter().map(|k| AkzKeypoint {
        x: k.point.0, y: k.point.1, response: k.response, size: k.size,
        octave: k.octave as u64, class_id: k.class_id as u64, angle: k.angle, _pad: 0,
    }).collect::<Vec<AkzKeypoint>>();
    let (d0, d1, k0, k1) = (flat(descriptors_0), flat(descriptors_1), raw(keypoints_0), raw(keypoints_1));
    let mut out = vec![Match { index_0: 0, index_1: 0, distance: 0.0 }; descriptors_0.len().max(1)];
    let mut n = 0u64;
    CTX.with(|ctx| check(unsafe {
        akz_match_features(*ctx, k0.as_ptr(), d0.as_ptr(), descriptors_0.len() as u64, k1.as_ptr(), d1.as_ptr(),
                           descriptors_1.len() as u64, nb as u64, lowes_ratio, ransac_trials as u64,
                           ransac_epsilon_inliers, out.as_mut_ptr(), &mut n)
    }));
    out.truncate(n as usize);
    out
}
