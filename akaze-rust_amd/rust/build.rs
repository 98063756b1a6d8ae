// Links the shim against libakaze_hip.so.  AKAZE_HIP_LIB_DIR = directory holding the library
// (akaze-rust_amd/ in this repository).
fn main() {
    let dir = std::env::var("AKAZE_HIP_LIB_DIR").unwrap_or_else(|_| "..".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=akaze_hip");
    println!("cargo:rerun-if-env-changed=AKAZE_HIP_LIB_DIR");
}
