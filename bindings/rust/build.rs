// Links libmiface.so (rs-face-detection-tflite_amd/build.sh builds it with hipcc for gfx950; it links libamdhip64 itself).
// MIFACE_LIB_DIR = directory holding libmiface.so (default: ../../rs-face-detection-tflite_amd relative to this crate).
fn main() {
    let dir = std::env::var("MIFACE_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../rs-face-detection-tflite_amd", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=miface");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=MIFACE_LIB_DIR");
}
