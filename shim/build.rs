// build.rs of the reference crate when the `hip` feature is on: where libsbn254_hip.so lives.
// SBN254_LIB_DIR = the directory holding the library built by `make -C spartan-bn254_amd` (hipcc --offload-arch=gfx950).
fn main() {
    println!("cargo:rerun-if-env-changed=SBN254_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("SBN254_LIB_DIR").expect("feature `hip`: set SBN254_LIB_DIR to the directory that holds libsbn254_hip.so");
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=sbn254_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
}
