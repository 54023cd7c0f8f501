// build.rs — link the soundscope crate against libsoundscope_hip.so (built by `make -C soundscope_amd/csrc`).
// SOUNDSCOPE_HIP_LIB_DIR overrides the directory; the default is the in-tree build of this repository.
fn main() {
    let dir = std::env::var("SOUNDSCOPE_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../soundscope_amd/lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=soundscope_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SOUNDSCOPE_HIP_LIB_DIR");
}
