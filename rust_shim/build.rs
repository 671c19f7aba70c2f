// Link against <repo>/fundsp_amd/libfundsp_hip.so (built by `make -C fundsp_amd/csrc` / __graft_entry__.build()).
// FUNDSP_HIP_DIR overrides the directory.
fn main() {
    let dir = std::env::var("FUNDSP_HIP_DIR")
        .unwrap_or_else(|_| format!("{}/../fundsp_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap()));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fundsp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FUNDSP_HIP_DIR");
}
