fn main() {
    // libvfind_hip.so is built by `make -C vfind_amd/csrc`
    let dir = std::env::var("VFIND_HIP_LIB_DIR").unwrap_or_else(|_| "../vfind_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vfind_hip");
    println!("cargo:rerun-if-env-changed=VFIND_HIP_LIB_DIR");
}
