// Links libmsbwt_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"` in the
// rust-msbwt_amd repository; it lands in rust-msbwt_amd/).
fn main() {
    let dir = std::env::var("MSBWT_HIP_LIB_DIR").expect("set MSBWT_HIP_LIB_DIR to the directory that holds libmsbwt_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=msbwt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=MSBWT_HIP_LIB_DIR");
}
