// Links libshodh_hip.so (built by `python -m shodh_memory_amd.build`, hipcc only). Point SHODH_HIP_LIB_DIR at the
// directory that holds it; the loader needs the same directory on LD_LIBRARY_PATH (or an rpath) at run time.
fn main() {
    println!("cargo:rerun-if-env-changed=SHODH_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("SHODH_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=shodh_hip");
}
