// Where libzkgpu.so lives: ZKGPU_LIB_DIR (e.g. <repo>/zkvm_amd/lib, where `python -m zkvm_amd.build` puts it), else the
// system's default search path.  The library itself needs the ROCm runtime (libamdhip64) at run time; RCCL is bound by
// dlopen only when a communicator is created.
fn main() {
    println!("cargo:rerun-if-env-changed=ZKGPU_LIB_DIR");
    if let Ok(dir) = std::env::var("ZKGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zkgpu");
}
