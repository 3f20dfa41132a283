// Tells rustc where librm_hip.so lives.  RM_HIP_LIB_DIR overrides; the default is the in-tree build product
// (<repo>/ray-marching_amd/librm_hip.so, built by `python -m ray_marching_amd.build`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = match env::var("RM_HIP_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => {
            let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR"));
            // bindings/rust -> repo root -> ray-marching_amd
            manifest.join("..").join("..").join("ray-marching_amd")
        }
    };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rm_hip");
    // the library's own dependencies (HIP runtime) resolve through its RUNPATH; add the directory to ours
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RM_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
