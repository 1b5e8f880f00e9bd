// Links libvoxbox_hip.so (built by `make -C vox_box.rs_amd`, or by __graft_entry__.build()).
// VOXBOX_HIP_LIB_DIR overrides the default location relative to this crate: ../../vox_box.rs_amd/lib.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("VOXBOX_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../vox_box.rs_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=voxbox_hip");
    // run-time search path, so that `cargo run` finds the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=VOXBOX_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
