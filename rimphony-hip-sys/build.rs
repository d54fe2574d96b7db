// rimphony-hip-sys/build.rs -- tell cargo where librimphony_hip.so lives.
//
// UNVERIFIED (no Rust toolchain in the image this was written in).
//
// Modelled on the two build scripts of the reference: leung-bessel/build.rs:1-14 compiles bessel.c with `cc`,
// gsl-sys/build.rs:1-33 probes the system GSL with pkg-config and runs bindgen.  This one does neither: the library
// is a prebuilt shared object (hipcc --offload-arch=gfx950, see rimphony_amd/_build.py), and the extern block in
// src/lib.rs is written by hand against include/rimphony_hip.h, so no bindgen/libclang is needed.
//
//   RIMPHONY_HIP_LIB_DIR   directory holding librimphony_hip.so   (default: ../rimphony_amd relative to this crate,
//                          i.e. the in-tree build of this repository)
//
// `links = "rimphony_hip"` in Cargo.toml makes cargo refuse two crates that link the library and lets dependants read
// DEP_RIMPHONY_HIP_LIB_DIR.

use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var_os("RIMPHONY_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|| {
            PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap())
                .join("..")
                .join("rimphony_amd")
        });

    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rimphony_hip");
    // the HIP runtime is a dependency of the shared object itself (DT_NEEDED libamdhip64.so); an rpath entry saves
    // the user an LD_LIBRARY_PATH for the in-tree library
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:lib_dir={}", dir.display());
    println!("cargo:rerun-if-env-changed=RIMPHONY_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
