// Links libsbn254.so (built by `make -C starky_bn254_amd/csrc`, hipcc --offload-arch=gfx950).
// SBN254_LIB_DIR overrides the default location relative to this crate.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("SBN254_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../starky_bn254_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=sbn254");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=SBN254_LIB_DIR");
}
