// Links libkmerhip.so (built by `make -C krust_amd/csrc`).  KMERHIP_LIB_DIR overrides the default
// in-tree location; /opt/rocm/lib is added to the rpath for libamdhip64.
use std::env;
use std::path::PathBuf;

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let lib_dir = env::var("KMERHIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| root.join("krust_amd/lib"));
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=kmerhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
    println!("cargo:rerun-if-env-changed=KMERHIP_LIB_DIR");
}
