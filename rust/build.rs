// build.rs — links the MI355X engine when the `mi355x` feature is on.  UNCOMPILED in the authoring environment (no Rust
// toolchain there): see rust/README.md.  ACT_MI355X_LIB_DIR = directory holding libact_mi355x.so
// (`make -C anonymous-credit-tokens_amd/csrc` with hipcc --offload-arch=gfx950 produces it).
fn main() {
    println!("cargo:rerun-if-env-changed=ACT_MI355X_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_MI355X").is_none() {
        return;
    }
    let dir = std::env::var("ACT_MI355X_LIB_DIR").expect("set ACT_MI355X_LIB_DIR to the directory of libact_mi355x.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=act_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
