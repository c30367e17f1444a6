// build.rs of the `gsw` crate once src/gsw/hip.rs is added: links the engine built by
// `python -c "import __graft_entry__ as g; g.build()"` (tfhe_amd/libgsw_hip.so).
fn main() {
    let root = std::env::var("GSW_MI355X_ROOT").expect("set GSW_MI355X_ROOT to the engine's checkout");
    println!("cargo:rustc-link-search=native={}/tfhe_amd", root);
    println!("cargo:rustc-link-lib=dylib=gsw_hip");
    println!("cargo:rerun-if-env-changed=GSW_MI355X_ROOT");
}
