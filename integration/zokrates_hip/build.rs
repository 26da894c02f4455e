// Links libzkhip.so (built by `python -m zokrates_amd.build`).  ZKHIP_LIB_DIR points at the directory holding it.
fn main() {
    if let Ok(dir) = std::env::var("ZKHIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zkhip");
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
}
