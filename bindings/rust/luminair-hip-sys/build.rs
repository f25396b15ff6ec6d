// Links libluminair_hip.so (or, with the `LUMINAIR_HIP_BATCH` environment variable set, libluminair_hip_batch.so, which
// exports the same ABI plus lmn_batch_*).  LUMINAIR_HIP_LIB_DIR = the directory holding the library
// (luminair_amd/csrc of this repository after `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    if let Ok(dir) = std::env::var("LUMINAIR_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    let lib = if std::env::var("LUMINAIR_HIP_BATCH").is_ok() { "luminair_hip_batch" } else { "luminair_hip" };
    println!("cargo:rustc-link-lib=dylib={lib}");
    println!("cargo:rerun-if-env-changed=LUMINAIR_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=LUMINAIR_HIP_BATCH");
}
