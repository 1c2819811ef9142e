// Links libsumma_gpu.so (built by `make -C circuits_halo2_amd/csrc`); SUMMA_GPU_LIB_DIR points at
// the directory that holds it.
fn main() {
    let dir = std::env::var("SUMMA_GPU_LIB_DIR").expect("set SUMMA_GPU_LIB_DIR to the directory of libsumma_gpu.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=summa_gpu");
    println!("cargo:rerun-if-env-changed=SUMMA_GPU_LIB_DIR");
}
