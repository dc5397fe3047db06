// Link against libhgi_hip.so; HGI_HIP_DIR points at the directory holding it (rustyhgi_amd/).
fn main() {
    let dir = std::env::var("HGI_HIP_DIR").unwrap_or_else(|_| "../rustyhgi_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=hgi_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
