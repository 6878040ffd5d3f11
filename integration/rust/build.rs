// circuits/build.rs (new): links libola_gpu.so when the crate is built with `--features hip` and stages the AIR-set blob the
// shim `include_bytes!`s.  Same mechanism as the reference's own GPU hook, plonky2/field/build.rs:4-27 (feature `cuda`:
// rustc-link-search + rustc-link-lib), with the library this repository builds (python __graft_entry__.py) in place of the
// prebuilt libcuda_lib.a.
//   OLA_GPU_LIB_DIR      directory holding libola_gpu.so            (<this repo>/olavm_amd/lib)
//   OLA_GPU_INCLUDE_DIR  directory holding ola_airset.bin           (<this repo>/include)
//   ROCM_PATH            ROCm root, default /opt/rocm               (libamdhip64.so; librccl.so is loaded at run time, on request)
#[cfg(not(feature = "hip"))]
fn main() {}

#[cfg(feature = "hip")]
fn main() {
    use std::{env, fs, path::PathBuf};
    let lib_dir = env::var("OLA_GPU_LIB_DIR").expect("set OLA_GPU_LIB_DIR to the directory that holds libola_gpu.so");
    let inc_dir = env::var("OLA_GPU_INCLUDE_DIR").expect("set OLA_GPU_INCLUDE_DIR to the directory that holds ola_airset.bin");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={lib_dir}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=ola_gpu");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=stdc++");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{lib_dir}");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    fs::copy(PathBuf::from(&inc_dir).join("ola_airset.bin"), out.join("ola_airset.bin")).expect("copy ola_airset.bin");
    println!("cargo:rerun-if-env-changed=OLA_GPU_LIB_DIR");
    println!("cargo:rerun-if-env-changed=OLA_GPU_INCLUDE_DIR");
    println!("cargo:rerun-if-changed={inc_dir}/ola_airset.bin");
}
