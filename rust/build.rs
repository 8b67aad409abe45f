// Builds the MI355X render library with hipcc and links it.
//
// RT_AMD_DIR  directory that holds the library's Makefile and csrc/ (default: ../ray-tracer_amd)
// HIPCC       compiler driver (default: /opt/rocm/bin/hipcc); the Makefile passes --offload-arch=gfx950
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let dir = env::var("RT_AMD_DIR").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../ray-tracer_amd"));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let jobs = env::var("NUM_JOBS").unwrap_or_else(|_| "4".to_string());

    // hipcc compiles csrc/*.hip for gfx950 without a GPU present; the objects are cached under build/
    let status = Command::new("make")
        .arg("-C").arg(&dir)
        .arg(format!("-j{}", jobs))
        .arg(format!("HIPCC={}", hipcc))
        .arg("lib/librt_amd.so")
        .status()
        .expect("failed to run make for librt_amd.so");
    assert!(status.success(), "make -C {:?} lib/librt_amd.so failed", dir);

    let lib = dir.join("lib");
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib.display());
    // a directory: cargo re-runs this script when any file below it changes
    for f in ["csrc", "../include/rt_amd.h", "Makefile"] {
        println!("cargo:rerun-if-changed={}", dir.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=RT_AMD_DIR");
    println!("cargo:rerun-if-env-changed=HIPCC");
}
