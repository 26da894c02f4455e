//! Hand-written bindings of include/zkhip.h (the subset the adapter needs).  Every function returns 0 on success.
#![allow(non_camel_case_types)]
use std::os::raw::c_char;

#[repr(C)] pub struct zkhip_ctx { _p: [u8; 0] }
#[repr(C)] pub struct zkhip_pk { _p: [u8; 0] }
#[repr(C)] pub struct zkhip_r1cs { _p: [u8; 0] }
#[repr(C)] pub struct zkhip_multi { _p: [u8; 0] }
#[repr(C)] pub struct zkhip_prog { _p: [u8; 0] }
/// 16 floats: h2d, matvec, ntt, msm_h, msm_z, finish, total, accum_g1, accum_g2, kernel_ntt, 6 reserved (milliseconds)
#[repr(C)] #[derive(Default, Clone, Copy)] pub struct zkhip_timings { pub ms: [f32; 16] }

pub const ZKHIP_CURVE_BN128: i32 = 0;
pub const ZKHIP_CURVE_BLS12_381: i32 = 1;

pub const ZKHIP_TUNE_PIPE_PLAN: i32 = 28;

extern "C" {
    /// process-wide, before the first context: ask the HIP runtime for `hw_queues` hardware queues (the library never sets
    /// GPU_MAX_HW_QUEUES on its own; 8 for a resident prover, 16 if it is the only context of its process, 0 = leave it)
    pub fn zkhip_init(hw_queues: i32) -> i32;
    pub fn zkhip_ctx_create(device: i32, out: *mut *mut zkhip_ctx) -> i32;
    pub fn zkhip_ctx_free(ctx: *mut zkhip_ctx);
    pub fn zkhip_ctx_tune(ctx: *mut zkhip_ctx, which: i32, value: i32) -> i32;
    pub fn zkhip_last_error(ctx: *const zkhip_ctx) -> *const c_char;
    pub fn zkhip_pk_load_g16(ctx: *mut zkhip_ctx, curve: i32, bytes: *const u8, len: usize, out: *mut *mut zkhip_pk) -> i32;
    pub fn zkhip_pk_load_gm17(ctx: *mut zkhip_ctx, curve: i32, bytes: *const u8, len: usize, out: *mut *mut zkhip_pk) -> i32;
    pub fn zkhip_pk_free(pk: *mut zkhip_pk);
    pub fn zkhip_r1cs_load(ctx: *mut zkhip_ctx, curve: i32, n: u64, l: u64, w: u64,
        rp_a: *const u64, col_a: *const u32, val_a: *const u8,
        rp_b: *const u64, col_b: *const u32, val_b: *const u8,
        rp_c: *const u64, col_c: *const u32, val_c: *const u8, out: *mut *mut zkhip_r1cs) -> i32;
    pub fn zkhip_r1cs_free(cs: *mut zkhip_r1cs);
    pub fn zkhip_prove_g16(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, cs: *const zkhip_r1cs, z: *const u8,
        r: *const u8, s: *const u8, proof_out: *mut u8, timings: *mut zkhip_timings) -> i32;
    pub fn zkhip_prove_gm17(ctx: *mut zkhip_ctx, pk: *const zkhip_pk, cs: *const zkhip_r1cs, z: *const u8,
        d1_d2_r: *const u8, proof_out: *mut u8, timings: *mut zkhip_timings) -> i32;
    // the files the CLI already holds: `out` -> R1CS in ark order, `witness` -> z + inputs (host only, no context)
    pub fn zkhip_prog_parse(bytes: *const u8, len: usize, out: *mut *mut zkhip_prog) -> i32;
    pub fn zkhip_prog_free(prog: *mut zkhip_prog);
    pub fn zkhip_prog_dims(prog: *const zkhip_prog, out: *mut u64 /* [8] */) -> i32;
    pub fn zkhip_prog_r1cs_load(ctx: *mut zkhip_ctx, prog: *const zkhip_prog, out: *mut *mut zkhip_r1cs) -> i32;
    pub fn zkhip_prog_assignment(prog: *const zkhip_prog, witness: *const u8, len: usize, z_out: *mut u8,
        inputs_out: *mut u8, inputs_cap: u64, n_inputs: *mut u64) -> i32;
    // key cache: the resident form of a loaded key as one image
    pub fn zkhip_pk_export_size(pk: *const zkhip_pk, bytes: *mut u64) -> i32;
    pub fn zkhip_pk_export(pk: *const zkhip_pk, out: *mut u8, cap: u64) -> i32;
    pub fn zkhip_pk_import(ctx: *mut zkhip_ctx, bytes: *const u8, len: usize, out: *mut *mut zkhip_pk) -> i32;
    // a resident prover binds its key to its constraint system once (four transforms per proof instead of six, no c)
    pub fn zkhip_pk_bind_r1cs(ctx: *mut zkhip_ctx, pk: *mut zkhip_pk, r1cs: *const zkhip_r1cs) -> i32;
    pub fn zkhip_pk_unbind(pk: *mut zkhip_pk) -> i32;
    pub fn zkhip_pk_is_bound(pk: *const zkhip_pk, r1cs: *const zkhip_r1cs) -> i32;
    /// a shard of a multi-GPU key (or a whole key, Groth16 or GM17) bound from the key FILE: the transforms need every base once
    pub fn zkhip_pk_bind_r1cs_shard(ctx: *mut zkhip_ctx, pk: *mut zkhip_pk, r1cs: *const zkhip_r1cs, key_bytes: *const u8, len: usize) -> i32;
    // one proof across several GPUs of this process
    pub fn zkhip_ctx_create_multi(devices: *const i32, n: i32, out: *mut *mut zkhip_multi) -> i32;
    pub fn zkhip_multi_free(m: *mut zkhip_multi);
    pub fn zkhip_multi_last_error(m: *const zkhip_multi) -> *const c_char;
    pub fn zkhip_multi_use_rccl(m: *mut zkhip_multi, on: i32) -> i32;
    pub fn zkhip_multi_exchange(m: *const zkhip_multi) -> *const c_char;
    pub fn zkhip_multi_r1cs_load(m: *mut zkhip_multi, curve: i32, n: u64, l: u64, w: u64,
        rp_a: *const u64, col_a: *const u32, val_a: *const u8,
        rp_b: *const u64, col_b: *const u32, val_b: *const u8,
        rp_c: *const u64, col_c: *const u32, val_c: *const u8) -> i32;
    pub fn zkhip_multi_pk_load_g16(m: *mut zkhip_multi, curve: i32, bytes: *const u8, len: usize) -> i32;
    pub fn zkhip_multi_pk_load_gm17(m: *mut zkhip_multi, curve: i32, bytes: *const u8, len: usize) -> i32;
    /// the members' keys bound to the members' system (one member computes, all install their ranges); bound Groth16 members then
    /// split a proof's witness map between them (zkhip_multi_transform_split, on by default)
    pub fn zkhip_multi_bind(m: *mut zkhip_multi, key_bytes: *const u8, len: usize) -> i32;
    pub fn zkhip_multi_unbind(m: *mut zkhip_multi) -> i32;
    pub fn zkhip_multi_transform_split(m: *mut zkhip_multi, on: i32) -> i32;
    pub fn zkhip_prove_g16_multi(m: *mut zkhip_multi, z: *const u8, r: *const u8, s: *const u8, proof_out: *mut u8,
        timings: *mut zkhip_timings) -> i32;
    pub fn zkhip_prove_gm17_multi(m: *mut zkhip_multi, z: *const u8, d1_d2_r: *const u8, proof_out: *mut u8,
        timings: *mut zkhip_timings) -> i32;
    // throughput mode: whole key per member, `count` independent proofs dealt over the members
    pub fn zkhip_multi_pk_load_g16_replicas(m: *mut zkhip_multi, curve: i32, bytes: *const u8, len: usize) -> i32;
    pub fn zkhip_prove_g16_multi_batch(m: *mut zkhip_multi, count: u32, z: *const u8, rs: *const u8, proofs_out: *mut u8,
        timings: *mut zkhip_timings) -> i32;
}
