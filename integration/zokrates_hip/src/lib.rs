//! `Hip`: a ZoKrates proving backend that hands `generate_proof` to libzkhip (AMD MI355X) and leaves `verify` / `setup`
//! with ark.  SOURCE ONLY — never compiled in this repository's image (no Rust toolchain there).
//!
//! What this file does and where the reference does it:
//!   * `Flat::build`   — the walk of `Computation::generate_constraints` (zokrates_ark/src/lib.rs:80-129): ark variable
//!                       order, one CSR row per linear combination, the assignment moved out of the witness map.
//!                       (Callers that still hold the `out` / `witness` *files* can skip it: `zkhip_prog_parse` and
//!                       `zkhip_prog_assignment` do the same on the C side.)
//!   * blinding scalars — the draws ark makes first: `Fr::rand(rng)` x2 (ark_groth16::create_random_proof),
//!                       x3 (ark_gm17::create_random_proof), from the caller's RNG, so `--entropy` replays.
//!   * proof points    — raw little-endian coordinates back from the library, hex-encoded big-endian as
//!                       `parse_g1` / `parse_g2` do (zokrates_ark/src/lib.rs:150-218).
mod ffi;

use ark_ec::PairingEngine;
use ark_ff::{ToBytes, UniformRand};
use rand_0_8::{CryptoRng, RngCore};
use std::collections::BTreeMap;
use std::ffi::CStr;
use std::io::Read;
use std::ptr::{null, null_mut};
use zokrates_ark::{parse_fr, Ark};
use zokrates_ast::common::flat::Variable;
use zokrates_ast::ir::{LinComb, ProgIterator, Statement, Witness};
use zokrates_field::{ArkFieldExtensions, Field};
use zokrates_proof_systems::gm17::GM17;
use zokrates_proof_systems::groth16::G16;
use zokrates_proof_systems::{Backend, G1Affine, G2Affine, G2AffineFq2, Proof, Scheme};

pub struct Hip;

type ArkFr<T> = <<T as ArkFieldExtensions>::ArkEngine as PairingEngine>::Fr;

/// Column of a variable during the walk: instance index, or witness index (resolved to `l + index` at the end).
#[derive(Clone, Copy)]
enum Slot { Instance(u32), Witness(u32) }

#[derive(Default)]
struct Csr { rp: Vec<u64>, col: Vec<u32>, val: Vec<u8> }

struct Flat { n: u64, l: u64, w: u64, a: Csr, b: Csr, c: Csr, z: Vec<u8> }

impl Flat {
    fn build<'a, T: Field + ArkFieldExtensions, I: IntoIterator<Item = Statement<'a, T>>>(
        program: ProgIterator<'a, T, I>, mut witness: Witness<T>,
    ) -> Flat {
        let mut symbols: BTreeMap<Variable, Slot> = BTreeMap::new();
        let mut instance: Vec<T> = vec![T::from(1)];          // ark instance variable 0 is the constant ONE
        let mut aux: Vec<T> = vec![];
        symbols.insert(Variable::one(), Slot::Instance(0));
        let mut take = |v: &Variable, w: &mut Witness<T>| w.0.remove(v).expect("AssignmentMissing");
        for p in program.arguments.iter() {                  // lib.rs:94-113
            let value = take(&p.id, &mut witness);
            let slot = if p.private { aux.push(value); Slot::Witness(aux.len() as u32 - 1) }
                       else { instance.push(value); Slot::Instance(instance.len() as u32 - 1) };
            symbols.insert(p.id, slot);
        }
        let mut rows: [Vec<Vec<(Slot, T)>>; 3] = [vec![], vec![], vec![]];
        for statement in program.statements {                // lib.rs:115-123
            if let Statement::Constraint(s) = statement {
                let lcs: [LinComb<T>; 3] = [s.quad.left, s.quad.right, s.lin];
                for (k, lc) in lcs.into_iter().enumerate() {
                    let mut row: Vec<(Slot, T)> = vec![];
                    for (var, coeff) in lc.value {            // stored order; first sight allocates (lib.rs:47-73)
                        let slot = *symbols.entry(var).or_insert_with(|| {
                            let value = take(&var, &mut witness);
                            if var.is_output() { instance.push(value); Slot::Instance(instance.len() as u32 - 1) }
                            else { aux.push(value); Slot::Witness(aux.len() as u32 - 1) }
                        });
                        row.push((slot, coeff));
                    }
                    rows[k].push(row);
                }
            }
        }
        let (l, w) = (instance.len() as u64, aux.len() as u64);
        let column = |s: Slot| match s { Slot::Instance(i) => i, Slot::Witness(j) => l as u32 + j };
        let mut out: [Csr; 3] = Default::default();
        for k in 0..3 {
            out[k].rp.push(0);
            for row in &rows[k] {
                // duplicates are summed and zero coefficients dropped, as ark's LinearCombination / to_matrices do
                let mut merged: BTreeMap<u32, T> = BTreeMap::new();
                for (slot, coeff) in row { let e = merged.entry(column(*slot)).or_insert_with(|| T::from(0)); *e = e.clone() + coeff.clone(); }
                for (col, coeff) in merged {
                    if coeff == T::from(0) { continue; }
                    out[k].col.push(col);
                    coeff.write(&mut out[k].val).unwrap();    // 32 bytes, canonical little-endian (zokrates_field lib.rs:222-226)
                }
                out[k].rp.push(out[k].col.len() as u64);
            }
        }
        let mut z = Vec::with_capacity(32 * (l + w) as usize);
        for v in instance.iter().chain(aux.iter()) { v.write(&mut z).unwrap(); }
        let [a, b, c] = out;
        Flat { n: rows[0].len() as u64, l, w, a, b, c, z }
    }
}

fn curve_id<T: Field>() -> (i32, usize) {
    match T::name() {
        "bn128" => (ffi::ZKHIP_CURVE_BN128, 32),
        "bls12_381" => (ffi::ZKHIP_CURVE_BLS12_381, 48),
        other => panic!("hip backend: unsupported curve {}", other),
    }
}

fn check(ctx: *const ffi::zkhip_ctx, rc: i32) {
    // the reference panics on every failure of this path (zokrates_ark/src/groth16.rs:41-44 `.unwrap()`)
    if rc != 0 { panic!("zkhip ({}): {}", rc, unsafe { CStr::from_ptr(ffi::zkhip_last_error(ctx)) }.to_string_lossy()) }
}

fn le32<F: ToBytes>(x: &F) -> Vec<u8> { let mut b = vec![]; x.write(&mut b).unwrap(); b }

fn hex_be(le: &[u8]) -> String { let mut v = le.to_vec(); v.reverse(); format!("0x{}", hex::encode(v)) }

/// raw = A.x A.y | B.x.c0 B.x.c1 B.y.c0 B.y.c1 | C.x C.y (canonical LE, `fq` bytes each) + 3 infinity flags
fn points_from_raw(raw: &[u8], fq: usize) -> (G1Affine, G2Affine, G1Affine) {
    // a flagged point at infinity arrives with all-zero coordinates; the ark backend prints `zero()` = (0, 1) through
    // parse_g1 / parse_g2 (zokrates_ark/src/lib.rs:150-218), so y (G2: y.c0) becomes 1 before the hex encoding
    let mut raw = raw.to_vec();
    for (flag, y) in [(8 * fq, 1usize), (8 * fq + 1, 4), (8 * fq + 2, 7)] {
        if raw[flag] != 0 { raw[y * fq] = 1; }
    }
    let e = |i: usize| hex_be(&raw[i * fq..(i + 1) * fq]);
    (G1Affine(e(0), e(1)), G2Affine::Fq2(G2AffineFq2((e(2), e(3)), (e(4), e(5)))), G1Affine(e(6), e(7)))
}

/// Shared body: constraint system + assignment on the device, key load, one proof.  `blind` = r|s (64 B) or d1|d2|r (96 B).
fn prove_raw<T: Field + ArkFieldExtensions>(flat: &Flat, pk_bytes: &[u8], gm17: bool, blind: &[u8]) -> Vec<u8> {
    let (curve, fq) = curve_id::<T>();
    let mut raw = vec![0u8; 8 * fq + 3];
    unsafe {
        let (mut ctx, mut pk, mut cs) = (null_mut(), null_mut(), null_mut());
        check(null(), ffi::zkhip_ctx_create(0, &mut ctx));
        check(ctx, if gm17 { ffi::zkhip_pk_load_gm17(ctx, curve, pk_bytes.as_ptr(), pk_bytes.len(), &mut pk) }
                   else { ffi::zkhip_pk_load_g16(ctx, curve, pk_bytes.as_ptr(), pk_bytes.len(), &mut pk) });
        check(ctx, ffi::zkhip_r1cs_load(ctx, curve, flat.n, flat.l, flat.w,
            flat.a.rp.as_ptr(), flat.a.col.as_ptr(), flat.a.val.as_ptr(),
            flat.b.rp.as_ptr(), flat.b.col.as_ptr(), flat.b.val.as_ptr(),
            flat.c.rp.as_ptr(), flat.c.col.as_ptr(), flat.c.val.as_ptr(), &mut cs));
        check(ctx, if gm17 { ffi::zkhip_prove_gm17(ctx, pk, cs, flat.z.as_ptr(), blind.as_ptr(), raw.as_mut_ptr(), null_mut()) }
                   else { ffi::zkhip_prove_g16(ctx, pk, cs, flat.z.as_ptr(), blind.as_ptr(), blind[32..].as_ptr(), raw.as_mut_ptr(), null_mut()) });
        ffi::zkhip_r1cs_free(cs);
        ffi::zkhip_pk_free(pk);
        ffi::zkhip_ctx_free(ctx);
    }
    raw
}

/// A prover that lives across calls (zokrates_js calls `generate_proof` many times per process, zokrates_js/src/lib.rs:380-452): the
/// context, the key and the constraint system stay on the GPU, and the Groth16 key is bound to the system once
/// (`zkhip_pk_bind_r1cs`: the quotient's inverse transforms applied to the key's bases, four transforms per proof afterwards, the
/// same proof bytes).  The C++ twin is `zokrates_hip::System` + `Hip::bind` (include/zkhip_backend.hpp).
pub struct Resident { ctx: *mut ffi::zkhip_ctx, pk: *mut ffi::zkhip_pk, cs: *mut ffi::zkhip_r1cs, fq: usize }

impl Resident {
    /// Process-wide, once, before the first `Resident` (and before anything else in the process starts the HIP runtime): the
    /// hardware queues a resident prover wants (zkhip_init; the library never touches the environment by itself).
    pub fn init(hw_queues: i32) { unsafe { check(null(), ffi::zkhip_init(hw_queues)) } }

    /// `flat`: the walk of one witness of the program (only its matrices are kept); `pk_bytes`: the `proving.key` file.
    /// The struct exists BEFORE the first call that can fail, so that a panic in `check` still frees what was made (Drop).
    pub fn new<T: Field + ArkFieldExtensions>(flat: &Flat, pk_bytes: &[u8], gm17: bool) -> Self {
        let (curve, fq) = curve_id::<T>();
        let mut me = Resident { ctx: null_mut(), pk: null_mut(), cs: null_mut(), fq };
        unsafe {
            check(null(), ffi::zkhip_ctx_create(0, &mut me.ctx));
            check(me.ctx, if gm17 { ffi::zkhip_pk_load_gm17(me.ctx, curve, pk_bytes.as_ptr(), pk_bytes.len(), &mut me.pk) }
                          else { ffi::zkhip_pk_load_g16(me.ctx, curve, pk_bytes.as_ptr(), pk_bytes.len(), &mut me.pk) });
            check(me.ctx, ffi::zkhip_r1cs_load(me.ctx, curve, flat.n, flat.l, flat.w,
                flat.a.rp.as_ptr(), flat.a.col.as_ptr(), flat.a.val.as_ptr(),
                flat.b.rp.as_ptr(), flat.b.col.as_ptr(), flat.b.val.as_ptr(),
                flat.c.rp.as_ptr(), flat.c.col.as_ptr(), flat.c.val.as_ptr(), &mut me.cs));
            // both schemes bind; not enough device memory for the two extra tables (-3): the key proves as it was loaded
            let rc = ffi::zkhip_pk_bind_r1cs(me.ctx, me.pk, me.cs);
            if rc != 0 && rc != -3 { check(me.ctx, rc) }
        }
        me
    }
    /// For long-lived provers, before the first proof: the context's streams placed on the GPU's four dispatchers by plan
    /// (`ZKHIP_TUNE_PIPE_PLAN`; level or better than streams in order of first use on every measured workload: DESIGN.md §3.7).
    pub fn separate_dispatchers(&self) { check(self.ctx, unsafe { ffi::zkhip_ctx_tune(self.ctx, ffi::ZKHIP_TUNE_PIPE_PLAN, 1) }) }
    /// one Groth16 proof over the resident pair: `z` the assignment in ark order (Flat::build's), r and s drawn by the caller as ark draws them
    pub fn prove(&self, z: &[u8], r: &[u8], s: &[u8]) -> Vec<u8> {
        let mut raw = vec![0u8; 8 * self.fq + 3];
        check(self.ctx, unsafe { ffi::zkhip_prove_g16(self.ctx, self.pk, self.cs, z.as_ptr(), r.as_ptr(), s.as_ptr(), raw.as_mut_ptr(), null_mut()) });
        raw
    }
    /// ... and a GM17 one (d1 | d2 | r, 96 bytes)
    pub fn prove_gm17(&self, z: &[u8], d1_d2_r: &[u8]) -> Vec<u8> {
        let mut raw = vec![0u8; 8 * self.fq + 3];
        check(self.ctx, unsafe { ffi::zkhip_prove_gm17(self.ctx, self.pk, self.cs, z.as_ptr(), d1_d2_r.as_ptr(), raw.as_mut_ptr(), null_mut()) });
        raw
    }
}
impl Drop for Resident {
    fn drop(&mut self) {          // (every free accepts a null handle: a Resident whose construction panicked half-way drops cleanly)
        unsafe { ffi::zkhip_r1cs_free(self.cs); ffi::zkhip_pk_free(self.pk); ffi::zkhip_ctx_free(self.ctx) }
    }
}

impl<T: Field + ArkFieldExtensions> Backend<T, G16> for Hip {
    fn generate_proof<'a, I: IntoIterator<Item = Statement<'a, T>>, R: Read, G: RngCore + CryptoRng>(
        program: ProgIterator<'a, T, I>, witness: Witness<T>, mut proving_key: R, rng: &mut G,
    ) -> Proof<T, G16> {
        // public inputs exactly as the ark backend reports them (zokrates_ark/src/groth16.rs:33-38)
        let inputs = program.public_inputs_values(&witness).iter().map(|v| parse_fr::<T>(&v.into_ark())).collect();
        let flat = Flat::build::<T, I>(program, witness);
        let (r, s) = (ArkFr::<T>::rand(rng), ArkFr::<T>::rand(rng));
        let mut pk_bytes = Vec::new();
        proving_key.read_to_end(&mut pk_bytes).unwrap();
        let raw = prove_raw::<T>(&flat, &pk_bytes, false, &[le32(&r), le32(&s)].concat());
        let (a, b, c) = points_from_raw(&raw, curve_id::<T>().1);
        Proof::new(zokrates_proof_systems::groth16::ProofPoints { a, b, c }, inputs)
    }
    fn verify(vk: <G16 as Scheme<T>>::VerificationKey, proof: Proof<T, G16>) -> bool {
        <Ark as Backend<T, G16>>::verify(vk, proof)          // the pairing check stays on the CPU
    }
}

impl<T: Field + ArkFieldExtensions> Backend<T, GM17> for Hip {
    fn generate_proof<'a, I: IntoIterator<Item = Statement<'a, T>>, R: Read, G: RngCore + CryptoRng>(
        program: ProgIterator<'a, T, I>, witness: Witness<T>, mut proving_key: R, rng: &mut G,
    ) -> Proof<T, GM17> {
        let inputs = program.public_inputs_values(&witness).iter().map(|v| parse_fr::<T>(&v.into_ark())).collect();
        let flat = Flat::build::<T, I>(program, witness);
        let (d1, d2, r) = (ArkFr::<T>::rand(rng), ArkFr::<T>::rand(rng), ArkFr::<T>::rand(rng));
        let mut pk_bytes = Vec::new();
        proving_key.read_to_end(&mut pk_bytes).unwrap();
        let raw = prove_raw::<T>(&flat, &pk_bytes, true, &[le32(&d1), le32(&d2), le32(&r)].concat());
        let (a, b, c) = points_from_raw(&raw, curve_id::<T>().1);
        Proof::new(zokrates_proof_systems::gm17::ProofPoints { a, b, c }, inputs)
    }
    fn verify(vk: <GM17 as Scheme<T>>::VerificationKey, proof: Proof<T, GM17>) -> bool {
        <Ark as Backend<T, GM17>>::verify(vk, proof)
    }
}
