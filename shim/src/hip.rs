//! `src/hip.rs` of the reference crate (feature `hip`): the Rust side of libsbn254_hip.so.
//!
//! Everything the reference's prover hot path needs from the MI355X library, behind the reference's own function
//! signatures: `shim/patches/*.diff` replace the bodies of
//!   GroupElement::msm_affine / vartime_multiscalar_mul      group.rs:143-158, 171-175
//!   <[Scalar] as Commitments>::commit                        commitments.rs:131-154
//!   DensePolynomial::commit_inner                            hyrax.rs:253-281
//!   SumcheckInstanceProof::{prove_cubic, prove_cubic_batched}          sumcheck.rs:89-161, 165-330
//!   ZKSumcheckInstanceProof::{prove_cubic_with_additive_term, prove_quad}   sumcheck.rs:465-649, 657-811
//!   BulletReductionProof::prove (through DotProductProofLog::prove)    nizk/bullet.rs:41-126, nizk/mod.rs:478-494
//! by calls into this module; nothing else in the crate changes.  The `extern "C"` block is generated from
//! include/sbn254.h (`tools/gen_rust_ffi.py`); `tests/test_shim_consistency.py` checks name, arity, argument order and
//! pointer-ness of every declaration against the header, so the two cannot drift.
//!
//! Not compiled in the build image (no Rust toolchain there): complete on paper, checked mechanically as far as that goes.
#![allow(non_camel_case_types, non_snake_case, clippy::too_many_arguments, clippy::missing_safety_doc)]

use std::borrow::Cow;
use std::collections::HashMap;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr::{null, null_mut};
use std::sync::{Arc, Mutex, OnceLock};

use ark_bn254::{Fq, G1Affine, G1Projective};
use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::PrimeField;
use merlin::Transcript;

use crate::commitments::MultiCommitGens;
use crate::group::GroupElement;
use crate::hyrax::DensePolynomial;
use crate::scalar::Scalar;
use crate::sumcheck::SumcheckInstanceProof;
use crate::transcript::{AppendToTranscript, ProofTranscript};
use crate::unipoly::{CompressedUniPoly, UniPoly};

// ---- opaque handles of the C ABI ------------------------------------------------------------------------------------
#[repr(C)] pub struct sbn_ctx { _p: [u8; 0] }
#[repr(C)] pub struct sbn_bases { _p: [u8; 0] }
#[repr(C)] pub struct sbn_table { _p: [u8; 0] }
#[repr(C)] pub struct sbn_sumcheck { _p: [u8; 0] }
#[repr(C)] pub struct sbn_bullet { _p: [u8; 0] }
#[repr(C)] pub struct sbn_group { _p: [u8; 0] }
#[repr(C)] pub struct sbn_group_bases { _p: [u8; 0] }

pub const SBN_OK: c_int = 0;
pub const SBN_SCALARS_MONT: u32 = 1;
pub const SBN_POINTS_MONT: u32 = 2;

/// MSMs below this many terms stay on arkworks: the Σ-protocol steps commit 1–5 scalars at a time (`nizk/mod.rs`,
/// `sumcheck.rs:539-634`) and a device call is a ~20 µs round trip + a ~0.3 ms latency-bound bucket pass.
pub const MIN_GPU_MSM: usize = 256;

#[link(name = "sbn254_hip")]
extern "C" {
    // GENERATED-BEGIN (tools/gen_rust_ffi.py from include/sbn254.h — do not edit by hand)
    pub fn sbn_ctx_create(device: c_int, out: *mut *mut sbn_ctx) -> c_int;
    pub fn sbn_ctx_destroy(ctx: *mut sbn_ctx);
    pub fn sbn_last_error(ctx: *const sbn_ctx) -> *const c_char;
    pub fn sbn_ctx_set_stream(ctx: *mut sbn_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn sbn_ctx_sync(ctx: *mut sbn_ctx) -> c_int;
    pub fn sbn_version() -> *const c_char;
    pub fn sbn_dev_alloc(ctx: *mut sbn_ctx, bytes: usize, out_dev: *mut *mut c_void) -> c_int;
    pub fn sbn_dev_free(ctx: *mut sbn_ctx, dev: *mut c_void) -> c_int;
    pub fn sbn_dev_upload(ctx: *mut sbn_ctx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn sbn_dev_download(ctx: *mut sbn_ctx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn sbn_msm(ctx: *mut sbn_ctx, scalars: *const u8, points: *const u8, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_msm_jacobian(ctx: *mut sbn_ctx, scalars: *const u8, points_xyz: *const u8, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_bases_upload(ctx: *mut sbn_ctx, g_xy: *const u8, n: usize, h_xy: *const u8, flags: u32, out: *mut *mut sbn_bases) -> c_int;
    pub fn sbn_bases_free(ctx: *mut sbn_ctx, b: *mut sbn_bases);
    pub fn sbn_bases_len(b: *const sbn_bases) -> usize;
    pub fn sbn_bases_precompute(ctx: *mut sbn_ctx, b: *mut sbn_bases, max_bytes: usize, window_bits: *mut c_int) -> c_int;
    pub fn sbn_gens_new(ctx: *mut sbn_ctx, n: usize, label: *const u8, label_len: usize, out_xy: *mut u8, out: *mut *mut sbn_bases) -> c_int;
    pub fn sbn_bases_synthetic(ctx: *mut sbn_ctx, n: usize, first: u64, s0: *const u8, d: *const u8, out: *mut *mut sbn_bases) -> c_int;
    pub fn sbn_scalars_synthetic(ctx: *mut sbn_ctx, seed: u64, first: u64, n: usize, out_dev: *mut c_void) -> c_int;
    pub fn sbn_bases_download(ctx: *mut sbn_ctx, b: *const sbn_bases, first: usize, count: usize, out_xy: *mut u8) -> c_int;
    pub fn sbn_bases_split_at(ctx: *mut sbn_ctx, b: *const sbn_bases, mid: usize, left: *mut *mut sbn_bases, right: *mut *mut sbn_bases) -> c_int;
    pub fn sbn_bases_scale(ctx: *mut sbn_ctx, b: *const sbn_bases, s: *const u8, out: *mut *mut sbn_bases) -> c_int;
    pub fn sbn_msm_bases(ctx: *mut sbn_ctx, b: *const sbn_bases, scalars: *const u8, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_msm_bases_dev(ctx: *mut sbn_ctx, b: *const sbn_bases, scalars_dev: *const c_void, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_commit_rows(ctx: *mut sbn_ctx, b: *const sbn_bases, z: *const u8, blinds: *const u8, l: usize, r: usize, flags: u32, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_commit_rows_dev(ctx: *mut sbn_ctx, b: *const sbn_bases, z_dev: *const c_void, blinds_dev: *const c_void, l: usize, r: usize, flags: u32, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_g1_compress(xy: *const u8, n: usize, out32: *mut u8) -> c_int;
    pub fn sbn_g1_sum(xy: *const u8, n: usize, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_unipoly_from_evals(evals: *const u8, n: usize, coeffs: *mut u8) -> c_int;
    pub fn sbn_unipoly_eval(coeffs: *const u8, n: usize, r: *const u8, out: *mut u8) -> c_int;
    pub fn sbn_factored_lens(ell: usize, left: *mut usize, right: *mut usize);
    pub fn sbn_table_upload(ctx: *mut sbn_ctx, z: *const u8, len: usize, flags: u32, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_table_from_dev(ctx: *mut sbn_ctx, z_dev: *const c_void, len: usize, flags: u32, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_table_free(ctx: *mut sbn_ctx, t: *mut sbn_table);
    pub fn sbn_table_len(t: *const sbn_table) -> usize;
    pub fn sbn_table_download(ctx: *mut sbn_ctx, t: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_table_read0(ctx: *mut sbn_ctx, t: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_table_read0_many(ctx: *mut sbn_ctx, ts: *const *const sbn_table, count: usize, out: *mut u8) -> c_int;
    pub fn sbn_bind_top(ctx: *mut sbn_ctx, t: *mut sbn_table, r: *const u8) -> c_int;
    pub fn sbn_bind_top_many(ctx: *mut sbn_ctx, ts: *const *mut sbn_table, count: usize, r: *const u8) -> c_int;
    pub fn sbn_sc_eval_cubic(ctx: *mut sbn_ctx, a: *const sbn_table, b: *const sbn_table, c: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_sc_eval_cubic_batched(ctx: *mut sbn_ctx, a: *const *const sbn_table, b: *const *const sbn_table, c: *const *const sbn_table, count: usize, out: *mut u8) -> c_int;
    pub fn sbn_sc_eval_r1cs(ctx: *mut sbn_ctx, tau: *const sbn_table, az: *const sbn_table, bz: *const sbn_table, cz: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_sc_eval_quad(ctx: *mut sbn_ctx, z: *const sbn_table, abc: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_sc_bind_eval_cubic_batched(ctx: *mut sbn_ctx, a: *const *mut sbn_table, b: *const *mut sbn_table, c: *const *mut sbn_table, count: usize, r: *const u8, out: *mut u8) -> c_int;
    pub fn sbn_sc_bind_eval_r1cs(ctx: *mut sbn_ctx, tau: *mut sbn_table, az: *mut sbn_table, bz: *mut sbn_table, cz: *mut sbn_table, r: *const u8, out: *mut u8) -> c_int;
    pub fn sbn_sc_bind_eval_quad(ctx: *mut sbn_ctx, z: *mut sbn_table, abc: *mut sbn_table, r: *const u8, out: *mut u8) -> c_int;
    pub fn sbn_sumcheck_begin(ctx: *mut sbn_ctx, a_par: *const *const sbn_table, b_par: *const *const sbn_table, c_par: *const sbn_table, n_par: usize, a_seq: *const *const sbn_table, b_seq: *const *const sbn_table, c_seq: *const *const sbn_table, n_seq: usize, coeffs: *const u8, out_evals: *mut u8, out: *mut *mut sbn_sumcheck) -> c_int;
    pub fn sbn_sumcheck_begin_eq(ctx: *mut sbn_ctx, a_par: *const *const sbn_table, b_par: *const *const sbn_table, n_par: usize, rand: *const u8, ell: usize, a_seq: *const *const sbn_table, b_seq: *const *const sbn_table, c_seq: *const *const sbn_table, n_seq: usize, coeffs: *const u8, out_evals: *mut u8, out: *mut *mut sbn_sumcheck) -> c_int;
    pub fn sbn_sumcheck_round(ctx: *mut sbn_ctx, st: *mut sbn_sumcheck, r: *const u8, out_evals: *mut u8) -> c_int;
    pub fn sbn_sumcheck_len(st: *const sbn_sumcheck) -> usize;
    pub fn sbn_sumcheck_finish(ctx: *mut sbn_ctx, st: *mut sbn_sumcheck, finals: *mut u8) -> c_int;
    pub fn sbn_sumcheck_free(ctx: *mut sbn_ctx, st: *mut sbn_sumcheck);
    pub fn sbn_eq_evals(ctx: *mut sbn_ctx, r: *const u8, ell: usize, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_table_dot(ctx: *mut sbn_ctx, a: *const sbn_table, b: *const sbn_table, out: *mut u8) -> c_int;
    pub fn sbn_table_evaluate(ctx: *mut sbn_ctx, z: *const sbn_table, r: *const u8, ell: usize, out: *mut u8) -> c_int;
    pub fn sbn_table_evaluate_many(ctx: *mut sbn_ctx, z: *const *const sbn_table, count: usize, r: *const u8, ell: usize, out: *mut u8) -> c_int;
    pub fn sbn_table_bound(ctx: *mut sbn_ctx, z: *const sbn_table, lvec: *const sbn_table, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_bullet_begin(ctx: *mut sbn_ctx, g: *const sbn_bases, q_xy: *const u8, a: *const sbn_table, b: *const sbn_table, blind: *const u8, gamma_xy: *mut u8, gamma_is_inf: *mut c_int, out: *mut *mut sbn_bullet) -> c_int;
    pub fn sbn_bullet_begin_scaled(ctx: *mut sbn_ctx, g: *const sbn_bases, q_base_xy: *const u8, q_scale: *const u8, a: *const sbn_table, b: *const sbn_table, blind: *const u8, gamma_xy: *mut u8, gamma_is_inf: *mut c_int, out: *mut *mut sbn_bullet) -> c_int;
    pub fn sbn_bullet_free(ctx: *mut sbn_ctx, st: *mut sbn_bullet);
    pub fn sbn_bullet_len(st: *const sbn_bullet) -> usize;
    pub fn sbn_bullet_cross(ctx: *mut sbn_ctx, st: *mut sbn_bullet, blind_l: *const u8, blind_r: *const u8, l_xy: *mut u8, l_is_inf: *mut c_int, r_xy: *mut u8, r_is_inf: *mut c_int, c_l: *mut u8, c_r: *mut u8) -> c_int;
    pub fn sbn_bullet_fold(ctx: *mut sbn_ctx, st: *mut sbn_bullet, u: *const u8, u_inv: *const u8) -> c_int;
    pub fn sbn_bullet_fold_cross(ctx: *mut sbn_ctx, st: *mut sbn_bullet, u: *const u8, u_inv: *const u8, blind_l: *const u8, blind_r: *const u8, l_xy: *mut u8, l_is_inf: *mut c_int, r_xy: *mut u8, r_is_inf: *mut c_int, c_l: *mut u8, c_r: *mut u8) -> c_int;
    pub fn sbn_bullet_finish(ctx: *mut sbn_ctx, st: *mut sbn_bullet, a_hat: *mut u8, b_hat: *mut u8, g_hat_xy: *mut u8, g_hat_is_inf: *mut c_int) -> c_int;
    pub fn sbn_hash_layer(ctx: *mut sbn_ctx, addr_dev: *const c_void, val: *const sbn_table, ts_dev: *const c_void, ts_add: u32, r_hash: *const u8, r_multiset: *const u8, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_hash_layer_pair(ctx: *mut sbn_ctx, addr_dev: *const c_void, val: *const sbn_table, ts_a_dev: *const c_void, ts_a_add: u32, ts_b_dev: *const c_void, ts_b_add: u32, r_hash: *const u8, r_multiset: *const u8, out_a: *mut *mut sbn_table, out_b: *mut *mut sbn_table) -> c_int;
    pub fn sbn_product_layer(ctx: *mut sbn_ctx, input: *const sbn_table, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_product_circuit(ctx: *mut sbn_ctx, input: *const sbn_table, layers: *mut *mut sbn_table, cap: usize, count: *mut usize) -> c_int;
    pub fn sbn_product_circuit_many(ctx: *mut sbn_ctx, ins: *const *const sbn_table, n: usize, layers: *mut *mut sbn_table, cap: usize, count: *mut usize) -> c_int;
    pub fn sbn_table_halves(ctx: *mut sbn_ctx, t: *const sbn_table, left: *mut *mut sbn_table, right: *mut *mut sbn_table) -> c_int;
    pub fn sbn_table_slice(ctx: *mut sbn_ctx, t: *const sbn_table, first: usize, len: usize, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_gather_merge(ctx: *mut sbn_ctx, mem: *const *const sbn_table, addr_dev: *const *const c_void, count: usize, n: usize, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_gather_merge_rows(ctx: *mut sbn_ctx, mem: *const *const sbn_table, addr_dev: *const *const c_void, count: usize, n: usize, r: usize, row0: usize, rstep: usize, nrows: usize, out: *mut *mut sbn_table) -> c_int;
    pub fn sbn_commit_table(ctx: *mut sbn_ctx, b: *const sbn_bases, t: *const sbn_table, blinds: *const u8, l: usize, r: usize, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_group_create(devices: *const c_int, n: usize, out: *mut *mut sbn_group) -> c_int;
    pub fn sbn_group_destroy(g: *mut sbn_group);
    pub fn sbn_group_size(g: *const sbn_group) -> usize;
    pub fn sbn_group_ctx(g: *mut sbn_group, i: usize) -> *mut sbn_ctx;
    pub fn sbn_group_last_error(g: *const sbn_group) -> *const c_char;
    pub fn sbn_group_bases_upload(g: *mut sbn_group, g_xy: *const u8, n: usize, h_xy: *const u8, flags: u32, out: *mut *mut sbn_group_bases) -> c_int;
    pub fn sbn_group_gens_new(g: *mut sbn_group, n: usize, label: *const u8, label_len: usize, out_xy: *mut u8, out: *mut *mut sbn_group_bases) -> c_int;
    pub fn sbn_group_bases_precompute(g: *mut sbn_group, gb: *mut sbn_group_bases, max_bytes_per_device: usize, window_bits: *mut c_int) -> c_int;
    pub fn sbn_group_bases_free(g: *mut sbn_group, gb: *mut sbn_group_bases);
    pub fn sbn_group_commit_rows(g: *mut sbn_group, gb: *const sbn_group_bases, z: *const u8, blinds: *const u8, l: usize, r: usize, flags: u32, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_group_commit_rows_dev(g: *mut sbn_group, gb: *const sbn_group_bases, z_dev: *const *const c_void, blinds_dev: *const *const c_void, l: usize, r: usize, flags: u32, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_group_gather_commit(g: *mut sbn_group, gb: *const sbn_group_bases, mem: *const *const sbn_table, addr_dev: *const *const c_void, count: usize, n: usize, l: usize, r: usize, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn sbn_group_msm(g: *mut sbn_group, scalars: *const u8, points: *const u8, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_group_bases_upload_ranges(g: *mut sbn_group, g_xy: *const u8, n: usize, flags: u32, out: *mut *mut sbn_group_bases) -> c_int;
    pub fn sbn_group_bases_synthetic_ranges(g: *mut sbn_group, n: usize, s0: *const u8, d: *const u8, out: *mut *mut sbn_group_bases) -> c_int;
    pub fn sbn_group_range(gb: *const sbn_group_bases, device: usize, lo: *mut usize, hi: *mut usize);
    pub fn sbn_group_msm_bases(g: *mut sbn_group, gb: *const sbn_group_bases, scalars: *const u8, n: usize, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_group_msm_bases_dev(g: *mut sbn_group, gb: *const sbn_group_bases, scalars_dev: *const *const c_void, flags: u32, out_xy: *mut u8, out_is_inf: *mut c_int) -> c_int;
    pub fn sbn_prof_enable(ctx: *mut sbn_ctx, on: c_int) -> c_int;
    pub fn sbn_prof_reset(ctx: *mut sbn_ctx) -> c_int;
    pub fn sbn_prof_count(ctx: *mut sbn_ctx) -> c_int;
    pub fn sbn_prof_get(ctx: *mut sbn_ctx, i: c_int, name: *mut *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn sbn_prof_last_job(ctx: *mut sbn_ctx, out: *mut u64) -> c_int;
    // GENERATED-END
}

// ---- errors: the reference's prover functions are infallible (assert! / panic on misuse, errors.rs:19-31 is verifier-only) ----
pub fn check(rc: c_int) {
    if rc != SBN_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(sbn_last_error(ctx())) }.to_string_lossy().into_owned();
        panic!("sbn254 (rc = {}): {}", rc, msg);
    }
}
pub fn check_group(rc: c_int) {
    if rc != SBN_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(sbn_group_last_error(group())) }.to_string_lossy().into_owned();
        panic!("sbn254 group (rc = {}): {}", rc, msg);
    }
}

// ---- one context per process; the library serialises calls on a context with a mutex (hyrax.rs:259-261 may enter from rayon workers) ----
struct CtxPtr(*mut sbn_ctx);
unsafe impl Send for CtxPtr {}
unsafe impl Sync for CtxPtr {}
pub fn ctx() -> *mut sbn_ctx {
    static CTX: OnceLock<CtxPtr> = OnceLock::new();
    CTX.get_or_init(|| {
        let dev: c_int = std::env::var("SBN_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut c = null_mut();
        let rc = unsafe { sbn_ctx_create(dev, &mut c) };
        assert_eq!(rc, SBN_OK, "sbn254: no usable gfx950 device (there is no CPU fallback in the library; build without --features hip)");
        CtxPtr(c)
    })
    .0
}
/// all devices listed in SBN_DEVICES (e.g. "0,1,2,3,4,5,6,7"); a single-device list when the variable is absent
struct GroupPtr(*mut sbn_group);
unsafe impl Send for GroupPtr {}
unsafe impl Sync for GroupPtr {}
pub fn group() -> *mut sbn_group {
    static G: OnceLock<GroupPtr> = OnceLock::new();
    G.get_or_init(|| {
        let devs: Vec<c_int> = std::env::var("SBN_DEVICES").ok().map(|s| s.split(',').filter_map(|x| x.trim().parse().ok()).collect()).unwrap_or_else(|| vec![0]);
        let mut g = null_mut();
        let rc = unsafe { sbn_group_create(devs.as_ptr(), devs.len(), &mut g) };
        assert_eq!(rc, SBN_OK, "sbn254: sbn_group_create failed for devices {:?}", devs);
        GroupPtr(g)
    })
    .0
}
pub fn group_size() -> usize { unsafe { sbn_group_size(group()) } }

// ---- owned handles --------------------------------------------------------------------------------------------------
/// device-resident generator table (MultiCommitGens.{G_affine, h_affine}, commitments.rs:17-27)
pub struct Bases(pub *mut sbn_bases);
unsafe impl Send for Bases {}
unsafe impl Sync for Bases {}
impl Drop for Bases { fn drop(&mut self) { if !self.0.is_null() { unsafe { sbn_bases_free(ctx(), self.0) } } } }
impl std::fmt::Debug for Bases { fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "Bases({:p})", self.0) } }

/// the same generator set on every device of the group
pub struct GroupBases(pub *mut sbn_group_bases);
unsafe impl Send for GroupBases {}
unsafe impl Sync for GroupBases {}
impl Drop for GroupBases { fn drop(&mut self) { if !self.0.is_null() { unsafe { sbn_group_bases_free(group(), self.0) } } } }
impl std::fmt::Debug for GroupBases { fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "GroupBases({:p})", self.0) } }

/// device-resident Fr table (DensePolynomial.Z, hyrax.rs:155-160)
pub struct Table(pub *mut sbn_table);
impl Drop for Table { fn drop(&mut self) { if !self.0.is_null() { unsafe { sbn_table_free(ctx(), self.0) } } } }
impl Table {
    pub fn upload(z: &[Scalar]) -> Table {
        let bytes = scalars_mont_bytes(z);
        let mut t = null_mut();
        check(unsafe { sbn_table_upload(ctx(), bytes.as_ptr(), z.len(), SBN_SCALARS_MONT, &mut t) });
        Table(t)
    }
    pub fn of(p: &DensePolynomial) -> Table { Table::upload(&p.vec()[..p.len()]) }
    pub fn len(&self) -> usize { unsafe { sbn_table_len(self.0) } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
}
struct SumcheckState(*mut sbn_sumcheck);
impl Drop for SumcheckState { fn drop(&mut self) { if !self.0.is_null() { unsafe { sbn_sumcheck_free(ctx(), self.0) } } } }
struct BulletState(*mut sbn_bullet);
impl Drop for BulletState { fn drop(&mut self) { if !self.0.is_null() { unsafe { sbn_bullet_free(ctx(), self.0) } } } }

/// the lazily created device twin of a `MultiCommitGens` (the field `dev` the commitments.rs patch adds next to the
/// `#[serde(skip)]` affine cache): shared by clones, rebuilt after deserialisation, dropped with the last clone
#[derive(Clone, Debug, Default)]
pub struct GensDev {
    one: Arc<OnceLock<Bases>>,
    all: Arc<OnceLock<GroupBases>>,
}
impl GensDev {
    pub fn bases(&self, gens: &MultiCommitGens) -> *const sbn_bases {
        self.one.get_or_init(|| {
            let g: Vec<u8> = gens.G_affine.iter().flat_map(point_xy_mont).collect();
            let h = point_xy_mont(&gens.h_affine);
            let mut b = null_mut();
            check(unsafe { sbn_bases_upload(ctx(), g.as_ptr(), gens.n, h.as_ptr(), SBN_POINTS_MONT, &mut b) });
            // a generator set that serves every proof of a circuit gets its fixed-base lookup table (SBN_LOOKUP_GB, default 0 = bucket method)
            let gb: usize = std::env::var("SBN_LOOKUP_GB").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            if gb > 0 && gens.n >= 1024 { let mut cbits = 0; check(unsafe { sbn_bases_precompute(ctx(), b, gb << 30, &mut cbits) }); }
            Bases(b)
        })
        .0
    }
    pub fn group_bases(&self, gens: &MultiCommitGens) -> *const sbn_group_bases {
        self.all.get_or_init(|| {
            let g: Vec<u8> = gens.G_affine.iter().flat_map(point_xy_mont).collect();
            let h = point_xy_mont(&gens.h_affine);
            let mut b = null_mut();
            check_group(unsafe { sbn_group_bases_upload(group(), g.as_ptr(), gens.n, h.as_ptr(), SBN_POINTS_MONT, &mut b) });
            GroupBases(b)
        })
        .0
    }
}

// ---- data conversion at the boundary ---------------------------------------------------------------------------------
// ark-ff keeps Fr / Fq as Montgomery limbs (R = 2^256), 4 x u64 little-endian: `Fp(pub BigInt<4>, PhantomData)`, `BigInt(pub [u64; 4])`.
// The library takes exactly those 32 bytes with SBN_SCALARS_MONT / SBN_POINTS_MONT (it converts with one product per element as
// it reads them).  `Scalar` is `#[repr(transparent)]` over `Fr` after patches/0002; whether `Fr` itself lies in memory as its four
// limbs is checked ONCE at run time (below) — if it ever does not, the slice is copied limb by limb instead of being reinterpreted.
fn limbs_le(x: &[u64; 4]) -> [u8; 32] {
    let mut o = [0u8; 32];
    for (i, l) in x.iter().enumerate() { o[8 * i..8 * i + 8].copy_from_slice(&l.to_le_bytes()); }
    o
}
fn scalar_layout_is_raw_limbs() -> bool {
    static OK: OnceLock<bool> = OnceLock::new();
    *OK.get_or_init(|| {
        if std::mem::size_of::<Scalar>() != 32 || std::mem::align_of::<Scalar>() > 8 || cfg!(target_endian = "big") { return false; }
        let probe = [Scalar::from_u64(0x0123_4567_89ab_cdef), Scalar::one(), Scalar::zero() - Scalar::one()];
        let raw = unsafe { core::slice::from_raw_parts(probe.as_ptr() as *const u8, 32 * probe.len()) };
        probe.iter().enumerate().all(|(i, s)| raw[32 * i..32 * i + 32] == limbs_le(&(s.0).0 .0))
    })
}
/// the slice as the library's SBN_SCALARS_MONT input: borrowed when the layout allows it (no copy of a 1 GiB matrix), copied otherwise
pub fn scalars_mont_bytes(s: &[Scalar]) -> Cow<'_, [u8]> {
    if scalar_layout_is_raw_limbs() {
        Cow::Borrowed(unsafe { core::slice::from_raw_parts(s.as_ptr() as *const u8, 32 * s.len()) })
    } else {
        Cow::Owned(s.iter().flat_map(|x| limbs_le(&(x.0).0 .0)).collect())
    }
}
/// canonical 32-byte little-endian encodings (Scalar::to_bytes, scalar.rs:75-84): challenges, blinds, coefficients
pub fn scalars_canonical(s: &[Scalar]) -> Vec<u8> { s.iter().flat_map(|x| x.to_bytes()).collect() }
/// a canonical 32-byte little-endian value coming back from the device
pub fn sc(b: &[u8]) -> Scalar {
    let a: [u8; 32] = b[..32].try_into().unwrap();
    Scalar::from_bytes(&a).expect("sbn254 returned a non-canonical scalar")
}
/// {x, y, infinity} -> x || y in ark-ff's Montgomery limbs; all-zero for infinity
pub fn point_xy_mont(p: &G1Affine) -> [u8; 64] {
    let mut o = [0u8; 64];
    if !p.is_zero() {
        o[..32].copy_from_slice(&limbs_le(&(p.x.0).0));
        o[32..].copy_from_slice(&limbs_le(&(p.y.0).0));
    }
    o
}
/// canonical x || y (what every entry point returns) -> GroupElement
pub fn group_from_xy(xy: &[u8], inf: bool) -> GroupElement {
    if inf { return GroupElement::identity(); }
    let x = Fq::from_le_bytes_mod_order(&xy[..32]);
    let y = Fq::from_le_bytes_mod_order(&xy[32..64]);
    GroupElement::from_affine(G1Affine::new_unchecked(x, y))
}
fn affine_of(g: &GroupElement) -> G1Affine { g.inner().into_affine() }

// ---- B1: single MSM (group.rs:143-158, 171-175) -----------------------------------------------------------------------
/// `None` when the call should stay on arkworks (tiny MSMs, length mismatch: the reference maps that to the identity)
pub fn msm_affine(scalars: &[Scalar], points: &[G1Affine]) -> Option<GroupElement> {
    if scalars.len() != points.len() || scalars.len() < MIN_GPU_MSM { return None; }
    let pts: Vec<u8> = points.iter().flat_map(point_xy_mont).collect();
    let sb = scalars_mont_bytes(scalars);
    let mut xy = [0u8; 64];
    let mut inf: c_int = 0;
    if group_size() > 1 && scalars.len() >= (1 << 22) {
        // a LARGE MSM (config 4): contiguous base-point ranges, one per device, N partial sums folded on the host
        check_group(unsafe { sbn_group_msm(group(), sb.as_ptr(), pts.as_ptr(), scalars.len(), SBN_SCALARS_MONT | SBN_POINTS_MONT, xy.as_mut_ptr(), &mut inf) });
    } else {
        check(unsafe { sbn_msm(ctx(), sb.as_ptr(), pts.as_ptr(), scalars.len(), SBN_SCALARS_MONT | SBN_POINTS_MONT, xy.as_mut_ptr(), &mut inf) });
    }
    Some(group_from_xy(&xy, inf != 0))
}
/// vartime_multiscalar_mul takes projective points; the n Jacobian triples are normalised on the device (group.rs:153 inverts per point)
pub fn msm_projective(scalars: &[Scalar], points: &[G1Projective]) -> Option<GroupElement> {
    if scalars.len() != points.len() || scalars.len() < MIN_GPU_MSM { return None; }
    let mut xyz = Vec::with_capacity(96 * points.len());
    for p in points {
        xyz.extend_from_slice(&limbs_le(&(p.x.0).0)); xyz.extend_from_slice(&limbs_le(&(p.y.0).0)); xyz.extend_from_slice(&limbs_le(&(p.z.0).0));
    }
    let sb = scalars_mont_bytes(scalars);
    let mut xy = [0u8; 64];
    let mut inf: c_int = 0;
    check(unsafe { sbn_msm_jacobian(ctx(), sb.as_ptr(), xyz.as_ptr(), scalars.len(), SBN_SCALARS_MONT | SBN_POINTS_MONT, xy.as_mut_ptr(), &mut inf) });
    Some(group_from_xy(&xy, inf != 0))
}

// ---- B2: Pedersen / Hyrax commitments (commitments.rs:144-154, hyrax.rs:253-308) ----------------------------------------
/// <[Scalar] as Commitments>::commit: MSM(self || blind, G || h) — one row
pub fn commit_row(scalars: &[Scalar], blind: &Scalar, gens: &MultiCommitGens) -> Option<GroupElement> {
    assert_eq!(gens.n, scalars.len());                                          // commitments.rs:146
    if scalars.len() < MIN_GPU_MSM { return None; }
    let sb = scalars_mont_bytes(scalars);
    let bb = scalars_mont_bytes(core::slice::from_ref(blind));
    let mut xy = [0u8; 64];
    let mut inf = [0u8; 1];
    check(unsafe { sbn_commit_rows(ctx(), gens.dev.bases(gens), sb.as_ptr(), bb.as_ptr(), 1, scalars.len(), SBN_SCALARS_MONT, xy.as_mut_ptr(), inf.as_mut_ptr()) });
    Some(group_from_xy(&xy, inf[0] != 0))
}
/// DensePolynomial::commit_inner: ONE call for the L x R matrix (the reference spawns L rayon tasks, each cloning R bases).
/// With several devices the rows are dealt i mod N over the group (rows are independent: no reduction).
pub fn commit_rows(z: &[Scalar], blinds: &[Scalar], gens: &MultiCommitGens) -> Vec<GroupElement> {
    let l = blinds.len();
    let r = z.len() / l;
    assert_eq!(l * r, z.len());                                                 // hyrax.rs:257
    assert_eq!(gens.n, r);                                                      // commitments.rs:146
    let all_zero = blinds.iter().all(|b| *b == Scalar::zero());                  // random_tape == None (hyrax.rs:301-305)
    let zb = scalars_mont_bytes(z);
    let bb = scalars_mont_bytes(blinds);
    let bp = if all_zero { null() } else { bb.as_ptr() };
    let mut xy = vec![0u8; 64 * l];
    let mut inf = vec![0u8; l];
    if group_size() > 1 && l >= 2 * group_size() {
        check_group(unsafe { sbn_group_commit_rows(group(), gens.dev.group_bases(gens), zb.as_ptr(), bp, l, r, SBN_SCALARS_MONT, xy.as_mut_ptr(), inf.as_mut_ptr()) });
    } else {
        check(unsafe { sbn_commit_rows(ctx(), gens.dev.bases(gens), zb.as_ptr(), bp, l, r, SBN_SCALARS_MONT, xy.as_mut_ptr(), inf.as_mut_ptr()) });
    }
    (0..l).map(|i| group_from_xy(&xy[64 * i..64 * i + 64], inf[i] != 0)).collect()
}

// ---- B3: the sumcheck prover loops --------------------------------------------------------------------------------------
// The `comb_func` closures cannot cross FFI; only three are ever passed (A*B*C at product_tree.rs:178-181, 275-278;
// tau*(Az*Bz - Cz) at r1csproof.rs:288-292; z*ABC at r1csproof.rs:389-390), so each prover has its fixed kernel.

fn triple(ev: &[u8]) -> (Scalar, Scalar, Scalar) { (sc(&ev[..32]), sc(&ev[32..64]), sc(&ev[64..96])) }

/// SumcheckInstanceProof::prove_cubic_batched (sumcheck.rs:165-330), comb_func = A*B*C.  The device state returns per round what the
/// transcript absorbs — the coeffs-combined (e0, e2, e3) of :269-271 — and binds every table (:289-299).  The reference's
/// DensePolynomials are left with their single final entry, as its own loop leaves them.
pub fn prove_cubic_batched(
    claim: &Scalar,
    num_rounds: usize,
    poly_vec_par: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut DensePolynomial),
    poly_vec_seq: (&mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>, &mut Vec<&mut DensePolynomial>),
    coeffs: &[Scalar],
    transcript: &mut Transcript,
) -> (SumcheckInstanceProof, Vec<Scalar>, (Vec<Scalar>, Vec<Scalar>, Scalar), (Vec<Scalar>, Vec<Scalar>, Vec<Scalar>)) {
    let (poly_A_vec_par, poly_B_vec_par, poly_C_par) = poly_vec_par;
    let (poly_A_vec_seq, poly_B_vec_seq, poly_C_vec_seq) = poly_vec_seq;
    let (n_par, n_seq) = (poly_A_vec_par.len(), poly_A_vec_seq.len());
    assert_eq!(coeffs.len(), n_par + n_seq);

    // the tables go up once; the state never writes them
    let ta_par: Vec<Table> = poly_A_vec_par.iter().map(|p| Table::of(p)).collect();
    let tb_par: Vec<Table> = poly_B_vec_par.iter().map(|p| Table::of(p)).collect();
    let tc_par = if n_par > 0 { Some(Table::of(poly_C_par)) } else { None };
    let ta_seq: Vec<Table> = poly_A_vec_seq.iter().map(|p| Table::of(p)).collect();
    let tb_seq: Vec<Table> = poly_B_vec_seq.iter().map(|p| Table::of(p)).collect();
    let tc_seq: Vec<Table> = poly_C_vec_seq.iter().map(|p| Table::of(p)).collect();
    let ptrs = |v: &Vec<Table>| -> Vec<*const sbn_table> { v.iter().map(|t| t.0 as *const sbn_table).collect() };
    let (pa, pb, sa, sb_, sc_) = (ptrs(&ta_par), ptrs(&tb_par), ptrs(&ta_seq), ptrs(&tb_seq), ptrs(&tc_seq));

    let mut ev = [0u8; 96];
    let mut st = null_mut();
    check(unsafe {
        sbn_sumcheck_begin(ctx(), pa.as_ptr(), pb.as_ptr(), tc_par.as_ref().map_or(null(), |t| t.0 as *const sbn_table), n_par,
                           sa.as_ptr(), sb_.as_ptr(), sc_.as_ptr(), n_seq, scalars_canonical(coeffs).as_ptr(), ev.as_mut_ptr(), &mut st)
    });
    let st = SumcheckState(st);

    let mut e = *claim;
    let mut r: Vec<Scalar> = Vec::new();
    let mut cubic_polys: Vec<CompressedUniPoly> = Vec::new();
    for _j in 0..num_rounds {
        let (c0, c2, c3) = triple(&ev);                                        // evals_combined_0 / _2 / _3 (:269-271)
        let poly = UniPoly::from_evals(&[c0, e - c0, c2, c3]);                  // :273-279
        poly.append_to_transcript(b"poly", transcript);                         // :282
        let r_j = transcript.challenge_scalar(b"challenge_nextround");          // :285
        r.push(r_j);
        check(unsafe { sbn_sumcheck_round(ctx(), st.0, r_j.to_bytes().as_ptr(), ev.as_mut_ptr()) });   // binds every table (:289-299), returns the next round's sums
        e = poly.evaluate(&r_j);
        cubic_polys.push(poly.compress());
    }
    // final claims (:302-318), in the order A_par.., B_par.., C_par, A_seq.., B_seq.., C_seq..
    let ntab = 2 * n_par + usize::from(n_par > 0) + 3 * n_seq;
    let mut fin = vec![0u8; 32 * ntab];
    check(unsafe { sbn_sumcheck_finish(ctx(), st.0, fin.as_mut_ptr()) });
    let f = |i: usize| sc(&fin[32 * i..32 * i + 32]);
    let a_par: Vec<Scalar> = (0..n_par).map(f).collect();
    let b_par: Vec<Scalar> = (0..n_par).map(|i| f(n_par + i)).collect();
    let c_par = if n_par > 0 { f(2 * n_par) } else { poly_C_par[0] };
    let o = 2 * n_par + usize::from(n_par > 0);
    let a_seq: Vec<Scalar> = (0..n_seq).map(|i| f(o + i)).collect();
    let b_seq: Vec<Scalar> = (0..n_seq).map(|i| f(o + n_seq + i)).collect();
    let c_seq: Vec<Scalar> = (0..n_seq).map(|i| f(o + 2 * n_seq + i)).collect();
    // leave the host polynomials as the reference's loop leaves them: one entry, the final claim
    for i in 0..n_par { poly_A_vec_par[i].set_final(a_par[i]); poly_B_vec_par[i].set_final(b_par[i]); }
    if n_par > 0 { poly_C_par.set_final(c_par); }
    for i in 0..n_seq { poly_A_vec_seq[i].set_final(a_seq[i]); poly_B_vec_seq[i].set_final(b_seq[i]); poly_C_vec_seq[i].set_final(c_seq[i]); }
    (SumcheckInstanceProof::new(cubic_polys), r, (a_par, b_par, c_par), (a_seq, b_seq, c_seq))
}

/// SumcheckInstanceProof::prove_cubic (sumcheck.rs:89-161), comb_func = A*B*C: the batched prover with ONE "seq" instance and
/// coefficient one (its combined sums ARE the instance's sums)
pub fn prove_cubic(
    claim: &Scalar,
    num_rounds: usize,
    poly_A: &mut DensePolynomial,
    poly_B: &mut DensePolynomial,
    poly_C: &mut DensePolynomial,
    transcript: &mut Transcript,
) -> (SumcheckInstanceProof, Vec<Scalar>, Vec<Scalar>) {
    let mut none_a: Vec<&mut DensePolynomial> = Vec::new();
    let mut none_b: Vec<&mut DensePolynomial> = Vec::new();
    let mut dummy_c = DensePolynomial::new(vec![Scalar::zero()]);
    let (proof, r, _claims_prod, claims_dotp) = {
        let mut va = vec![&mut *poly_A];
        let mut vb = vec![&mut *poly_B];
        let mut vc = vec![&mut *poly_C];
        prove_cubic_batched(claim, num_rounds, (&mut none_a, &mut none_b, &mut dummy_c), (&mut va, &mut vb, &mut vc), &[Scalar::one()], transcript)
    };
    (proof, r, vec![claims_dotp.0[0], claims_dotp.1[0], claims_dotp.2[0]])
}

/// The table side of ZKSumcheckInstanceProof::prove_cubic_with_additive_term (sumcheck.rs:465-649): f = tau * (Az * Bz - Cz).
/// `evals()` is the loop of :502-530 for the current round; `bind(r_j)` is :551-554 and prepares the next round's sums in the same
/// pass.  Everything between (commitments, Σ-protocol, transcript: :531-548, :556-640) stays the reference's host code.
pub struct R1csRounds { t: [Table; 4], ev: [u8; 96] }
impl R1csRounds {
    pub fn new(tau: &DensePolynomial, az: &DensePolynomial, bz: &DensePolynomial, cz: &DensePolynomial) -> Self {
        let t = [Table::of(tau), Table::of(az), Table::of(bz), Table::of(cz)];
        let mut ev = [0u8; 96];
        check(unsafe { sbn_sc_eval_r1cs(ctx(), t[0].0, t[1].0, t[2].0, t[3].0, ev.as_mut_ptr()) });
        R1csRounds { t, ev }
    }
    /// (eval_point_0, eval_point_2, eval_point_3) of the current round
    pub fn evals(&self) -> (Scalar, Scalar, Scalar) { triple(&self.ev) }
    /// bound_poly_var_top(r_j) on all four tables; the next round's sums are ready afterwards
    pub fn bind(&mut self, r_j: &Scalar) {
        let rb = r_j.to_bytes();
        if self.t[0].len() >= 4 {
            check(unsafe { sbn_sc_bind_eval_r1cs(ctx(), self.t[0].0, self.t[1].0, self.t[2].0, self.t[3].0, rb.as_ptr(), self.ev.as_mut_ptr()) });
        } else {
            let ts: Vec<*mut sbn_table> = self.t.iter().map(|t| t.0).collect();
            check(unsafe { sbn_bind_top_many(ctx(), ts.as_ptr(), ts.len(), rb.as_ptr()) });
        }
    }
    /// [poly_tau[0], poly_Az[0], poly_Bz[0], poly_Cz[0]] after the last round (:645)
    pub fn finals(&self) -> Vec<Scalar> {
        let ts: Vec<*const sbn_table> = self.t.iter().map(|t| t.0 as *const sbn_table).collect();
        let mut out = [0u8; 128];
        check(unsafe { sbn_table_read0_many(ctx(), ts.as_ptr(), ts.len(), out.as_mut_ptr()) });
        (0..4).map(|i| sc(&out[32 * i..32 * i + 32])).collect()
    }
}
/// The table side of ZKSumcheckInstanceProof::prove_quad (sumcheck.rs:657-811): f = z * ABC, points 0 and 2 (:691-699), bind (:715-716)
pub struct QuadRounds { t: [Table; 2], ev: [u8; 64] }
impl QuadRounds {
    pub fn new(z: &DensePolynomial, abc: &DensePolynomial) -> Self {
        let t = [Table::of(z), Table::of(abc)];
        let mut ev = [0u8; 64];
        check(unsafe { sbn_sc_eval_quad(ctx(), t[0].0, t[1].0, ev.as_mut_ptr()) });
        QuadRounds { t, ev }
    }
    pub fn evals(&self) -> (Scalar, Scalar) { (sc(&self.ev[..32]), sc(&self.ev[32..64])) }
    pub fn bind(&mut self, r_j: &Scalar) {
        let rb = r_j.to_bytes();
        if self.t[0].len() >= 4 {
            check(unsafe { sbn_sc_bind_eval_quad(ctx(), self.t[0].0, self.t[1].0, rb.as_ptr(), self.ev.as_mut_ptr()) });
        } else {
            let ts: Vec<*mut sbn_table> = self.t.iter().map(|t| t.0).collect();
            check(unsafe { sbn_bind_top_many(ctx(), ts.as_ptr(), ts.len(), rb.as_ptr()) });
        }
    }
    pub fn finals(&self) -> Vec<Scalar> {
        let ts: Vec<*const sbn_table> = self.t.iter().map(|t| t.0 as *const sbn_table).collect();
        let mut out = [0u8; 64];
        check(unsafe { sbn_table_read0_many(ctx(), ts.as_ptr(), ts.len(), out.as_mut_ptr()) });
        vec![sc(&out[..32]), sc(&out[32..64])]
    }
}

// ---- BulletReductionProof::prove (nizk/bullet.rs:41-126) ---------------------------------------------------------------------
// The generators arrive as a slice (`&gens.gens_n.G`); their device handle is cached by (address, length, first and last point), i.e.
// per MultiCommitGens in practice — built on the first opening over a generator set, with h = H.
fn bases_for_slice(G_vec: &[GroupElement], H: &GroupElement) -> Arc<Bases> {
    static CACHE: OnceLock<Mutex<HashMap<(usize, usize, [u8; 64], [u8; 64]), Arc<Bases>>>> = OnceLock::new();
    let first = point_xy_mont(&affine_of(&G_vec[0]));
    let last = point_xy_mont(&affine_of(&G_vec[G_vec.len() - 1]));
    let key = (G_vec.as_ptr() as usize, G_vec.len(), first, last);
    let mut m = CACHE.get_or_init(|| Mutex::new(HashMap::new())).lock().unwrap();
    if let Some(b) = m.get(&key) { return b.clone(); }
    let proj: Vec<G1Projective> = G_vec.iter().map(|g| *g.inner()).collect();
    let aff = G1Projective::normalize_batch(&proj);
    let g: Vec<u8> = aff.iter().flat_map(point_xy_mont).collect();
    let h = point_xy_mont(&affine_of(H));
    let mut b = null_mut();
    check(unsafe { sbn_bases_upload(ctx(), g.as_ptr(), G_vec.len(), h.as_ptr(), SBN_POINTS_MONT, &mut b) });
    let b = Arc::new(Bases(b));
    m.insert(key, b.clone());
    b
}

/// BulletReductionProof::prove with Q = q_scale * Q_base: DotProductProofLog::prove hands the reduction Q = gens_1.scale(r).G[0]
/// (nizk/mod.rs:478-494), a new point per proof over the FIXED base gens_1.G[0]; given as (base, scalar) the device builds its
/// derived generator set once per circuit.  Returns (L_vec, R_vec, Gamma, a_hat, b_hat, g_hat, rhat_Gamma) — the fields and values of
/// bullet.rs:118-125.  The generators are never folded on the device (bullet.rs:87-91 costs n scalar multiplications per round):
/// round j's L, R are MSMs over the ORIGINAL generators; same group elements.
pub fn bullet_prove(
    transcript: &mut Transcript,
    Q_base: &GroupElement,
    q_scale: &Scalar,
    G_vec: &[GroupElement],
    H: &GroupElement,
    a_vec: &[Scalar],
    b_vec: &[Scalar],
    blind: &Scalar,
    blinds_vec: &[(Scalar, Scalar)],
) -> (Vec<GroupElement>, Vec<GroupElement>, GroupElement, Scalar, Scalar, GroupElement, Scalar) {
    let n = G_vec.len();
    assert_eq!(a_vec.len(), n);                                                 // bullet.rs:42
    assert_eq!(b_vec.len(), n);                                                 // :43
    assert!(n.is_power_of_two());                                               // :44
    let lg_n = n.trailing_zeros() as usize;
    assert_eq!(blinds_vec.len(), lg_n);                                         // :47

    let bases = bases_for_slice(G_vec, H);
    let (ta, tb) = (Table::upload(a_vec), Table::upload(b_vec));                // the reference clones a and b too (:50-52)
    let qb = point_xy_canonical(&affine_of(Q_base));
    let mut gamma = [0u8; 64];
    let mut ginf: c_int = 0;
    let mut st = null_mut();
    check(unsafe {
        sbn_bullet_begin_scaled(ctx(), bases.0, qb.as_ptr(), q_scale.to_bytes().as_ptr(), ta.0, tb.0, blind.to_bytes().as_ptr(), gamma.as_mut_ptr(), &mut ginf, &mut st)
    });                                                                         // Gamma (:58-60)
    let st = BulletState(st);
    let Gamma = group_from_xy(&gamma, ginf != 0);

    let mut blind_Gamma = *blind;
    let mut L_vec: Vec<GroupElement> = Vec::with_capacity(lg_n);
    let mut R_vec: Vec<GroupElement> = Vec::with_capacity(lg_n);
    let (mut lxy, mut rxy, mut cl, mut cr) = ([0u8; 64], [0u8; 64], [0u8; 32], [0u8; 32]);
    let (mut li, mut ri): (c_int, c_int) = (0, 0);
    if lg_n > 0 {
        let (bl, br) = blinds_vec[0];
        check(unsafe {
            sbn_bullet_cross(ctx(), st.0, bl.to_bytes().as_ptr(), br.to_bytes().as_ptr(), lxy.as_mut_ptr(), &mut li, rxy.as_mut_ptr(), &mut ri, cl.as_mut_ptr(), cr.as_mut_ptr())
        });                                                                     // c_L, c_R, L, R of round 0 (:72-78)
    }
    for i in 0..lg_n {
        let L = group_from_xy(&lxy, li != 0);
        let R = group_from_xy(&rxy, ri != 0);
        L.append_to_transcript(b"L", transcript);                               // :80
        R.append_to_transcript(b"R", transcript);                               // :81
        let u = transcript.challenge_scalar(b"u");                              // :83
        let u_inv = u.invert().unwrap();                                        // :84
        let (blind_L, blind_R) = blinds_vec[i];
        if i + 1 < lg_n {
            // the folds with u (:86-106) and the NEXT round's cross terms in one call
            let (bl, br) = blinds_vec[i + 1];
            check(unsafe {
                sbn_bullet_fold_cross(ctx(), st.0, u.to_bytes().as_ptr(), u_inv.to_bytes().as_ptr(), bl.to_bytes().as_ptr(), br.to_bytes().as_ptr(),
                                      lxy.as_mut_ptr(), &mut li, rxy.as_mut_ptr(), &mut ri, cl.as_mut_ptr(), cr.as_mut_ptr())
            });
        } else {
            check(unsafe { sbn_bullet_fold(ctx(), st.0, u.to_bytes().as_ptr(), u_inv.to_bytes().as_ptr()) });
        }
        blind_Gamma = u * u * blind_L + blind_Gamma + u_inv * u_inv * blind_R;  // :108
        L_vec.push(L);
        R_vec.push(R);
    }
    let (mut ah, mut bh, mut gh) = ([0u8; 32], [0u8; 32], [0u8; 64]);
    let mut gi: c_int = 0;
    check(unsafe { sbn_bullet_finish(ctx(), st.0, ah.as_mut_ptr(), bh.as_mut_ptr(), gh.as_mut_ptr(), &mut gi) });   // :114-120
    (L_vec, R_vec, Gamma, sc(&ah), sc(&bh), group_from_xy(&gh, gi != 0), blind_Gamma)
}
/// canonical x || y (no *_MONT flag on this argument of the bullet entry points)
fn point_xy_canonical(p: &G1Affine) -> [u8; 64] {
    let mut o = [0u8; 64];
    if !p.is_zero() {
        o[..32].copy_from_slice(&limbs_le(&p.x.into_bigint().0));
        o[32..].copy_from_slice(&limbs_le(&p.y.into_bigint().0));
    }
    o
}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::commitments::Commitments;

    /// the device MSM against arkworks on the reference's own generator derivation (two thirds of the points equal G)
    #[test]
    fn msm_matches_arkworks() {
        let n = 4096;
        let gens = MultiCommitGens::new(n, b"gens_r1cs_eval");
        let mut rng = rand::rngs::OsRng;
        let s: Vec<Scalar> = (0..n).map(|_| Scalar::random(&mut rng)).collect();
        let want = {
            let fr: Vec<ark_bn254::Fr> = s.iter().map(|x| x.0).collect();
            GroupElement(<G1Projective as ark_ec::VariableBaseMSM>::msm(&gens.G_affine, &fr).unwrap())
        };
        assert_eq!(msm_affine(&s, &gens.G_affine).unwrap(), want);
        let blind = Scalar::random(&mut rng);
        assert_eq!(commit_row(&s, &blind, &gens).unwrap(), want + blind * gens.h);
        assert_eq!(s.commit(&blind, &gens), want + blind * gens.h);
    }
}
