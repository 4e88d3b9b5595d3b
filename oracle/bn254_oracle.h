/*
 * oracle/bn254_oracle.h — CPU restatement of the Spartan-BN254 prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (spartan-bn254_amd/,
 * include/) may include, link or call this.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker.
 *
 * Parity status: the reference (Rust) delegates the MSM arithmetic to
 * ark-ec / ark-ff / ark-bn254 0.5 (Cargo.toml:9-11, not vendored, Cargo.lock
 * ignored) and ships NO golden vectors for this path.  Byte-level MSM /
 * commitment output is therefore "parity unpinned" by reference fixtures.  What
 * IS pinned (tests/test_oracle_*.py): the reference's own unit-test relations
 * (group.rs:313-321 2G+3G=5G; unipoly.rs:130-184 interpolation KATs;
 * hyrax.rs:427-470; sumcheck.rs:818-861; scalar.rs:206-215), the published
 * wire formats, and an independent pure-Python big-integer
 * model of the mathematical definition (tests/pyref.py -> tests/golden/).
 * Canonical affine coordinates of a group element are unique, so agreement with
 * the mathematical definition is agreement with arkworks' result.
 *
 * All byte interfaces: scalars = 32 B little-endian canonical integers < r
 * (scalar.rs:75-95); points = x||y, each 32 B little-endian canonical < p,
 * the all-zero 64 B string meaning the point at infinity.
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- field helpers on canonical LE bytes; which = 0 -> Fq, 1 -> Fr ---- */
int  orc_fe_is_canonical(int which, const uint8_t a[32]);
void orc_fe_add(int which, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void orc_fe_sub(int which, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void orc_fe_mul(int which, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void orc_fe_inv(int which, const uint8_t a[32], uint8_t out[32]);
/* 64 bytes LE reduced mod r (transcript.rs:56-67 from_le_bytes_mod_order) */
void orc_fr_from_wide(const uint8_t in[64], uint8_t out[32]);
/* sum_i k_i * s_i mod r  (the dlog oracle of SURVEY §0.7) */
void orc_fr_dot(const uint8_t* k, const uint8_t* s, size_t n, uint8_t out[32]);

/* ---- G1 on canonical affine bytes ---- */
int  orc_g1_on_curve(const uint8_t p[64]);
void orc_g1_add(const uint8_t p[64], const uint8_t q[64], uint8_t out[64]);
void orc_g1_neg(const uint8_t p[64], uint8_t out[64]);
void orc_g1_mul(const uint8_t p[64], const uint8_t k[32], uint8_t out[64]);
/* arkworks serialize_compressed (group.rs:135-140): x LE, bit7 of byte31 = y>p-y, bit6 = infinity */
void orc_g1_compress(const uint8_t p[64], uint8_t out[32]);
int  orc_g1_decompress(const uint8_t in[32], uint8_t out[64]);
/* n fixed-base multiples s_i*G of the generator (1,2); used for fixtures with known dlogs */
void orc_g1_mul_gen_batch(const uint8_t* s, size_t n, uint8_t* out_xy, int threads);

/* ---- MSM (group.rs:143-158, 171-175) ---- */
/* definition: double-and-add per term */
void orc_msm_naive(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out[64]);
/* arkworks-0.5-style signed-digit Pippenger, `threads` over windows like ark-ec/parallel */
void orc_msm_pippenger(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out[64], int threads);
int  orc_msm_window_bits(size_t n);

/* ---- Pedersen generators + commit (commitments.rs:31-62, 118-154; group.rs:110-131) ---- */
/* writes (n+1) points: G[0..n) then h; dlogs (optional, may be NULL) gets the (n+1) scalars s_i with P_i = s_i*G */
void orc_gens_new(size_t n, const uint8_t* label, size_t label_len, uint8_t* out_xy, uint8_t* out_dlogs, int threads);
/* commit = MSM(scalars || blind, G || h)  (commitments.rs:144-154) */
void orc_commit(const uint8_t* scalars, size_t n, const uint8_t blind[32], const uint8_t* G_xy, const uint8_t h_xy[64], uint8_t out[64]);
/* Hyrax row commit (hyrax.rs:253-267): C[i] = commit(Z[i*R..(i+1)*R], blinds[i]); blinds NULL = zeros; threads over rows like rayon */
void orc_commit_rows(const uint8_t* Z, const uint8_t* blinds, size_t L, size_t R, const uint8_t* G_xy, const uint8_t h_xy[64], uint8_t* out_xy, int threads);
/* hyrax.rs:371-373 */
void orc_factored_lens(size_t ell, size_t* left, size_t* right);

/* ---- multilinear / sumcheck (hyrax.rs:195-203, 355-369; sumcheck.rs) ---- */
void orc_eq_evals(const uint8_t* r, size_t ell, uint8_t* out /* 2^ell x 32 */);
void orc_bind_top(uint8_t* Z, size_t len, const uint8_t r[32]);  /* in place; first len/2 entries valid after */
void orc_sc_eval_cubic(const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, uint8_t out[96]);        /* e0,e2,e3 of A*B*C      (sumcheck.rs:111-135) */
void orc_sc_eval_r1cs(const uint8_t* T, const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, uint8_t out[96]); /* T*(A*B-C) (sumcheck.rs:502-530, r1csproof.rs:288-292) */
void orc_sc_eval_quad(const uint8_t* Z, const uint8_t* ABC, size_t len, uint8_t out[64]);                        /* e0,e2 of Z*ABC         (sumcheck.rs:691-699) */
/* prove_cubic_batched (sumcheck.rs:165-330), comb = A*B*C, challenges supplied by the caller; see the .c file for the layouts */
void orc_sc_prove_cubic_batched(const uint8_t* const* A_par, const uint8_t* const* B_par, const uint8_t* C_par, size_t n_par,
                                const uint8_t* const* A_seq, const uint8_t* const* B_seq, const uint8_t* const* C_seq, size_t n_seq,
                                size_t len, size_t num_rounds, const uint8_t* coeffs, const uint8_t* challenges,
                                uint8_t* evals, uint8_t* combined, uint8_t* finals, int threads);
/* unipoly.rs:28-59: evals at 0,1,2[,3] -> coefficients low..high; n = 3 or 4 */
void orc_unipoly_from_evals(const uint8_t* evals, size_t n, uint8_t* coeffs);
void orc_unipoly_eval(const uint8_t* coeffs, size_t n, const uint8_t r[32], uint8_t out[32]);
void orc_dotproduct(const uint8_t* a, const uint8_t* b, size_t n, uint8_t out[32]);
/* DensePolynomial::bound (hyrax.rs:311-324): out[i] = sum_j L[j] * Z[j*R_size + i], i < R_size */
void orc_bound(const uint8_t* Z, const uint8_t* L, size_t L_size, size_t R_size, uint8_t* out);

/* Layers::build_hash_layer (sparse_mlpoly_full.rs:745-796): out[j] = (ts[j] + ts_add) * g^2 + val[j] * g + addr[j] - tau.
 * addr / ts: n uint32 each; NULL addr = identity (cell index j), NULL ts = zeros */
void orc_hash_layer(const uint32_t* addr, const uint8_t* val, const uint32_t* ts, uint32_t ts_add, const uint8_t g[32], const uint8_t tau[32], size_t n, uint8_t* out);
/* ProductCircuit::compute_layer (product_tree.rs:21-37): out[i] = in[i] * in[i + len/2], i < len/2 */
void orc_product_layer(const uint8_t* in, size_t len, uint8_t* out);

/* BulletReductionProof::prove (nizk/bullet.rs:41-126) with the Fiat-Shamir challenges u_i supplied by the caller (the
 * reference draws them from the transcript, bullet.rs:80-83).  n a power of two; blinds_vec = lg n pairs (blind_L, blind_R);
 * outputs: L_vec, R_vec (lg n points each), Gamma, a_hat, b_hat, g_hat, rhat_Gamma. */
void orc_bullet_prove(const uint8_t* G_xy, const uint8_t Q_xy[64], const uint8_t H_xy[64], const uint8_t* a, const uint8_t* b, size_t n,
                      const uint8_t blind[32], const uint8_t* blinds_vec, const uint8_t* us,
                      uint8_t* L_vec, uint8_t* R_vec, uint8_t Gamma[64], uint8_t a_hat[32], uint8_t b_hat[32], uint8_t g_hat[64], uint8_t blind_hat[32]);

/* ---- Keccak (sha3 crate call sites: commitments.rs:33-44, group.rs:113-128) ---- */
void orc_sha3_256(const uint8_t* in, size_t len, uint8_t out[32]);
void orc_shake256(const uint8_t* in, size_t len, uint8_t* out, size_t outlen);

#ifdef __cplusplus
}
#endif
#endif
