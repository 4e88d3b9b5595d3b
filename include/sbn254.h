/*
 * sbn254.h — C ABI of libsbn254_hip.so: the MI355X (gfx950) implementation of Spartan-BN254's prover
 * hot path.  These are the entry points a Rust `extern "C"` block binds (INTEGRATION.md shows the shim);
 * every function cites the reference interface it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - return 0 on success, a negative SBN_E* code otherwise; sbn_last_error() has the text.
 *     The reference's prover functions are infallible (assert!/panic on misuse, errors.rs:19-31 is
 *     verifier-only), so the shim turns a non-zero code into panic!.
 *   - scalars: 32 B little-endian.  Default = canonical integer < r (Scalar::to_bytes, scalar.rs:75-84);
 *     with SBN_SCALARS_MONT the 4 x u64 Montgomery limbs ark-ff keeps in memory (R = 2^256) are taken as-is.
 *   - points: 64 B = x || y, each 32 B little-endian canonical integer < p; all-zero = point at infinity.
 *     With SBN_POINTS_MONT the coordinates are ark-ff Montgomery limbs instead (G1Affine's in-memory x, y).
 *   - `*_dev` variants take DEVICE pointers (hipMalloc'ed by the caller or by sbn_dev_alloc); the others take
 *     host pointers and stage through HBM themselves.
 *   - a context owns one HIP stream and a workspace; calls on one context are serialised by a mutex
 *     (hyrax.rs:259-261 may enter B1 from many rayon workers at once); use one context per thread for overlap.
 *   - there is NO CPU fallback: without a gfx950 device sbn_ctx_create fails.
 */
#ifndef SBN254_H
#define SBN254_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SBN_OK 0
#define SBN_EINVAL (-1)   /* bad argument (NULL, size mismatch, non-canonical input when checked) */
#define SBN_EHIP (-2)     /* HIP runtime error */
#define SBN_ENODEV (-3)   /* no usable gfx950 device */
#define SBN_ENOMEM (-4)

#define SBN_SCALARS_MONT 1u
#define SBN_POINTS_MONT 2u

typedef struct sbn_ctx sbn_ctx;
typedef struct sbn_bases sbn_bases;   /* device-resident generator table: MultiCommitGens.{G_affine,h_affine} (commitments.rs:17-27) */
typedef struct sbn_table sbn_table;   /* device-resident Fr table: DensePolynomial.Z (hyrax.rs:155-160) */

/* ---- context ---- */
int sbn_ctx_create(int device, sbn_ctx** out);
void sbn_ctx_destroy(sbn_ctx* ctx);
const char* sbn_last_error(const sbn_ctx* ctx);
/* run on the caller's HIP stream (hipStream_t as void*); NULL restores the context's own stream */
int sbn_ctx_set_stream(sbn_ctx* ctx, void* hip_stream);
int sbn_ctx_sync(sbn_ctx* ctx);
const char* sbn_version(void);

/* ---- raw device memory for callers without their own allocator (the Rust shim) ---- */
int sbn_dev_alloc(sbn_ctx* ctx, size_t bytes, void** out_dev);
int sbn_dev_free(sbn_ctx* ctx, void* dev);
int sbn_dev_upload(sbn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sbn_dev_download(sbn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- B1: single MSM — GroupElement::msm_affine(scalars, points) (group.rs:171-175) and
 *      vartime_multiscalar_mul (group.rs:143-158).  out_is_inf may be NULL.
 *      n == 0 returns the identity, as arkworks' msm of empty slices does. ---- */
int sbn_msm(sbn_ctx* ctx, const uint8_t* scalars, const uint8_t* points, size_t n, uint32_t flags,
            uint8_t out_xy[64], int* out_is_inf);

/* vartime_multiscalar_mul (group.rs:143-158) takes PROJECTIVE points and converts each with its own inversion on the CPU
 * (group.rs:153); here the n Jacobian triples X||Y||Z (96 B each; Z = 0 is the identity) are normalised on the device. */
int sbn_msm_jacobian(sbn_ctx* ctx, const uint8_t* scalars, const uint8_t* points_xyz, size_t n, uint32_t flags,
                     uint8_t out_xy[64], int* out_is_inf);

/* ---- generator tables — MultiCommitGens (commitments.rs:17-27).  G: n points, h: 1 point or NULL.
 *      Stored on the device in Montgomery form; duplicates are detected so that commit() can merge the
 *      scalars of equal bases (the reference's derivation makes ~66 % of them equal to G, group.rs:110-131). ---- */
int sbn_bases_upload(sbn_ctx* ctx, const uint8_t* G_xy, size_t n, const uint8_t* h_xy, uint32_t flags, sbn_bases** out);
void sbn_bases_free(sbn_ctx* ctx, sbn_bases* b);
size_t sbn_bases_len(const sbn_bases* b);          /* n (without h) */
/* Fixed-base precomputation for a generator set that will serve many commitments (the per-circuit gens_derefs / gens_ops of
 * SparseMatPolyCommitmentGens, sparse_mlpoly_full.rs:619-627): a table of every digit multiple d * 2^(c w) * G_j in HBM, so a row
 * commitment costs ceil(254/c) mixed additions per scalar and no bucket work.  The largest window c <= 16 whose table fits
 * max_bytes is built (the reference's 8193 gens_r1cs_eval generators: 2814 unique x 16 windows x 32768 x 64 B = 94 GB at c = 16);
 * commits on this handle use it from then on.  Results are unchanged (same group element).  *window_bits (optional) <- c. */
int sbn_bases_precompute(sbn_ctx* ctx, sbn_bases* b, size_t max_bytes, int* window_bits);
/* MultiCommitGens::new(n, label) (commitments.rs:31-62): SHAKE256 stream -> from_uniform_bytes (group.rs:110-131);
 * builds G[0..n) and h on the device and, if out_xy != NULL, also returns the n+1 canonical points. */
int sbn_gens_new(sbn_ctx* ctx, size_t n, const uint8_t* label, size_t label_len, uint8_t* out_xy, sbn_bases** out);

/* Synthetic benchmark bases with known discrete logs: P_i = (s0 + (first + i) * d) * G for
 * i in [0, n), all distinct, built on the device; the expected MSM result is then (sum k_i (s0 + (first+i) d)) * G. */
int sbn_bases_synthetic(sbn_ctx* ctx, size_t n, uint64_t first, const uint8_t s0[32], const uint8_t d[32], sbn_bases** out);
/* Synthetic benchmark scalars, written to a DEVICE buffer of n x 32 B: scalar t = the SplitMix64 outputs of
 * counters 4*(first+t)+1 .. +4 (state = seed + counter * 0x9E3779B97F4A7C15) as four little-endian u64 limbs, the top limb cut
 * to 62 bits, minus r when >= r: canonical, full-width values of Fr (the same stream bench.py's numpy generator produces). */
int sbn_scalars_synthetic(sbn_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* out_dev);
/* copy `count` points starting at `first` back to the host as canonical x||y (for tests of resident tables) */
int sbn_bases_download(sbn_ctx* ctx, const sbn_bases* b, size_t first, size_t count, uint8_t* out_xy);

/* MultiCommitGens::split_at(mid) (commitments.rs:78-98): (G[..mid], h) and (G[mid..], h) */
int sbn_bases_split_at(sbn_ctx* ctx, const sbn_bases* b, size_t mid, sbn_bases** left, sbn_bases** right);
/* MultiCommitGens::scale(s) (commitments.rs:64-76): G_i <- s * G_i, h unchanged */
int sbn_bases_scale(sbn_ctx* ctx, const sbn_bases* b, const uint8_t s[32], sbn_bases** out);

/* MSM of n scalars against the first n points of a resident table (no h): msm_affine with cached G_affine */
int sbn_msm_bases(sbn_ctx* ctx, const sbn_bases* b, const uint8_t* scalars, size_t n, uint32_t flags,
                  uint8_t out_xy[64], int* out_is_inf);
int sbn_msm_bases_dev(sbn_ctx* ctx, const sbn_bases* b, const void* scalars_dev, size_t n, uint32_t flags,
                      uint8_t out_xy[64], int* out_is_inf);

/* ---- B2: Pedersen / Hyrax commitments ----
 * <[Scalar] as Commitments>::commit(blind, gens_n) = MSM(scalars || blind, G || h) (commitments.rs:144-154):
 *   sbn_commit_rows with L = 1.
 * DensePolynomial::commit -> commit_inner (hyrax.rs:253-267, 283-308): C[i] = commit(Z[i*R..(i+1)*R], blinds[i]),
 *   Z row-major L x R, R == sbn_bases_len(b), blinds NULL = all zero (random_tape == None, hyrax.rs:301-305).
 * out_xy: L x 64 B canonical affine; out_inf: L flags or NULL. */
int sbn_commit_rows(sbn_ctx* ctx, const sbn_bases* b, const uint8_t* Z, const uint8_t* blinds, size_t L, size_t R,
                    uint32_t flags, uint8_t* out_xy, uint8_t* out_inf);
int sbn_commit_rows_dev(sbn_ctx* ctx, const sbn_bases* b, const void* Z_dev, const void* blinds_dev, size_t L, size_t R,
                        uint32_t flags, uint8_t* out_xy, uint8_t* out_inf);
/* arkworks serialize_compressed of n affine points (group.rs:135-140; what transcript.rs:102-108 absorbs) */
int sbn_g1_compress(const uint8_t* xy, size_t n, uint8_t* out32);
/* sum of n canonical affine points on the host (no device needed): the fold of per-GPU partial MSM results after
 * the RCCL all-gather, i.e. the `+` of GroupElement (group.rs:199-262) applied n-1 times */
int sbn_g1_sum(const uint8_t* xy, size_t n, uint8_t out_xy[64], int* out_is_inf);
/* UniPoly::from_evals (unipoly.rs:28-59): the round polynomial from its values at 0, 1, 2[, 3] (n = 3 or 4 canonical scalars,
 * in the order [e0, claim - e0, e2, e3] of sumcheck.rs:137); coeffs low..high.  Host only.  UniPoly::evaluate (unipoly.rs:74-82). */
int sbn_unipoly_from_evals(const uint8_t* evals, size_t n, uint8_t* coeffs);
int sbn_unipoly_eval(const uint8_t* coeffs, size_t n, const uint8_t r[32], uint8_t out[32]);
/* EqPolynomial::compute_factored_lens (hyrax.rs:371-373) */
void sbn_factored_lens(size_t ell, size_t* left, size_t* right);

/* ---- B3: sumcheck rounds on device-resident tables ----
 * A table is a DensePolynomial's Z vector (len a power of two).  Each eval returns the per-round values the
 * host needs for UniPoly::from_evals (unipoly.rs:28-59) as canonical 32 B scalars. */
int sbn_table_upload(sbn_ctx* ctx, const uint8_t* Z, size_t len, uint32_t flags, sbn_table** out);
int sbn_table_from_dev(sbn_ctx* ctx, const void* Z_dev, size_t len, uint32_t flags, sbn_table** out);
void sbn_table_free(sbn_ctx* ctx, sbn_table* t);
size_t sbn_table_len(const sbn_table* t);                                   /* current len (halves per bind) */
int sbn_table_download(sbn_ctx* ctx, const sbn_table* t, uint8_t* out /* len x 32 canonical */);
int sbn_table_read0(sbn_ctx* ctx, const sbn_table* t, uint8_t out[32]);     /* poly[0] after the last round (sumcheck.rs:157) */
/* the same for `count` tables in one launch and one wait (out: count x 32 bytes): the products of several product circuits
 * (ProductCircuit::evaluate, product_tree.rs:59-64, called per circuit at sparse_mlpoly_full.rs:1326-1345) after all of them were enqueued */
int sbn_table_read0_many(sbn_ctx* ctx, const sbn_table* const* ts, size_t count, uint8_t* out);
/* DensePolynomial::bound_poly_var_top(r) (hyrax.rs:195-203) */
int sbn_bind_top(sbn_ctx* ctx, sbn_table* t, const uint8_t r[32]);
int sbn_bind_top_many(sbn_ctx* ctx, sbn_table* const* ts, size_t count, const uint8_t r[32]);
/* prove_cubic inner loop, comb = A*B*C (sumcheck.rs:111-135; product_tree.rs:178-181): out = e0,e2,e3 */
int sbn_sc_eval_cubic(sbn_ctx* ctx, const sbn_table* A, const sbn_table* B, const sbn_table* C, uint8_t out[96]);
/* prove_cubic_batched inner loops (sumcheck.rs:201-267): instance i uses (A[i], B[i], C[i]); pass the shared
 * poly_C_par for the "par" instances.  out = count x (e0,e2,e3) */
int sbn_sc_eval_cubic_batched(sbn_ctx* ctx, const sbn_table* const* A, const sbn_table* const* B,
                              const sbn_table* const* C, size_t count, uint8_t* out /* count x 96 */);
/* prove_cubic_with_additive_term inner loop, comb = tau*(Az*Bz - Cz) (sumcheck.rs:502-530; r1csproof.rs:288-292) */
int sbn_sc_eval_r1cs(sbn_ctx* ctx, const sbn_table* tau, const sbn_table* Az, const sbn_table* Bz, const sbn_table* Cz,
                     uint8_t out[96]);
/* prove_quad inner loop, comb = z*ABC (sumcheck.rs:691-699; r1csproof.rs:389-390): out = e0,e2 */
int sbn_sc_eval_quad(sbn_ctx* ctx, const sbn_table* Z, const sbn_table* ABC, uint8_t out[64]);
/* Fused round: bind every distinct table among the arguments to r (bound_poly_var_top, as sumcheck.rs:148-150 / 289-299 /
 * 551-554 / 715-716 do after each challenge) AND return the NEXT round's sums on the bound tables — the same values the
 * separate bind + eval calls give, in one pass over the data.  Needs len >= 4.  Argument order as the eval calls. */
int sbn_sc_bind_eval_cubic_batched(sbn_ctx* ctx, sbn_table* const* A, sbn_table* const* B, sbn_table* const* C, size_t count,
                                   const uint8_t r[32], uint8_t* out /* count x 96 */);
int sbn_sc_bind_eval_r1cs(sbn_ctx* ctx, sbn_table* tau, sbn_table* Az, sbn_table* Bz, sbn_table* Cz, const uint8_t r[32], uint8_t out[96]);
int sbn_sc_bind_eval_quad(sbn_ctx* ctx, sbn_table* Z, sbn_table* ABC, const uint8_t r[32], uint8_t out[64]);
/* ---- SumcheckInstanceProof::prove_cubic_batched (sumcheck.rs:165-330; comb_func = A*B*C, its only call sites product_tree.rs:275-278)
 *      as a device-resident state; the transcript stays with the caller:
 *        begin -> evals of round 0;  per round: (UniPoly::from_evals, absorb, squeeze r_j) -> round(r_j) -> evals of round j+1;  finish.
 * What crosses the boundary per round is what the transcript absorbs: the coeffs-weighted combination of the instances' sums
 * (sumcheck.rs:269-271), i.e. (e0, e2, e3) = sum_i coeffs[i] * (e0, e2, e3)_i as three canonical scalars — not the per-instance
 * triples.  The "par" instances share C_par (:201-235), the "seq" instances have their own C (:238-267); coeffs: n_par + n_seq
 * canonical scalars, "par" first.  n_par + n_seq <= 24.  The caller's tables are only read: the state binds into buffers of its own.
 * Same field elements as the per-instance calls + a host-side combination, bit for bit. */
typedef struct sbn_sumcheck sbn_sumcheck;
int sbn_sumcheck_begin(sbn_ctx* ctx, const sbn_table* const* A_par, const sbn_table* const* B_par, const sbn_table* C_par, size_t n_par,
                       const sbn_table* const* A_seq, const sbn_table* const* B_seq, const sbn_table* const* C_seq, size_t n_seq,
                       const uint8_t* coeffs, uint8_t out_evals[96], sbn_sumcheck** out);
/* The same with poly_C_par = EqPolynomial::new(rand).evals() built inside the call (product_tree.rs:267-275: how every layer of
 * ProductCircuitEvalProofBatched::prove makes its C): rand = ell canonical scalars, 2^ell = the tables' length; the eq table belongs to the
 * state (its final claim is listed where C_par's is).  One call per layer instead of sbn_eq_evals + sbn_sumcheck_begin + sbn_table_free. */
int sbn_sumcheck_begin_eq(sbn_ctx* ctx, const sbn_table* const* A_par, const sbn_table* const* B_par, size_t n_par, const uint8_t* rand, size_t ell,
                          const sbn_table* const* A_seq, const sbn_table* const* B_seq, const sbn_table* const* C_seq, size_t n_seq,
                          const uint8_t* coeffs, uint8_t out_evals[96], sbn_sumcheck** out);
/* bind every table to r_j (sumcheck.rs:289-299); out_evals = the combined sums of the next round (zeros after the last bind) */
int sbn_sumcheck_round(sbn_ctx* ctx, sbn_sumcheck* st, const uint8_t r[32], uint8_t out_evals[96]);
size_t sbn_sumcheck_len(const sbn_sumcheck* st);                       /* current table length (halves per round) */
/* after the last round (length 1): poly[0] of every table (sumcheck.rs:302-318), 32 B each, in the order
 * A_par[0..n_par), B_par[0..n_par), C_par, A_seq[..], B_seq[..], C_seq[..] */
int sbn_sumcheck_finish(sbn_ctx* ctx, sbn_sumcheck* st, uint8_t* finals);
void sbn_sumcheck_free(sbn_ctx* ctx, sbn_sumcheck* st);
/* EqPolynomial::evals (hyrax.rs:355-369) built on the device */
int sbn_eq_evals(sbn_ctx* ctx, const uint8_t* r, size_t ell, sbn_table** out);

/* ---- Hyrax opening pieces ----
 * compute_dotproduct (hyrax.rs:409-415) of two equally long tables */
int sbn_table_dot(sbn_ctx* ctx, const sbn_table* a, const sbn_table* b, uint8_t out[32]);
/* DensePolynomial::evaluate(r) (hyrax.rs:217-222) = <Z, eq(r)>; the eq table is built on the device (r: ell scalars, 2^ell == len) */
int sbn_table_evaluate(sbn_ctx* ctx, const sbn_table* Z, const uint8_t* r, size_t ell, uint8_t out[32]);
/* the same for `count` tables of equal length at ONE point: HashLayerProof::prove evaluates 6 derefs + 15 addr/val/ts polynomials at
 * rand_ops and 2 at rand_mem (sparse_mlpoly_full.rs:907-976), each call rebuilding the eq table in the reference; here it is built once.
 * out = count x 32 B */
int sbn_table_evaluate_many(sbn_ctx* ctx, const sbn_table* const* Z, size_t count, const uint8_t* r, size_t ell, uint8_t* out);
/* DensePolynomial::bound(L) (hyrax.rs:311-324), the L*Z of PolyEvalProof::prove (hyrax.rs:101): Z viewed as L_size x R_size,
 * out[i] = sum_j Lvec[j] * Z[j*R_size + i]  (a new table of R_size entries) */
int sbn_table_bound(sbn_ctx* ctx, const sbn_table* Z, const sbn_table* Lvec, sbn_table** out);

/* ---- BulletReductionProof::prove (nizk/bullet.rs:41-126) as a device-resident state; the transcript stays with the caller:
 *      begin, cross (the first round's L, R), then per challenge u: fold_cross(u) -> the next L, R (fold(u) for the last one), then finish.
 * The generators are never folded (bullet.rs:87-91 costs n 254-bit scalar multiplications per round): round j's L and R are
 * MSMs over the ORIGINAL generators with scalars a[..] * s_t, s_t the running product of the u / u_inv the fold would have
 * applied to G_t (the verifier's compute_s, bullet.rs:181-199); g_hat = MSM(s, G).  Same group elements, bit for bit. ----
 * begin: G = the n generators (+ h = H if the handle has one; it must outlive the state), Q_xy = canonical affine Q or NULL
 *   (term left out), a / b = the two vectors (copied, as bullet.rs:50-52 clones them).  If Gamma_xy != NULL it receives
 *   Gamma = MSM(a, G) + <a, b>*Q + blind*H (bullet.rs:58-60; blind NULL = 0). */
typedef struct sbn_bullet sbn_bullet;
int sbn_bullet_begin(sbn_ctx* ctx, const sbn_bases* G, const uint8_t* Q_xy, const sbn_table* a, const sbn_table* b, const uint8_t* blind,
                     uint8_t* Gamma_xy, int* Gamma_is_inf, sbn_bullet** out);
/* The same with Q = q_scale * Q_base: DotProductProofLog::prove (nizk/mod.rs:478-494) passes Q = gens_1.scale(r).G[0] with r fresh from the
 * transcript — a new point per proof over a FIXED base.  Given as (Q_base, q_scale) the derived generator set G || Q_base and its lookup
 * table are built once per generator set, not once per proof; Q_base's column carries c * q_scale.  Same Gamma, L, R, bit for bit. */
int sbn_bullet_begin_scaled(sbn_ctx* ctx, const sbn_bases* G, const uint8_t* Q_base_xy, const uint8_t q_scale[32], const sbn_table* a, const sbn_table* b,
                            const uint8_t* blind, uint8_t* Gamma_xy, int* Gamma_is_inf, sbn_bullet** out);
void sbn_bullet_free(sbn_ctx* ctx, sbn_bullet* st);
size_t sbn_bullet_len(const sbn_bullet* st);       /* current n (halves per fold) */
/* One round's cross terms (bullet.rs:72-78), h = n/2:
 *   c_L = <a[..h], b[h..]>,  c_R = <a[h..], b[..h]>,
 *   L = MSM(a[..h], G[h..]) + c_L*Q + blind_L*H,   R = MSM(a[h..], G[..h]) + c_R*Q + blind_R*H   (blinds: canonical or NULL = 0) */
int sbn_bullet_cross(sbn_ctx* ctx, sbn_bullet* st, const uint8_t* blind_L, const uint8_t* blind_R, uint8_t L_xy[64], int* L_is_inf,
                     uint8_t R_xy[64], int* R_is_inf, uint8_t c_L[32], uint8_t c_R[32]);
/* The folds after the challenge u (bullet.rs:86-106): G <- u_inv*G_L + u*G_R (kept as coefficients), a <- u*a_L + u_inv*a_R,
 * b <- u_inv*b_L + u*b_R; the length halves. */
int sbn_bullet_fold(sbn_ctx* ctx, sbn_bullet* st, const uint8_t u[32], const uint8_t u_inv[32]);
/* The loop body of bullet.rs:63-108 as ONE call per challenge: fold with u (as sbn_bullet_fold), then the cross terms of the NEXT round
 * (as sbn_bullet_cross) on the folded vectors — one launch ahead of the commit, one host wait.  Needs length >= 4 (after the fold at
 * least one more round follows); the last challenge goes to sbn_bullet_fold.  Same L, R, c_L, c_R as the two separate calls. */
int sbn_bullet_fold_cross(sbn_ctx* ctx, sbn_bullet* st, const uint8_t u[32], const uint8_t u_inv[32], const uint8_t* blind_L, const uint8_t* blind_R,
                          uint8_t L_xy[64], int* L_is_inf, uint8_t R_xy[64], int* R_is_inf, uint8_t c_L[32], uint8_t c_R[32]);
/* After the last fold (length 1): a_hat, b_hat, g_hat (bullet.rs:114-120) */
int sbn_bullet_finish(sbn_ctx* ctx, sbn_bullet* st, uint8_t a_hat[32], uint8_t b_hat[32], uint8_t g_hat_xy[64], int* g_hat_is_inf);

/* ---- network construction pieces ----
 * sbn_hash_layer and sbn_product_layer only enqueue work (their outputs are consumed by later calls on the same context, which
 * are ordered behind them); every call that returns data to the host waits for it. ----
 * Layers::build_hash_layer (sparse_mlpoly_full.rs:745-796): out[j] = (ts[j] + ts_add) * r_hash^2 + val[j] * r_hash + addr[j] - r_multiset
 * addr_dev / ts_dev: DEVICE arrays of n uint32 (NULL addr = the cell index j, as for poly_init/audit_hashed; NULL ts = zeros);
 * ts_add = 1 gives the write set (read_ts + 1).  val: table of n entries (eval_table or a derefs poly). */
int sbn_hash_layer(sbn_ctx* ctx, const void* addr_dev, const sbn_table* val, const void* ts_dev, uint32_t ts_add,
                   const uint8_t r_hash[32], const uint8_t r_multiset[32], sbn_table** out);
/* Two hashed sets over the same (addr, val) in one pass: the read and write sets of one sparse polynomial (ts_a = ts_b = read_ts, adds 0 and 1) or
 * the init and audit sets of a memory (addr NULL, ts_a NULL, ts_b = audit_ts) — sparse_mlpoly_full.rs:762-790 builds each pair from the same inputs.
 * out_a / out_b = what two sbn_hash_layer calls with (ts_a, ts_a_add) / (ts_b, ts_b_add) return. */
int sbn_hash_layer_pair(sbn_ctx* ctx, const void* addr_dev, const sbn_table* val, const void* ts_a_dev, uint32_t ts_a_add, const void* ts_b_dev, uint32_t ts_b_add,
                        const uint8_t r_hash[32], const uint8_t r_multiset[32], sbn_table** out_a, sbn_table** out_b);
/* ProductCircuit::compute_layer (product_tree.rs:21-37): the next layer's full vector out[i] = in[i] * in[i + len/2] */
int sbn_product_layer(sbn_ctx* ctx, const sbn_table* in, sbn_table** out);
/* ProductCircuit::new (product_tree.rs:39-57): every layer above `in` in one call — layers[0] = compute_layer(in) (len/2 entries),
 * layers[k] = compute_layer(layers[k-1]), down to the single-entry layer whose value is the circuit's product (evaluate(), :59-66).
 * cap = size of the caller's array (log2(len) layers are produced), *count <- number written.  Enqueued only, like sbn_product_layer. */
int sbn_product_circuit(sbn_ctx* ctx, const sbn_table* in, sbn_table** layers, size_t cap, size_t* count);
/* n product circuits over tables of ONE length, built together: layers[i * cap + k] = layer k of circuit i, *count <- layers per circuit.
 * The hashed sets of a sparse-polynomial evaluation proof all get their circuit at the same point (sparse_mlpoly_full.rs:813-823: row / col x
 * read / write over the operations, init / audit over the memories); one launch per layer serves all of them — below ~2^15 entries a layer's
 * launch costs more than its arithmetic.  Same values as n sbn_product_circuit calls. */
int sbn_product_circuit_many(sbn_ctx* ctx, const sbn_table* const* ins, size_t n, sbn_table** layers, size_t cap, size_t* count);
/* DensePolynomial::split(len/2) (hyrax.rs:186-192) as views: left = first half, right = second half of `t` (the A and B tables of
 * a product-circuit layer).  Views share t's memory and must be freed before t. */
int sbn_table_halves(sbn_ctx* ctx, const sbn_table* t, sbn_table** left, sbn_table** right);

/* entries [first, first + len) of t as a view (len a power of two; freed before t): the polynomials DensePolynomial::merge laid end to end
 * (hyrax.rs:237-247) are slices of the merged table — e.g. Derefs' row_ops_val / col_ops_val inside `comb` (sparse_mlpoly_full.rs:286-297),
 * which Layers::new hashes again (sparse_mlpoly_full.rs:745-796) */
int sbn_table_slice(sbn_ctx* ctx, const sbn_table* t, size_t first, size_t len, sbn_table** out);

/* ---- derefs on the device ----
 * MultiSparseMatPolynomialAsDense::deref -> AddrTimestamps::deref_mem (sparse_mlpoly_full.rs:245-257, 275-279) followed by
 * Derefs::new -> DensePolynomial::merge (sparse_mlpoly_full.rs:293-297, hyrax.rs:237-247):
 *   out = concat_k [ mem[k][addr[k][i]] for i < n ], zero-padded to the next power of two.
 * mem[k]: the table instance k reads (the eq(rx) table for the row_ops_val polys, eq(ry) for col_ops_val: build them with
 * sbn_eq_evals).  addr[k]: DEVICE arrays of n uint32 cell indices (fixed per circuit: upload once with sbn_dev_upload).
 * The result is a table (the `comb` polynomial), ready for sbn_commit_table — the 1 GiB scalar matrix of the keyless
 * derefs commitment never crosses PCIe. */
int sbn_gather_merge(sbn_ctx* ctx, const sbn_table* const* mem, const void* const* addr_dev, size_t count, size_t n, sbn_table** out);
/* One device's share of the same polynomial when its L x R view (R a power of two, hyrax.rs:371-373) is committed by interleaved rows over
 * several devices: only the rows row0, row0 + rstep, ... (nrows of them) are gathered, as one table of nrows x R entries for
 * sbn_commit_table(…, L = nrows, R).  row0 = 0, rstep = 1, nrows = L gives sbn_gather_merge's table. */
int sbn_gather_merge_rows(sbn_ctx* ctx, const sbn_table* const* mem, const void* const* addr_dev, size_t count, size_t n, size_t R,
                          size_t row0, size_t rstep, size_t nrows, sbn_table** out);
/* DensePolynomial::commit (hyrax.rs:283-308) of a device-resident table viewed as L x R (L*R == len, R == gens n); blinds as
 * in sbn_commit_rows (host pointer or NULL) */
int sbn_commit_table(sbn_ctx* ctx, const sbn_bases* b, const sbn_table* t, const uint8_t* blinds, size_t L, size_t R,
                     uint8_t* out_xy, uint8_t* out_inf);

/* ---- device groups: ONE host process, several GPUs behind one call ----
 * The reference parallelises inside one process (rayon over the rows of the Hyrax matrix, hyrax.rs:259-261; arkworks over the windows of
 * an MSM), so its drop-in has one process too: a group owns one context per listed device (a device may be listed more than once) and
 * runs every call with one (persistent) host thread per device.  Nothing here needs a launcher or a collective library.
 *   rows of ONE matrix: row i on device i mod N, no exchange (rows are independent);
 *   ONE MSM: contiguous base-point ranges, the N 64-byte partial sums folded on the host with sbn_g1_sum (group.rs:199-262). */
typedef struct sbn_group sbn_group;
typedef struct sbn_group_bases sbn_group_bases;
int sbn_group_create(const int* devices, size_t n, sbn_group** out);
void sbn_group_destroy(sbn_group* g);
size_t sbn_group_size(const sbn_group* g);
sbn_ctx* sbn_group_ctx(sbn_group* g, size_t i);                 /* the i-th device's context (for the single-device calls above) */
const char* sbn_group_last_error(const sbn_group* g);
/* MultiCommitGens (commitments.rs:17-27) replicated on every device of the group */
int sbn_group_bases_upload(sbn_group* g, const uint8_t* G_xy, size_t n, const uint8_t* h_xy, uint32_t flags, sbn_group_bases** out);
int sbn_group_gens_new(sbn_group* g, size_t n, const uint8_t* label, size_t label_len, uint8_t* out_xy, sbn_group_bases** out);
int sbn_group_bases_precompute(sbn_group* g, sbn_group_bases* gb, size_t max_bytes_per_device, int* window_bits);
void sbn_group_bases_free(sbn_group* g, sbn_group_bases* gb);
/* DensePolynomial::commit -> commit_inner (hyrax.rs:253-267) of ONE L x R matrix (host pointer) over the group; arguments as sbn_commit_rows */
int sbn_group_commit_rows(sbn_group* g, const sbn_group_bases* gb, const uint8_t* Z, const uint8_t* blinds, size_t L, size_t R,
                          uint32_t flags, uint8_t* out_xy, uint8_t* out_inf);
/* The same with the matrix already ON the devices: Z_dev[d] = device pointer on device d to ITS rows (d, d + N, d + 2N, ... in this order,
 * contiguous, R x 32 B each); blinds_dev[d] likewise (32 B per row) or blinds_dev == NULL.  Only the L x 64 B of results cross PCIe. */
int sbn_group_commit_rows_dev(sbn_group* g, const sbn_group_bases* gb, const void* const* Z_dev, const void* const* blinds_dev, size_t L, size_t R,
                              uint32_t flags, uint8_t* out_xy, uint8_t* out_inf);
/* Derefs::commit (sparse_mlpoly_full.rs:301-304) over the group from device-resident inputs: deref_mem + merge (sparse_mlpoly_full.rs:245-257,
 * 293-297) and commit_inner (hyrax.rs:253-267) in one call.  mem[d * count + k] / addr_dev[d * count + k]: table k (the eq(rx) / eq(ry) tables,
 * built per device with sbn_eq_evals on sbn_group_ctx(g, d)) and its n uint32 cell addresses ON device d.  Device d gathers and commits only
 * the rows d, d + N, ... of the L x R view of the merged polynomial (L * R = next power of two of count * n); no blinds. */
int sbn_group_gather_commit(sbn_group* g, const sbn_group_bases* gb, const sbn_table* const* mem, const void* const* addr_dev, size_t count, size_t n,
                            size_t L, size_t R, uint8_t* out_xy, uint8_t* out_inf);
/* GroupElement::msm_affine (group.rs:171-175) of ONE MSM over the group: device d takes the pairs [d n / N, (d+1) n / N) */
int sbn_group_msm(sbn_group* g, const uint8_t* scalars, const uint8_t* points, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf);
/* the same with resident points: cut once into the per-device ranges (no h), then only scalars travel */
int sbn_group_bases_upload_ranges(sbn_group* g, const uint8_t* G_xy, size_t n, uint32_t flags, sbn_group_bases** out);
int sbn_group_bases_synthetic_ranges(sbn_group* g, size_t n, const uint8_t s0[32], const uint8_t d[32], sbn_group_bases** out);   /* sbn_bases_synthetic, range by range */
void sbn_group_range(const sbn_group_bases* gb, size_t device, size_t* lo, size_t* hi);
int sbn_group_msm_bases(sbn_group* g, const sbn_group_bases* gb, const uint8_t* scalars, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf);
/* scalars_dev[d]: device pointer ON device d to the (hi - lo) x 32 B of its range */
int sbn_group_msm_bases_dev(sbn_group* g, const sbn_group_bases* gb, const void* const* scalars_dev, uint32_t flags, uint8_t out_xy[64], int* out_is_inf);

/* ---- per-kernel timing (HIP events on the context's stream), for bench.py's roofline line ---- */
int sbn_prof_enable(sbn_ctx* ctx, int on);
int sbn_prof_reset(sbn_ctx* ctx);
/* number of distinct kernel names seen; i-th name, summed ms and launch count */
int sbn_prof_count(sbn_ctx* ctx);
int sbn_prof_get(sbn_ctx* ctx, int i, const char** name, double* total_ms, uint64_t* launches);
/* shape of the context's most recent bucket job (an MSM or a row commit): out = {window bits c, windows W, (digit, point)
 * slots = mixed additions when no digit is zero, buckets}; bench.py prices the accumulate kernel against the ALU roofline with it */
int sbn_prof_last_job(sbn_ctx* ctx, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
