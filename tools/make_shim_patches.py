#!/usr/bin/env python3
"""shim/patches/*.diff — the edits to the reference crate that put its prover hot path behind libsbn254_hip.so (feature `hip`).

Run in the build container, where the reference lies at /root/reference (read as text only):
    python tools/make_shim_patches.py            regenerate shim/patches/
    python tools/make_shim_patches.py --check    fail if the committed patches no longer apply to the reference (tests/test_shim_consistency.py)
Every edit is an (anchor, replacement) pair applied to the reference's text; the output is a plain unified diff per file
(`patch -p1 < shim/patches/NNNN-*.diff` from the crate root; then copy shim/src/hip.rs to src/hip.rs and shim/build.rs to build.rs).
The anchors are the reference's own lines, quoted as short as uniqueness allows; the inserted code is this repository's."""
import difflib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SBN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "shim", "patches")


def sub_once(text, anchor, repl, what):
    n = text.count(anchor)
    if n != 1:
        raise SystemExit(f"{what}: anchor occurs {n} times, expected exactly once:\n{anchor}")
    return text.replace(anchor, repl)


def edit_cargo(t):
    return sub_once(t, '# Use KZG instead of Hyrax for polynomial commitments (requires trusted setup)\nkzg = []\n',
                    '# Use KZG instead of Hyrax for polynomial commitments (requires trusted setup)\nkzg = []\n'
                    '# MI355X (gfx950) offload of the prover hot path through libsbn254_hip.so: src/hip.rs + build.rs (SBN254_LIB_DIR)\nhip = []\n', "Cargo.toml")


def edit_lib(t):
    return sub_once(t, "pub mod group;\n", 'pub mod group;\n#[cfg(feature = "hip")]\npub mod hip;\n', "lib.rs")


def edit_scalar(t):
    return sub_once(t, "#[derive(Clone, Copy, Debug, PartialEq, Eq, Default)]\npub struct Scalar(pub Fr);",
                    "#[derive(Clone, Copy, Debug, PartialEq, Eq, Default)]\n#[repr(transparent)] // one field, Fr: slices of Scalar are handed to the device as they lie in memory (src/hip.rs checks Fr's own layout at run time)\npub struct Scalar(pub Fr);", "scalar.rs")


def edit_group(t):
    t = sub_once(t, """        let scalars_vec: Vec<Fr> = scalars.into_iter().map(|s| s.borrow().0).collect();
        let points_vec: Vec<G1Affine> = points
""", """        #[cfg(feature = "hip")]
        {
            // the n projective points are normalised on the device (the loop below inverts once per point on the CPU)
            let s: Vec<Scalar> = scalars.into_iter().map(|s| *s.borrow()).collect();
            let p: Vec<G1Projective> = points.into_iter().map(|p| p.borrow().0).collect();
            if let Some(g) = crate::hip::msm_projective(&s, &p) {
                return g;
            }
            let fr: Vec<Fr> = s.iter().map(|x| x.0).collect();
            let aff = G1Projective::normalize_batch(&p);
            return GroupElement(G1Projective::msm(&aff, &fr).unwrap_or_default());
        }
        #[cfg(not(feature = "hip"))]
        {
        let scalars_vec: Vec<Fr> = scalars.into_iter().map(|s| s.borrow().0).collect();
        let points_vec: Vec<G1Affine> = points
""", "group.rs vartime_multiscalar_mul")
    t = sub_once(t, """        let result = G1Projective::msm(&points_vec, &scalars_vec).unwrap_or_default();
        GroupElement(result)
    }

    /// Get the underlying projective point""", """        let result = G1Projective::msm(&points_vec, &scalars_vec).unwrap_or_default();
        GroupElement(result)
        }
    }

    /// Get the underlying projective point""", "group.rs vartime_multiscalar_mul end")
    t = sub_once(t, """    pub fn msm_affine(scalars: &[Scalar], points: &[G1Affine]) -> Self {
""", """    pub fn msm_affine(scalars: &[Scalar], points: &[G1Affine]) -> Self {
        // MSMs of MIN_GPU_MSM terms and more run on the GPU; a length mismatch falls through to arkworks' `unwrap_or_default`
        #[cfg(feature = "hip")]
        if let Some(g) = crate::hip::msm_affine(scalars, points) {
            return g;
        }
""", "group.rs msm_affine")
    return t


def edit_commitments(t):
    t = sub_once(t, """    #[serde(skip)]
    pub h_affine: G1Affine,
}
""", """    #[serde(skip)]
    pub h_affine: G1Affine,
    /// device twin of the affine cache: the generator table on the GPU, created on first use, shared by clones
    #[cfg(feature = "hip")]
    #[serde(skip)]
    pub dev: crate::hip::GensDev,
}
""", "commitments.rs struct")
    # every struct literal of MultiCommitGens gets the new field
    lines = t.split("\n")
    out = []
    i = 0
    n_lit = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(\s*)(\(\s*)?MultiCommitGens \{$", ln)
        if m and "struct" not in ln and "impl" not in ln:
            indent = m.group(1)
            j = i + 1
            while not re.match(r"^" + indent + r"\}[,)]?$", lines[j]):
                j += 1
            out.extend(lines[i:j])
            out.append(indent + '    #[cfg(feature = "hip")]')
            out.append(indent + "    dev: Default::default(),")
            n_lit += 1
            i = j
            continue
        out.append(ln)
        i += 1
    if n_lit != 5:
        raise SystemExit(f"commitments.rs: {n_lit} MultiCommitGens literals found, expected 5")
    t = "\n".join(out)
    for ty in ("Vec<Scalar>", "[Scalar]"):
        t = sub_once(t, f"""impl Commitments for {ty} {{
    fn commit(&self, blind: &Scalar, gens_n: &MultiCommitGens) -> GroupElement {{
        assert_eq!(gens_n.n, self.len());
""", f"""impl Commitments for {ty} {{
    fn commit(&self, blind: &Scalar, gens_n: &MultiCommitGens) -> GroupElement {{
        assert_eq!(gens_n.n, self.len());
        // rows of MIN_GPU_MSM scalars and more: MSM(self || blind, G || h) over the resident generator table, no per-call copies
        #[cfg(feature = "hip")]
        if let Some(c) = crate::hip::commit_row(self, blind, gens_n) {{
            return c;
        }}
""", f"commitments.rs commit for {ty}")
    return t


def edit_hyrax(t):
    t = sub_once(t, """    #[cfg(feature = "parallel")]
    fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
""", """    /// leave the polynomial as a finished sumcheck leaves it — one entry, the final claim (the hip provers bind on the device)
    #[cfg(feature = "hip")]
    pub(crate) fn set_final(&mut self, v: Scalar) {
        self.Z.clear();
        self.Z.push(v);
        self.num_vars = 0;
        self.len = 1;
    }

    /// ONE device call for the whole L x R matrix (rows dealt over all GPUs of the group when there are several)
    #[cfg(feature = "hip")]
    fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
        PolyCommitment { C: crate::hip::commit_rows(&self.Z, blinds, gens) }
    }

    #[cfg(all(feature = "parallel", not(feature = "hip")))]
    fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
""", "hyrax.rs commit_inner parallel")
    t = sub_once(t, """    #[cfg(not(feature = "parallel"))]
    fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
""", """    #[cfg(all(not(feature = "parallel"), not(feature = "hip")))]
    fn commit_inner(&self, blinds: &[Scalar], gens: &MultiCommitGens) -> PolyCommitment {
""", "hyrax.rs commit_inner serial")
    return t


def edit_sumcheck(t):
    # the two non-ZK provers get a second definition under the feature (same signature, body = the device loop); the reference's
    # own definitions stay as they are behind `not(feature = "hip")`
    t = sub_once(t, """    /// Prove sumcheck for cubic polynomial (A * B * C)
    pub fn prove_cubic<F>(
""", """    /// Prove sumcheck for cubic polynomial (A * B * C) — feature `hip`: the rounds run on the GPU (src/hip.rs).
    /// The only comb_func ever passed here is A * B * C (product_tree.rs:178-181); the closure itself cannot cross the FFI.
    #[cfg(feature = "hip")]
    pub fn prove_cubic<F>(
        claim: &Scalar,
        num_rounds: usize,
        poly_A: &mut DensePolynomial,
        poly_B: &mut DensePolynomial,
        poly_C: &mut DensePolynomial,
        _comb_func: F,
        transcript: &mut Transcript,
    ) -> (Self, Vec<Scalar>, Vec<Scalar>)
    where
        F: Fn(&Scalar, &Scalar, &Scalar) -> Scalar,
    {
        crate::hip::prove_cubic(claim, num_rounds, poly_A, poly_B, poly_C, transcript)
    }

    /// Prove sumcheck for cubic polynomial (A * B * C)
    #[cfg(not(feature = "hip"))]
    pub fn prove_cubic<F>(
""", "sumcheck.rs prove_cubic")
    t = sub_once(t, """    /// Prove batched sumcheck for cubic polynomials
    /// This is used in the product circuit evaluation
    pub fn prove_cubic_batched<F>(
""", """    /// Prove batched sumcheck for cubic polynomials — feature `hip`: a device-resident state (src/hip.rs); what crosses the
    /// boundary per round is what the transcript absorbs, the coeffs-combined triple.  comb_func is always A * B * C (product_tree.rs:275-278).
    #[cfg(feature = "hip")]
    pub fn prove_cubic_batched<F>(
        claim: &Scalar,
        num_rounds: usize,
        poly_vec_par: (
            &mut Vec<&mut DensePolynomial>,
            &mut Vec<&mut DensePolynomial>,
            &mut DensePolynomial,
        ),
        poly_vec_seq: (
            &mut Vec<&mut DensePolynomial>,
            &mut Vec<&mut DensePolynomial>,
            &mut Vec<&mut DensePolynomial>,
        ),
        coeffs: &[Scalar],
        _comb_func: F,
        transcript: &mut Transcript,
    ) -> (
        Self,
        Vec<Scalar>,
        (Vec<Scalar>, Vec<Scalar>, Scalar),
        (Vec<Scalar>, Vec<Scalar>, Vec<Scalar>),
    )
    where
        F: Fn(&Scalar, &Scalar, &Scalar) -> Scalar,
    {
        crate::hip::prove_cubic_batched(claim, num_rounds, poly_vec_par, poly_vec_seq, coeffs, transcript)
    }

    /// Prove batched sumcheck for cubic polynomials
    /// This is used in the product circuit evaluation
    #[cfg(not(feature = "hip"))]
    pub fn prove_cubic_batched<F>(
""", "sumcheck.rs prove_cubic_batched")

    # ---- prove_cubic_with_additive_term: tau * (Az * Bz - Cz) on the device, the Sigma-protocol around it unchanged
    t = sub_once(t, """        let mut claim_per_round = *claim;
        let mut comm_claim_per_round = claim_per_round.commit(blind_claim, gens_1);

        let mut r: Vec<Scalar> = Vec::new();
        let mut comm_polys: Vec<GroupElement> = Vec::new();
        let mut comm_evals: Vec<GroupElement> = Vec::new();
        let mut proofs: Vec<DotProductProof> = Vec::new();

        for j in 0..num_rounds {
            // Compute polynomial evaluations
            let (poly, comm_poly) = {
                let mut eval_point_0 = Scalar::zero();
                let mut eval_point_2 = Scalar::zero();
                let mut eval_point_3 = Scalar::zero();

                let len = poly_tau.len() / 2;
                for i in 0..len {
""", """        let mut claim_per_round = *claim;
        let mut comm_claim_per_round = claim_per_round.commit(blind_claim, gens_1);

        let mut r: Vec<Scalar> = Vec::new();
        let mut comm_polys: Vec<GroupElement> = Vec::new();
        let mut comm_evals: Vec<GroupElement> = Vec::new();
        let mut proofs: Vec<DotProductProof> = Vec::new();

        // feature `hip`: the four tables live on the GPU; the only comb_func ever passed is tau * (Az * Bz - Cz) (r1csproof.rs:288-292)
        #[cfg(feature = "hip")]
        let mut hip_rounds = {
            let _ = &comb_func;
            crate::hip::R1csRounds::new(poly_tau, poly_Az, poly_Bz, poly_Cz)
        };

        for j in 0..num_rounds {
            // Compute polynomial evaluations
            let (poly, comm_poly) = {
                let mut eval_point_0 = Scalar::zero();
                let mut eval_point_2 = Scalar::zero();
                let mut eval_point_3 = Scalar::zero();

                #[cfg(feature = "hip")]
                {
                    let (h0, h2, h3) = hip_rounds.evals();
                    eval_point_0 = h0;
                    eval_point_2 = h2;
                    eval_point_3 = h3;
                }
                #[cfg(not(feature = "hip"))]
                let len = poly_tau.len() / 2;
                #[cfg(not(feature = "hip"))]
                for i in 0..len {
""", "sumcheck.rs additive_term eval loop")
    t = sub_once(t, """            // bound all tables to the verifier's challenge
            poly_tau.bound_poly_var_top(&r_j);
            poly_Az.bound_poly_var_top(&r_j);
            poly_Bz.bound_poly_var_top(&r_j);
            poly_Cz.bound_poly_var_top(&r_j);
""", """            // bound all tables to the verifier's challenge (hip: on the device, in the pass that also prepares the next round's sums)
            #[cfg(feature = "hip")]
            hip_rounds.bind(&r_j);
            #[cfg(not(feature = "hip"))]
            {
                poly_tau.bound_poly_var_top(&r_j);
                poly_Az.bound_poly_var_top(&r_j);
                poly_Bz.bound_poly_var_top(&r_j);
                poly_Cz.bound_poly_var_top(&r_j);
            }
""", "sumcheck.rs additive_term binds")
    t = sub_once(t, """        (
            ZKSumcheckInstanceProof::new(comm_polys, comm_evals, proofs),
            r,
            vec![poly_tau[0], poly_Az[0], poly_Bz[0], poly_Cz[0]],
""", """        // hip: bring the final claims back and leave the host polynomials as the loop above leaves them (one entry each)
        #[cfg(feature = "hip")]
        {
            let f = hip_rounds.finals();
            poly_tau.set_final(f[0]);
            poly_Az.set_final(f[1]);
            poly_Bz.set_final(f[2]);
            poly_Cz.set_final(f[3]);
        }

        (
            ZKSumcheckInstanceProof::new(comm_polys, comm_evals, proofs),
            r,
            vec![poly_tau[0], poly_Az[0], poly_Bz[0], poly_Cz[0]],
""", "sumcheck.rs additive_term finals")

    # ---- prove_quad: z * ABC
    t = sub_once(t, """        for j in 0..num_rounds {
            // Compute polynomial evaluations
            let (poly, comm_poly) = {
                let mut eval_point_0 = Scalar::zero();
                let mut eval_point_2 = Scalar::zero();

                let len = poly_z.len() / 2;
                for i in 0..len {
""", """        // feature `hip`: both tables live on the GPU; the only comb_func ever passed is z * ABC (r1csproof.rs:389-390)
        #[cfg(feature = "hip")]
        let mut hip_rounds = {
            let _ = &comb_func;
            crate::hip::QuadRounds::new(poly_z, poly_ABC)
        };

        for j in 0..num_rounds {
            // Compute polynomial evaluations
            let (poly, comm_poly) = {
                let mut eval_point_0 = Scalar::zero();
                let mut eval_point_2 = Scalar::zero();

                #[cfg(feature = "hip")]
                {
                    let (h0, h2) = hip_rounds.evals();
                    eval_point_0 = h0;
                    eval_point_2 = h2;
                }
                #[cfg(not(feature = "hip"))]
                let len = poly_z.len() / 2;
                #[cfg(not(feature = "hip"))]
                for i in 0..len {
""", "sumcheck.rs quad eval loop")
    t = sub_once(t, """            // bound tables to the verifier's challenge
            poly_z.bound_poly_var_top(&r_j);
            poly_ABC.bound_poly_var_top(&r_j);
""", """            // bound tables to the verifier's challenge (hip: on the device)
            #[cfg(feature = "hip")]
            hip_rounds.bind(&r_j);
            #[cfg(not(feature = "hip"))]
            {
                poly_z.bound_poly_var_top(&r_j);
                poly_ABC.bound_poly_var_top(&r_j);
            }
""", "sumcheck.rs quad binds")
    t = sub_once(t, """        (
            ZKSumcheckInstanceProof::new(comm_polys, comm_evals, proofs),
            r,
            vec![poly_z[0], poly_ABC[0]],
""", """        #[cfg(feature = "hip")]
        {
            let f = hip_rounds.finals();
            poly_z.set_final(f[0]);
            poly_ABC.set_final(f[1]);
        }

        (
            ZKSumcheckInstanceProof::new(comm_polys, comm_evals, proofs),
            r,
            vec![poly_z[0], poly_ABC[0]],
""", "sumcheck.rs quad finals")
    return t


def edit_bullet(t):
    t = sub_once(t, """        let mut n = G_vec.len();
        assert_eq!(a_vec.len(), n);
        assert_eq!(b_vec.len(), n);
        assert!(n.is_power_of_two());
""", """        // feature `hip`: from MIN_GPU_MSM generators on, the reduction runs on the GPU (Q as its own base, scale one)
        #[cfg(feature = "hip")]
        if G_vec.len() >= crate::hip::MIN_GPU_MSM {
            return Self::prove_scaled(transcript, Q, &Scalar::one(), G_vec, H, a_vec, b_vec, blind, blinds_vec);
        }
        let mut n = G_vec.len();
        assert_eq!(a_vec.len(), n);
        assert_eq!(b_vec.len(), n);
        assert!(n.is_power_of_two());
""", "bullet.rs prove head")
    t = sub_once(t, """    /// Verify the bullet reduction proof
    #[allow(non_snake_case)]
    pub fn verify(
""", """    /// `prove` with Q = q_scale * Q_base.  DotProductProofLog::prove hands the reduction Q = gens_1.scale(r).G[0] with r fresh from the
    /// transcript (nizk/mod.rs): a new point per proof over a FIXED base.  Given as (base, scalar) the device keeps one derived
    /// generator set and lookup table per circuit instead of building them per proof.  Same proof, bit for bit.
    #[cfg(feature = "hip")]
    #[allow(clippy::too_many_arguments)]
    #[allow(non_snake_case)]
    pub fn prove_scaled(
        transcript: &mut Transcript,
        Q_base: &GroupElement,
        q_scale: &Scalar,
        G_vec: &[GroupElement],
        H: &GroupElement,
        a_vec: &[Scalar],
        b_vec: &[Scalar],
        blind: &Scalar,
        blinds_vec: &[(Scalar, Scalar)],
    ) -> (BulletReductionProof, GroupElement, Scalar, Scalar, GroupElement, Scalar) {
        if G_vec.len() < crate::hip::MIN_GPU_MSM {
            // short vectors (the sumchecks' 3-4 coefficient dot products never come here; tests do): the host loop
            return Self::prove(transcript, &(*q_scale * *Q_base), G_vec, H, a_vec, b_vec, blind, blinds_vec);
        }
        let (L_vec, R_vec, Gamma, a_hat, b_hat, g_hat, rhat_Gamma) =
            crate::hip::bullet_prove(transcript, Q_base, q_scale, G_vec, H, a_vec, b_vec, blind, blinds_vec);
        (BulletReductionProof { L_vec, R_vec }, Gamma, a_hat, b_hat, g_hat, rhat_Gamma)
    }

    /// Verify the bullet reduction proof
    #[allow(non_snake_case)]
    pub fn verify(
""", "bullet.rs prove_scaled")
    return t


def edit_nizk_mod(t):
    return sub_once(t, """        let (bullet_reduction_proof, _Gamma_hat, x_hat, a_hat, g_hat, rhat_Gamma) =
            BulletReductionProof::prove(
                transcript,
                &gens_1_scaled.G[0],
                &gens.gens_n.G,
""", """        // feature `hip`: Q = r * gens_1.G[0] goes to the device as (base, scalar) — see BulletReductionProof::prove_scaled
        #[cfg(feature = "hip")]
        let (bullet_reduction_proof, _Gamma_hat, x_hat, a_hat, g_hat, rhat_Gamma) =
            BulletReductionProof::prove_scaled(
                transcript,
                &gens.gens_1.G[0],
                &r,
                &gens.gens_n.G,
                &gens.gens_n.h,
                x_vec,
                a_vec,
                &blind_Gamma,
                &blinds_vec,
            );
        #[cfg(not(feature = "hip"))]
        let (bullet_reduction_proof, _Gamma_hat, x_hat, a_hat, g_hat, rhat_Gamma) =
            BulletReductionProof::prove(
                transcript,
                &gens_1_scaled.G[0],
                &gens.gens_n.G,
""", "nizk/mod.rs DotProductProofLog::prove")


EDITS = [
    ("0001-cargo-feature-hip", "Cargo.toml", edit_cargo),
    ("0002-lib-mod-hip", "src/lib.rs", edit_lib),
    ("0003-scalar-repr-transparent", "src/scalar.rs", edit_scalar),
    ("0004-group-msm", "src/group.rs", edit_group),
    ("0005-commitments-device-handle", "src/commitments.rs", edit_commitments),
    ("0006-hyrax-commit-inner", "src/hyrax.rs", edit_hyrax),
    ("0007-sumcheck-provers", "src/sumcheck.rs", edit_sumcheck),
    ("0008-bullet-prove-scaled", "src/nizk/bullet.rs", edit_bullet),
    ("0009-dotproductprooflog-scaled-q", "src/nizk/mod.rs", edit_nizk_mod),
]


def make():
    res = {}
    for name, rel, fn in EDITS:
        old = open(os.path.join(REF, rel)).read()
        new = fn(old)
        d = "".join(difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True), "a/" + rel, "b/" + rel, n=3))
        res[name + ".diff"] = d
    return res


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} is not here: the patches are generated (and checked) where the reference lies")
    res = make()
    if "--check" in sys.argv:
        bad = [k for k, v in res.items() if not os.path.exists(os.path.join(OUT, k)) or open(os.path.join(OUT, k)).read() != v]
        if bad:
            raise SystemExit(f"stale or missing patches: {bad}; run tools/make_shim_patches.py")
        print(f"{len(res)} patches are what the edits produce on the reference at {REF}")
        return
    os.makedirs(OUT, exist_ok=True)
    for k, v in res.items():
        open(os.path.join(OUT, k), "w").write(v)
        print(f"{k}: {v.count(chr(10))} lines")


if __name__ == "__main__":
    main()
