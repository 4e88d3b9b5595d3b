"""BASELINE config 5 from COMPILED code: spartan-bn254_amd/harness/prove_stages.cpp issues the device-side stages of a keyless-shaped
prove through the C ABI exactly as the Rust shim would (one call per sumcheck round, host-side UniPoly / hash in between).  Here it
runs at 2^-14 of the keyless sizes with its trace on, and tests/harness_model.py replays the trace against the CPU oracle: every
commitment, every round's sums, every final claim, every bullet-reduction L / R of every stage.  Integer work: bit-exact."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 7, 6), (9, 8, 6)])
def test_harness_small_prove_vs_oracle(ctx, sbn, shape):
    import harness_model
    from spartan_bn254_amd import binding
    lo, lm, lc = shape
    digests = []
    for stateful in (True, False):
        stages, digest, trace, rounds = binding.harness_prove(ctx, lo, lm, lc, stateful=stateful, seed=3, trace_cap=8 << 20)
        got = harness_model.replay(trace, lo, lm, lc, seed=3)
        assert got["records_checked"] > 100 and got["digest"] == digest
        assert got["sumcheck_rounds_ops"] == rounds["sumcheck_rounds_ops"] == sum(range(1, lo))
        assert got["sumcheck_rounds_mem"] == rounds["sumcheck_rounds_mem"] == sum(range(1, lm))
        assert all(v >= 0 for v in stages.values()) and stages["network_proof"] > 0
        digests.append(digest)
    assert digests[0] == digests[1], "the stateful sumcheck and the per-instance calls absorbed different values"


def test_harness_lookup_tables_same_transcript(ctx, sbn):
    """with the fixed-base lookup tables (sbn_bases_precompute) the commitments, hence the whole transcript, are unchanged"""
    from spartan_bn254_amd import binding
    _, d0, _, _ = binding.harness_prove(ctx, 8, 7, 6, seed=5)
    _, d1, _, _ = binding.harness_prove(ctx, 8, 7, 6, seed=5, lookup_bytes_sat=8 << 20, lookup_bytes_eval=64 << 20)
    _, d2, _, _ = binding.harness_prove(ctx, 8, 7, 6, seed=6)
    assert d0 == d1 and d0 != d2
