#!/usr/bin/env python3
"""Sumcheck-round throughput at keyless sizes: the batched cubic sumcheck of the ops product circuits,
layer 0 — 12 "par" instances (A, B of 2^21) sharing one C table + 6 "seq" instances with their own C: 43 tables of
64 MiB = 2.7 GiB touched in round 0, 21 rounds, tables halving.  Reports the whole-sumcheck time with separate
eval + bind launches (the reference's structure) and with the fused bind+eval round, and HBM GB/s against 8 TB/s using
ALGORITHMIC bytes: eval reads every live table once; bind reads it once and writes half; the fused round reads once and
writes half.  Dev/measurement tool (bench.py stays the MSM metric)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from __graft_entry__ import load_pkg  # noqa: E402

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = 1 << logn
    sbn = load_pkg()
    ctx = sbn.Context(0)
    dev = torch.device("cuda:0")
    NPAR, NSEQ = 12, 6
    ntab = 2 * NPAR + 1 + 3 * NSEQ

    def fresh():
        g = torch.Generator(device=dev); g.manual_seed(7)
        ts = []
        for k in range(ntab):
            x = torch.empty(32 * n, dtype=torch.uint8, device=dev)
            ctx.scalars_synthetic(1000 + k, 0, n, x.data_ptr())          # uniform in Fr, full width
            ts.append(ctx.table_from_dev(x.data_ptr(), n, 0)); del x
        par_a, par_b, c_par = ts[:NPAR], ts[NPAR:2 * NPAR], ts[2 * NPAR]
        rest = ts[2 * NPAR + 1:]
        seq_a, seq_b, seq_c = rest[:NSEQ], rest[NSEQ:2 * NSEQ], rest[2 * NSEQ:]
        As, Bs, Cs = par_a + seq_a, par_b + seq_b, [c_par] * NPAR + seq_c
        return ts, As, Bs, Cs

    def challenge(ev):
        return (int.from_bytes(hashlib.sha3_256(ev).digest(), "little") % R_MOD).to_bytes(32, "little")

    res = {}
    for mode in ("separate", "fused"):
        times = []
        for rep in range(reps + 1):
            ts, As, Bs, Cs = fresh()
            ctx.prof_enable(True); ctx.prof_reset()
            ctx.sync(); t0 = time.perf_counter()
            ev = ctx.sc_eval_cubic_batched(As, Bs, Cs)
            for rnd in range(logn):
                r = challenge(ev)
                if mode == "fused" and len(ts[0]) >= 4:
                    ev = ctx.sc_bind_eval_cubic_batched(As, Bs, Cs, r)
                else:
                    ctx.bind_top_many(ts, r)
                    if len(ts[0]) >= 2:
                        ev = ctx.sc_eval_cubic_batched(As, Bs, Cs)
            ctx.sync(); dt = time.perf_counter() - t0
            prof = ctx.prof_get(); ctx.prof_enable(False)
            finals = [ctx.table_read0(t) for t in ts[:3]]
            for t in ts:
                t.free()
            if rep:
                times.append(dt)
        res[mode] = {"ms_per_sumcheck": round(1e3 * sum(times) / len(times), 3), "finals": [f.hex()[:16] for f in finals],
                     "kernels_ms_total": {k: round(v[0], 3) for k, v in prof.items()}, "kernels_launches": {k: v[1] for k, v in prof.items()}}
    # the stateful form (sbn_sumcheck_*): what crosses the boundary per round is the coeffs-combined triple (sumcheck.rs:269-271)
    coeffs = b"".join(((0x9E3779B97F4A7C15 * (i + 1)) % R_MOD).to_bytes(32, "little") for i in range(NPAR + NSEQ))
    times = []
    for rep in range(reps + 1):
        ts, As, Bs, Cs = fresh()
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.sync(); t0 = time.perf_counter()
        st, ev = ctx.sumcheck_begin(As[:NPAR], Bs[:NPAR], Cs[0], As[NPAR:], Bs[NPAR:], Cs[NPAR:], coeffs)
        for rnd in range(logn):
            ev = st.round(challenge(ev + bytes([rnd])))
        fin = st.finish()
        ctx.sync(); dt = time.perf_counter() - t0
        prof = ctx.prof_get(); ctx.prof_enable(False)
        st.free()
        for t in ts:
            t.free()
        if rep:
            times.append(dt)
    res["stateful"] = {"ms_per_sumcheck": round(1e3 * sum(times) / len(times), 3), "finals": [f.hex()[:16] for f in fin[:3]],
                       "kernels_ms_total": {k: round(v[0], 3) for k, v in prof.items()}, "kernels_launches": {k: v[1] for k, v in prof.items()}}
    assert res["separate"]["finals"] == res["fused"]["finals"], "fused and separate rounds disagree"
    table_bytes = ntab * n * 32
    # sum over rounds of live bytes: 2 * table_bytes (geometric)
    alg_sep = 2 * table_bytes * (1 + 1 + 0.5)          # eval read + bind read + bind write
    alg_fused = table_bytes + 2 * table_bytes * (1 + 0.5) - table_bytes * 0  # first eval + fused rounds (read + half write) over all rounds
    for mode, alg in (("separate", alg_sep), ("fused", alg_fused), ("stateful", alg_fused)):
        ms = res[mode]["ms_per_sumcheck"]
        res[mode]["algorithmic_GB"] = round(alg / 1e9, 3)
        res[mode]["GBps_vs_8000"] = round(alg / (ms * 1e-3) / 1e9, 1)
    kf = res["fused"]["kernels_ms_total"]
    stream_ms = kf.get("k_sc_bind_eval_cubic_stream", 0.0) + kf.get("k_bind_oop", 0.0)
    npf = max(0, logn - 15)
    if stream_ms and npf:
        res["fused"]["streaming_rounds_ms"] = round(stream_ms, 3)
        res["fused"]["streaming_rounds_GBps"] = round(table_bytes * 1.5 * sum(0.5 ** j for j in range(npf)) / (stream_ms * 1e-3) / 1e9, 1)
    fused_ms = stream_ms + kf.get("k_sc_bind_eval_cubic", 0.0) + kf.get("k_bind_top", 0.0)
    res["fused"]["kernel_only_rounds_ms"] = round(fused_ms, 3)
    res["fused"]["kernel_only_rounds_GBps"] = round(2 * table_bytes * 1.5 / (fused_ms * 1e-3) / 1e9, 1) if fused_ms else None
    ks = res["stateful"]["kernels_ms_total"]
    st_stream = ks.get("k_sc_round_mixed", 0.0) + ks.get("k_sc_round_mixed_first", 0.0) + ks.get("k_sc_comb_bind_eval", 0.0) + ks.get("k_sc_comb_bind_eval_first", 0.0) + ks.get("k_sc_bind_eval_cubic_stream", 0.0) + ks.get("k_bind_oop", 0.0)
    if st_stream and npf:
        res["stateful"]["streaming_rounds_ms"] = round(st_stream, 3)
        res["stateful"]["streaming_rounds_GBps"] = round(table_bytes * 1.5 * sum(0.5 ** j for j in range(npf)) / (st_stream * 1e-3) / 1e9, 1)
    res["env"] = {k: v for k, v in os.environ.items() if k.startswith("SBN_")}
    print(json.dumps({"workload": f"batched cubic sumcheck, {NPAR} par + {NSEQ} seq instances, tables of 2^{logn}, {logn} rounds, {round(table_bytes / 2**30, 2)} GiB in round 0", **res}))
    ctx.close()


if __name__ == "__main__":
    main()
