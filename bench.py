#!/usr/bin/env python3
"""bench.py — throughput of the MI355X MSM hot path (BASELINE.json metric: BN254 G1 MSM points/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm|hyrax] [--log-n 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  msm    (default; BASELINE configs[1]) one MSM of 2^log_n points per GPU: uniform scalars, distinct bases with known
         discrete logs.  With N > 1 the MSM of N*2^log_n points is sharded by base-point range, each
         rank computes its partial sum, and ONE all-gather of the 64-byte partials over RCCL + a local fold finishes it
         (weak scaling: per-GPU work fixed).
  hyrax  (BASELINE configs[2]) the derefs commitment shape: 4096 x 8192 scalars against the reference's 8192(+h) shared
         generators (~66 % of them equal to G), rows 3072.. zero; with N > 1 every rank commits its own matrix (rows are
         independent: no data-path collective).
Before timing, the result is checked bit-for-bit against the discrete-log identity / the CPU oracle.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from __graft_entry__ import load_pkg  # noqa: E402

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MADD_PEAK = 1.2e10             # xyzz mixed additions/s with operands in registers, whole chip (tools/micro/ecbench.hip, measured on MI355X)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable streaming)
S0 = 0x1234567890abcdef1234567890abcdef
DSTEP = 0x0fedcba987654321


def splitmix_scalars(n, seed):
    """n uniform canonical Fr scalars (32 B LE each) from SplitMix64, vectorised"""
    with np.errstate(over="ignore"):
        idx = np.arange(4 * n, dtype=np.uint64) + np.uint64(1)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    limbs = z.reshape(n, 4).copy()
    limbs[:, 3] &= np.uint64((1 << 62) - 1)                      # < 2^254
    r = [np.uint64((R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    ge = np.zeros(n, dtype=bool); decided = np.zeros(n, dtype=bool)
    for i in (3, 2, 1, 0):
        gt = (limbs[:, i] > r[i]) & ~decided; lt = (limbs[:, i] < r[i]) & ~decided
        ge |= gt; decided |= gt | lt
    ge |= ~decided
    borrow = np.zeros(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(4):                                        # subtract r where value >= r  (value < 2^254 < 2r)
            sub = np.where(ge, r[i], np.uint64(0))
            t = limbs[:, i] - sub
            b1 = (limbs[:, i] < sub).astype(np.uint64)
            t2 = t - borrow
            b2 = (t < borrow).astype(np.uint64)
            limbs[:, i] = t2; borrow = b1 | b2
    return limbs.tobytes()


def fr_dot_arith(scalars, first, n):
    """sum_i k_i * (S0 + (first+i)*DSTEP) mod r  =  S0*sum(k_i) + DSTEP*sum((first+i)*k_i), exact, vectorised:
    the scalars are split into 16-bit digits so that every numpy partial sum stays below 2^64."""
    k16 = np.frombuffer(scalars, dtype=np.uint16).reshape(n, 16).astype(np.uint64)       # little-endian 16-bit digits
    idx = np.arange(n, dtype=np.uint64)
    ilo, ihi = idx & np.uint64(0xFFFF), idx >> np.uint64(16)                             # n <= 2^32
    sum_k = 0; sum_ik = 0
    for j in range(16):
        col = k16[:, j]
        sj = int(col.sum(dtype=np.uint64))                      # < 2^16 * n
        lo = int((col * ilo).sum(dtype=np.uint64))              # < 2^32 * n  (n <= 2^31)
        hi = int((col * ihi).sum(dtype=np.uint64))
        sum_k += sj << (16 * j)
        sum_ik += (lo + (hi << 16)) << (16 * j)
    tot = (S0 * sum_k + DSTEP * (sum_ik + first * sum_k)) % R_MOD
    return tot.to_bytes(32, "little")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=["msm", "hyrax"], default="msm")
    ap.add_argument("--log-n", type=int, default=20, help="msm: log2 of the points per GPU")
    ap.add_argument("--rows", type=int, default=4096, help="hyrax: matrix rows per GPU")
    ap.add_argument("--cols", type=int, default=8192, help="hyrax: matrix columns")
    ap.add_argument("--inflight", type=int, default=4, help="independent steps kept in flight on separate HIP streams (contexts); 1 = strictly serial")
    ap.add_argument("--fixed-base", action="store_true", help="msm: treat the resident bases as fixed generators (per-window multiples precomputed once, "
                    "as the commit path does): sbn_commit_rows with L = 1 instead of sbn_msm_bases")
    ap.add_argument("--const-tail", type=float, default=0.0, help="hyrax: fraction of each 512-row block whose rows repeat one constant "
                    "(the padded tail of every derefs matrix repeats mem[0], sparse_mlpoly_full.rs:89-101; ~0.43 at keyless size)")
    ap.add_argument("--precompute-gb", type=float, default=100.0, help="hyrax: HBM budget (GiB) of the fixed-base lookup table of the generator set "
                    "(sbn_bases_precompute; built once before the timed region like any commitment-key setup); 0 = bucket method")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices and the
    # 64-byte partials travel as CPU tensors); the driver's runs use the default: "nccl", which is RCCL on ROCm.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    sbn = load_pkg()
    from spartan_bn254_amd import sharding
    import oracle_lib as ol          # the checker and the cpu_baseline leg only
    ctx = sbn.Context(dev_index)    # raises if the HIP library / device is missing: no fallback
    M = max(1, args.inflight)
    ctxs = [ctx] + [sbn.Context(dev_index) for _ in range(M - 1)]     # one HIP stream + workspace per step in flight

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for cx in ctxs:
            cx.sync()

    G_XY = bytes([1]) + bytes(31) + bytes([2]) + bytes(31)
    cpu_baseline = None
    if args.workload == "msm":
        n = 1 << args.log_n
        first = rank * n                                           # this rank's base-point range of the N*n-point MSM
        scal = splitmix_scalars(n, 0x5BA27A2B4E254 + 7919 * rank)
        d_scal = torch.frombuffer(bytearray(scal), dtype=torch.uint8).to(dev)
        bases = ctx.bases_synthetic(n, first, S0.to_bytes(32, "little"), DSTEP.to_bytes(32, "little"))
        torch.cuda.synchronize()

        def local_step(cx):
            if args.fixed_base:
                xy, infs = cx.commit_rows_dev(bases, d_scal.data_ptr(), 0, 1, n)
                return xy, bool(infs[0])
            return cx.msm_bases_dev(bases, d_scal.data_ptr(), n)

        def finish(parts):
            """one RCCL all-gather for the partial sums of the steps that just completed, then the local folds"""
            if world == 1:
                return parts
            xy, _ = zip(*parts)
            t = torch.frombuffer(bytearray(b"".join(xy)), dtype=torch.uint8).to(coll_dev)
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            allb = [o.cpu().numpy().tobytes() for o in outs]
            return [sbn.g1_sum(b"".join(a[64 * j:64 * j + 64] for a in allb)) for j in range(len(parts))]

        def step():
            return finish([local_step(ctx)])[0]

        # parity gate: partial == (sum k_i s_i) G, exact
        part, _ = local_step(ctx)
        want = ol.g1_mul(G_XY, fr_dot_arith(scal, first, n))
        if part != want:
            raise SystemExit(f"rank {rank}: GPU MSM result differs from the discrete-log oracle")
        total, _ = step()
        if world > 1:                                             # the folded result must be the sum of all ranks' expectations
            wants = [None] * world
            dist.all_gather_object(wants, want)
            if total != sbn.g1_sum(b"".join(wants))[0]:
                raise SystemExit("sharded MSM result differs from the folded oracle partials")
        units_per_step = n
        alg_bytes_per_launch = 96.0 * n                            # 32 B scalar + 64 B affine base per point
        active_fraction = 1.0                                      # uniform scalars: a digit is zero with probability 2^-c
        dominant = "k_acc_first"
        workload = f"synthetic BN254 G1 MSM, 2^{args.log_n} uniform scalars x distinct bases per GPU, inputs resident in HBM" + (
            " (fixed-base mode: per-window multiples of the bases precomputed once)" if args.fixed_base else "")
        metric = "msm_points_per_s"
        unit = "points/s"
    else:
        L, Rc = args.rows, args.cols
        bases, _ = ctx.gens_new(Rc, b"gens_r1cs_eval", want_points=False)    # the reference's gens_derefs set (sparse_mlpoly_full.rs:625-627)
        comb_c = 0
        if args.precompute_gb > 0:
            tpre = time.perf_counter()
            try:
                comb_c = ctx.bases_precompute(bases, int(args.precompute_gb * (1 << 30)))
            except sbn.SbnError as e:                              # e.g. another tenant holds the HBM: keep the bucket method
                print(f"[bench] precompute skipped: {e}", file=sys.stderr)
            tpre = time.perf_counter() - tpre
        g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
        Z = torch.randint(0, 2**31 - 1, (L * Rc, 8), dtype=torch.int32, device=dev, generator=g)
        Z[:, 7] &= 0x0fffffff                                     # canonical (< 2^252)
        Z[(3 * L // 4) * Rc:] = 0                                  # rows 3072.. are zero padding (hyrax.rs:245)
        if args.const_tail > 0:                                    # SURVEY 8d config 3 variant: constant suffix of every 512-row block
            blk = max(1, L // 8)
            for b0 in range(0, 3 * L // 4, blk):
                k0 = b0 + int(blk * (1.0 - args.const_tail))
                Z[k0 * Rc:(b0 + blk) * Rc] = Z[k0 * Rc]
        torch.cuda.synchronize()

        def local_step(cx):
            return cx.commit_rows_dev(bases, Z.data_ptr(), 0, L, Rc)

        def finish(parts):
            return parts

        def step():
            return local_step(ctx)

        out, infs = step()
        gxy, _ = ol.gens_new(Rc, b"gens_r1cs_eval")
        for i in (0, L // 2, 3 * L // 4 - 1, L - 1):
            row = Z[i * Rc:(i + 1) * Rc].cpu().numpy().tobytes()
            if out[64 * i:64 * i + 64] != ol.commit(row, bytes(32), gxy[:64 * Rc], gxy[64 * Rc:]):
                raise SystemExit(f"rank {rank}: Hyrax row {i} differs from the oracle")
        units_per_step = L * Rc
        alg_bytes_per_launch = L * Rc * 32.0 + (Rc + 1) * 64.0 + L * 64.0     # 32.02 B/pair at 4096 x 8192
        active_fraction = 0.75 * (1.0 - args.const_tail)           # zero rows and constant rows add (almost) nothing to the buckets
        dominant = "k_comb_rows" if comb_c else "k_acc_first"
        workload = f"Hyrax derefs commitment shape: {L} x {Rc} scalars per GPU, {Rc}+1 shared reference generators, last quarter of rows zero, inputs resident in HBM" + (
            f", fixed-base lookup table c={comb_c} (built once in {tpre:.2f} s)" if comb_c else ", bucket method") + (
            f", constant tail {args.const_tail:.2f} of every {max(1, L // 8)}-row block" if args.const_tail > 0 else "")
        metric = "msm_points_per_s"
        unit = "points/s"

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=M)

    def run_steps(count):
        """`count` steps with up to M in flight.  N = 1: each stream free-runs its share.  N > 1: rounds of up to 4*M local steps,
        then one collective per round issued from this thread (collectives must be ordered identically on every rank)."""
        if M == 1:
            for _ in range(count):
                step()
            return
        if world == 1:
            shares = [count // M + (1 if j < count % M else 0) for j in range(M)]
            futs = [pool.submit(lambda cx=cx, k=k: [local_step(cx) for _ in range(k)]) for cx, k in zip(ctxs, shares)]
            for f in futs:
                f.result()
            return
        done = 0
        while done < count:                       # rounds of up to 4 steps per stream, then ONE all-gather carrying all their partials
            g = min(4 * M, count - done)
            shares = [g // M + (1 if j < g % M else 0) for j in range(M)]
            futs = [pool.submit(lambda cx=cx, k=k: [local_step(cx) for _ in range(k)]) for cx, k in zip(ctxs, shares) if k]
            finish([p for f in futs for p in f.result()])
            done += g

    for cx in ctxs:                       # every stream's workspace is sized before the timed region
        local_step(cx)
    run_steps(args.warmup)
    for cx in ctxs:
        cx.prof_enable(True); cx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    for cx in ctxs:
        for k, (ms, cnt) in cx.prof_get().items():
            a, b = prof.get(k, (0.0, 0)); prof[k] = (a + ms, b + cnt)
        cx.prof_enable(False)
    # serial reference: a few steps one at a time on one stream — the latency of a single call and per-kernel durations that
    # are not stretched by other streams' kernels running beside them
    serial = None
    if M > 1:
        ns_ser = max(2, min(16, args.steps))
        ctx.prof_enable(True); ctx.prof_reset()
        barrier(); ts0 = time.perf_counter()
        for _ in range(ns_ser):
            step()
        barrier(); ts = time.perf_counter() - ts0
        sp = ctx.prof_get(); ctx.prof_enable(False)
        serial = {"steps": ns_ser, "ms_per_step": round(ts / ns_ser * 1e3, 4), "kernels_avg_ms": {k: round(ms / max(cnt, 1), 4) for k, (ms, cnt) in sp.items()}}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = units_per_step * world * args.steps / dt
        kern = {k: round(ms / max(cnt, 1), 4) for k, (ms, cnt) in prof.items()}
        dom_ms, dom_cnt = prof.get(dominant, (0.0, 0))
        dom_avg_ms = dom_ms / max(dom_cnt, 1)
        achieved = alg_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else None
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # HBM bytes/launch from separate rocprofv3 --pmc passes
        if os.path.exists(pmc_path):
            try:
                per_kernel = json.load(open(pmc_path)).get(args.workload, {})
                traffic = per_kernel.get(dominant)
                if traffic is None:                     # template instances are recorded as name<args>: take the busiest one
                    inst = [v for k, v in per_kernel.items() if k.startswith(dominant + "<")]
                    traffic = max(inst) if inst else None
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None, "traffic": traffic,
                    "kernel_avg_ms": round(dom_avg_ms, 4),
                    "note": "MSM is integer-ALU bound (~170 modular products per point vs 96 B): see DESIGN.md for the ALU roofline"}
        # the bound that actually applies: mixed additions per second of the accumulate kernel against the in-register
        # ceiling measured by tools/micro/ecbench.hip on MI355X (every slot of a non-zero digit is one 8M+2S mixed addition)
        job = ctx.prof_last_job()
        madds = job["slots"] * active_fraction
        ref_ms = serial["kernels_avg_ms"].get(dominant) if serial else dom_avg_ms
        if ref_ms:
            roofline["alu"] = {"unit": "mixed additions/s", "achieved": round(madds / (ref_ms * 1e-3), 1), "peak": MADD_PEAK, "frac": round(madds / (ref_ms * 1e-3) / MADD_PEAK, 4),
                               "window_bits": job["c"], "windows": job["W"], "mixed_additions_per_launch": int(madds),
                               "note": "peak = xyzz_madd in registers, all CUs busy (tools/micro/ecbench.hip: 1.2e10/s = 1.29e11 Montgomery products/s); kernel time from the one-step-in-flight pass"}
        if serial and serial["kernels_avg_ms"].get(dominant):
            sa = alg_bytes_per_launch / (serial["kernels_avg_ms"][dominant] * 1e-3) / 1e9
            roofline["serial_pass"] = {"kernel_avg_ms": serial["kernels_avg_ms"][dominant], "achieved": round(sa, 2), "frac": round(sa / HBM_PEAK_GBS, 5),
                                       "note": "same kernel with ONE step in flight: with several streams a launch shares the chip, so its event-to-event time is longer"}
        if not args.no_cpu_baseline and world == 1:
            cores = min(len(os.sched_getaffinity(0)), 16)
            if args.workload == "msm":
                ns = min(n, 1 << 20)
                pts = ctx.bases_download(bases, 0, ns)
                tb = time.perf_counter(); got = ol.msm_pippenger(scal[:32 * ns], pts, cores); tcpu = time.perf_counter() - tb
                if ns == n and got != want:
                    raise SystemExit("CPU oracle and GPU disagree")
                cpu_baseline = {"value": round(ns / tcpu, 1), "unit": unit, "cores": cores, "kind": "port",
                                "sample": f"the same MSM on its first 2^{ns.bit_length() - 1} points: oracle/ arkworks-style signed-digit Pippenger (c={ol.window_bits(ns)}), threads over windows"}
            else:
                rows = min(L, 2 * cores)
                Zs = Z[:rows * Rc].cpu().numpy().tobytes()
                tb = time.perf_counter(); got = ol.commit_rows(Zs, None, rows, Rc, gxy[:64 * Rc], gxy[64 * Rc:], cores); tcpu = time.perf_counter() - tb
                if got != out[:64 * rows]:
                    raise SystemExit("CPU oracle and GPU disagree")
                cpu_baseline = {"value": round(rows * Rc / tcpu, 1), "unit": unit, "cores": cores, "kind": "port",
                                "sample": f"first {rows} rows of the same matrix: oracle/ per-row Pippenger, threads over rows (hyrax.rs:259-261)"}
        line = {"metric": metric, "value": round(value, 1), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic",
                "config": {"workload": workload, "units_per_step_per_gpu": units_per_step, "steps_in_flight": M, "collective_backend": backend if world > 1 else None,
                           "sharding": "base-point ranges + one RCCL all-gather of 64-B partial sums" if args.workload == "msm" else "independent matrices per GPU, no collective",
                           "parity": "bit-exact vs discrete-log oracle, checked before timing"},
                "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels_avg_ms": kern, "serial_reference": serial}
        print(json.dumps(line), flush=True)
    bases.free()
    pool.shutdown()
    for cx in ctxs:
        cx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
