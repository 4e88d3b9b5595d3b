#!/usr/bin/env python3
"""per-round latency of the stateful sumcheck (sbn_sumcheck_*) on small tables, one launch per round, ctypes calls (round 3 measured a
resident "tail" kernel against it — tables in LDS, host hand-shake through a pinned mailbox, commit af333d1 — and dropped it: 22.0 vs
17.5 us per round at 2^10 x 18 instances; the PCIe poll / publish hand-shake costs more than a launch).
usage: python tools/bench_stateful_latency.py [logn] [n_par] [n_seq]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import rand_scalars  # noqa: E402

sbn = load_pkg(); ctx = sbn.Context(0); L = sbn.lib()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_par = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n_seq = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = 1 << logn
tabs = [ctx.table_upload(rand_scalars(n, 5 + k)) for k in range(2 * n_par + 1 + 3 * n_seq)]
o = 2 * n_par + 1
mk = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.h for t in ts])
Ap, Bp, As, Bs, Cs = mk(tabs[:n_par]), mk(tabs[n_par:2 * n_par]), mk(tabs[o:o + n_seq]), mk(tabs[o + n_seq:o + 2 * n_seq]), mk(tabs[o + 2 * n_seq:])
co = rand_scalars(n_par + n_seq, 99); ch = [rand_scalars(1, 200 + j) for j in range(logn)]
out = (C.c_uint8 * 96)(); fin = (C.c_uint8 * (32 * len(tabs)))()
for mode in ("launch per round",):
    best = None
    for rep in range(30):
        st = C.c_void_p()
        t0 = time.perf_counter()
        assert L.sbn_sumcheck_begin(ctx.h, Ap, Bp, tabs[2 * n_par].h, C.c_size_t(n_par), As, Bs, Cs, C.c_size_t(n_seq), co, out, C.byref(st)) == 0
        t1 = time.perf_counter()
        for j in range(logn):
            assert L.sbn_sumcheck_round(ctx.h, st, ch[j], out) == 0
        t2 = time.perf_counter()
        assert L.sbn_sumcheck_finish(ctx.h, st, fin) == 0
        L.sbn_sumcheck_free(ctx.h, st)
        t3 = time.perf_counter()
        cur = (t1 - t0, (t2 - t1) / logn, t3 - t2, t3 - t0)
        if rep >= 5 and (best is None or cur[3] < best[3]):
            best = cur
    print(f"{mode:18s} 2^{logn} x ({n_par} par + {n_seq} seq): begin {best[0] * 1e6:7.1f} us, per round {best[1] * 1e6:6.1f} us, finish+free {best[2] * 1e6:6.1f} us, whole sumcheck {best[3] * 1e6:7.1f} us  finals {bytes(fin[:8]).hex()}")
ctx.close()
