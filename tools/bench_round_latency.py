"""Per-call latency of a sumcheck round on small tables (the launch-bound regime: 18 instances, as the ops product circuits).
Calls the C ABI with pre-built handle arrays, i.e. what a compiled caller pays."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg, rand_scalars  # noqa: E402

sbn = load_pkg()
from spartan_bn254_amd import binding as B  # noqa: E402

ctx = sbn.Context(0)
L = B.lib()
cnt = 18
out = (C.c_uint8 * (96 * cnt))()
rp = (C.c_uint8 * 32).from_buffer_copy(rand_scalars(1, 99))


def arrays(n):
    ts = [ctx.table_upload(rand_scalars(4, 7 + i) * (n // 4)) for i in range(3 * cnt)]
    mk = lambda xs: (C.c_void_p * cnt)(*[t.h for t in xs])
    return ts, mk(ts[:cnt]), mk(ts[cnt:2 * cnt]), mk(ts[2 * cnt:])


ts, A, Bb, Cc = arrays(1024)
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(2000):
        L.sbn_sc_eval_cubic_batched(ctx.h, A, Bb, Cc, C.c_size_t(cnt), out)
    print("eval_cubic_batched, 18 instances of 2^10: %.1f us/call" % ((time.perf_counter() - t0) / 2000 * 1e6))
for t in ts:
    t.free()
for big in (1 << 10, 1 << 16):
    for rep in range(3):
        ts, A, Bb, Cc = arrays(big)
        ctx.sync(); t0 = time.perf_counter(); k = 0; m = big
        while m >= 4:
            L.sbn_sc_bind_eval_cubic_batched(ctx.h, A, Bb, Cc, C.c_size_t(cnt), rp, out); m //= 2; k += 1
        dt = time.perf_counter() - t0
        print("bind_eval_cubic_batched, 18 instances from 2^%d down: %.1f us/round over %d rounds" % (big.bit_length() - 1, dt / k * 1e6, k))
        for t in ts:
            t.free()
