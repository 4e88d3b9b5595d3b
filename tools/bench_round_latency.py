import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg, rand_scalars
sbn = load_pkg(); from spartan_bn254_amd import binding as B
ctx = sbn.Context(0)
n = 1024; cnt = 18
ts = [ctx.table_upload(rand_scalars(n, i)) for i in range(3 * cnt)]
A = (C.c_void_p * cnt)(*[t.h for t in ts[:cnt]]); Bb = (C.c_void_p * cnt)(*[t.h for t in ts[cnt:2*cnt]]); Cc = (C.c_void_p * cnt)(*[t.h for t in ts[2*cnt:]])
out = (C.c_uint8 * (96 * cnt))()
L = B.lib()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(2000):
        L.sbn_sc_eval_cubic_batched(ctx.h, A, Bb, Cc, C.c_size_t(cnt), out)
    dt = time.perf_counter() - t0
    print("eval_cubic_batched count=18 len=1024: %.1f us/call" % (dt / 2000 * 1e6))
r = rand_scalars(1, 99); rp = (C.c_uint8 * 32).from_buffer_copy(r)
big = 1 << 16
for rep in range(3):
    ts2 = [ctx.table_upload(rand_scalars(4, 7) * (big // 4)) for i in range(3 * cnt)]
    A2 = (C.c_void_p * cnt)(*[t.h for t in ts2[:cnt]]); B2 = (C.c_void_p * cnt)(*[t.h for t in ts2[cnt:2*cnt]]); C2 = (C.c_void_p * cnt)(*[t.h for t in ts2[2*cnt:]])
    ctx.sync(); t0 = time.perf_counter(); k = 0; m = big
    while m >= 4:
        L.sbn_sc_bind_eval_cubic_batched(ctx.h, A2, B2, C2, C.c_size_t(cnt), rp, out); m //= 2; k += 1
    dt = time.perf_counter() - t0
    print("bind_eval_cubic_batched count=18 from 2^16 down: %.1f us/round over %d rounds" % (dt / k * 1e6, k))
    for t in ts2: t.free()
