#!/usr/bin/env python3
"""Which XCD each block of the one-launch sumcheck round (k_sc_round_mixed) ran on and when each kind of block finished.
Runs tools/bench_sumcheck.py's stateful leg once with SBN_SC_DEBUG_BLOCKS (the library then records kind, HW_REG_XCC_ID and the
100 MHz clock at entry / exit of every block) and prints, per launch: blocks of each kind per XCD, and per kind the time from the
launch's first block start to the kind's last block end."""
import collections
import os
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = tempfile.mktemp(suffix=".csv")
env = dict(os.environ, SBN_SC_DEBUG_BLOCKS=path)
logn = sys.argv[1] if len(sys.argv) > 1 else "21"
subprocess.run([sys.executable, os.path.join(root, "tools", "bench_sumcheck.py"), logn, "1"], env=env, check=True, stdout=subprocess.DEVNULL)
launches = collections.OrderedDict()
for line in open(path):
    seq, q, b, kind, xcc, t0, t1 = (int(x) for x in line.strip().split(","))
    launches.setdefault((seq, q), []).append((b, kind, xcc, t0, t1))
os.unlink(path)
seen_q = set()
for (seq, q), rows in launches.items():
    if q in seen_q:
        continue                      # warm-up and timed repetitions repeat the same rounds: first of each size only
    seen_q.add(q)
    start = min(r[3] for r in rows)
    per = {0: collections.Counter(), 1: collections.Counter()}
    end = {0: 0, 1: 0}; busy = {0: 0, 1: 0}
    for b, kind, xcc, t0, t1 in rows:
        per[kind][xcc] += 1; end[kind] = max(end[kind], t1 - start); busy[kind] += t1 - t0
    n = {k: sum(per[k].values()) for k in per}
    print(f"q=2^{q.bit_length() - 1}: {len(rows)} blocks; par-group blocks per XCD {[per[0][x] for x in range(8)]}, seq blocks per XCD {[per[1][x] for x in range(8)]}")
    print(f"    last par block ends at {end[0] / 100:.1f} us, last seq block at {end[1] / 100:.1f} us; mean block time par {busy[0] / max(n[0], 1) / 100:.1f} us, seq {busy[1] / max(n[1], 1) / 100:.1f} us")
