#!/usr/bin/env python3
"""One keyless-shaped prove from the compiled harness (spartan-bn254_amd/harness/prove_stages.cpp), for profilers:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_harness -- python3 tools/trace_harness.py [passes]
then `python tools/trace_summary.py gpurun_out/prof_harness` for kernel time vs gaps per kernel kind."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from __graft_entry__ import load_pkg  # noqa: E402

sbn = load_pkg()
from spartan_bn254_amd import binding  # noqa: E402
ctx = sbn.Context(0)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lookup = int(os.environ.get("HARNESS_LOOKUP_GB", "200")) << 30
stages, digest, _, rounds = binding.harness_prove(ctx, 22, 21, 20, stateful=True, lookup_bytes_sat=16 << 30, lookup_bytes_eval=lookup, seed=11, passes=passes, trace_markers=bool(os.environ.get("HARNESS_MARKERS", "1") != "0"))
print(json.dumps({"stage_ms": {k: round(v, 3) for k, v in stages.items()}, "rounds": rounds, "digest": digest.hex()[:16]}))
ctx.close()
