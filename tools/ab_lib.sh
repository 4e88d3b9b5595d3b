#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_lib.sh <variant .so> [bench args...]  (runs base, variant, base, variant)
var=$1; shift
cd spartan-bn254_amd && cp libsbn254_hip.so /tmp/base.so && cp $var /tmp/var.so && cd ..
for rep in 1 2; do for which in base var; do
  cp /tmp/$which.so spartan-bn254_amd/libsbn254_hip.so
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['summary']
print('$which', json.dumps({k: s[k] for k in ('sumcheck','prove') if k in s}))"
done; done
cp /tmp/base.so spartan-bn254_amd/libsbn254_hip.so
