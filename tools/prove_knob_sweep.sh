# prove-level A/B of the library's experiment overrides: total device-side ms of the keyless-shaped prove (compiled harness) per setting
# usage: tools/prove_knob_sweep.sh VAR v1 v2 ...   ("-" = unset)
VAR=$1; shift
for V in "$@"; do
  if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
  python bench.py --no-cpu-baseline --blocks prove_stages 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ps=d['prove_stages']; print('$VAR=$V', ps.get('total_device_side_ms'), ps.get('detail_ms',{}).get('ops_sumchecks'), ps.get('detail_ms',{}).get('mem_sumchecks'), ps.get('error','')[:200] if isinstance(ps.get('error'),str) else '')"
done
