# stateful sumcheck (12 par + 6 seq x 2^21): how many instance groups the small streaming rounds are dealt into (SBN_SC_COMB_LANES = lanes to fill)
for L in 131072 262144 524288 1048576; do
  export SBN_SC_COMB_LANES=$L
  python tools/bench_sumcheck.py 21 3 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stateful']; k=s['kernels_ms_total']; print(os.environ['SBN_SC_COMB_LANES'], s['ms_per_sumcheck'], k.get('k_sc_round_mixed_first'), k.get('k_sc_round_mixed'), k.get('k_sc_comb_bind_eval'))"
done
