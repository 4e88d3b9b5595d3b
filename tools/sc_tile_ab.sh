# stateful sumcheck, 12 par + 6 seq x 2^21: the tile-interleaved layout of the state's tables against plain tables, same box, alternating
for rep in 1 2; do for e in "SBN_SC_X=1" "SBN_SC_NO_TILE=1"; do
  env $e python tools/bench_sumcheck.py 21 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stateful']; k=s['kernels_ms_total']; print('$e', s['ms_per_sumcheck'], k.get('k_sc_round_mixed_first'), k.get('k_sc_round_mixed'), k.get('k_bind_oop'))"
done; done
