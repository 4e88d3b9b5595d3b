# stateful sumcheck (12 par + 6 seq x 2^21): blocks per launch of the one-launch round kernel (SBN_SC_COMB_BLOCKS for the "par" groups,
# SBN_SC_SEQ_BLOCKS for the "seq" instances); each block ends in a ~7 us epilogue (wave sums, write-through stores, ticket)
PAIRS=${PAIRS:-256:256 384:384 512:256 512:512 512:768 768:768 1024:1024 1024:2048}
for P in $PAIRS; do
  export SBN_SC_COMB_BLOCKS=${P%%:*} SBN_SC_SEQ_BLOCKS=${P##*:}
  python tools/bench_sumcheck.py 21 ${REPS:-8} 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stateful']; k=s['kernels_ms_total']; print(os.environ['SBN_SC_COMB_BLOCKS'], os.environ['SBN_SC_SEQ_BLOCKS'], s['ms_per_sumcheck'], k.get('k_sc_round_mixed_first'), k.get('k_sc_round_mixed'), k.get('k_sc_comb_bind_eval'))"
done
