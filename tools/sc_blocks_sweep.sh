# stateful sumcheck (12 par + 6 seq x 2^21): blocks per launch.  Every block ends in a ~7 us epilogue (wave sums, write-through stores, ticket).
#   MODE=round (default): the one-launch round kernel, PAIRS = SBN_SC_COMB_BLOCKS:SBN_SC_SEQ_BLOCKS ("par" groups : "seq" instances)
#   MODE=begin: round 0's evaluation launch (one mixed launch; two with SBN_SC_NO_MIXED_EVAL=1), PAIRS = SBN_SC_COMB_EVAL_BLOCKS:SBN_SC_EVAL_BLOCKS
if [ "${MODE:-round}" = begin ]; then A=SBN_SC_COMB_EVAL_BLOCKS; B=SBN_SC_EVAL_BLOCKS; PAIRS=${PAIRS:-2048:4096 1024:4096 512:4096 2048:2048 2048:1024 1024:1024 512:512 1024:2048}
else A=SBN_SC_COMB_BLOCKS; B=SBN_SC_SEQ_BLOCKS; PAIRS=${PAIRS:-256:256 384:384 512:256 512:512 512:768 768:768 1024:1024 1024:2048}; fi
for P in $PAIRS; do
  export $A=${P%%:*} $B=${P##*:}
  python tools/bench_sumcheck.py 21 ${REPS:-8} 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stateful']; k=s['kernels_ms_total']; print('$P', s['ms_per_sumcheck'], 'first', k.get('k_sc_round_mixed_first'), 'rounds', k.get('k_sc_round_mixed'), 'eval_mixed', k.get('k_sc_eval_mixed'), 'comb_eval', k.get('k_sc_comb_eval'), 'eval', k.get('k_sc_eval_cubic'))"
done
