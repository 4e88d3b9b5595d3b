for Q in 2048 4096 8192 16384 32768 8192; do
  export SBN_SC_COMB_MIN_Q=$Q
  python tools/bench_sumcheck.py 21 8 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stateful']; k=s['kernels_ms_total']; print(os.environ['SBN_SC_COMB_MIN_Q'], s['ms_per_sumcheck'], {x:k[x] for x in k if 'mixed' in x or 'comb' in x or 'cubic' in x})"
done
