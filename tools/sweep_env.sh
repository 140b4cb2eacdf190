#!/bin/bash
# usage: [ROUNDS=3] tools/sweep_env.sh VAR v1 v2 ...
# bench.py once per value of an environment tuning hook, ROUNDS interleaved rounds, median per value
var=$1; shift
rounds=${ROUNDS:-3}
tmp=$(mktemp)
for r in $(seq $rounds); do
  for v in "$@"; do
    env $var=$v python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        if 'value' in d: print('$v', d['encode_ms'], d['decode_ms'], d['value']/1e6)
" >> $tmp
  done
done
python - $tmp "$var" <<'PY'
import sys, statistics, collections
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    v, e, d, val = line.split()
    rows.setdefault(v, []).append((float(e), float(d), float(val)))
for v, r in rows.items():
    e = statistics.median(x[0] for x in r); d = statistics.median(x[1] for x in r); val = statistics.median(x[2] for x in r)
    print('%-22s enc %.3f ms  dec %.3f ms  value %.1f M  (median of %d; enc min %.3f, dec min %.3f)'
          % (sys.argv[2] + '=' + v, e, d, val, len(r), min(x[0] for x in r), min(x[1] for x in r)))
PY
rm -f $tmp
