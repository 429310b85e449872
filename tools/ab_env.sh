#!/bin/bash
# usage: tools/ab_env.sh VAR "<M list>"   -- interleaved same-box A/B of a library environment switch (VAR=0 vs VAR=1), store kernel
VAR=$1; MS=$2
for rep in 1 2 3; do for v in 0 1; do
  env $VAR=$v python tools/msweep.py 1000000000 $MS 2>&1 >/dev/null | python -c "
import sys, json
out = []
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); out.append('M=%d %.4f' % (r['M'], r['store']['ms']))
print('$VAR=$v rep$rep store ms:', '  '.join(out))"
done; done
