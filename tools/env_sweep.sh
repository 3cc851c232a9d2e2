#!/bin/bash
# usage: tools/env_sweep.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench.py run per environment setting, one summary line each
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-cpu-baseline --steps 50 2>/dev/null | tail -1)
  echo "$cfg :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["stages_us"])')"
done
