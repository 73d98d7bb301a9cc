#!/bin/bash
# one compact line per bench.py run: bash tools/bench_line.sh <label> [bench.py args]
L=$1; shift
python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$L', c['workload'][:32], d['value'], d['ms_per_step'], c.get('fwd_ms'), c.get('bwd_ms'))"
