#!/bin/bash
# scratch: rounds of error removal vs closed / correct gaps and assembly time (bench.py, GF_BENCH_ASM_SIMPLIFY)
for cfg in C5 C4; do
  for r in 2 4 8; do
    GF_BENCH_ASM_SIMPLIFY=$r python bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg rounds $r', round(d['ms_per_step'],2), round(d['phases_ms']['assemble'],2), d['counts']['contigs'], d['counts']['gaps_closed'], d['counts']['gaps_closed_correct'])"
  done
done
