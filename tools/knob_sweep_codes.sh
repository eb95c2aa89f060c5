#!/bin/bash
for w in 5 6 7 8; do for c in 1 0; do PCX_WAVES_PER_CU=$w PCX_SM_CODES=$c python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves_per_cu=$w codes=$c  %.5f ms  frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; done; done
