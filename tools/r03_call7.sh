#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for c in warehouse_L0:262144 hello_world:262144; do
  echo "== $c"; PCX_DEBUG=8 timeout 120 python tools/generic_timing.py $c 2>&1 | grep -E "cycles per group" | tail -1
done
