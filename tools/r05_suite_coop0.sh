#!/bin/bash
# the whole GPU suite with every kernel forced into its large-batch launch shape (PCX_COOP_BELOW=0): the persistent workers of
# pcx_scrolly_maze_step / pcx_warehouse_step / pcx_hello_world_step then step every fixture of those games
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_suite_coop0; mkdir -p $OUT
cd $ROOT
PCX_COOP_BELOW=0 timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest.txt | tail -30
