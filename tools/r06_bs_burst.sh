#!/bin/bash
# Round 6: pcx_better_scrolly_step with the stores of KB iterations regrouped plane by plane (pcx_stream.h stream_planes_burst):
# parity with the loop forced on, then same-box A/B (PCX_DEBUG bits 64 / 128 / 256: KB = 4 / 8 / 2).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_bs_burst
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
for d in 64 128; do
  PCX_DEBUG=$d PCX_COOP_BELOW=0 python -m pytest tests/test_hip_parity.py tests/test_gate_digests.py tests/test_random_levels.py -m gpu -q -x -k "better" 2>&1 | tail -3 >> $OUT/tests.txt
done
cat $OUT/tests.txt
{
python tools/env_sweep.py --game better_scrolly_maze --batches 65536,262144 --steps 30 --variants "default;b4:PCX_DEBUG=64;b8:PCX_DEBUG=128;b2:PCX_DEBUG=256" 2>&1 | $Q
python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 262144 --steps 30 --variants "default;b4:PCX_DEBUG=64;b8:PCX_DEBUG=128;b2:PCX_DEBUG=256" 2>&1 | $Q
python tools/env_sweep.py --fixture better_scrolly_maze_L2 --batches 131072 --steps 30 --variants "default;b4:PCX_DEBUG=64;b8:PCX_DEBUG=128;b2:PCX_DEBUG=256" 2>&1 | $Q
} > $OUT/r06_bs_burst_sweep.txt 2>&1
cat $OUT/r06_bs_burst_sweep.txt
