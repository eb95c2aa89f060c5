#!/bin/bash
# Round 6: the evidence committed under profiles/r06_* from ONE box (run on the GPU box through gpurun):
# the sweeps behind profiles/r06_tuning.md / r06_generic.md, bench.py's default line and the driver's flags, and the per-row
# rocprofv3 passes (tools/profile_r06.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
{
echo "# pcx_warehouse_step: mask loop / owner codes and worker shapes (tools/env_sweep.py, interleaved on one engine)"
python tools/env_sweep.py --game warehouse --batches 131072,262144,1048576 --variants "mask:PCX_WM_CODES=0;codes;mask_pw0:PCX_WM_CODES=0,PCX_WM_PW=0;codes_pw0:PCX_WM_PW=0;w4k1:PCX_WM_WORKERS=4,PCX_WM_LOCK=1;w4k2:PCX_WM_WORKERS=4,PCX_WM_LOCK=2;w6k2:PCX_WM_WORKERS=6,PCX_WM_LOCK=2;w6k3:PCX_WM_WORKERS=6,PCX_WM_LOCK=3;w8k3:PCX_WM_WORKERS=8,PCX_WM_LOCK=3;w2k2x2:PCX_WM_WORKERS=2,PCX_WM_LOCK=2,PCX_WM_PER_CU=2;w1k0x4:PCX_WM_WORKERS=1,PCX_WM_LOCK=0,PCX_WM_PER_CU=4;mask_w8k4:PCX_WM_CODES=0,PCX_WM_WORKERS=8,PCX_WM_LOCK=4;mask_w4k1:PCX_WM_CODES=0,PCX_WM_WORKERS=4,PCX_WM_LOCK=1" 2>&1 | $Q
} > $OUT/r06_warehouse_codes_sweep.txt
{
echo "# pcx_hello_world_step: mask loop / owner codes (nibbles) and worker shapes"
python tools/env_sweep.py --game hello_world --batches 262144,1048576 --steps 30 --variants "mask:PCX_HW_CODES=0;codes;w1k0x4:PCX_HW_WORKERS=1,PCX_HW_PER_CU=4,PCX_HW_LOCK=0;w2k1x2:PCX_HW_WORKERS=2,PCX_HW_PER_CU=2,PCX_HW_LOCK=1;w2k1x3:PCX_HW_WORKERS=2,PCX_HW_PER_CU=3,PCX_HW_LOCK=1;w1k0x3:PCX_HW_WORKERS=1,PCX_HW_PER_CU=3,PCX_HW_LOCK=0;pw0:PCX_HW_PW=0;pw0mask:PCX_HW_PW=0,PCX_HW_CODES=0" 2>&1 | $Q
echo "# pcx_better_scrolly_step: workgroups per CU, two waves per group"
python tools/env_sweep.py --game better_scrolly_maze --batches 65536,262144 --steps 30 --variants "default;w8:PCX_WAVES_PER_CU=8;w4:PCX_WAVES_PER_CU=4;pair:PCX_BS_WAVES=2;coop:PCX_COOP_BELOW=100;logic:PCX_DEBUG=2" 2>&1 | $Q
echo "# pcx_marauders_step: workgroups per CU"
python tools/env_sweep.py --game marauders --batches 262144 --steps 30 --variants "default;w3:PCX_WAVES_PER_CU=3;w5:PCX_WAVES_PER_CU=5;w6:PCX_WAVES_PER_CU=6;w8:PCX_WAVES_PER_CU=8" 2>&1 | $Q
echo "# pcx_scrolly_maze_step, the headline: worker shapes (workers per workgroup, workgroups per CU, streaming slots)"
python tools/env_sweep.py --game scrolly_maze --batches 1048576 --steps 40 --variants "auto;w4p1k2:PCX_SM_WAVES=4,PCX_SM_PER_CU=1,PCX_SM_LOCK=2;w2p3k1:PCX_SM_WAVES=2,PCX_SM_PER_CU=3,PCX_SM_LOCK=1;w6p1k3:PCX_SM_WAVES=6,PCX_SM_PER_CU=1,PCX_SM_LOCK=3;w4p2k2:PCX_SM_WAVES=4,PCX_SM_PER_CU=2,PCX_SM_LOCK=2;w3p1k2:PCX_SM_WAVES=3,PCX_SM_PER_CU=1,PCX_SM_LOCK=2;w5p1k2:PCX_SM_WAVES=5,PCX_SM_PER_CU=1,PCX_SM_LOCK=2;w6p1k2:PCX_SM_WAVES=6,PCX_SM_PER_CU=1,PCX_SM_LOCK=2;w4p1k3:PCX_SM_WAVES=4,PCX_SM_PER_CU=1,PCX_SM_LOCK=3;w8p1k2:PCX_SM_WAVES=8,PCX_SM_PER_CU=1,PCX_SM_LOCK=2" 2>&1 | $Q
} > $OUT/r06_stream_kernels_sweeps.txt
{
echo "# pcx_generic_step (specialised build): what the engine settles on (auto: the tuner picks waves per workgroup and the render loop),"
echo "# owner codes / masks forced, the round-5 build (no sprite registers, masks), the logic phase alone (PCX_DEBUG=2), pcx_generic_step_pw"
export PCX_FORCE_GENERIC=1
V="auto;codes:PCX_GENERIC_CODES=1;masks:PCX_GENERIC_CODES=0;r5_build:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_NO_SPRITE_REGS,PCX_GENERIC_CODES=0;logic:PCX_DEBUG=2;pw_l6r2:!PCX_GENERIC_PW=1,PCX_GENERIC_PW_LOGIC=6,PCX_GENERIC_PW_RENDER=2;w1:PCX_GENERIC_WAVES=1;w2:PCX_GENERIC_WAVES=2;w4:PCX_GENERIC_WAVES=4"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "auto;codes:PCX_GENERIC_CODES=1;masks:PCX_GENERIC_CODES=0" 2>&1 | $Q
echo "# phase timers (PCX_DEBUG=8; +2: logic only), cycles per group of 64 environments"
for f in warehouse_L0 walkers_scroll_groups marauders_custom_A; do
  cf=0; [ $f = walkers_scroll_groups ] && cf=2
  for defs in "" "-DPCX_X_NO_SPRITE_REGS"; do
    echo "== $f PCX_GENERIC_SPEC_DEFS='$defs'"; PCX_GENERIC_SPEC_DEFS="$defs" PCX_DEBUG=8 python tools/env_sweep.py --fixture $f --cardinal-fields $cf --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
  done
done
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_sweeps.txt
python bench.py > $OUT/r06_bench_n1.json 2> $OUT/r06_bench_n1.err
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_n1_driver_flags.json 2> $OUT/r06_bench_n1_driver_flags.err
bash tools/profile_r06.sh > $OUT/r06_profile_log.txt 2>&1
tail -c 300 $OUT/r06_bench_n1.json; echo; tail -5 $OUT/r06_generic_sweeps.txt
