# round 6: where pcx_generic_step's time goes on its three bench fixtures (same box, interleaved)
V="base;logic:PCX_DEBUG=2;w1:PCX_GENERIC_WAVES=1;w2:PCX_GENERIC_WAVES=2;w4:PCX_GENERIC_WAVES=4;w1logic:PCX_GENERIC_WAVES=1,PCX_DEBUG=2;w2logic:PCX_GENERIC_WAVES=2,PCX_DEBUG=2;w4logic:PCX_GENERIC_WAVES=4,PCX_DEBUG=2"
python tools/env_sweep.py --game warehouse --fixture warehouse_L0 --batches 262144 --variants "$V;hand:!PCX_FORCE_GENERIC=0" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
for f in warehouse_L0 walkers_scroll_groups; do PCX_FORCE_GENERIC=1 PCX_DEBUG=8 python tools/env_sweep.py --fixture $f --cardinal-fields 2 --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -2; done
