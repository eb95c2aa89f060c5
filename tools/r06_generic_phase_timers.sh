export PCX_FORCE_GENERIC=1
for f in warehouse_L0 walkers_scroll_groups; do
for v in "PCX_GENERIC_PW=0" "PCX_GENERIC_PW_LOGIC=6 PCX_GENERIC_PW_RENDER=2" ; do
for d in 8 10; do
echo "== $f $v PCX_DEBUG=$d"; env $v PCX_DEBUG=$d python tools/env_sweep.py --fixture $f --cardinal-fields 2 --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic\|262144" | tail -2
done; done; done
