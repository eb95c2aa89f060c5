export PCX_FORCE_GENERIC=1
for f in warehouse_L0; do
for d in 8 40 10; do
echo "== $f PCX_DEBUG=$d"; PCX_GENERIC_PW_LOGIC=6 PCX_GENERIC_PW_RENDER=2 PCX_DEBUG=$d python tools/env_sweep.py --fixture $f --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
done; done
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "old:PCX_GENERIC_PW=0;l6r2:PCX_GENERIC_PW_LOGIC=6,PCX_GENERIC_PW_RENDER=2;l6r2nowait:PCX_GENERIC_PW_LOGIC=6,PCX_GENERIC_PW_RENDER=2,PCX_DEBUG=32;l6r2logic:PCX_GENERIC_PW_LOGIC=6,PCX_GENERIC_PW_RENDER=2,PCX_DEBUG=2" 2>&1 | grep -v amdgpu.ids
