#!/bin/bash
# generic kernel with the LDS-DMA state load: probe above 64 KB, parity (whole GPU suite through the table-driven kernel), timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_generic; mkdir -p $OUT
cd $ROOT
tools/bin/ldsdma_probe > $OUT/ldsdma_probe.txt 2>&1; echo "probe rc=$?" >> $OUT/ldsdma_probe.txt
cat $OUT/ldsdma_probe.txt
PCX_FORCE_GENERIC=1 timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/suite_generic.txt 2>&1; echo "suite rc=$?"
tail -5 $OUT/suite_generic.txt
python tools/generic_timing.py > $OUT/after_dma.txt 2>&1
cat $OUT/after_dma.txt
for fx in warehouse_L0:262144 marauders_custom_A:32768 walkers_scroll_groups:262144; do
  PCX_DEBUG=8 python tools/generic_timing.py $fx 2>&1 | tail -2
done
