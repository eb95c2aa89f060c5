#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
PCX_EPI_TWO_PASS=0 python tools/fusion_bench.py epi 2>&1 | grep -v amdgpu.ids
PCX_EPI_TWO_PASS=1 python tools/fusion_bench.py epi 2>&1 | grep -v amdgpu.ids

