#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2i; mkdir -p $O
python tools/dbg_lde.py 22 94 2>&1 | grep -v amdgpu.ids | tee $O/dbg_lde.log
