#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2s; mkdir -p $O
(echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; echo "loadavg $(cat /proc/loadavg)"; python -c "import sys; sys.path.insert(0,'.'); from tests import oracle_lib; print('oracle threads', oracle_lib.load().lib.oracle_num_threads())") 2>&1 | tee $O/host.log
(time timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -24) > $O/pytest_all.log 2>&1
tail -26 $O/pytest_all.log
