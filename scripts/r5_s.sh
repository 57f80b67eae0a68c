#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
cat /proc/self/cgroup | head -3; ls /sys/fs/cgroup | head -30
nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
