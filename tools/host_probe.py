#!/usr/bin/env python3
"""What the GPU box gives a process: CPUs visible / allowed / cgroup quota, NUMA nodes, memory."""
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/memory.max"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e.__class__.__name__)
try:
    print("nodes", sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")))
except OSError:
    pass
os.system("nproc; grep -c processor /proc/cpuinfo; cat /proc/loadavg")
