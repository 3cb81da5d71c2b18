"""What the GPU box's host gives this process: online CPUs, affinity mask, cgroup CPU quota, memory."""
import os
print("os.cpu_count()", os.cpu_count())
print("sched_getaffinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
print(open("/proc/meminfo").read().splitlines()[0])
import subprocess
print(subprocess.run(["bash", "-c", "lscpu | head -20"], capture_output=True, text=True).stdout)
