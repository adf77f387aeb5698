#!/usr/bin/env python3
"""What the GPU box offers besides HIP: every OpenCL platform / device with type, image support and version (can ANY real
OpenCL runtime run the reference's unmodified image kernels there? -- DESIGN.md section 5), and the CPU share of the job
(cgroup quota along the hierarchy, affinity mask).  Prints one JSON object; touches no GPU through HIP."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_share():
    out = {"cpu_count": os.cpu_count()}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except Exception as e:  # pragma: no cover
        out["affinity"] = repr(e)
    try:
        out["proc_self_cgroup"] = open("/proc/self/cgroup").read().strip().splitlines()
    except Exception as e:
        out["proc_self_cgroup"] = repr(e)
    quotas = {}
    for line in out.get("proc_self_cgroup", []) if isinstance(out.get("proc_self_cgroup"), list) else []:
        parts = line.split(":", 2)
        if len(parts) != 3:
            continue
        rel = parts[2]
        for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
            p = rel
            while True:
                for fn in ("cpu.max", "cpu.cfs_quota_us", "cpu.cfs_period_us", "cpuset.cpus.effective", "cpuset.cpus"):
                    path = os.path.join(base, p.lstrip("/"), fn)
                    if os.path.exists(path):
                        try:
                            quotas[path] = open(path).read().strip()
                        except Exception as e:
                            quotas[path] = repr(e)
                if p in ("", "/"):
                    break
                p = os.path.dirname(p)
    out["cgroup_files"] = quotas
    try:
        out["loadavg"] = open("/proc/loadavg").read().strip()
    except Exception:
        pass
    return out


if __name__ == "__main__":
    import bench  # the same probe functions the bench line uses
    print(json.dumps({"opencl": bench.opencl_devices(), "cpu": cpu_share(), "host_cores_used_by_bench": bench.host_cores()}, indent=1))
