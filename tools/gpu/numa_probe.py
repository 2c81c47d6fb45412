"""Which NUMA node is the GPU on, which CPUs may this process use, and what does the step cost when the launching thread is
pinned to the GPU's node vs the other one?  usage: python tools/gpu/numa_probe.py"""
import glob, os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    cpus = json.loads(sys.argv[2])
    if cpus:
        os.sched_setaffinity(0, cpus)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-ops", "--no-roofline"],
                       capture_output=True, text=True)
    print(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"])
    sys.exit(0)

import torch
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip() if os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bdf) else "?"
print("gpu", bdf, "numa_node", node, "allowed cpus", len(os.sched_getaffinity(0)))
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("nodes", {k: (v[0], v[-1], len(v)) for k, v in nodes.items()})
allowed = os.sched_getaffinity(0)
for k, v in nodes.items():
    use = sorted(set(v) & allowed)
    if not use:
        continue
    for rep in range(2):
        r = subprocess.run([sys.executable, __file__, "--child", json.dumps(use)], capture_output=True, text=True)
        print("pinned to node", k, "->", r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else "")
for rep in range(3):
    r = subprocess.run([sys.executable, __file__, "--child", "[]"], capture_output=True, text=True)
    print("unpinned ->", r.stdout.strip())
