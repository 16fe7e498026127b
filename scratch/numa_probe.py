"""Does the NUMA node the process runs on decide the witness upload?  Pins the whole process (before the library creates any thread or pinned
buffer) to the CPUs of the GPU's own node / of another node / not at all and runs the three entry points (scratch/paths_loop.py).
usage: numa_probe.py local|remote|none"""
import ctypes as C, glob, os, subprocess, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
hip = C.CDLL("libamdhip64.so")
buf = C.create_string_buffer(64)
assert hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0
bdf = buf.value.decode().lower()
node = -1
try:
    node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
except OSError:
    pass
def cpus_of(n):
    s = open(f"/sys/devices/system/node/node{n}/cpulist").read().strip()
    out = set()
    for part in s.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print(f"GPU 0 at {bdf}: NUMA node {node}; nodes {nodes}; mode {mode}", flush=True)
if mode != "none" and node >= 0 and len(nodes) > 1:
    target = node if mode == "local" else next(n for n in nodes if n != node)
    os.sched_setaffinity(0, cpus_of(target) & os.sched_getaffinity(0))
    print(f"pinned to node {target}: {len(os.sched_getaffinity(0))} CPUs", flush=True)
sys.argv = ["paths_loop.py", "30"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "paths_loop.py")).read())
