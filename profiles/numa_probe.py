#!/usr/bin/env python3
"""Does the NUMA node this process runs on matter for the end-to-end rate (pack into page-locked memory + PCIe upload)?
Runs the rescale estimator's batch call pinned to each node's CPUs in turn (a fresh process per node).
    python profiles/numa_probe.py"""
import glob, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", [(os.path.basename(n), open(n + "/cpulist").read().strip()) for n in nodes])
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
child = r'''
import os, sys, time
sys.path.insert(0, %r)
cpus = set()
for part in sys.argv[1].split(","):
    a, _, b = part.partition("-"); cpus |= set(range(int(a), int(b or a) + 1))
os.sched_setaffinity(0, cpus)
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator
F, N = 32768, 2000
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(4096)]
f3 = [pool[i %% 4096][0] for i in range(F)]; f2 = [pool[i %% 4096][1] for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=1)
est.scale_calculation_batch(f3, f2)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); est.scale_calculation_batch(f3, f2); ts.append(time.perf_counter() - t0)
print("cpus %%s: %%s k frames/s" %% (sys.argv[1], " ".join("%%.0f" %% (F / t / 1e3) for t in ts)), flush=True)
''' % R
for rep in range(2):
    for n in nodes:
        cl = open(n + "/cpulist").read().strip()
        subprocess.run([sys.executable, "-c", child, cl], check=False)
subprocess.run([sys.executable, "-c", child, "0-%d" % (os.cpu_count() - 1)], check=False)
