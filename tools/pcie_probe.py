import sys, time
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes
parts = scenes.seed_block((0,0,0),(32,32,32))
s = lfa.Sim((64,64,64))
t=time.perf_counter(); s.upload_particles(parts); print("upload ms", 1e3*(time.perf_counter()-t))
s.hash()
out = parts.copy()
for k in range(3):
    t=time.perf_counter(); s.download_particles(into=out, write_positions=True); print("download ms", 1e3*(time.perf_counter()-t))
for k in range(2):
    t=time.perf_counter(); c = s.cells(); print("cells ms", 1e3*(time.perf_counter()-t), c.nbytes/1e6)
