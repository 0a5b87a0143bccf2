"""Probes on the GPU box: how long does loading librccl take; does RCCL accept two ranks of one communicator on ONE device?"""
import ctypes, os, sys, threading, time
sys.path.insert(0, ".")
t = time.perf_counter()
for name in ("/opt/rocm/lib/librccl.so.1",):
    ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
    print("dlopen", name, round(time.perf_counter() - t, 2), "s", flush=True)
import numpy as np
import libfluid_amd as lfa
t = time.perf_counter()
uid = lfa.rccl_unique_id()
print("unique id", round(time.perf_counter() - t, 2), "s", flush=True)
size, block = (16, 16, 32), ((2, 0, 3), (14, 10, 29))
res = {}
def run(rank):
    try:
        s = lfa.Sim(size)
        s.init_rccl_slab(rank, 2, uid, [0, 2, 4])
        s.seed_block(*block)
        res[rank] = s.step_hot(0.01)
        s.close()
    except Exception as e:
        res[rank] = repr(e)
os.environ.setdefault("NCCL_DEBUG", "WARN")
th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
t = time.perf_counter()
[x.start() for x in th]
[x.join(120) for x in th]
print("two ranks on one device:", res, round(time.perf_counter() - t, 2), "s", flush=True)
