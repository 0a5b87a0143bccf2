"""Phase timeline of the single-launch coarse levels (k_mg_coarse): LFA_MG_CO_STAMPS=1 makes workgroup 0 stamp every phase of
the last launch; lfa_mg_free prints them at handle destruction. usage: python tools/mg_coarse_stamps.py [C2|C4]"""
import os, sys
os.environ["LFA_MG_CO_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfluid_amd as lfa
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
size, block = {"C2": ((128, 128, 128), ((0, 0, 0), (32, 64, 128))), "C4": ((512, 512, 512), ((0, 0, 0), (128, 256, 256)))}[cfg]
s = lfa.Sim(size, method=lfa.APIC)
s.seed_block(*block)
for _ in range(3):
    print(cfg, s.step_hot(0.01), s.solver_stats())
s.close()
