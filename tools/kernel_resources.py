"""VGPRs / SGPRs / spills / LDS of the kernels of one source file, from the compiler's own metadata (CPU container: hipcc cross-compiles).
usage: python tools/kernel_resources.py particles.hip [name filter] [-DXYZ ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libfluid_amd import build as B  # noqa: E402

src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    subprocess.run([B._hipcc(), *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, src), "-o", os.path.join(d, "x.o"), "--save-temps=obj"],
                   check=True, capture_output=True)
    asm = open(next(os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s"))).read()
for blk in asm.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
    name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    if flt in name:
        print(f"{name[5:name.index('(')] if name.startswith('void ') else name[:90]:60s} vgpr {get('vgpr_count'):>4} sgpr {get('sgpr_count'):>4} spill {get('vgpr_spill_count'):>3} "
              f"lds {get('group_segment_fixed_size'):>7} scratch {get('private_segment_fixed_size'):>5}")
