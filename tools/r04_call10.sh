cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
LFA_DEBUG_ABORT=1 python3 - <<'P'
import sys
sys.path.insert(0, ".")
import libfluid_amd as lfa
for size, block in (((64,64,64), ((0,0,0),(32,40,32))), ((128,128,128), ((0,0,0),(64,64,64)))):
    s = lfa.Sim(size)
    s.seed_block(*block)
    for k in range(3):
        r, it, rc = s.step_hot(0.005)
        print(size, k, it, rc, s.solver_stats())
    s.close()
P
