#!/bin/bash
# tools/ab_lib.sh ALT.so [CONFIGS...] -- the library as built against a variant of it (a source compiled with another -D switch, linked
# with the other objects): stage medians of the full step, twice each, alternating.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
ALT=$1; shift
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
cp libfluid_amd/libfluid_amd.so /tmp/lfa_keep.so
for C in ${@:-C4}; do
  for V in base alt base alt; do
    if [ $V = alt ]; then cp $ALT libfluid_amd/libfluid_amd.so; else cp /tmp/lfa_keep.so libfluid_amd/libfluid_amd.so; fi
    python3 bench.py --config $C --steps 20 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/abl.json
    python3 - <<P
import json
o=json.load(open("/tmp/abl.json")); sm=o["stage_ms_median"]
print("$V $C ms/step %.3f serial %.3f" % (o["ms_per_step"], o.get("ms_per_step_serial_stages") or 0), {k: round(v, 3) for k, v in sm.items() if k in ("advect_collide", "bin", "p2g", "p2g_scatter_kernel", "build_system", "pcg_loop", "correct_collide", "g2p")})
P
  done
done
cp /tmp/lfa_keep.so libfluid_amd/libfluid_amd.so
