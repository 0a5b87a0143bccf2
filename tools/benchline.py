import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]/1e9,3), round(d["ms_per_step"],3), d["pcg"]["iterations_per_step"], {k: round(v,3) for k,v in d["stage_ms"].items()})
if "kernels" in d: print({k: round(v["ms"]*1e3,1) for k, v in d["kernels"].items()})
