import tempfile, numpy as np, sys
sys.path.insert(0,'.')
from tests import callers_util as cu, util
g=util.load_golden("ref_callers")
exe=cu.build_device(tempfile.mkdtemp()+"/own", False)
d=tempfile.mkdtemp()
for name in ("testbed_scene4","gridnode_source"):
    rec,out,_=cu.run(exe,name,d)
    print("==",name)
    print(out[-1500:])
    for k in sorted(rec):
        gk=g.get(name+"/"+k)
        if k.endswith((".pos",".points")):
            cnt = int(g[name+"/"+k.replace(".pos",".count")]) if (name+"/"+k.replace(".pos",".count")) in g and gk is None else (len(gk)//3 if gk is not None else None)
            print(k, len(rec[k])//3, "ref", cnt)
        elif k.endswith(".energy") or k in ("dts","iterations"):
            print(k, rec[k], "ref", gk)
