import os, sys; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mg_transfer_study as m

class HLevel(m.Level):
    """per-tile smoother: neighbours across an 8^3 tile face are FROZEN at their values from before the smoothing step"""
    def __init__(self,t):
        super().__init__(t)
        i,j,k=np.indices(t.shape)
        # direction order of nbr/coup: +x,-x,+y,-y,+z,-z
        self.same=[(i%8!=7),(i%8!=0),(j%8!=7),(j%8!=0),(k%8!=7),(k%8!=0)]
    def nsum2(self,x,xf):
        xp=m.pad(x); fp=m.pad(xf)
        s=lambda a,dx,dy,dz: a[1+dx:a.shape[0]-1+dx,1+dy:a.shape[1]-1+dy,1+dz:a.shape[2]-1+dz]
        d=[(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)]
        return sum(c*np.where(sm,s(xp,*dd),s(fp,*dd)) for c,sm,dd in zip(self.coup,self.same,d))
    def half2(self,x,xf,b,colour,omega):
        mk=self.on&(self.red==(colour==0))
        new=(b+self.nsum2(x,xf))*self.inv
        x[mk]=x[mk]+omega*(new[mk]-x[mk])

class HMG(m.MG):
    def __init__(self,t,omegas=(1.15,1.15),hyb_levels=99):
        self.lv=[HLevel(t)]
        while max(self.lv[-1].t.shape)>8: self.lv.append(HLevel(m.coarsen(self.lv[-1].t)))
        self.tri=set(); self.om=omegas; self.hl=hyb_levels
    def vcycle(self,l,b):
        L=self.lv[l]; x=np.zeros_like(b)
        if l==len(self.lv)-1: return m.MG.vcycle(self,l,b)
        xf=x.copy()
        for k in range(2): L.half2(x,xf,b,0,self.om[k]); L.half2(x,xf,b,1,self.om[k])
        r=np.where(L.unk,b-L.apply(x),0).astype(np.float32)
        cs=self.lv[l+1].t.shape
        bc=np.where(self.lv[l+1].unk,m.restrict_const(r,cs),0).astype(np.float32)
        e=self.vcycle(l+1,bc)
        x=np.where(L.unk,x+m.prolong_const(e,b.shape),x).astype(np.float32)
        xf=x.copy()
        for k in (1,0): L.half2(x,xf,b,1,self.om[k]); L.half2(x,xf,b,0,self.om[k])
        return x

if __name__=="__main__":
    n=int(sys.argv[1]) if len(sys.argv)>1 else 64
    t,b=m.dam(n)
    print("global RB:", m.pcg(m.MG(t,set()),b))
    for om in ((1.15,1.15),(1.0,1.0),(1.0,1.3),(1.3,1.0),(1.2,1.1),(1.1,1.2),(1.25,1.25),(1.0,1.5),(0.9,1.4)):
        print("tile hybrid omega",om, m.pcg(HMG(t,om),b),flush=True)
