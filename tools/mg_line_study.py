"""CPU model study (numpy): would y-LINE relaxation (zebra: columns along gravity's axis solved exactly inside an 8^3 tile, red-black over
(x + z)) cut the iteration count of the multigrid-preconditioned CG - above all late in a run, when the fluid is a thin floor sheet
whose stiff direction is the vertical one? Same model as tools/mg_hybrid_study.py (per-tile smoother, tile faces frozen during a
smoothing step, the shipped coarsening and transfers); systems dumped from the device by tools/dump_system.py.
  python tools/mg_line_study.py gpurun_out/system_C3_550.npz [point|line|both]       (or: python tools/mg_line_study.py dam 64)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mg_transfer_study as m
import mg_hybrid_study as h


class LLevel(h.HLevel):
    def line_half(self, x, xf, b, colour, omega, axis=1):
        """columns along `axis` of colour ((i + k) & 1) == colour solved exactly inside their tile, everything else frozen"""
        d = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
        xp, fp = m.pad(x), m.pad(xf)
        s = lambda a, dx, dy, dz: a[1 + dx:a.shape[0] - 1 + dx, 1 + dy:a.shape[1] - 1 + dy, 1 + dz:a.shape[2] - 1 + dz]
        rhs = b.astype(np.float64).copy()
        up = lo = None
        for c, sm, dd in zip(self.coup, self.same, d):
            along = dd[axis] != 0
            if along:
                inside = c * sm                      # coupling to the next / previous cell of the line inside the tile
                rhs += (c * (~sm)) * s(fp, *dd)      # across the tile face: frozen
                if dd[axis] > 0: up = inside.astype(np.float64)
                else: lo = inside.astype(np.float64)
            else:
                rhs += c * np.where(sm, s(xp, *dd), s(fp, *dd))
        diag = np.where(self.on, self.diag, 1.0).astype(np.float64)
        rhs = np.where(self.on, rhs, 0.0)
        up = np.where(self.on, up, 0.0); lo = np.where(self.on, lo, 0.0)
        # Thomas along `axis`
        mv = lambda a: np.moveaxis(a, axis, 0)
        D, R, U, L = mv(diag), mv(rhs), mv(up), mv(lo)
        n = D.shape[0]
        cp = np.zeros_like(D); dp = np.zeros_like(D)
        den = D[0]
        cp[0] = U[0] / den; dp[0] = R[0] / den
        for j in range(1, n):
            den = D[j] - L[j] * cp[j - 1]
            cp[j] = U[j] / den
            dp[j] = (R[j] + L[j] * dp[j - 1]) / den
        sol = np.zeros_like(D)
        sol[n - 1] = dp[n - 1]
        for j in range(n - 2, -1, -1):
            sol[j] = dp[j] + cp[j] * sol[j + 1]
        new = np.moveaxis(sol, 0, axis).astype(np.float32)
        i, j, k = np.indices(x.shape, sparse=True)
        other = [a for a in range(3) if a != axis]
        idx = [i, j, k]
        col = ((idx[other[0]] + idx[other[1]]) & 1) == colour
        mk = self.on & col
        x[mk] = x[mk] + omega * (new[mk] - x[mk])


class LMG(m.MG):
    def __init__(self, t, mode="line", omega=1.15, line_levels=99, sweeps=2, axis=1):
        self.lv = [LLevel(t)]
        while max(self.lv[-1].t.shape) > 8:
            self.lv.append(LLevel(m.coarsen(self.lv[-1].t)))
        self.tri = set(); self.mode = mode; self.om = omega; self.ll = line_levels; self.sw = sweeps; self.axis = axis

    def smooth(self, L, l, x, b, forward):
        xf = x.copy()
        line = self.mode == "line" and l < self.ll
        for _ in range(self.sw):
            for colour in ((0, 1) if forward else (1, 0)):
                if line: L.line_half(x, xf, b, colour, self.om, self.axis)
                else: L.half2(x, xf, b, colour, self.om)

    def vcycle(self, l, b):
        L = self.lv[l]
        x = np.zeros_like(b)
        if l == len(self.lv) - 1:
            return m.MG.vcycle(self, l, b)
        self.smooth(L, l, x, b, True)
        r = np.where(L.unk, b - L.apply(x), 0).astype(np.float32)
        cs = self.lv[l + 1].t.shape
        bc = np.where(self.lv[l + 1].unk, m.restrict_const(r, cs), 0).astype(np.float32)
        e = self.vcycle(l + 1, bc)
        x = np.where(L.unk, x + m.prolong_const(e, b.shape), x).astype(np.float32)
        self.smooth(L, l, x, b, False)
        return x


def pcg_abs(mg, b, tol_abs, maxit=200):
    """pressure_solver::solve's stopping rule: signed max of r below an absolute tolerance (src/pressure_solver.cpp:54)"""
    L = mg.lv[0]
    x = np.zeros_like(b); r = b.copy()
    z = mg.vcycle(0, r); p = z.copy()
    sigma = float(np.vdot(z.astype(np.float64), r.astype(np.float64)))
    for it in range(1, maxit + 1):
        q = L.apply(p)
        alpha = sigma / float(np.vdot(p.astype(np.float64), q.astype(np.float64)))
        x += np.float32(alpha) * p; r -= np.float32(alpha) * q
        if float(r.max()) < tol_abs:
            return it
        z = mg.vcycle(0, r)
        s2 = float(np.vdot(z.astype(np.float64), r.astype(np.float64)))
        p = z + np.float32(s2 / sigma) * p; sigma = s2
    return maxit


def load(path):
    d = np.load(path)
    nx, ny, nz = [int(v) for v in d["size"]]
    ty = d["types"].reshape(nz, ny, nx)              # reference layout: x fastest; cell::type air 1 / fluid 2 / solid 4
    fc = d["fluid_cells"].astype(np.int64)
    t = np.full((nz, ny, nx), m.AIR, dtype=np.uint8)
    t[ty == 4] = m.SOLID
    t.reshape(-1)[fc] = m.FLUID                      # the unknowns (cells that hold particles)
    b = np.zeros((nz, ny, nx), dtype=np.float32)
    b.reshape(-1)[fc] = d["b"]
    # model axes (x, y, z)
    t = np.ascontiguousarray(np.transpose(t, (2, 1, 0))); b = np.ascontiguousarray(np.transpose(b, (2, 1, 0)))
    # the model's operator is unscaled (A' = A / scale): A p = b <=> A' (scale p) = b, the residual is the same vector
    return t, b, int(d["device_iterations"])


if __name__ == "__main__":
    src = sys.argv[1]
    which = sys.argv[2] if len(sys.argv) > 2 and src != "dam" else "both"
    if src == "dam":
        t, b = m.dam(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
        tol, dev = 1e-6 * float(np.abs(b).max()), None
    else:
        t, b, dev = load(src)
        tol = 1e-6
    print(src, "unknowns", int((t == m.FLUID).sum()), "device iterations", dev, flush=True)
    runs = []
    if which in ("point", "both"): runs.append(("point RB-SOR 1.15 (shipped smoother)", dict(mode="point")))
    if which in ("line", "both"):
        runs += [("y-line zebra, omega 1.0", dict(mode="line", omega=1.0)), ("y-line zebra, omega 1.15", dict(mode="line", omega=1.15)),
                 ("y-line zebra on the finest level only, omega 1.0", dict(mode="line", omega=1.0, line_levels=1)),
                 ("y-line zebra, one sweep, omega 1.0", dict(mode="line", omega=1.0, sweeps=1))]
    for name, kw in runs:
        t0 = time.time()
        it = pcg_abs(LMG(t, **kw), b, tol)
        print(f"  {name}: {it} iterations ({time.time() - t0:.0f} s)", flush=True)
