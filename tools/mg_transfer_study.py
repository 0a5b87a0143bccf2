"""CPU model of the solver's multigrid-preconditioned CG (dense numpy grids, no tiles) to compare TRANSFER OPERATORS:
piecewise-constant prolongation / restriction with the factor 0.5 (what the library ships) against cell-centred trilinear
P and R = 0.5 P^T (McAdams et al. 2010), on every level or between the two finest levels only. Same operator (unscaled 7-point
Laplacian: diagonal = non-solid neighbours, -1 to fluid neighbours), same coarsening of the cell types (air if any child is air,
else fluid if any is fluid, else solid), red-black SOR (omega 1.15), V(1,1) with two sweeps, stopping rule max |r| < tol * max |b|.
The smoother here is global red-black (the library's is per tile with frozen faces): iteration counts are a few lower than the
library's, the COMPARISON between transfers is what this is for.   python tools/mg_transfer_study.py [n ...]"""
import sys
import numpy as np

SOLID, FLUID, AIR = 0, 1, 2
OMEGA, SWEEPS = 1.15, 2


def pad(a, v=0):
    return np.pad(a, 1, constant_values=v)


class Level:
    def __init__(self, t):
        self.t = t
        tp = pad(t, SOLID)
        c = (slice(1, -1),) * 3
        sh = lambda dx, dy, dz: tp[1 + dx:tp.shape[0] - 1 + dx, 1 + dy:tp.shape[1] - 1 + dy, 1 + dz:tp.shape[2] - 1 + dz]
        self.unk = t == FLUID
        self.nbr = [sh(1, 0, 0), sh(-1, 0, 0), sh(0, 1, 0), sh(0, -1, 0), sh(0, 0, 1), sh(0, 0, -1)]
        self.diag = sum((n != SOLID).astype(np.float32) for n in self.nbr)
        self.coup = [((n == FLUID) & self.unk).astype(np.float32) for n in self.nbr]
        self.on = self.unk & (self.diag > 0)
        self.inv = np.where(self.on, 1.0 / np.maximum(self.diag, 1), 0).astype(np.float32)
        i, j, k = np.indices(t.shape)
        self.red = ((i + j + k) & 1) == 0

    def nsum(self, x):
        xp = pad(x)
        s = lambda dx, dy, dz: xp[1 + dx:xp.shape[0] - 1 + dx, 1 + dy:xp.shape[1] - 1 + dy, 1 + dz:xp.shape[2] - 1 + dz]
        sh = [s(1, 0, 0), s(-1, 0, 0), s(0, 1, 0), s(0, -1, 0), s(0, 0, 1), s(0, 0, -1)]
        return sum(c * v for c, v in zip(self.coup, sh))

    def apply(self, x):
        return np.where(self.unk, self.diag * x - self.nsum(x), 0).astype(np.float32)

    def half(self, x, b, colour):
        m = self.on & (self.red == (colour == 0))
        new = (b + self.nsum(x)) * self.inv
        x[m] = x[m] + OMEGA * (new[m] - x[m])


def coarsen(t):
    n = [(s + 1) // 2 * 2 for s in t.shape]
    tp = np.full(n, SOLID, dtype=t.dtype)
    tp[:t.shape[0], :t.shape[1], :t.shape[2]] = t
    blocks = tp.reshape(n[0] // 2, 2, n[1] // 2, 2, n[2] // 2, 2)
    any_air = (blocks == AIR).any(axis=(1, 3, 5))
    any_fluid = (blocks == FLUID).any(axis=(1, 3, 5))
    return np.where(any_air, AIR, np.where(any_fluid, FLUID, SOLID)).astype(t.dtype)


def restrict_const(r, cshape):
    n = [2 * s for s in cshape]
    rp = np.zeros(n, dtype=r.dtype)
    rp[:r.shape[0], :r.shape[1], :r.shape[2]] = r
    return 0.5 * rp.reshape(cshape[0], 2, cshape[1], 2, cshape[2], 2).sum(axis=(1, 3, 5))


def prolong_const(e, fshape):
    return np.repeat(np.repeat(np.repeat(e, 2, 0), 2, 1), 2, 2)[:fshape[0], :fshape[1], :fshape[2]]


def prolong_tri_axis(e, axis, nf):
    """fine cell 2i -> 3/4 e[i] + 1/4 e[i-1], fine cell 2i+1 -> 3/4 e[i] + 1/4 e[i+1] (zero outside)"""
    ep = np.pad(e, [(1, 1) if a == axis else (0, 0) for a in range(3)])
    sl = lambda lo, hi: tuple(slice(lo, hi) if a == axis else slice(None) for a in range(3))
    n = e.shape[axis]
    even = 0.75 * ep[sl(1, n + 1)] + 0.25 * ep[sl(0, n)]
    odd = 0.75 * ep[sl(1, n + 1)] + 0.25 * ep[sl(2, n + 2)]
    out = np.stack([even, odd], axis=axis + 1)
    shp = list(e.shape)
    shp[axis] = 2 * n
    out = out.reshape(shp)
    return out[sl(0, nf)]


def prolong_tri(e, fshape):
    x = e
    for a in range(3):
        x = prolong_tri_axis(x, a, fshape[a])
    return x


def restrict_tri_axis(r, axis, nc):
    """transpose of prolong_tri_axis"""
    n = 2 * nc
    rp = np.zeros([n if a == axis else s for a, s in enumerate(r.shape)], dtype=r.dtype)
    sl = lambda lo, hi, st=1: tuple(slice(lo, hi, st) if a == axis else slice(None) for a in range(3))
    rp[sl(0, r.shape[axis])] = r
    even, odd = rp[sl(0, n, 2)], rp[sl(1, n, 2)]
    out = 0.75 * (even + odd)
    o = np.zeros_like(out)
    o[sl(0, nc - 1)] += 0.25 * even[sl(1, nc)]   # even fine cell 2(i+1) gives 1/4 to coarse i
    o[sl(1, nc)] += 0.25 * odd[sl(0, nc - 1)]    # odd fine cell 2(i-1)+1 gives 1/4 to coarse i
    return out + o


def restrict_tri(r, cshape):
    x = r
    for a in range(3):
        x = restrict_tri_axis(x, a, cshape[a])
    return 0.5 * x


class MG:
    def __init__(self, t, tri_levels):
        self.lv = [Level(t)]
        while max(self.lv[-1].t.shape) > 8:
            self.lv.append(Level(coarsen(self.lv[-1].t)))
        self.tri = tri_levels  # set of fine-level indices l whose transfer l <-> l + 1 is trilinear

    def vcycle(self, l, b):
        L = self.lv[l]
        x = np.zeros_like(b)
        if l == len(self.lv) - 1:
            for _ in range(2):
                L.half(x, b, 0); L.half(x, b, 1)
            for _ in range(2):
                L.half(x, b, 1); L.half(x, b, 0)
            return x
        for _ in range(SWEEPS):
            L.half(x, b, 0); L.half(x, b, 1)
        r = np.where(L.unk, b - L.apply(x), 0).astype(np.float32)
        cs = self.lv[l + 1].t.shape
        bc = restrict_tri(r, cs) if l in self.tri else restrict_const(r, cs)
        bc = np.where(self.lv[l + 1].unk, bc, 0).astype(np.float32)
        e = self.vcycle(l + 1, bc)
        pe = prolong_tri(e, b.shape) if l in self.tri else prolong_const(e, b.shape)
        x = np.where(L.unk, x + pe, x).astype(np.float32)
        for _ in range(SWEEPS):
            L.half(x, b, 1); L.half(x, b, 0)
        return x


def pcg(mg, b, tol=1e-6, maxit=200):
    L = mg.lv[0]
    x = np.zeros_like(b)
    r = b.copy()
    z = mg.vcycle(0, r)
    p = z.copy()
    sigma = float(np.vdot(z.astype(np.float64), r.astype(np.float64)))
    bmax = float(np.abs(b).max())
    for it in range(1, maxit + 1):
        q = L.apply(p)
        alpha = sigma / float(np.vdot(p.astype(np.float64), q.astype(np.float64)))
        x += np.float32(alpha) * p
        r -= np.float32(alpha) * q
        if float(np.abs(r).max()) < tol * bmax:
            return it
        z = mg.vcycle(0, r)
        s2 = float(np.vdot(z.astype(np.float64), r.astype(np.float64)))
        p = z + np.float32(s2 / sigma) * p
        sigma = s2
    return maxit


def dam(n):
    t = np.full((n, n, n), AIR, dtype=np.uint8)
    # a dam that has started to collapse: block + a wedge in front of it, floor film
    t[:n // 4, :n // 2, :n // 2] = FLUID
    for i in range(n // 4, n // 2):
        h = max(2, int((n // 2) * (1.0 - (i - n // 4) / (n // 4))))
        t[i, :h, :n // 2] = FLUID
    rng = np.random.default_rng(1)
    b = np.where(t == FLUID, rng.standard_normal(t.shape), 0).astype(np.float32)
    # smooth-ish right-hand side like a divergence field: gravity term on the columns
    b += np.where(t == FLUID, 0.2, 0).astype(np.float32)
    return t, b


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128]
    for n in sizes:
        t, b = dam(n)
        nl = 0
        s = n
        while s > 8:
            s = (s + 1) // 2
            nl += 1
        res = {}
        for name, tri in (("constant (shipped)", set()), ("trilinear 0<->1 only", {0}), ("trilinear every level", set(range(nl)))):
            res[name] = pcg(MG(t, tri), b)
        print(n, "^3:", int((t == FLUID).sum()), "unknowns;", res, flush=True)
