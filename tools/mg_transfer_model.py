"""CPU model (numpy) of the multigrid-preconditioned CG of mg.hip on a dam-break-like geometry: how many iterations do the
piecewise-constant transfers (what ships: restriction = half the sum of the 8 children, prolongation = injection) cost against the
cell-centred tensor-product pair (restriction [1 3 3 1]/8 per axis x 4, prolongation = 8 x its transpose, i.e. trilinear)?
Same everything else: AIR-if-any-child-air coarsening, rediscretised 7-point operator, red-black Gauss-Seidel (omega 1.15, 2 sweeps
before and 2 after in reversed colour order), 2 + 2 sweeps on the coarsest level, stopping rule max|r| <= 1e-6 max|b|... of the
reference (pressure_solver.cpp:54-58 uses the same relative infinity norm). Not a product path: a design experiment for DESIGN 9.
usage: python tools/mg_transfer_model.py [N] [scene ...] [--tiled]      scene: column | sheet | obstacle;
--tiled: the values across the faces of 8^3 tiles stay frozen during the sweeps of a smoothing step, as in mg.hip"""
import sys
import numpy as np

AIR, FLUID, SOLID = 0, 1, 2


def build(types):
    """diag (non-solid neighbours) and the six 'fluid neighbour' masks of the unknowns (FLUID cells); outside the array = SOLID"""
    t = np.pad(types, 1, constant_values=SOLID)
    c = t[1:-1, 1:-1, 1:-1]
    unk = c == FLUID
    nb = [t[2:, 1:-1, 1:-1], t[:-2, 1:-1, 1:-1], t[1:-1, 2:, 1:-1], t[1:-1, :-2, 1:-1], t[1:-1, 1:-1, 2:], t[1:-1, 1:-1, :-2]]
    diag = sum((n != SOLID).astype(np.float64) for n in nb) * unk
    off = [((n == FLUID) & unk) for n in nb]
    return unk, diag, off


def neighbours_sum(x, off):
    xp = np.pad(x, 1)
    nb = [xp[2:, 1:-1, 1:-1], xp[:-2, 1:-1, 1:-1], xp[1:-1, 2:, 1:-1], xp[1:-1, :-2, 1:-1], xp[1:-1, 1:-1, 2:], xp[1:-1, 1:-1, :-2]]
    return sum(n * o for n, o in zip(nb, off))


def apply_A(x, lv):
    unk, diag, off = lv["A"]
    return diag * x - neighbours_sum(x, off)


def colour_masks(shape):
    i, j, k = np.indices(shape)
    red = ((i + j + k) & 1) == 0
    return red, ~red


TILE = 0  # 8: the values across the faces of 8^3 tiles stay frozen during the sweeps of one smoothing step (what mg.hip does)


def neighbours_sum_tiled(x, x_frozen, off):
    """as neighbours_sum, but a neighbour in another 8^3 tile contributes its value of before the smoothing step"""
    xp, fp = np.pad(x, 1), np.pad(x_frozen, 1)
    sl = [(slice(2, None), slice(1, -1), slice(1, -1)), (slice(None, -2), slice(1, -1), slice(1, -1)),
          (slice(1, -1), slice(2, None), slice(1, -1)), (slice(1, -1), slice(None, -2), slice(1, -1)),
          (slice(1, -1), slice(1, -1), slice(2, None)), (slice(1, -1), slice(1, -1), slice(None, -2))]
    idx = np.indices(x.shape)
    tot = 0.0
    for k, (s_, o) in enumerate(zip(sl, off)):
        ax, up = k // 2, k % 2 == 0
        pos = idx[ax] % TILE
        cross = (pos == TILE - 1) if up else (pos == 0)
        tot = tot + np.where(cross, fp[s_], xp[s_]) * o
    return tot


def smooth(x, b, lv, order, sweeps, omega=1.15):
    unk, diag, off = lv["A"]
    inv = np.where(diag > 0, 1.0 / np.maximum(diag, 1), 0.0)
    frozen = x.copy()
    for _ in range(sweeps):
        for col in order:
            m = lv["col"][col] & unk & (diag > 0)
            nsum = neighbours_sum_tiled(x, frozen, off) if TILE else neighbours_sum(x, off)
            gs = (b + nsum) * inv
            x = np.where(m, x + omega * (gs - x), x)
    return x


def coarsen_types(t):
    n = [(s + 1) // 2 * 2 for s in t.shape]
    tp = np.full(n, SOLID, dtype=t.dtype)
    tp[: t.shape[0], : t.shape[1], : t.shape[2]] = t
    blocks = tp.reshape(n[0] // 2, 2, n[1] // 2, 2, n[2] // 2, 2)
    any_air = (blocks == AIR).any(axis=(1, 3, 5))
    any_fluid = (blocks == FLUID).any(axis=(1, 3, 5))
    return np.where(any_air, AIR, np.where(any_fluid, FLUID, SOLID)).astype(t.dtype)


def restrict_const(r, shape_c):
    n = [2 * s for s in shape_c]
    rp = np.zeros(n)
    rp[: r.shape[0], : r.shape[1], : r.shape[2]] = r
    return 0.5 * rp.reshape(shape_c[0], 2, shape_c[1], 2, shape_c[2], 2).sum(axis=(1, 3, 5))


def prolong_const(e, shape_f):
    return np.repeat(np.repeat(np.repeat(e, 2, 0), 2, 1), 2, 2)[: shape_f[0], : shape_f[1], : shape_f[2]]


def axis_weights(nf, nc):
    """P (nf x nc): cell-centred linear interpolation, fine cell 2I, 2I+1 from coarse I (3/4) and I-1 / I+1 (1/4)"""
    P = np.zeros((nf, nc))
    for f in range(nf):
        I = f // 2
        o = I - 1 if f % 2 == 0 else I + 1
        P[f, I] += 0.75
        if 0 <= o < nc:
            P[f, o] += 0.25
    return P


def prolong_lin(e, shape_f, Ps):
    x = np.einsum("ai,ijk->ajk", Ps[0], e)
    x = np.einsum("bj,ajk->abk", Ps[1], x)
    return np.einsum("ck,abk->abc", Ps[2], x)


def restrict_lin(r, Ps):
    # 4 x Rbar with Rbar = P^T / 8  =>  0.5 P^T
    x = np.einsum("ai,ajk->ijk", Ps[0], r)
    x = np.einsum("bj,ibk->ijk", Ps[1], x)
    return 0.5 * np.einsum("ck,ijc->ijk", Ps[2], x)


def hierarchy(types):
    lv = []
    t = types
    while True:
        L = dict(types=t, A=build(t), col=colour_masks(t.shape))
        lv.append(L)
        if max(t.shape) <= 2 or not (t == FLUID).any():
            break
        tc = coarsen_types(t)
        L["P"] = [axis_weights(t.shape[d], tc.shape[d]) for d in range(3)]
        t = tc
    return lv


def vcycle(b, lv, l, linear):
    L = lv[l]
    unk = L["A"][0]
    x = np.zeros_like(b)
    if l == len(lv) - 1 or not (lv[l + 1]["types"] == FLUID).any():
        x = smooth(x, b, L, (0, 1), 2)
        return smooth(x, b, L, (1, 0), 2)
    x = smooth(x, b, L, (0, 1), 2)
    r = (b - apply_A(x, L)) * unk
    shape_c = lv[l + 1]["types"].shape
    unk_c = lv[l + 1]["A"][0]
    if linear:
        bc = restrict_lin(r, L["P"]) * unk_c
    else:
        bc = restrict_const(r, shape_c) * unk_c
    ec = vcycle(bc, lv, l + 1, linear) * unk_c
    e = prolong_lin(ec, b.shape, L["P"]) if linear else prolong_const(ec, b.shape)
    x = x + e * unk
    return smooth(x, b, L, (1, 0), 2)


def pcg(b, lv, linear, tol=1e-6, maxit=200):
    unk = lv[0]["A"][0]
    x = np.zeros_like(b)
    r = b.copy()
    z = vcycle(r, lv, 0, linear)
    s = z.copy()
    sigma = (z * r).sum()
    b0 = np.abs(b).max()
    for it in range(1, maxit + 1):
        q = apply_A(s, lv[0]) * unk
        alpha = sigma / (s * q).sum()
        x += alpha * s
        r -= alpha * q
        if np.abs(r).max() <= tol * b0:
            return it, x
        z = vcycle(r, lv, 0, linear)
        sn = (z * r).sum()
        s = z + (sn / sigma) * s
        sigma = sn
    return maxit, x


def scene(n, kind):
    t = np.full((n, n, n), AIR, dtype=np.int8)
    if kind == "column":      # the dam of the benchmark: a quarter x half x half block in a corner of the tank
        t[: n // 4, : n // 2, : n // 2] = FLUID
    elif kind == "sheet":     # late in the run: a shallow layer over the floor with a bump against the far wall
        t[:, : max(2, n // 16), :] = FLUID
        t[3 * n // 4 :, : n // 4, :] = FLUID
    elif kind == "obstacle":  # the column with a solid box standing in it
        t[: n // 4, : n // 2, : n // 2] = FLUID
        t[n // 16 : n // 8, : n // 4, n // 8 : n // 4] = SOLID
    return t


def main():
    global TILE
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--tiled" in sys.argv:
        TILE = 8
    n = int(args[0]) if args else 64
    kinds = args[1:] or ["column", "sheet", "obstacle"]
    rng = np.random.default_rng(3)
    for kind in kinds:
        t = scene(n, kind)
        lv = hierarchy(t)
        unk = lv[0]["A"][0]
        # right-hand side: the divergence of a falling column (constant) plus noise, like gravity * dt on a column at rest
        b = (1.0 + 0.2 * rng.standard_normal(t.shape)) * unk
        out = {}
        for linear in (False, True):
            it, x = pcg(b.copy(), lv, linear)
            res = np.abs((b - apply_A(x, lv[0])) * unk).max() / np.abs(b).max()
            out["linear" if linear else "const"] = (it, res)
        print(("tile-frozen smoother " if TILE else "global smoother      ") + f"{kind:9s} {n}^3  unknowns {int(unk.sum()):8d}  levels {len(lv)}  piecewise constant: {out['const'][0]:3d} iterations"
              f"   [1 3 3 1]/8 + trilinear: {out['linear'][0]:3d} iterations   (residuals {out['const'][1]:.1e} / {out['linear'][1]:.1e})")


if __name__ == "__main__":
    main()
