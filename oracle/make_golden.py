"""Generate tests/golden/*.npz — seeded inputs and expected outputs for the hot path.

The reference holds no golden vectors and Julia is unavailable, so these are
produced by oracle/covgram_oracle.py and, for every small case, cross-checked here against an
INDEPENDENT 50-digit mpmath evaluation of the definitions the reference's source states (per pair:
direct-difference r², the closed-form profile, the block formula of src/gradient.jl:86-92/109-115).
The script refuses to write a file whose oracle values disagree with mpmath by more than 1e-13 (rel).

    python oracle/make_golden.py          # rewrites tests/golden/
"""
from __future__ import annotations

import math
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import covgram_oracle as o  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
mp.mp.dps = 50

KERNELS = {
    "EQ": o.Kernel(o.EQ),
    "Exp": o.Kernel(o.EXP),
    "RQ1": o.Kernel(o.RQ, param=1.0),
    "Cauchy": o.Kernel(o.CAUCHY),
    "IMQ08": o.Kernel(o.IMQ, param=0.8),
    "gExp15": o.Kernel(o.GAMMAEXP, param=1.5),
    "MaternP0": o.Kernel(o.MATERNP, p=0),
    "MaternP1": o.Kernel(o.MATERNP, p=1),
    "MaternP2": o.Kernel(o.MATERNP, p=2),
    "MaternP3": o.Kernel(o.MATERNP, p=3),
    "EQ_l07": o.Kernel(o.EQ, lengthscale=0.7),
    "Dot3": o.Kernel(o.DOT, power=3),
    "ExpDot": o.Kernel(o.EXPDOT),
}
GRAD_KERNELS = ["EQ", "RQ1", "MaternP2", "MaternP3", "Dot3", "Cauchy", "EQ_l07", "ExpDot"]


def kernel_fields(k):
    return np.array([k.family, k.p, k.power, k.param, k.lengthscale, k.scale], dtype=np.float64)


# ---- independent mpmath evaluation ------------------------------------------------------------
def mp_phi(k, s):
    s = mp.mpf(s)
    if isinstance(k, o.Composite):   # src/algebra.jl:17,36
        return mp.mpf(k.scale) * sum(mp.fprod(mp_phi(f, s) for f in term) for term in k.terms)
    if k.family == o.CONSTANT:
        return mp.mpf(k.scale)
    if k.trait == o.ISOTROPIC:
        s = s / mp.mpf(k.lengthscale) ** 2
    f = k.family
    if f == o.EQ: v = mp.exp(-s / 2)
    elif f == o.EXP: v = mp.exp(-mp.sqrt(s))
    elif f == o.RQ: v = (1 + s / (2 * mp.mpf(k.param))) ** (-mp.mpf(k.param))
    elif f == o.GAMMAEXP: v = mp.exp(-(s ** (mp.mpf(k.param) / 2)) / 2) if s != 0 else mp.mpf(1)
    elif f == o.CAUCHY: v = 1 / (1 + s)
    elif f == o.IMQ: v = 1 / mp.sqrt(s + mp.mpf(k.param) ** 2)
    elif f == o.MATERNP:
        p = k.p
        r = mp.sqrt((2 * p + 1) * s)   # naive closed form, src/stationary.jl:162-169
        v = sum(mp.mpf(math.factorial(p + i)) / (math.factorial(p - i) * math.factorial(i)) * (2 * r) ** (p - i) for i in range(p + 1))
        v = v * mp.exp(-r) / (mp.mpf(math.factorial(2 * p)) / math.factorial(p))
    elif f == o.MATERN:   # src/stationary.jl:111-112 (no Taylor branch here: the inputs stay far above taylor_bound)
        nu = mp.mpf(k.param)
        r = mp.sqrt(2 * nu * s)
        v = mp.mpf(1) if r == 0 else 2 ** (1 - nu) / mp.gamma(nu) * r ** nu * mp.besselk(nu, r)
    elif f == o.DOT: v = s
    else: v = mp.exp(s)
    return mp.mpf(k.scale) * v ** k.power


def mp_arg(k, x, y):
    x = [mp.mpf(float(t)) for t in np.atleast_1d(x)]; y = [mp.mpf(float(t)) for t in np.atleast_1d(y)]
    if k.trait == o.ISOTROPIC:
        return sum((a - b) ** 2 for a, b in zip(x, y))
    return sum(a * b for a, b in zip(x, y))


def mp_mul(k, X, Y, A, y0, alpha, beta):
    n, m = X.shape[0], Y.shape[0]
    A2 = A.reshape(m, -1)
    out = np.zeros((n, A2.shape[1]))
    for i in range(n):
        row = [mp_phi(k, mp_arg(k, X[i], Y[j])) for j in range(m)]
        for c in range(A2.shape[1]):
            acc = sum(row[j] * mp.mpf(float(A2[j, c])) for j in range(m)) * mp.mpf(alpha)
            if beta != 0:
                acc += mp.mpf(beta) * mp.mpf(float(y0.reshape(n, -1)[i, c]))
            out[i, c] = float(acc)
    return out.reshape((n,) + A.shape[1:])


def mp_grad_mul(k, X, Y, a, y0, alpha, beta):
    n, d = X.shape; m = Y.shape[0]
    A = a.reshape(m, d)
    out = np.zeros((n, d))
    for i in range(n):
        acc = [mp.mpf(beta) * mp.mpf(float(y0.reshape(n, d)[i, l])) if beta != 0 else mp.mpf(0) for l in range(d)]
        xi = [mp.mpf(float(t)) for t in X[i]]
        for j in range(m):
            yj = [mp.mpf(float(t)) for t in Y[j]]; aj = [mp.mpf(float(t)) for t in A[j]]
            s = mp_arg(k, X[i], Y[j])
            k1 = mp.diff(lambda t: mp_phi(k, t), s)
            k2 = mp.diff(lambda t: mp_phi(k, t), s, 2)
            if k.trait == o.ISOTROPIC:
                r = [p_ - q_ for p_, q_ in zip(xi, yj)]
                ra = sum(p_ * q_ for p_, q_ in zip(r, aj))
                for l in range(d):
                    acc[l] += mp.mpf(alpha) * -2 * (k1 * aj[l] + 2 * k2 * r[l] * ra)
            else:
                xa = sum(p_ * q_ for p_, q_ in zip(xi, aj))
                for l in range(d):
                    acc[l] += mp.mpf(alpha) * (k1 * aj[l] + k2 * yj[l] * xa)
        out[i] = [float(v) for v in acc]
    return out.reshape(-1)


def mp_valgrad_mul(k, X, Y, a, y0, alpha, beta):
    """50-digit ValueGradientKernel MVM: value / gradient covariances of src/gradient.jl:441-463 with k0, k1, k2 by
    numerical differentiation of the profile, blocks applied as src/gradient.jl:319-351."""
    n, d = X.shape; m = Y.shape[0]
    A = a.reshape(m, d + 1)
    out = np.zeros((n, d + 1))
    for i in range(n):
        acc = [mp.mpf(beta) * mp.mpf(float(y0.reshape(n, d + 1)[i, l])) if beta != 0 else mp.mpf(0) for l in range(d + 1)]
        xi = [mp.mpf(float(t)) for t in X[i]]
        for j in range(m):
            yj = [mp.mpf(float(t)) for t in Y[j]]; aj = [mp.mpf(float(t)) for t in A[j]]
            a0, ag = aj[0], aj[1:]
            s = mp_arg(k, X[i], Y[j])
            k0 = mp_phi(k, s)
            k1 = mp.diff(lambda t: mp_phi(k, t), s)
            k2 = mp.diff(lambda t: mp_phi(k, t), s, 2)
            al = mp.mpf(alpha)
            if k.trait == o.ISOTROPIC:
                r = [p_ - q_ for p_, q_ in zip(xi, yj)]
                ra = sum(p_ * q_ for p_, q_ in zip(r, ag))
                acc[0] += al * (k0 * a0 - 2 * k1 * ra)
                for l in range(d):
                    acc[1 + l] += al * (2 * k1 * r[l] * a0 - 2 * (k1 * ag[l] + 2 * k2 * r[l] * ra))
            else:
                xa = sum(p_ * q_ for p_, q_ in zip(xi, ag))
                acc[0] += al * (k0 * a0 + k1 * xa)
                for l in range(d):
                    acc[1 + l] += al * (k1 * yj[l] * a0 + k1 * ag[l] + k2 * yj[l] * xa)
        out[i] = [float(v) for v in acc]
    return out.reshape(-1)


# composite kernels (src/algebra.jl:5-63) and the kernels of the value-gradient fixtures
COMPOSITES = {
    # 1.7 * (1.3 MaternP(2; l=0.8) * RQ(1.5)^2 + EQ(l=2) + 0.5)
    "iso_sum_of_products": o.Composite(((o.Kernel(o.MATERNP, p=2, lengthscale=0.8, scale=1.3), o.Kernel(o.RQ, param=1.5, power=2)),
                                        (o.Kernel(o.EQ, lengthscale=2.0),), (o.Kernel(o.CONSTANT, scale=0.5),)), o.ISOTROPIC, 1.7),
    # EQ * Cauchy (test/algebra.jl-style product of two isotropic kernels)
    "iso_product": o.Composite(((o.Kernel(o.EQ), o.Kernel(o.CAUCHY, lengthscale=1.5)),), o.ISOTROPIC, 1.0),
    # Dot^2 + 0.3 * ExponentialDot  (test/gradient_algebra.jl:33-47 style sum of dot-product kernels)
    "dot_sum": o.Composite(((o.Kernel(o.DOT, power=2),), (o.Kernel(o.EXPDOT, scale=0.3),)), o.DOTPRODUCT, 1.0),
}
VALGRAD_KERNELS = ["EQ", "RQ1", "MaternP2", "Dot3", "ExpDot", "EQ_l07"]
# Matern with real nu (src/stationary.jl:87-114): Bessel-function profile, checked here against mpmath's besselk
MATERN_KERNELS = {"Matern0.8": o.Kernel(o.MATERN, param=0.8), "Matern1.3": o.Kernel(o.MATERN, param=1.3, lengthscale=0.6),
                  "Matern2.7": o.Kernel(o.MATERN, param=2.7, scale=1.5), "Matern2.5": o.Kernel(o.MATERN, param=2.5)}


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    os.makedirs(OUT, exist_ok=True)
    worst = 0.0
    # ---- (1) dense ------------------------------------------------------------------------------
    dense = {}
    names = list(KERNELS)
    dense["kernel_names"] = np.array(names)
    dense["kernel_fields"] = np.stack([kernel_fields(KERNELS[k]) for k in names])
    for d in (1, 2, 3, 8, 32):
        for (n, m) in ((8, 16), (257, 129)):
            rng = np.random.default_rng(0xC0F * 1000 + 100 * d + n)
            X = rng.standard_normal((n, d)) / math.sqrt(d); Y = rng.standard_normal((m, d)) / math.sqrt(d)
            a = rng.standard_normal(m); A3 = rng.standard_normal((m, 3)); y0 = rng.standard_normal(n); Y3 = rng.standard_normal((n, 3))
            tag = f"d{d}_n{n}_m{m}"
            dense[f"{tag}_X"], dense[f"{tag}_Y"], dense[f"{tag}_a"], dense[f"{tag}_A3"], dense[f"{tag}_y0"], dense[f"{tag}_Y3"] = X, Y, a, A3, y0, Y3
            for name in names:
                k = KERNELS[name]
                b10 = o.mul(None, k, X, Y, a, 1.0, 0.0)
                b2 = o.mul(y0, k, X, Y, a, -0.7, 1.3)
                B3 = o.mul(Y3, k, X, Y, A3, -0.7, 1.3)
                if n == 8:   # independent 50-digit check
                    for got, want in ((b10, mp_mul(k, X, Y, a, y0, 1.0, 0.0)), (b2, mp_mul(k, X, Y, a, y0, -0.7, 1.3)),
                                      (B3, mp_mul(k, X, Y, A3, Y3, -0.7, 1.3))):
                        e = rel(got, want); worst = max(worst, e)
                        assert e < 1e-13, (name, tag, e)
                dense[f"{tag}_{name}_b10"], dense[f"{tag}_{name}_b2"], dense[f"{tag}_{name}_B3"] = b10, b2, B3
    np.savez_compressed(os.path.join(OUT, "dense.npz"), **dense)

    # ---- (2) gradient (mirrors test/gradient.jl:16-52) -------------------------------------------
    grad = {"kernel_names": np.array(GRAD_KERNELS), "kernel_fields": np.stack([kernel_fields(KERNELS[k]) for k in GRAD_KERNELS])}
    for d in (1, 5, 32):
        for n in (2, 33):
            rng = np.random.default_rng(0xC0F * 2000 + 100 * d + n)
            X = rng.standard_normal((n, d)) / math.sqrt(d)
            a = rng.standard_normal(n * d); b0 = rng.standard_normal(n * d)
            alpha, beta = (float(v) for v in rng.standard_normal(2))
            tag = f"d{d}_n{n}"
            grad[f"{tag}_X"], grad[f"{tag}_a"], grad[f"{tag}_b0"], grad[f"{tag}_ab"] = X, a, b0, np.array([alpha, beta])
            for name in GRAD_KERNELS:
                k = KERNELS[name]
                out = o.grad_mul(b0, k, X, X, a, alpha, beta)
                if n == 2 and d <= 5:
                    # independent 50-digit check on a rectangular twin (Y != X): numerical differentiation of the
                    # sqrt-type profiles is not defined at s = 0; the diagonal is pinned by closed forms in tests/test_oracle.py
                    Yt = X[::-1] * 0.7 + 0.31
                    e = rel(o.grad_mul(b0, k, X, Yt, a, alpha, beta), mp_grad_mul(k, X, Yt, a, b0, alpha, beta)); worst = max(worst, e)
                    assert e < 1e-12, (name, tag, e)
                grad[f"{tag}_{name}"] = out
    np.savez_compressed(os.path.join(OUT, "gradient.npz"), **grad)

    # ---- (3) Toeplitz (mirrors test/gramian.jl:143-178) -------------------------------------------
    toep = {}
    for n in (32, 1000, 4096):
        rg = o.srange(-1, 1, n)
        x = o.srange_points(rg)
        rng = np.random.default_rng(0xC0F * 3000 + n)
        a = rng.standard_normal(n); y0 = rng.standard_normal(n)
        toep[f"n{n}_a"], toep[f"n{n}_y0"] = a, y0
        for name in ("EQ", "Exp"):
            k = KERNELS[name]
            vc, _ = o.toeplitz_vectors(k, rg)
            dense_T = o.toeplitz_dense(vc)
            assert rel(dense_T, o.matrix(k, x, x)) < 1e-14      # Matrix(G) ≈ Matrix(Gramian(k, x))
            b = dense_T @ a
            assert rel(o.toeplitz_mul(None, vc, None, a), b) < 1e-11
            toep[f"n{n}_{name}_vc"], toep[f"n{n}_{name}_b"] = vc, b
            toep[f"n{n}_{name}_b_ab"] = 0.3 * b - 1.1 * y0
            if n == 32:
                yr = (rg[0] + 0.37, rg[1], rg[2])
                vc2, vr2 = o.toeplitz_vectors(k, rg, yr)
                T2 = o.toeplitz_dense(vc2, vr2)
                assert rel(T2, o.matrix(k, x, o.srange_points(yr))) < 1e-14
                toep[f"n{n}_{name}_shift_vc"], toep[f"n{n}_{name}_shift_vr"], toep[f"n{n}_{name}_shift_b"] = vc2, vr2, T2 @ a
                toep[f"n{n}_{name}_circ_b"] = o.toeplitz_dense(vc, circulant=True) @ a
    np.savez_compressed(os.path.join(OUT, "toeplitz.npz"), **toep)

    # ---- (4) Kronecker / separable (mirrors test/algebra.jl:70-89, test/separable.jl:9-28) ---------
    rng = np.random.default_rng(0xC0F * 4000)
    n, d = 4, 3
    x = rng.standard_normal(n); y = rng.standard_normal(2 * n)
    F = o.matrix(KERNELS["EQ"], x, y)
    a = rng.standard_normal((2 * n) ** d)
    kr = {"x": x, "y": y, "a": a, "b": o.kron_mul(None, [F, F, F], a)}
    assert rel(kr["b"], o.kron_dense([F, F, F]) @ a) < 1e-13
    # the Kronecker Gramian of a separable EQ equals the isotropic EQ Gramian on the grid points (row-major order)
    gx, gy = o.lazy_grid_points([x] * d), o.lazy_grid_points([y] * d)
    assert rel(o.matrix(KERNELS["EQ"], gx, gy) @ a, kr["b"]) < 1e-13
    f1, f2, f3 = rng.standard_normal((3, 5)), rng.standard_normal((4, 2)), rng.standard_normal((2, 6))
    av = rng.standard_normal(5 * 2 * 6)
    kr.update(f1=f1, f2=f2, f3=f3, av=av, bv=np.kron(np.kron(f1, f2), f3) @ av)
    assert rel(o.kron_mul(None, [f1, f2, f3], av), kr["bv"]) < 1e-13
    B = rng.standard_normal((3, 3)); B = B.T @ B
    xs = rng.standard_normal(3); v = rng.standard_normal(9)
    kr.update(B=B, xs=xs, v=v, sep_b=np.kron(o.matrix(KERNELS["EQ"], xs, xs), B) @ v)
    np.savez_compressed(os.path.join(OUT, "kronecker.npz"), **kr)

    # ---- (5) composite kernels and ValueGradientKernel (mirrors test/gradient.jl:87-125, test/gradient_algebra.jl:13-47) ----
    comp = {"names": np.array(list(COMPOSITES)), "valgrad_names": np.array(VALGRAD_KERNELS), "matern_names": np.array(list(MATERN_KERNELS))}
    for nm, km in MATERN_KERNELS.items():
        comp[f"matern_{nm}_fields"] = kernel_fields(km)
    for d in (1, 3, 8):
        for (n, m) in ((4, 6), (65, 33)):
            rng = np.random.default_rng(0xC0F * 5000 + 100 * d + n)
            X = rng.standard_normal((n, d)) / math.sqrt(d); Y = rng.standard_normal((m, d)) / math.sqrt(d)
            a = rng.standard_normal(m); y0 = rng.standard_normal(n)
            ag = rng.standard_normal(m * d); yg0 = rng.standard_normal(n * d)
            av = rng.standard_normal(m * (d + 1)); yv0 = rng.standard_normal(n * (d + 1))
            alpha, beta = (float(v) for v in rng.standard_normal(2))
            tag = f"d{d}_n{n}_m{m}"
            for key, val in (("X", X), ("Y", Y), ("a", a), ("y0", y0), ("ag", ag), ("yg0", yg0), ("av", av), ("yv0", yv0),
                             ("ab", np.array([alpha, beta]))):
                comp[f"{tag}_{key}"] = val
            kernels = dict(COMPOSITES); kernels.update({nm: KERNELS[nm] for nm in VALGRAD_KERNELS}); kernels.update(MATERN_KERNELS)
            for name, k in kernels.items():
                b = o.mul(y0, k, X, Y, a, alpha, beta)
                bg = o.grad_mul(yg0, k, X, Y, ag, alpha, beta)
                bv = o.valgrad_mul(yv0, k, X, Y, av, alpha, beta)
                if n == 4 and d <= 3:
                    for got, want in ((b, mp_mul(k, X, Y, a, y0, alpha, beta)), (bg, mp_grad_mul(k, X, Y, ag, yg0, alpha, beta)),
                                      (bv, mp_valgrad_mul(k, X, Y, av, yv0, alpha, beta))):
                        e = rel(got, want); worst = max(worst, e)
                        assert e < 1e-12, (name, tag, e)
                if name in COMPOSITES or name in MATERN_KERNELS:
                    comp[f"{tag}_{name}_b"], comp[f"{tag}_{name}_bg"] = b, bg
                comp[f"{tag}_{name}_bv"] = bv
    np.savez_compressed(os.path.join(OUT, "composite.npz"), **comp)

    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)}
    print("worst oracle-vs-mpmath relative error:", worst)
    print("wrote", sizes)


if __name__ == "__main__":
    main()
