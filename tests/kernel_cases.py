"""Shared kernel zoo for the parity tests: (name, covgram kernel factory, oracle Kernel)."""
import covgram_oracle as o


def cases(cg):
    return [
        ("EQ", cg.EQ(), o.Kernel(o.EQ)),
        ("Exp", cg.Exp(), o.Kernel(o.EXP)),
        ("RQ(1.0)", cg.RQ(1.0), o.Kernel(o.RQ, param=1.0)),
        ("RQ(0.37)", cg.RQ(0.37), o.Kernel(o.RQ, param=0.37)),
        ("gammaExp(1.5)", cg.GammaExp(1.5), o.Kernel(o.GAMMAEXP, param=1.5)),
        ("Cauchy", cg.Cauchy(), o.Kernel(o.CAUCHY)),
        ("IMQ(0.8)", cg.InverseMultiQuadratic(0.8), o.Kernel(o.IMQ, param=0.8)),
        ("MaternP(0)", cg.MaternP(0), o.Kernel(o.MATERNP, p=0)),
        ("MaternP(1)", cg.MaternP(1), o.Kernel(o.MATERNP, p=1)),
        ("MaternP(2)", cg.MaternP(2), o.Kernel(o.MATERNP, p=2)),
        ("MaternP(3)", cg.MaternP(3), o.Kernel(o.MATERNP, p=3)),
        ("Lengthscale(EQ,0.7)", cg.Lengthscale(cg.EQ(), 0.7), o.Kernel(o.EQ, lengthscale=0.7)),
        ("2.5*Lengthscale(MaternP(2),1.3)", 2.5 * cg.Lengthscale(cg.MaternP(2), 1.3), o.Kernel(o.MATERNP, p=2, lengthscale=1.3, scale=2.5)),
        ("Dot()^3", cg.Dot() ** 3, o.Kernel(o.DOT, power=3)),
        ("Dot()", cg.Dot(), o.Kernel(o.DOT)),
        ("ExponentialDot", cg.ExponentialDot(), o.Kernel(o.EXPDOT)),
        ("EQ^2", cg.EQ() ** 2, o.Kernel(o.EQ, power=2)),
        # Matern with real nu (Bessel-function profile, src/stationary.jl:87-114): below 1, between 1 and 2, above 2, half-integer
        ("Matern(0.8)", cg.Matern(0.8), o.Kernel(o.MATERN, param=0.8)),
        ("Lengthscale(Matern(1.3),0.6)", cg.Lengthscale(cg.Matern(1.3), 0.6), o.Kernel(o.MATERN, param=1.3, lengthscale=0.6)),
        ("1.5*Matern(2.7)", 1.5 * cg.Matern(2.7), o.Kernel(o.MATERN, param=2.7, scale=1.5)),
        ("Matern(2.5)", cg.Matern(2.5), o.Kernel(o.MATERN, param=2.5)),
        ("Matern(7.25)", cg.Matern(7.25), o.Kernel(o.MATERN, param=7.25)),
    ]


# Sum / Product / Power of kernels sharing one input trait (src/algebra.jl:5-63): same definitions as
# oracle/make_golden.py::COMPOSITES, written with the host algebra on one side and the oracle dataclass on the other
def composite_cases(cg):
    return [
        ("iso_sum_of_products",
         1.7 * (1.3 * cg.Lengthscale(cg.MaternP(2), 0.8) * cg.RQ(1.5) ** 2 + cg.Lengthscale(cg.EQ(), 2.0) + 0.5),
         o.Composite(((o.Kernel(o.MATERNP, p=2, lengthscale=0.8, scale=1.3), o.Kernel(o.RQ, param=1.5, power=2)),
                      (o.Kernel(o.EQ, lengthscale=2.0),), (o.Kernel(o.CONSTANT, scale=0.5),)), o.ISOTROPIC, 1.7)),
        ("iso_product", cg.EQ() * cg.Lengthscale(cg.Cauchy(), 1.5),
         o.Composite(((o.Kernel(o.EQ), o.Kernel(o.CAUCHY, lengthscale=1.5)),), o.ISOTROPIC, 1.0)),
        ("dot_sum", cg.Dot() ** 2 + 0.3 * cg.ExponentialDot(),
         o.Composite(((o.Kernel(o.DOT, power=2),), (o.Kernel(o.EXPDOT, scale=0.3),)), o.DOTPRODUCT, 1.0)),
    ]


def valgrad_cases(cg):
    keep = {"EQ", "RQ(1.0)", "MaternP(2)", "Dot()^3", "ExponentialDot", "Lengthscale(EQ,0.7)", "Cauchy", "EQ^2",
            "2.5*Lengthscale(MaternP(2),1.3)", "1.5*Matern(2.7)"}
    return [c for c in cases(cg) if c[0] in keep] + composite_cases(cg)


# kernels whose phi', phi'' are finite at s = 0 (gradient Gramian well defined on the diagonal)
def grad_cases(cg):
    keep = {"EQ", "RQ(1.0)", "RQ(0.37)", "Cauchy", "IMQ(0.8)", "MaternP(2)", "MaternP(3)", "Lengthscale(EQ,0.7)",
            "2.5*Lengthscale(MaternP(2),1.3)", "Dot()^3", "Dot()", "ExponentialDot", "EQ^2", "1.5*Matern(2.7)", "Matern(2.5)",
            "Matern(7.25)"}
    return [c for c in cases(cg) if c[0] in keep]
