"""Constructor arguments of the equation-class cases of tests/golden/equations.json (used by the generating script on the
reference's classes and by tests/test_equations.py on this package's)."""
CASES = {
    # case: (class, reference module under ppsci/equation/pde/, kwargs)
    "laplace2": ("Laplace", "laplace", dict(dim=2)),
    "laplace3_detach": ("Laplace", "laplace", dict(dim=3, detach_keys=("u__x__x",))),
    "poisson2": ("Poisson", "poisson", dict(dim=2)),
    "ns2_steady": ("NavierStokes", "navier_stokes", dict(nu=0.01, rho=1.0, dim=2, time=False)),
    "ns3_unsteady_symbolic_nu": ("NavierStokes", "navier_stokes", dict(nu="nu", rho=1.0, dim=3, time=True)),
    "ns2_detach": ("NavierStokes", "navier_stokes", dict(nu=0.01, rho=1.0, dim=2, time=False, detach_keys=("u", "v__y"))),
    "biharmonic1": ("Biharmonic", "biharmonic", dict(dim=1, q=-1.0, D=1.0)),
    "biharmonic2": ("Biharmonic", "biharmonic", dict(dim=2, q=2.0, D=0.5)),
    "nlsmb": ("NLSMB", "nls_m_b", dict(alpha_1=0.5, alpha_2=-1, omega_0=-1, time=True)),
    "nlsmb_float": ("NLSMB", "nls_m_b", dict(alpha_1=0.25, alpha_2=1.5, omega_0=0.5, time=True, detach_keys=("eta",))),
    "normal_dot_vec3": ("NormalDotVec", "normal_dot_vec", dict(vec_keys=("u", "v", "w"))),
    "normal_dot_vec2": ("NormalDotVec", "normal_dot_vec", dict(vec_keys=("u", "v"))),
    "heat_exchanger": ("HeatExchanger", "heat_exchanger", dict(alpha_h=0.5, alpha_c=0.25, v_h=1.5, v_c=2.0, w_h=0.1, w_c=0.3)),
    "elasticity3_lame": ("LinearElasticity", "linear_elasticity", dict(E=None, nu=None, lambda_=1e4, mu=100, dim=3)),
    "elasticity2_E_nu_time": ("LinearElasticity", "linear_elasticity", dict(E=10.0, nu=0.3, rho=2.0, dim=2, time=True)),
    "elasticity3_fields": ("LinearElasticity", "linear_elasticity", dict(lambda_="lambda_f", mu="mu_f", rho="rho_f", dim=3,
                                                                         detach_keys=("sigma_xx",))),
}
