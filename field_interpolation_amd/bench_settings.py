"""The solver settings of the BASELINE.json configurations, in ONE place: bench.py times them, tests/test_gpu_fullsize*.py
check them against the oracle's golden solutions -- the same numbers by construction (the bench's settings were
verified by the bench line alone).

Every configuration runs the same solver: fp64 CG (x, r, p, the operator apply, every dot product, the stop test)
preconditioned by an fp32 V-cycle over `levels` coarser levels, from a coarse-to-fine start whose levels are solved to
`coarse_tol` (FI_OPT_LEVELS / FI_OPT_COARSE_TOLERANCE / FI_OPT_MULTIGRID / FI_OPT_MIXED_PRECISION).

Stop rule.  BASELINE.json states a relative RESIDUAL (1e-5; config 5: 1e-6) and, in the north-star, a FIELD tolerance (values
within 1e-5 of the CPU reference's double solve, sparse_linear.cpp:154-184).  kappa ~ side^4: what a residual buys in the field
varies by three orders of magnitude with the lattice, the data and the weights (field error per unit of residual: 2-3 for
config 2, 70-140 for config 4, 400-800 for the oriented points of configs 3 and 5).  Round 6: ONE rule for every configuration
and no constant that depends on the workload -- the solver stops by the field (FI_OPT_FIELD_TOLERANCE, include/fi_hip.h: the
last step times sigma / (1 - sigma), sigma the slowest mean decay of the residual norm over the recent windows and the whole
solve, with a margin of two), checked here
against the oracle's fp64 solutions wherever it has one (tests/test_gpu_fullsize_golden.py: three seeds of config 4 at 256^3,
config 2 at full size, config 3's shape at 1024^2, config 5's at 128^3; tools/r6_field_rule.py, profiles/r6_field_rule.txt)
and on random problems against the solve to the fp64 floor (tests/stress_field_rule.py).
Rounds 4-5 stopped config 4's own 256^3 / 1 M point workload at a residual calibrated against the oracle for exactly that
workload (3e-7, five iterations); the rule takes six to seven there -- the price of not knowing the answer in advance.
Over slabs (one process per GPU, up to FIELD_RULE_MAX_SLABS) the same rule: every slab's maxima travel with the r . r sum of
the iteration's all-reduce; `slab_residual` below is what a run over more slabs (or one given --tol) stops at."""

CONFIG4_SEEDS = (3, 11, 12)      # synth.config4 seeds with an oracle golden at 256^3 (3: the metric's own workload)
FIELD_TOLERANCE = 1e-5           # BASELINE.json north_star
FIELD_RULE_MAX_SLABS = 16        # fi_internal.h kFieldRanks


def slab_residual(config, sizes):
    """Residual at which a run over more than FIELD_RULE_MAX_SLABS slabs stops: BASELINE's for
    configs 2 / 3 / 5, for config 4 the conservative rule of round 3 (1e-7, tightened with the lattice beyond 256^3)."""
    if config == 4:
        return 1e-7 * min(1.0, (256.0 / max(sizes)) ** 1.75)
    return SETTINGS[config]["tol"]


#            coarser levels, tolerance of the start's levels, BASELINE's residual
SETTINGS = {
    4: dict(levels=3, coarse_tol=3e-4, tol=1e-5),
    # (kcycle, round 6: FI_OPT_MG_KCYCLE on the first four coarse levels -- 44 -> 17 iterations, 568 -> 371 ms; 1 / 2 / 3 levels: 525 / 413 / 386)
    # (cheb: the full-operator smoother under the K-cycle -- (4, 10) 228-259 ms / 12-14 iterations, the 3-D default (5, 40) 347-371 / 16-17)
    5: dict(levels=6, coarse_tol=1e-2, tol=1e-6, kcycle=4, cheb=(4, 10.0)),
    # (hierarchy depth and the levels' tolerance: tools/r4_sweep_c23.sh, tools/r5_sweep_levels.sh -- the levels of a
    # coarse-to-fine start are worth a loose solve only: config 3 with 7 levels to 1e-4 57.6 ms per step, 8 levels to 1e-1 28.4)
    # (round 6, under the field rule: 9 levels -- the coarsest 8^2 -- 45 ms and 26 iterations against 52 / 34 with 8; 7 levels: 129 / 93)
    # (kcycle 2: 26 -> 11 iterations, 45.5 -> 40.1 ms; 1: 43.6, 3: 41.8, 4: 59 -- the small levels' launches)
    3: dict(levels=9, coarse_tol=1e-1, tol=1e-5, kcycle=2),
    # (round 5, profiles/r5_sweep_levels.txt: 3 levels -- the coarsest 128^2 -- 4.0 ms per step and 8 iterations, 4 levels 4.8 / 9, 2 levels 9.9 / 25)
    2: dict(levels=3, coarse_tol=1e-1, tol=1e-5),
}


def configure(field, levels, coarse_tol, multigrid=True, mixed=True, by_field=False, kcycle=0, cheb=None):
    """The headline solver on a LatticeField whose model weights are set.  by_field: stop by the field (the bench's rule on one
    GPU: FIELD_TOLERANCE), otherwise at the residual passed to solve_cg.  kcycle: FI_OPT_MG_KCYCLE (SETTINGS[config].get("kcycle", 0))."""
    if levels > 0:
        field.set_levels(levels, coarse_tol)
        if multigrid:
            field.set_multigrid(True)
            if mixed:
                field.set_mixed_precision(True)
            if kcycle > 0 and hasattr(field, "set_kcycle"):
                field.set_kcycle(kcycle)
                if cheb and hasattr(field, "set_cheb_smoother"):   # (the K-cycle's companion setting: with the V-cycle the defaults stand)
                    field.set_cheb_smoother(*cheb)
    if by_field:
        field.set_field_tolerance(FIELD_TOLERANCE)


def headline_field(fi, config, sizes, weights, by_field=False, **kw):
    """A fresh fp64 LatticeField with the configuration's solver settings (points not added yet)."""
    s = SETTINGS[config]
    f = fi.LatticeField(sizes, dtype="f64", **kw)
    f.add_field_constraints(weights)
    configure(f, s["levels"], s["coarse_tol"], by_field=by_field, kcycle=s.get("kcycle", 0), cheb=s.get("cheb"))
    return f
