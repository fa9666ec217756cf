"""The tutorial's multigrid run (docs/src/index.md:162-304) as the pin for the smoother / V-cycle rows: it is the only
V-cycle observable the reference publishes.  CPU: the oracle reproduces the published tail contraction; GPU: the device
path follows the oracle cycle by cycle."""
import numpy as np
import pytest

import _docs_example as D


def _tail_ratio(norms):
    return (norms[-1] / norms[-3]) ** 0.5


BAND = (0.74, 0.90)     # measured: 0.78-0.86 over seeds, mesh sizes and lambda (see _docs_example.py)


def test_oracle_contraction_on_the_tutorial_problem(oracle):
    """The restated V-cycle on the tutorial's problem.  The published tail ratio (0.9105) is NOT reproduced -- the
    restatement converges faster -- and the test says so instead of hiding it: it pins the measured band and checks
    that the published ratio lies inside the envelope spanned by the coefficient contrast (1 -> ~0.34, 100 -> ~0.97)."""
    ref = _tail_ratio(D.REFERENCE_TAIL)
    assert abs(ref - 0.9105) < 1e-3                                    # what the reference prints
    norms, _, _ = D.run_oracle(oracle)
    ratio = _tail_ratio(norms)
    assert BAND[0] <= ratio <= BAND[1], ratio
    assert all(b < a for a, b in zip(norms[5:], norms[6:]))            # monotone, as printed (5.18 > 4.72 > 4.30)
    lo = _tail_ratio(D.run_oracle(oracle, cycles=12, high=1.0)[0])     # constant coefficient: fast
    hi = _tail_ratio(D.run_oracle(oracle, cycles=60, high=100.0)[0])   # contrast 100: slow
    assert lo < 0.5 and hi > 0.93
    assert lo < ratio < ref < hi                                       # the printout is inside the contrast envelope
    assert ratio < ref                                                 # ... and slower than the restatement (FINDING)


@pytest.mark.parametrize("seed", [1, 7])
def test_contraction_is_insensitive_to_the_random_field(oracle, seed):
    norms, _, _ = D.run_oracle(oracle, seed=seed)
    assert BAND[0] <= _tail_ratio(norms) <= BAND[1], _tail_ratio(norms)


@pytest.mark.gpu
def test_device_follows_the_oracle_cycle_by_cycle(oracle):
    import homogenization_jl_amd as hmg
    ctx = hmg.Context(0)
    try:
        want, xs_o, _ = D.run_oracle(oracle)
        got, xs_d = D.run_device(hmg, ctx, oracle)
        assert BAND[0] <= _tail_ratio(got) <= BAND[1]
        # Measured (tools/dev/docs_drift.py): norm(r) agrees to 1e-13 for the first 40 cycles, then the difference grows
        # by ~10x per 3 cycles and saturates at 1-6 % of the (by then 1e-3 ... 1e-7) residual (it differs from run to run: the
        # oracle's OpenMP dot products are not bit-reproducible either): the CG coefficients
        # depend on the iterate, and the direction of the tail residual is a sensitive function of rounding (any two
        # FP64 implementations -- e.g. another dot-product summation order -- separate the same way).  x itself
        # stays within 3e-7 of max|x| throughout.
        for i, (a, b) in enumerate(zip(want, got)):
            assert abs(a - b) <= (1e-10 if i < 40 else 0.25) * a, (i, a, b)
        for i in range(100):
            assert np.abs(xs_d[i] - xs_o[i]).max() <= (1e-12 if i < 40 else 2e-6) * np.abs(xs_o[i]).max(), i
    finally:
        ctx.close()
