"""-m gpu: every BASELINE configuration at its FULL particle count against the one-thread oracle, whole arrays, bit for bit
("ancestor index arrays bit-exact" at C3 and C4 had rested on slices and on the floor / ceil offspring property;
C2 at full N had been checked for 6 observations).  The oracle needs about 70 s of one host core for the three cases together: the
series are short (C3, C4) or the particle count moderate (C2), which is what makes the comparison affordable -- the kernels, launch
geometries and group-sum layouts are those of the full-size runs:
  C2  N = 2^20, T = 100   one tile per unit, 1025 k_offspring blocks resident at once, group sums layout 1
  C3  N = 2^22, T = 3     d = 9: one particle per thread, FINE geometry (one tile per block + k_reduce_units)
  C4  N = 2^24, T = 3     LGCP, 4-tile units, 4096 units: group sums layout 2; first event dt = 0 (level = its max: k_tile_sums),
                          second and third ~ 9-12 sub-steps with the level predicted from the event before (contract v8)
"""
import numpy as np
import pytest

import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize("name,n,T,lgcp", [("c2_model", 1 << 20, 100, 0), ("c3_model", 1 << 22, 3, 0), ("c4_model", 1 << 24, 3, 2)])
def test_full_size_whole_arrays_bit_exact(name, n, T, lgcp):
    model = getattr(cases, name)()
    t, y, has = cases.event_times(T, horizon=0.1 * T) if lgcp else cases.poisson_counts(T)
    g = NativePf(model, n, cases.SEED, lgcp_precision=lgcp)
    gl, gll, gess, _ = g.run(t, y, has)
    o = oracle.OraclePf(model.descriptor(lgcp), n, cases.SEED)
    ol, oll, oess, _ = o.filter(t, y, has)
    assert gl == ol, (gl, ol)
    np.testing.assert_array_equal(gll, oll)                      # the likelihood after every observation
    np.testing.assert_array_equal(gess, oess)
    anc = g.ancestors()
    np.testing.assert_array_equal(anc, o.ancestors())           # the FULL ancestor array of the last observation
    np.testing.assert_array_equal(g.particles(), o.particles())  # ... and the resampled cloud it selects
    assert np.all(np.diff(anc.astype(np.int64)) >= 0)
    g.close()
