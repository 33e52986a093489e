"""Property tests (hypothesis) of the HIP path against the oracle: arbitrary small tables with
NULs, high bytes, empty values, heavy duplicates, 1-3 key columns, prefix joins, Find."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from csvplus_amd import DeviceIndex, StrCol, join_chain
from oracle import orc
from tests.helpers import assert_bounds_equal, assert_join_equal

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_build_paths")]

value = st.one_of(
    st.binary(min_size=0, max_size=6),
    st.text(alphabet="0123456789", min_size=0, max_size=9).map(str.encode),
    st.sampled_from([b"", b"a", b"a\x00", b"ab", b"\xff", b"\x00", b"Smith", b"Jones"]),
    st.binary(min_size=20, max_size=40),
)


@st.composite
def tables(draw):
    ncols = draw(st.integers(1, 3))
    n = draw(st.integers(0, 60))
    m = draw(st.integers(0, 60))
    pool = [draw(st.lists(value, min_size=1, max_size=8)) for _ in range(ncols)]
    build = [[draw(st.sampled_from(pool[c])) for _ in range(n)] for c in range(ncols)]
    extra = [draw(st.lists(value, min_size=1, max_size=3)) for _ in range(ncols)]
    probe = [[draw(st.sampled_from(pool[c] + extra[c])) for _ in range(m)] for c in range(ncols)]
    return build, probe


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture,
                                                                  HealthCheck.too_slow, HealthCheck.data_too_large])
@given(tables())
def test_index_join_find_match_oracle(ctx, t):
    build, probe = t
    bcols = [StrCol.from_values(c) for c in build]
    pcols = [StrCol.from_values(c) for c in probe]
    g = DeviceIndex(ctx, bcols)
    o = orc.OracleIndex(bcols)
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    for k in range(1, len(bcols) + 1):
        assert_join_equal(g.probe(pcols[:k]), o.join(pcols[:k]))
        assert_bounds_equal(g, pcols[:k], o.join(pcols[:k]))   # bounds only: the rank table when the index has distinct keys
    for r in range(min(5, len(build[0]))):
        for k in range(1, len(bcols) + 1):
            vals = [build[c][r] for c in range(k)]
            assert g.find(*vals) == o.find(*vals)
    if len(bcols) == 1 and probe[0]:
        ch = join_chain(ctx, [(g, pcols)])
        j = o.join(pcols)
        np.testing.assert_array_equal(ch.stream_row, j["probe_idx"])
        np.testing.assert_array_equal(ch.build_row(0), j["build_row"])
        ch.release()
    if probe[0]:   # the chain reporting sorted positions, on 1-3 key columns: perm[position] is the oracle's row
        chp = join_chain(ctx, [(g, pcols)], positions=True)
        j = o.join(pcols)
        np.testing.assert_array_equal(chp.stream_row, j["probe_idx"])
        np.testing.assert_array_equal(g.perm()[chp.build_row(0)] if chp.nrows else np.zeros(0, np.uint32), j["build_row"])
        chp.release()
    keys = [tuple(build[c][r] for c in range(len(bcols))) for r in range(min(8, len(build[0])))]
    if keys:
        lo, hi = g.find_many(keys)
        for (a, b), key in zip(zip(lo, hi), keys):
            assert (int(a), int(b)) == o.find(*key)
    g.close()
