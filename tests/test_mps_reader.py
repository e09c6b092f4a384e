"""MPS / perPlex reader (pysparselp_amd/MPSparser.py, netlib.py) against what the reference's parser returns for
the five netlib problems it ships (fixtures: tests/golden/make_netlib_golden.py), plus the parts of the format the
reference does not read (free format, RANGES).  CPU only."""
import io
import os

import numpy as np
import pytest
import scipy.sparse

from conftest import GOLDEN
from pysparselp_amd.MPSparser import MpsError, mps_parser
from pysparselp_amd.netlib import get_problem

NETLIB = os.path.join(GOLDEN, "netlib")


@pytest.mark.parametrize("name", ["AFIRO", "KB2", "SC50A", "SC50B", "SC105"])
def test_reader_matches_reference_parser(name):
    ref = np.load(os.path.join(GOLDEN, "netlib_parsed.npz"))
    d = get_problem(name, data_dir=NETLIB)
    for key in ("cost_vector", "upper_bounds", "lower_bounds", "b_eq", "b_lower", "b_upper", "solution"):
        assert np.array_equal(d[key], ref[f"{name}_{key}"]), key
    for key in ("a_eq", "a_ineq"):
        m = scipy.sparse.csr_matrix(d[key])
        m.sort_indices()
        assert tuple(ref[f"{name}_{key}_shape"]) == m.shape
        assert np.array_equal(m.indptr, ref[f"{name}_{key}_indptr"]) and np.array_equal(m.indices, ref[f"{name}_{key}_indices"])
        assert np.array_equal(m.data, ref[f"{name}_{key}_data"])
    # the reference cuts the NAME line at the data-field columns and returns '' for these files (MPSparser.py:55-57)
    assert d["problem_name"] == name and ref[f"{name}_names"][0] in ("", name) and d["costname"] == ref[f"{name}_names"][1]
    # the perPlex point is feasible and optimal for the parsed LP
    x = d["solution"]
    assert np.all(x >= d["lower_bounds"] - 1e-9) and np.all(x <= d["upper_bounds"] + 1e-9)
    assert np.max(np.abs(d["a_eq"] @ x - d["b_eq"])) < 1e-7
    ax = d["a_ineq"] @ x
    assert np.all(ax <= d["b_upper"] + 1e-7) and np.all(ax >= d["b_lower"] - 1e-7)


def test_sc50a_objective_is_the_perplex_value():
    d = get_problem("sc50a", data_dir=NETLIB)
    assert abs(d["cost_vector"] @ d["solution"] - (-146650 / 2271)) < 1e-12  # data/perPlex/sc50a.txt "Objvalue"


FREE = """NAME TINY
ROWS
 N COST
 L LIM1
 G LIM2
 E MYEQN
COLUMNS
 X COST 1 LIM1 1
 X LIM2 1
 Y COST 2 LIM1 1
 Y MYEQN -1
 Z COST -1 MYEQN 1
RHS
 RHS LIM1 4 LIM2 1
 RHS MYEQN 7
RANGES
 RNG LIM1 2.5 LIM2 3
BOUNDS
 UP BND X 4
 LO BND Y -1
 UP BND Y 1
 MI BND Z
ENDATA
"""


def test_free_format_and_ranges():
    d = mps_parser(io.StringIO(FREE))
    assert d["variable_names"] == ["X", "Y", "Z"] and d["costname"] == "COST" and d["problem_name"] == "TINY"
    assert np.array_equal(d["cost_vector"], [1, 2, -1])
    assert np.array_equal(d["lower_bounds"], [0, -1, -np.inf]) and np.array_equal(d["upper_bounds"], [4, 1, np.inf])
    assert np.array_equal(d["a_ineq"].toarray(), [[1, 1, 0], [1, 0, 0]]) and np.array_equal(d["a_eq"].toarray(), [[0, -1, 1]])
    assert np.array_equal(d["b_eq"], [7])
    assert np.array_equal(d["b_upper"], [4, 4]) and np.array_equal(d["b_lower"], [1.5, 1])  # L: [rhs-|R|, rhs], G: [rhs, rhs+|R|]
    assert d["solution"] is None


def test_rejects_what_the_reference_rejects():
    with pytest.raises(MpsError):
        mps_parser(io.StringIO("NAME X\nROWS\n N C\n L R\nCOLUMNS\n A C 1 R 1\nBOUNDS\n BV B A\nENDATA\n"))
    with pytest.raises(MpsError):
        mps_parser(io.StringIO("NAME X\nROWS\n N C\n L R\n L R\nENDATA\n"))
    with pytest.raises(FileNotFoundError):
        get_problem("NOSUCHLP", data_dir=NETLIB)


def test_netlib_problem_through_the_modelling_layer():
    """tests/test_netlib.py:19-48 of the reference: build the SparseLP from the parsed dictionary; the perPlex point checks out."""
    from pysparselp_amd.SparseLP import SparseLP

    d = get_problem("AFIRO", data_dir=NETLIB)
    gt = d["solution"]
    lp = SparseLP()
    lp.add_variables_array(len(d["cost_vector"]), lower_bounds=d["lower_bounds"],
                           upper_bounds=np.minimum(d["upper_bounds"], np.max(gt) * 2), costs=d["cost_vector"])
    lp.add_equality_constraints_sparse(d["a_eq"], d["b_eq"])
    lp.add_inequality_constraints_sparse(d["a_ineq"], d["b_lower"], d["b_upper"])
    assert lp.check_solution(gt)
    assert abs(lp.costsvector @ gt - (-406659 / 875)) < 1e-9  # data/perPlex/afiro.txt "Objvalue"


@pytest.mark.parametrize("name", ["AFIRO", "KB2", "SC50A", "SC50B", "SC105"])
def test_save_mps_round_trips_through_the_reader(name, tmp_path):
    """``SparseLP.save_mps`` (reference SparseLP.py:280-366, which cannot run as written: :310): the five netlib problems
    written and read back give the same LP array for array (rows and variables are renamed E<i> / I<i> / X<j> in order)."""
    from conftest import lp_from_golden  # noqa: F401  (same construction, from the parsed dictionary)
    from pysparselp_amd.SparseLP import SparseLP

    d = get_problem(name, data_dir=NETLIB)
    lp = SparseLP()
    lp.nb_variables = d["cost_vector"].size
    lp.costsvector, lp.lower_bounds, lp.upper_bounds = d["cost_vector"], d["lower_bounds"], d["upper_bounds"]
    lp.is_integer = np.zeros(lp.nb_variables, dtype=bool)
    lp.a_equalities, lp.b_equalities = scipy.sparse.csr_matrix(d["a_eq"]), d["b_eq"]
    lp.a_inequalities, lp.b_lower, lp.b_upper = scipy.sparse.csr_matrix(d["a_ineq"]), d["b_lower"], d["b_upper"]
    path = tmp_path / (name + ".mps")
    lp.save_mps(str(path))
    with open(path) as f:
        back = mps_parser(f)
    for key in ("cost_vector", "upper_bounds", "lower_bounds", "b_eq", "b_lower", "b_upper"):
        assert np.array_equal(back[key], d[key]), key
    for key in ("a_eq", "a_ineq"):
        m0, m1 = scipy.sparse.csr_matrix(d[key]), scipy.sparse.csr_matrix(back[key])
        m0.sort_indices()
        m1.sort_indices()
        assert m0.shape == m1.shape and np.array_equal(m0.indptr, m1.indptr) and np.array_equal(m0.indices, m1.indices)
        assert np.array_equal(m0.data, m1.data)


def test_save_mps_two_sided_rows_free_and_fixed_variables(tmp_path):
    from pysparselp_amd.SparseLP import SparseLP

    lp = SparseLP()
    x = lp.add_variables_array(5, lower_bounds=None, upper_bounds=None, costs=np.array([1.0, -2.5, 0.1, 0.0, 1e-17]))
    lp.set_bounds_on_variables(x[1:2], 0.25, 0.25)          # fixed
    lp.set_bounds_on_variables(x[2:3], -np.inf, 3.0)        # MI + UP
    lp.set_bounds_on_variables(x[3:4], 1.0 / 3.0, np.inf)   # LO only
    lp.set_bounds_on_variables(x[4:5], 0.0, np.inf)         # the MPS default: nothing written
    a = scipy.sparse.csr_matrix(np.array([[1.0, 2.0, 0, 0, 0.1], [0, 1.0 / 7.0, -1.0, 0, 0], [0, 0, 0, 5.0, 1.0]]))
    lp.add_inequality_constraints_sparse(a, lower_bounds=np.array([-1.0, -np.inf, 0.5]), upper_bounds=np.array([2.0, 4.0, np.inf]))
    lp.add_equality_constraints_sparse(scipy.sparse.csr_matrix(np.array([[1.0, 1.0, 1.0, 1.0, 1.0]])), np.array([2.0 / 3.0]))
    path = tmp_path / "small.mps"
    lp.save_mps(str(path))
    with open(path) as f:
        back = mps_parser(f)
    assert np.array_equal(back["cost_vector"], lp.costsvector)
    assert np.array_equal(back["lower_bounds"], lp.lower_bounds) and np.array_equal(back["upper_bounds"], lp.upper_bounds)
    assert np.array_equal(back["b_eq"], lp.b_equalities)
    assert np.array_equal(back["b_upper"], lp.b_upper) and np.array_equal(back["b_lower"], lp.b_lower)
    assert np.array_equal(scipy.sparse.csr_matrix(back["a_ineq"]).toarray(), lp.a_inequalities.toarray())
    assert np.array_equal(scipy.sparse.csr_matrix(back["a_eq"]).toarray(), lp.a_equalities.toarray())


def test_save_mps_refuses_a_two_sided_row_it_cannot_write_exactly(tmp_path):
    """A row whose two bounds are reproduced neither by rhs - range nor by rhs + range in fp64 would come back off by an ulp:
    the writer refuses instead of breaking its bit-for-bit promise."""
    from pysparselp_amd.SparseLP import SparseLP

    rng = np.random.RandomState(0)
    lo = up = None
    for _ in range(100000):  # such pairs are common: search one
        l, u = rng.randn(), rng.randn()
        l, u = min(l, u), max(l, u)
        if u - (u - l) != l and l + (u - l) != u:
            lo, up = l, u
            break
    assert lo is not None
    lp = SparseLP()
    lp.add_variables_array(2, lower_bounds=0, upper_bounds=1, costs=np.array([1.0, 2.0]))
    lp.add_inequality_constraints_sparse(scipy.sparse.csr_matrix(np.array([[1.0, 1.0]])), lower_bounds=np.array([lo]),
                                         upper_bounds=np.array([up]))
    with pytest.raises(ValueError, match="cannot be written exactly"):
        lp.save_mps(str(tmp_path / "inexact.mps"))
