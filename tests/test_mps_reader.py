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
