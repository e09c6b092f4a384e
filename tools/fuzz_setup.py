"""Randomised cross-check of the device-side setup transforms (csrc/slp_spgemm.hip) against the numpy restatement
(pysparselp_amd/tools.py) and the oracle: row scaling, slack standard form (unsorted rows, duplicates, empty rows, entries that
underflow), second scaling, M = 2 A^T A + 3 I, column removal -- bit for bit.   python tools/fuzz_setup.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle  # noqa: E402
from pysparselp_amd import tools  # noqa: E402
from pysparselp_amd.device import DeviceMatrix  # noqa: E402


def same(dev, ref, what):
    got = dev.download()
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices), what
    assert np.array_equal(got.data, ref.data), (what, float(np.max(np.abs(got.data - ref.data))))


def run(cases, seed):
    rng = np.random.RandomState(seed)
    for case in range(cases):
        n, me, mi = int(rng.randint(1, 80)), int(rng.randint(0, 60)), int(rng.randint(1, 90))
        maxlen = int(rng.choice([3, 6, 12]))

        def block(rows):
            lens = rng.randint(0, maxlen, size=rows)
            indptr = np.concatenate(([0], np.cumsum(lens)))
            indices = rng.randint(0, n, size=indptr[-1]).astype(np.int32)
            data = np.round(rng.randn(indptr[-1]) * 8) / 8
            if data.size > 3 and rng.rand() < 0.5:
                data[0], data[1] = 5e-324, 1e150
            return scipy.sparse.csr_matrix((data, indices, indptr), shape=(rows, n))

        a_eq, a_in = block(me), block(mi)
        be = rng.randn(me)
        bl = np.where(rng.rand(mi) < 0.3, -np.inf, rng.randn(mi))
        bu = np.where(rng.rand(mi) < 0.3, np.inf, rng.randn(mi) + 3)
        c, lb, ub, x0 = rng.randn(n), -rng.rand(n), rng.rand(n), rng.randn(n)
        hae, hbe = tools.precondition_constraints(a_eq, be)
        hai, hbl, hbu = tools.precondition_constraints(a_in, bl, bu)
        _, ha2, hb2, _, _, hx = tools.convert_to_standard_form_with_bounds(c, hae if me else None, hbe if me else None, hai, hbl, hbu, lb, ub, x0)
        ha3, hb3 = tools.precondition_constraints(ha2, hb2)
        dae, dai = DeviceMatrix.from_csr(a_eq), DeviceMatrix.from_csr(a_in)
        dae2, dbe, _ = dae.precondition_rows(be)
        dai2, dbl, dbu = dai.precondition_rows(bl, bu)
        same(dae2, hae, (case, "A_eq scaled"))
        same(dai2, hai, (case, "A_ineq scaled"))
        assert np.array_equal(dbe, hbe, equal_nan=True) and np.array_equal(dbl, hbl, equal_nan=True) and np.array_equal(dbu, hbu, equal_nan=True), case
        da2 = DeviceMatrix.standard_form(dae2 if me else None, dai2)
        same(da2, ha2, (case, "standard form"))
        da3, db3, _ = da2.precondition_rows(hb2)
        same(da3, ha3, (case, "A3"))
        assert np.array_equal(db3, hb3, equal_nan=True), case
        if np.all(np.isfinite(ha3.data)):
            m_ref = oracle.normal_matrix(oracle.as_csr(ha3.tocsr()), 2.0, 3.0)
            dm = da3.normal_matrix(2.0, 3.0)
            got = dm.download()
            assert np.array_equal(got.indptr, m_ref.indptr) and np.array_equal(got.indices, m_ref.indices) and np.array_equal(got.data, m_ref.data), (case, "M")
            dm.close()
        keep = rng.rand(n) > 0.3
        red, a_shift = dai.remove_columns(keep, np.where(keep, 0.0, rng.randn(n)))
        same(red, a_in[:, keep], (case, "remove_columns"))
        for m in (dae, dai, dae2, dai2, da2, da3, red):
            m.close()
    return cases


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=200)
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args()
    print("ok:", run(args.cases, args.seed), "cases")
