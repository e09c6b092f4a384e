"""Fixtures for the MPS / perPlex reader: what the reference's parser
(MPSparser.py:10-271, netlib.py:36-72) returns for the five netlib problems it ships.

Run in the build container only (imports the reference from /root/reference through make_golden's /tmp build):

    python tests/golden/make_netlib_golden.py

Writes tests/golden/netlib_parsed.npz (arrays only) and copies the public netlib data files the
reference's tests read (pysparselp/data/netlib/*.SIF, pysparselp/data/perPlex/*.txt) to
tests/golden/netlib/ -- data files, not source.
"""
import os
import shutil
import sys

import numpy as np
import scipy.sparse

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

NAMES = ("AFIRO", "KB2", "SC50A", "SC50B", "SC105")


def main():
    make_golden.build_reference()
    make_golden.install_shims()
    from pysparselp.netlib import get_problem

    out = {}
    dst = os.path.join(HERE, "netlib")
    os.makedirs(dst, exist_ok=True)
    for name in NAMES:
        d = get_problem(name)
        for key in ("cost_vector", "upper_bounds", "lower_bounds", "b_eq", "b_lower", "b_upper", "solution"):
            out[f"{name}_{key}"] = np.asarray(d[key], dtype=np.float64)
        for key in ("a_eq", "a_ineq"):
            m = scipy.sparse.csr_matrix(d[key])
            m.sort_indices()
            out[f"{name}_{key}_indptr"], out[f"{name}_{key}_indices"] = m.indptr.astype(np.int64), m.indices.astype(np.int32)
            out[f"{name}_{key}_data"], out[f"{name}_{key}_shape"] = m.data, np.array(m.shape)
        out[f"{name}_names"] = np.array([d["problem_name"].strip(), d["costname"].strip()])
        ref_data = os.path.join(make_golden.REF_SRC, "pysparselp", "data")
        shutil.copyfile(os.path.join(ref_data, "netlib", name + ".SIF"), os.path.join(dst, name + ".SIF"))
        shutil.copyfile(os.path.join(ref_data, "perPlex", name.lower() + ".txt"), os.path.join(dst, name.lower() + ".txt"))
        print(name, d["a_eq"].shape, d["a_ineq"].shape, float(d["cost_vector"] @ d["solution"]))
    np.savez_compressed(os.path.join(HERE, "netlib_parsed.npz"), **out)


if __name__ == "__main__":
    main()
