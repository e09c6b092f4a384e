"""Netlib problems from local files (reference netlib.py:36-72 without the download).

``get_problem("SC105")`` reads ``SC105.SIF`` and, when present, the perPlex exact solution
``sc105.txt`` from ``data_dir`` (argument, else ``$PYSPARSELP_NETLIB_DIR``, else the
``data/netlib`` + ``data/perPlex`` folders next to this file -- the reference's layout).
The reference fetches missing files from ftp.numerical.rl.ac.uk / zib.de; this build runs
without a network, so a missing file is an error that says where to put it.
"""
import gzip
import os

from .MPSparser import mps_parser


def _find(names, folders):
    for folder in folders:
        for name in names:
            path = os.path.join(folder, name)
            if os.path.isfile(path):
                return path
    return None


def get_problem(problem_name, data_dir=None):
    here = os.path.dirname(os.path.abspath(__file__))
    folders = [d for d in (data_dir, os.environ.get("PYSPARSELP_NETLIB_DIR")) if d]
    lp_folders = folders + [os.path.join(here, "data", "netlib")]
    sol_folders = folders + [os.path.join(here, "data", "perPlex")]
    up, low = problem_name.upper(), problem_name.lower()
    filename_lp = _find([up + ".SIF", up + ".mps", low + ".mps", up + ".MPS"], lp_folders)
    if filename_lp is None:
        raise FileNotFoundError(
            f"{up}.SIF not found in {lp_folders}; get it from the netlib LP collection "
            "(ftp://ftp.numerical.rl.ac.uk/pub/cuter/netlib/) and point data_dir or PYSPARSELP_NETLIB_DIR at it")
    filename_sol = _find([low + ".txt", low + ".txt.gz"], sol_folders)
    with open(filename_lp, "r") as file_lp:
        if filename_sol is None:
            return mps_parser(file_lp, None)
        opener = gzip.open if filename_sol.endswith(".gz") else open
        with opener(filename_sol, "rt") as f_sol:
            return mps_parser(file_lp, f_sol)
