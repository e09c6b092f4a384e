"""MPS / SIF linear-program reader and perPlex solution reader (host side, numpy only).

Drop-in for ``pysparselp.MPSparser.mps_parser`` (reference MPSparser.py:10-271): same
call, same dictionary -- ``cost_vector, upper_bounds, lower_bounds, a_eq, b_eq,
a_ineq, b_lower, b_upper, problem_name, costname, solution`` with the rows numbered
in the order of the ROWS section (equalities and inequalities counted separately)
and the variables in the order of their first appearance in COLUMNS.  ``a_eq`` /
``a_ineq`` are scipy DOK matrices like the reference's.

Conventions mirrored from the reference: an ``L`` row is ``-inf <= a x <= rhs``, a ``G`` row
``rhs <= a x <= +inf``, missing right-hand sides are 0, variables are ``[0, +inf)`` unless a
BOUNDS line says otherwise (``UP, LO, FX, FR, MI, PL``), integer bound types are rejected.
Differences, all on inputs the reference cannot read: fields are split on white space when
that is unambiguous (free-format files), with the fixed columns 2-3 / 5-12 / 15-22 / 25-36 /
40-47 / 50-61 as the fallback for names that contain blanks; a RANGES section is applied
(the reference raises "not coded yet", MPSparser.py:70-72); a negative ``UP`` bound on a
variable whose lower bound is still the default 0 is left as written, like the reference.
"""
import numpy as np
import scipy.sparse

_SECTIONS = ("NAME", "ROWS", "COLUMNS", "RHS", "RANGES", "BOUNDS", "ENDATA", "OBJSENSE", "OBJSENSE", "GROUPS", "VARIABLES",
             "CONSTANTS", "ELEMENT", "GROUP", "OBJECT")
_NO_VALUE_BOUNDS = ("FR", "MI", "PL")


def _fixed_fields(line):
    """The six fixed-format fields of a data line (MPS columns 2-3, 5-12, 15-22, 25-36, 40-47, 50-61)."""
    line = line.ljust(61)
    return [line[1:3].strip(), line[4:12].strip(), line[14:22].strip(), line[24:36].strip(), line[39:47].strip(),
            line[49:61].strip()]


def _is_number(tok):
    try:
        float(tok)
        return True
    except ValueError:
        return False


class MpsError(ValueError):
    pass


def _data_fields(line, section):
    """[type-or-empty, name1, name2, value1, name3, value2] for a data line of `section`."""
    tok = line.split()
    if section == "ROWS":
        if len(tok) == 2:
            return [tok[0], tok[1], "", "", "", ""]
        return _fixed_fields(line)
    if section in ("COLUMNS", "RHS", "RANGES"):
        # name, then (row, value) once or twice; RHS / RANGES lines may leave the set name out
        if len(tok) in (3, 5) and _is_number(tok[2]) and (len(tok) == 3 or _is_number(tok[4])):
            return ["", tok[0], tok[1], tok[2], tok[3] if len(tok) == 5 else "", tok[4] if len(tok) == 5 else ""]
        if section != "COLUMNS" and len(tok) in (2, 4) and _is_number(tok[1]) and (len(tok) == 2 or _is_number(tok[3])):
            return ["", "", tok[0], tok[1], tok[2] if len(tok) == 4 else "", tok[3] if len(tok) == 4 else ""]
        return _fixed_fields(line)
    if section == "BOUNDS":
        if len(tok) == 4 and _is_number(tok[3]):
            return [tok[0], tok[1], tok[2], tok[3], "", ""]
        if len(tok) == 3 and tok[0] in _NO_VALUE_BOUNDS:
            return [tok[0], tok[1], tok[2], "", "", ""]
        if len(tok) == 3 and _is_number(tok[2]):  # bound set name left out
            return [tok[0], "", tok[1], tok[2], "", ""]
        return _fixed_fields(line)
    return _fixed_fields(line)


def mps_parser(f, fsol=None):
    """Parse an MPS (or MPS-compatible SIF) file object; `fsol`: optional perPlex solution file object."""
    row_type, row_id, row_order = {}, {}, []
    nb_eq = nb_ineq = 0
    b_eq, b_lower, b_upper = {}, {}, {}
    var_id, var_names = {}, []
    cost, lo, up = [], [], []
    eq_entries, ineq_entries = [], []
    problem_name, costname = "", None
    section = None
    for raw in f:
        line = raw.rstrip("\n").rstrip("\r")
        if not line.strip() or line.startswith("*"):
            continue
        if not line[0].isspace():  # section header
            head = line.split()
            section = head[0].upper()
            if section == "NAME":
                problem_name = head[1] if len(head) > 1 else ""
            if section == "ENDATA":
                break
            if section not in ("NAME", "ROWS", "COLUMNS", "RHS", "RANGES", "BOUNDS"):
                raise MpsError(f"unsupported MPS section {head[0]!r}")
            continue
        t = _data_fields(line, section)
        if section == "ROWS":
            kind, name = t[0].upper(), t[1]
            if name in row_type:
                raise MpsError(f"row {name!r} declared twice")
            if kind not in ("N", "L", "G", "E"):
                raise MpsError(f"unknown row type {kind!r}")
            row_type[name] = kind
            row_order.append(name)
            if kind == "N":
                if costname is None:
                    costname = name
            elif kind == "E":
                row_id[name] = nb_eq
                b_eq[nb_eq] = 0.0
                nb_eq += 1
            else:
                row_id[name] = nb_ineq
                b_lower[nb_ineq], b_upper[nb_ineq] = (0.0, np.inf) if kind == "G" else (-np.inf, 0.0)
                nb_ineq += 1
        elif section == "COLUMNS":
            name = t[1]
            if t[2].upper() == "'MARKER'" or t[2] == "MARKER":
                raise MpsError("integer markers are not supported (the reference raises on integer constraints too)")
            if name not in var_id:
                var_id[name] = len(var_names)
                var_names.append(name)
                cost.append(0.0)
                lo.append(0.0)
                up.append(np.inf)
            j = var_id[name]
            for rname, val in ((t[2], t[3]), (t[4], t[5])):
                if rname == "":
                    continue
                if rname not in row_type:
                    raise MpsError(f"COLUMNS refers to the undeclared row {rname!r}")
                v = float(val)
                kind = row_type[rname]
                if kind == "N":
                    if rname == costname:
                        cost[j] = v
                elif kind == "E":
                    eq_entries.append((row_id[rname], j, v))
                else:
                    ineq_entries.append((row_id[rname], j, v))
        elif section == "RHS":
            for rname, val in ((t[2], t[3]), (t[4], t[5])):
                if rname == "":
                    continue
                if rname not in row_type:
                    raise MpsError(f"RHS refers to the undeclared row {rname!r}")
                kind, v = row_type[rname], float(val)
                if kind == "N":
                    if rname == costname:
                        raise MpsError("a right-hand side on the objective row is not supported")
                elif kind == "L":
                    b_upper[row_id[rname]] = v
                elif kind == "G":
                    b_lower[row_id[rname]] = v
                else:
                    b_eq[row_id[rname]] = v
        elif section == "RANGES":
            for rname, val in ((t[2], t[3]), (t[4], t[5])):
                if rname == "":
                    continue
                kind, r = row_type[rname], float(val)
                if kind == "L":
                    b_lower[row_id[rname]] = b_upper[row_id[rname]] - abs(r)
                elif kind == "G":
                    b_upper[row_id[rname]] = b_lower[row_id[rname]] + abs(r)
                elif kind == "E":
                    raise MpsError("RANGES on an equality row would turn it into an inequality: not supported")
        elif section == "BOUNDS":
            kind, vname = t[0].upper(), t[2]
            if vname not in var_id:
                raise MpsError(f"BOUNDS refers to the unknown variable {vname!r}")
            j = var_id[vname]
            if kind == "UP":
                up[j] = float(t[3])
            elif kind == "LO":
                lo[j] = float(t[3])
            elif kind == "FX":
                lo[j] = up[j] = float(t[3])
            elif kind == "FR":
                lo[j], up[j] = -np.inf, np.inf
            elif kind == "MI":
                lo[j] = -np.inf
            elif kind == "PL":
                up[j] = np.inf
            else:
                raise MpsError(f"bound type {kind!r} (integer / binary) is not supported")
    nb_var = len(var_names)

    def dok(entries, nrows):
        m = scipy.sparse.dok_matrix((nrows, nb_var))
        for i, j, v in entries:
            m[i, j] = v
        return m

    r = {
        "cost_vector": np.array(cost, dtype=float),
        "upper_bounds": np.array(up, dtype=float),
        "lower_bounds": np.array(lo, dtype=float),
        "a_eq": dok(eq_entries, nb_eq),
        "b_eq": np.array([b_eq[i] for i in range(nb_eq)], dtype=float),
        "a_ineq": dok(ineq_entries, nb_ineq),
        "b_lower": np.array([b_lower[i] for i in range(nb_ineq)], dtype=float),
        "b_upper": np.array([b_upper[i] for i in range(nb_ineq)], dtype=float),
        "problem_name": problem_name,
        "costname": costname,
        "variable_names": list(var_names),
        "solution": None,
    }
    if fsol is not None:
        r["solution"] = perplex_solution(fsol, var_id, r["lower_bounds"], r["upper_bounds"])
    return r


def _perplex_number(text):
    """'0.255e2 = 51/2' -> 25.5 (the exact fraction when it evaluates, else the decimal)."""
    dec, _, frac = text.partition("=")
    frac = frac.strip()
    val = np.nan
    if frac:
        num, _, den = frac.partition("/")
        try:
            val = float(num) / float(den) if den else float(num)
        except (ValueError, ZeroDivisionError):
            val = np.nan
    return float(dec) if np.isnan(val) else val


def perplex_solution(fsol, var_id, lower_bounds, upper_bounds):
    """Primal values from a perPlex 1.00 solution file (MPSparser.py:196-262): `V Value` lines, or the bound a
    variable sits on (`V State : on lower / on upper / on both`)."""
    sol = np.full(len(var_id), np.nan)
    part, j = None, None
    for raw in fsol:
        line = raw.decode() if isinstance(raw, bytes) else raw
        line = line.rstrip("\n")
        if line.startswith("- EOF"):
            break
        if line.startswith("- Variables"):
            part = "V"
            continue
        if line.startswith("- Constraints"):
            part = "C"
            continue
        if part != "V":
            continue
        if line.startswith("V Name"):
            j = var_id[line.split(": ", 1)[1].strip()]
        elif line.startswith("V Value"):
            sol[j] = _perplex_number(line.split(":", 1)[1])
        elif line.startswith("V State    : on lower"):
            sol[j] = lower_bounds[j]
        elif line.startswith("V State    : on upper"):
            sol[j] = upper_bounds[j]
        elif line.startswith("V State    : on both"):
            assert upper_bounds[j] == lower_bounds[j]
            sol[j] = upper_bounds[j]
    return sol
