"""Removes the kernel-lab preprocessor switches from a source file (a minimal unifdef for the macros named below): the shipped
sources carry none of them; the lab forms live on as patches under tools/lab/patches/ (``patch -p1 < ...`` puts them back for a
``make variant`` build).

    python tools/lab/strip_lab_switches.py in.hip out.hip

UNDEFINED macros are taken as not defined, VALUES as defined to the given number (their ``#ifndef X / #define X v / #endif`` default
blocks go too).  Conditionals on anything else are left alone."""
import re
import sys

UNDEFINED = {"SLP_TALL_ABL", "SLP_TALL_FLAT", "SLP_TALL_X64", "SLP_TALL_WHOLE_ISSUE", "SLP_TALL_FULL_ISSUE", "SLP_TALL_BUILD_PROF",
             "SLP_TALL_DEAL_CEIL", "SLP_GS_ABLATE", "SLP_GS_BANDS_ABLATE_FETCH", "SLP_GS_BANDS_ABLATE_PUBLISH", "SLP_GS_BANDS_ABLATE_SC1"}
VALUES = {"SLP_TALL_XLOAD": 0, "SLP_TALL_STAGE": 0}
KNOWN = UNDEFINED | set(VALUES)


def evaluate(expr):
    """True / False, or None when the expression names a macro this script does not know."""
    expr = re.sub(r"//.*$", "", expr).strip()
    names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
    if not names or not names <= KNOWN:
        return None
    py = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "True" if m.group(1) in VALUES else "False", expr)
    py = re.sub(r"defined\s+(\w+)", lambda m: "True" if m.group(1) in VALUES else "False", py)
    py = py.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    py = re.sub(r"\b(\w+)\b", lambda m: str(VALUES.get(m.group(1), 0)) if m.group(1) in KNOWN else m.group(1), py)
    return bool(eval(py))  # noqa: S307 -- numbers, comparisons and boolean operators only


def strip(lines):
    out = []
    stack = []   # per open conditional: [ours, emitting, taken]; ours = False: left alone
    for line in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        live = all(f[1] for f in stack if f[0])
        if not m:
            if live:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                v = evaluate(rest)
            else:
                name = re.match(r"\s*(\w+)", rest).group(1)
                v = None if name not in KNOWN else ((name in VALUES) == (kind == "ifdef"))
            if v is None:
                stack.append([False, True, True])
                if live:
                    out.append(line)
            else:
                stack.append([True, v, v])
        elif kind == "elif":
            f = stack[-1]
            if not f[0]:
                if live:
                    out.append(line)
            else:
                v = evaluate(rest)
                assert v is not None, line
                f[1] = (not f[2]) and v
                f[2] = f[2] or v
        elif kind == "else":
            f = stack[-1]
            if not f[0]:
                if live:
                    out.append(line)
            else:
                f[1] = not f[2]
                f[2] = True
        else:
            f = stack.pop()
            if not f[0] and all(g[1] for g in stack if g[0]):
                out.append(line)
    assert not stack
    return out


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        src = f.readlines()
    with open(sys.argv[2], "w") as f:
        f.writelines(strip(src))
