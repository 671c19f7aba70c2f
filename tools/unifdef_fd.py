#!/usr/bin/env python3
"""tools/unifdef_fd.py -- partial C preprocessor for RETIRING A/B switches: given `NAME=value` pairs, every #if / #ifdef / #ifndef / #elif
that the values decide is resolved in the source text (the dead branch and the directives go, the live branch stays), the switches'
own `#ifndef NAME / #define NAME v / #endif` blocks go (`NAME=undef`: a name that is never defined), and conditions the values do not decide are left alone (with the known names
replaced by their values, flagged on stderr for a hand edit).  Uses of a retired name OUTSIDE preprocessor lines (`if constexpr (FD_X)`,
template defaults) are reported, not touched.  Design tool.

    python3 tools/unifdef_fd.py FD_KNOCK=0 FD_WT_PAIRS=0 -- fundsp_amd/csrc/fd_device.hpp fundsp_amd/csrc/fd_nodes.hpp
"""
import itertools
import re
import sys

IDENT = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")


def evaluate(expr, known):
    """-> (True | False | None, expression text with the known names substituted)"""
    expr = re.sub(r"//.*$", "", expr)
    expr = re.sub(r"/\*.*?\*/", "", expr).strip()
    atoms = {}

    def atom(text):
        return atoms.setdefault(text, f"__a{len(atoms)}")

    def sub_defined(m):
        name = m.group(1) or m.group(2)
        return ("0" if known[name] == "undef" else "1") if name in known else atom(f"defined({name})")
    work = re.sub(r"defined\s*(?:\(\s*([A-Za-z_]\w*)\s*\)|([A-Za-z_]\w*))", sub_defined, expr)
    shown = re.sub(r"defined\s*(?:\(\s*([A-Za-z_]\w*)\s*\)|([A-Za-z_]\w*))",
                   lambda m: ("0" if known[m.group(1) or m.group(2)] == "undef" else "1") if (m.group(1) or m.group(2)) in known else m.group(0), expr)
    value_of = lambda n: "0" if known[n] == "undef" else str(known[n])   # an undefined name is 0 in #if arithmetic
    shown = IDENT.sub(lambda m: value_of(m.group(0)) if m.group(0) in known else m.group(0), shown)

    def sub_ident(m):
        t = m.group(0)
        if t.startswith("__a"):
            return t
        if t in known:
            return value_of(t)
        return atom(t)
    work = IDENT.sub(sub_ident, work)
    py = work.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    py = re.sub(r"(\d+)[uUlL]+\b", r"\1", py)
    names = sorted(set(atoms.values()))
    results = set()
    try:
        for combo in itertools.product((0, 1, 5), repeat=len(names)):
            results.add(bool(eval(py, {"__builtins__": {}}, dict(zip(names, combo)))))
    except Exception:
        return None, shown
    if len(results) == 1:
        return results.pop(), shown
    return None, shown


def process(text, known, path):
    out, notes = [], []
    # stack entries: dict(emit=parent emitting?, state='undecided'|'taken'|'open', keep=directives kept?, live=this branch's body is emitted)
    stack = []
    lines = text.split("\n")
    i = 0

    def emitting():
        return all(f["live"] for f in stack)
    while i < len(lines):
        line = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$", line)
        if not m:
            if emitting():
                if not re.match(r"\s*#", line):
                    for name in known:
                        if re.search(r"\b%s\b" % re.escape(name), re.sub(r"//.*$", "", line)):
                            notes.append(f"{path}:{i + 1}: use outside the preprocessor: {line.strip()[:140]}")
                elif re.match(r"\s*#\s*define\s+(\w+)", line) and re.match(r"\s*#\s*define\s+(\w+)", line).group(1) in known:
                    notes.append(f"{path}:{i + 1}: stray #define of a retired name: {line.strip()[:120]}")
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            parent = emitting()
            if kind == "ifdef":
                cond = f"defined({rest.strip().split()[0]})"
            elif kind == "ifndef":
                cond = f"!defined({rest.strip().split()[0]})"
            else:
                cond = rest
            val, shown = evaluate(cond, known)
            touched = any(re.search(r"\b%s\b" % re.escape(n), cond) for n in known)
            frame = {"parent": parent, "decided": False, "keep": False, "live": False, "first_kept": False}
            if not parent:
                frame["live"] = False
            elif val is True:
                frame.update(decided=True, live=True)
            elif val is False:
                frame.update(live=False)
            else:
                frame.update(keep=True, live=True, first_kept=True)
                if touched:
                    out.append(f"#if {shown.strip()}")
                    notes.append(f"{path}:{i + 1}: condition only partly decided: #if {shown.strip()}")
                else:
                    out.append(line)
            stack.append(frame)
        elif kind == "elif":
            f = stack[-1]
            if not f["parent"]:
                pass
            elif f["decided"]:
                f["live"] = False
            else:
                val, shown = evaluate(rest, known)
                touched = any(re.search(r"\b%s\b" % re.escape(n), rest) for n in known)
                if val is True:
                    if f["keep"]:
                        out.append("#else")
                        f["live"] = True
                        f["decided"] = True
                    else:
                        f.update(decided=True, live=True)
                elif val is False:
                    f["live"] = False
                else:
                    if f["keep"]:
                        out.append(f"#elif {shown.strip()}" if touched else line)
                    else:
                        out.append(f"#if {shown.strip()}" if touched else re.sub(r"#\s*elif", "#if", line))
                        f["keep"] = True
                    if touched:
                        notes.append(f"{path}:{i + 1}: condition only partly decided: {shown.strip()}")
                    f["live"] = True
        elif kind == "else":
            f = stack[-1]
            if not f["parent"]:
                pass
            elif f["decided"]:
                f["live"] = False
            elif f["keep"]:
                out.append(line)
                f["live"] = True
            else:
                f.update(decided=True, live=True)
        else:  # endif
            f = stack.pop()
            if f["parent"] and f["keep"]:
                out.append(line)
        i += 1
    return "\n".join(out), notes


def main(argv):
    if "--" not in argv:
        sys.exit(__doc__)
    k = argv.index("--")
    known = {}
    for a in argv[:k]:
        name, _, value = a.partition("=")
        known[name] = int(value) if re.fullmatch(r"-?\d+", value) else value
    for path in argv[k + 1:]:
        text = open(path).read()
        new, notes = process(text, known, path)
        if new != text:
            open(path, "w").write(new)
        for n in notes:
            print(n, file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
