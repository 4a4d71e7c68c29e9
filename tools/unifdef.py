#!/usr/bin/env python3
"""Resolve preprocessor conditionals on a fixed set of macros and delete the losing arms (a small `unifdef`):

    python tools/unifdef.py FILE NAME=VALUE [NAME=VALUE ...] [--keep-define]

Handles `#if EXPR` / `#elif EXPR` / `#else` / `#endif` whose EXPR mentions only the listed names, integers and the
operators of C integer expressions, and `#ifdef NAME` / `#ifndef NAME` of a listed name.  A block of the form

    #ifndef NAME
    #define NAME value      (one or more lines, comments allowed)
    #endif

becomes the bare `#define NAME value` lines for the listed value (the macro stays usable in C code such as
`if (NAME)`), so the switch can no longer be thrown from the command line.  Conditionals on anything else are left alone.
Used once per round to remove A/B arms from the product sources after the measurement; the arms' record is the diff
kept under profiles/."""
import re
import sys


def main():
    path = sys.argv[1]
    defs = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
    lines = open(path).read().split("\n")
    out = []
    # stack entries: [known, taken_before, active_now, parent_active]
    stack = []

    def active():
        return all(e[2] for e in stack if e[0])

    def evaluate(expr):
        names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
        if not names or not names <= set(defs):
            return None
        e = expr
        for n in sorted(names, key=len, reverse=True):
            e = re.sub(r"\b%s\b" % n, "(%s)" % defs[n], e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        e = re.sub(r"(\d+)[uU][lL]*\b", r"\1", e)
        return bool(eval(e))

    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line)
        if not m:
            if active():
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), re.sub(r"/\*.*?\*/|//.*", "", m.group(2)).strip()
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "ifndef" and rest in defs:
                # the "#ifndef NAME / #define NAME v / #endif" idiom -> bare defines
                j = i + 1
                body = []
                while not re.match(r"\s*#\s*endif\b", lines[j]):
                    body.append(lines[j])
                    j += 1
                if all(re.match(r"\s*(#\s*define\b.*|//.*|)$", b) for b in body):
                    if active():
                        for b in body:
                            mm = re.match(r"(\s*#\s*define\s+%s\s+)(\S+)(.*)" % re.escape(rest), b)
                            out.append(mm.group(1) + defs[rest] + mm.group(3) if mm else b)
                    i = j + 1
                    continue
                val = False
            elif kind == "ifdef" and rest in defs:
                val = True
            elif kind == "if":
                val = evaluate(rest)
            else:
                val = None
            if val is None:
                stack.append([False, False, True])
                if active():
                    out.append(line)
            else:
                stack.append([True, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if active():
                    out.append(line)
            else:
                val = evaluate(rest)
                assert val is not None, "mixed #elif at line %d" % (i + 1)
                top[2] = (not top[1]) and val
                top[1] = top[1] or val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if active():
                    out.append(line)
            else:
                top[2] = not top[1]
                top[1] = True
        else:
            top = stack.pop()
            if not top[0] and active():
                out.append(line)
        i += 1
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
