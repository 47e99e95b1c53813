#!/usr/bin/env python3
"""Minimal `unifdef -U`: resolve the preprocessor conditionals whose condition mentions ONLY macros of a given set, treating
those macros as undefined, and leave every other directive alone.  Used once in round 6 to move the retired experiment
macros of csrc/ out of the product sources into tools/probes/*.patch (tools/make_probe_patches.sh).

    python tools/unifdef_lite.py -U NJF_A -U NJF_B in.h > out.h
"""
import re
import sys


def resolve(cond: str, undef: set):
    """value of a #if condition when every macro of `undef` is undefined; None when it mentions anything else"""
    names = set(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", cond)) - {"defined"}
    if not names or not names <= undef:
        return None
    e = re.sub(r"defined\s*\(\s*[A-Za-z_][A-Za-z0-9_]*\s*\)|defined\s+[A-Za-z_][A-Za-z0-9_]*", "0", cond)
    e = re.sub(r"[A-Za-z_][A-Za-z0-9_]*", "0", e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e))


def strip_comment(s: str) -> str:
    return s.split("//")[0].strip()


def run(lines, undef):
    out = []
    # frame: [resolved, any_taken, emitting_this_branch, parent_emitting]
    stack = []
    emitting = True
    for line in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), strip_comment(m.group(2))
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "ifdef":
                val = False if rest in undef else None
            elif kind == "ifndef":
                val = True if rest in undef else None
            else:
                val = resolve(rest, undef)
            parent = emitting
            if val is None:
                stack.append([False, False, True, parent])
                if emitting:
                    out.append(line)
            else:
                stack.append([True, val, val, parent])
                emitting = parent and val
        elif kind == "elif":
            fr = stack[-1]
            if not fr[0]:
                if fr[3]:
                    out.append(line)
                continue
            val = resolve(rest, undef)
            if val is None:
                raise SystemExit(f"mixed conditional chain not supported: {line.strip()}")
            take = (not fr[1]) and val
            fr[1] = fr[1] or take
            fr[2] = take
            emitting = fr[3] and take
        elif kind == "else":
            fr = stack[-1]
            if not fr[0]:
                if fr[3]:
                    out.append(line)
                continue
            take = not fr[1]
            fr[1] = True
            fr[2] = take
            emitting = fr[3] and take
        else:  # endif
            fr = stack.pop()
            if not fr[0] and fr[3]:
                out.append(line)
            emitting = fr[3]
    assert not stack
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    undef, files = set(), []
    while args:
        a = args.pop(0)
        if a == "-U":
            undef.add(args.pop(0))
        elif a.startswith("-U"):
            undef.add(a[2:])
        else:
            files.append(a)
    src = open(files[0]).read().splitlines(keepends=True)
    sys.stdout.write("".join(run(src, undef)))
