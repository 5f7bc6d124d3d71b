#!/usr/bin/env python
"""Removes the diagnostic build switches (ACX_SLAB_*, ACX_FSLAB_*, ACX_DBG_*, ACX_LAB_*, ACX_GS_*, ACX_SPLIT_*,
ACX_FS_DRAIN, ACX_FE_DEBUG ...) from a kernel source: every such macro is taken as UNDEFINED and the dead branches
are dropped, so that the shipped file contains only code the parity suite executes.  The copies with the switches
live in tools/lab_src/ (what tools/*_lab.hip and tools/race2 build).
    python tools/strip_lab.py tools/lab_src/gemm_split.hip > audioset-convnext-inf_amd/csrc/gemm_split.hip"""
import re
import sys

LAB = re.compile(r"ACX_(SLAB|FSLAB|DBG|LAB|GS|SPLIT|FS_DRAIN|FE_DEBUG|NO_TAIL)\w*")
KEEP_DEFAULT = re.compile(r"^\s*#\s*ifndef\s+(ACX_\w+)\s*$")


def is_lab(cond):
    names = re.findall(r"ACX_\w+", cond)
    return bool(names) and all(LAB.fullmatch(n) for n in names)


def strip(lines):
    out = []
    stack = []          # per open conditional: [lab?, emitting_now, any_branch_taken, parent_emit]
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", ln)
        emit = all(s[1] for s in stack)
        if not m:
            if emit:
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).split("//")[0].split("/*")[0].strip()
        if kind in ("ifdef", "ifndef", "if"):
            # default-value guard:  #ifndef X / #define X v / #endif  -> keep the #define
            if kind == "ifndef" and i + 2 < len(lines) and re.match(r"^\s*#\s*define\s+" + re.escape(rest) + r"\b", lines[i + 1]) \
                    and re.match(r"^\s*#\s*endif", lines[i + 2]) and LAB.fullmatch(rest) is None and emit:
                out.extend(lines[i:i + 3])
                i += 3
                continue
            if kind == "ifndef" and i + 2 < len(lines) and re.match(r"^\s*#\s*define\s+" + re.escape(rest) + r"\b", lines[i + 1]) \
                    and re.match(r"^\s*#\s*endif", lines[i + 2]) and emit:
                out.append(lines[i + 1])          # a lab-overridable default: keep the plain #define
                i += 3
                continue
            if is_lab(rest):
                taken = kind == "ifndef"          # macro undefined: #ifndef branch is live, #ifdef / #if defined dead
                stack.append([True, taken, taken, emit])
            else:
                stack.append([False, True, True, emit])
                if emit:
                    out.append(ln)
        elif kind == "elif":
            s = stack[-1]
            if s[0]:
                if s[2]:
                    s[1] = False
                elif is_lab(rest):
                    s[1] = False
                else:
                    raise SystemExit("mixed lab / non-lab #elif: " + ln)
            elif emit or s[3]:
                out.append(ln)
        elif kind == "else":
            s = stack[-1]
            if s[0]:
                s[1] = not s[2]
                s[2] = True
            elif all(t[1] for t in stack[:-1]):
                out.append(ln)
        else:
            s = stack.pop()
            if not s[0] and all(t[1] for t in stack):
                out.append(ln)
        i += 1
    assert not stack
    return out


if __name__ == "__main__":
    sys.stdout.write("".join(strip(open(sys.argv[1]).readlines())))
