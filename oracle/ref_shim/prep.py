#!/usr/bin/env python3
"""oracle/ref_shim/prep.py -- rewrite CUDA launch syntax of the reference sources for g++.

    kernel<<<grid, block, shm, stream>>>( args );   ->   SHIM_LAUNCH("kernel", (grid), (block), [&]{ kernel( args ); });

Reads  /root/reference/src/popsift/**  and writes the transformed tree into the directory given
on the command line (oracle/_ref/gen, git-ignored).  Nothing else in the sources is touched.
TEST INFRASTRUCTURE ONLY.
"""
import os
import re
import sys

LAUNCH = re.compile(r"([A-Za-z_][\w:]*)(\s*<[^<>;(){}]*>)?\s*<<<")


def split_top(s):
    """split a comma separated argument list at nesting depth 0"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def transform(text):
    pos, out = 0, []
    while True:
        m = LAUNCH.search(text, pos)
        if not m:
            out.append(text[pos:])
            break
        name = m.group(1) + (m.group(2) or "")
        name = re.sub(r"\s+", "", name)
        cfg_start = m.end()
        cfg_end = text.index(">>>", cfg_start)
        cfg = split_top(text[cfg_start:cfg_end])
        # argument list
        p = cfg_end + 3
        while text[p].isspace():
            p += 1
        assert text[p] == "(", "launch without argument list near: " + text[m.start():m.start() + 80]
        depth, q = 0, p
        while True:
            if text[q] == "(":
                depth += 1
            elif text[q] == ")":
                depth -= 1
                if depth == 0:
                    break
            q += 1
        args = text[p + 1:q]
        out.append(text[pos:m.start()])
        out.append('SHIM_LAUNCH("%s", (%s), (%s), [&]{ %s(%s); })' % (name, cfg[0], cfg[1], name, args))
        pos = q + 1
    return "".join(out)


def main():
    src_root, dst_root = sys.argv[1], sys.argv[2]
    n = 0
    for base, _, files in os.walk(src_root):
        for f in files:
            if not f.endswith((".cu", ".h", ".cpp", ".hpp")):
                continue
            sp = os.path.join(base, f)
            rel = os.path.relpath(sp, src_root)
            dp = os.path.join(dst_root, rel)
            if dp.endswith(".cu"):
                dp = dp[:-3] + ".cu.cpp"
            os.makedirs(os.path.dirname(dp), exist_ok=True)
            txt = open(sp, errors="replace").read()
            new = transform(txt) if "<<<" in txt else txt
            n += new != txt
            open(dp, "w").write(new)
    print("prep: %d files with launches rewritten -> %s" % (n, dst_root))


if __name__ == "__main__":
    main()
