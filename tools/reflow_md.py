#!/usr/bin/env python3
"""Reflow the prose of a Markdown file to 120 columns (tables, headings and fenced blocks untouched; word sequence unchanged).
    python tools/reflow_md.py HISTORY.md"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, par, fence = [], [], False


def flush():
    if not par:
        return
    m = re.match(r"^(\s*)([*\-] |\d+\. )?", par[0])
    ind, bullet = m.group(1), m.group(2) or ""
    body = " ".join([par[0][len(ind) + len(bullet):].strip()] + [x.strip() for x in par[1:]])
    w = textwrap.wrap(body, width=width - len(ind) - len(bullet), break_long_words=False, break_on_hyphens=False) or [""]
    out.append(ind + bullet + w[0])
    out.extend(ind + " " * len(bullet) + x for x in w[1:])
    par.clear()


for ln in lines:
    st = ln.strip()
    if st.startswith("```"):
        flush()
        fence = not fence
        out.append(ln)
    elif fence:
        out.append(ln)
    elif st == "" or st.startswith("#") or st.startswith("|") or st.startswith("---"):
        flush()
        out.append(ln)
    elif re.match(r"^\s*([*\-] |\d+\. )", ln):
        flush()
        par.append(ln)
    else:
        par.append(ln)
flush()
open(path, "w").write("\n".join(out))
print(path, len(lines), "->", len(out), "lines; longest prose line", max(len(l) for l in out if not l.lstrip().startswith("|")))
