#!/usr/bin/env python3
"""Re-wraps the prose of a markdown file at 120 columns (DESIGN.md had lines of 1-3 K characters).
Tables, code fences, headings and reference-style lines are left alone; list items and block continuation keep their indent.
usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, para, fence = [], [], False


def flush():
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*+-] |\d+\. )?", first)
    indent = m.group(1) + (" " * len(m.group(2)) if m.group(2) else "")
    text = " ".join(l.strip() for l in para)
    lead = m.group(1) + (m.group(2) or "")
    body = text[len((m.group(2) or "")):] if m.group(2) else text
    wrapped = textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)
    out.extend(wrapped or [lead.rstrip()])
    para.clear()


for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        flush()
        fence = not fence
        out.append(line)
        continue
    if fence or line.startswith("|") or line.startswith("#") or not line.strip():
        flush()
        out.append(line)
        continue
    starts_item = re.match(r"^\s*([*+-] |\d+\. )", line) is not None
    if starts_item:
        flush()
    elif para:
        # a continuation line with a different indent than the paragraph's continuation starts a new block
        pass
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
print(path, "longest line now", max(len(l) for l in out), "; table rows over the width:", sum(1 for l in out if l.startswith("|") and len(l) > width))
