#!/usr/bin/env python3
"""One-off maintenance tool (round 6): cut the 157 KB DESIGN.md of rounds 1-5 into per-topic files under docs/ with lines of at
most 160 columns. Paragraphs and list items are re-wrapped; a markdown table whose rows do not fit becomes a list (one item per row, one sub-item per
column, headed by the column's title), which keeps every figure and loses only the grid. Code blocks are left alone.

    python tools/split_design.py DESIGN.md            # writes docs/*.md from the section map below"""
import re
import sys
import textwrap

W = 140   # characters; the text is full of multi-byte signs, 140 characters stay below 160 BYTES
SECTIONS = [   # (heading prefix in the old file, output file, title)
    ("## 3. Kernels", "docs/kernels.md", "Kernels"),
    ("## 4. Oracle and parity", "docs/parity.md", "Oracle and parity"),
    ("## 5. Multi-GPU", "docs/multigpu.md", "Multi-GPU"),
    ("## 6. Measurements", "docs/measurements_r1_r5.md", "Measurements, rounds 1-5"),
    ("## 7. Known gaps", "docs/budget_r1_r5.md", "Per-kernel budget and the gaps of rounds 1-5"),
]


def wrap_par(text, indent="", first=None):
    first = indent if first is None else first
    return textwrap.fill(" ".join(text.split()), width=W, initial_indent=first, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)


def table_to_list(rows):
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    head, body = cells[0], [c for c in cells[2:]]
    if all(len(r) <= W for r in rows):
        return rows
    out = []
    for r in body:
        title = r[0] if r and r[0] else "(row)"
        out.append(wrap_par(f"**{title}**", "  ", "- "))
        for h, c in zip(head[1:], r[1:]):
            if c:
                out.append(wrap_par(f"*{h}:* {c}" if h else c, "    ", "  - "))
    return out


def convert(lines):
    out, i = [], 0
    while i < len(lines):
        ln = lines[i].rstrip("\n")
        if ln.startswith("```"):
            j = i + 1
            while j < len(lines) and not lines[j].startswith("```"):
                j += 1
            out += [l.rstrip("\n") for l in lines[i:j + 1]]
            i = j + 1
            continue
        if ln.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            out += table_to_list([l.rstrip("\n") for l in lines[i:j]])
            i = j
            continue
        if not ln.strip() or ln.startswith("#"):
            out.append(ln)
            i += 1
            continue
        # a paragraph or list item: gather continuation lines (indented, or plain text following plain text)
        m = re.match(r"^(\s*)([-*]|\d+\.)\s+", ln)
        if m:
            indent = " " * len(m.group(0))
            j = i + 1
            while j < len(lines) and lines[j].strip() and not re.match(r"^\s*([-*]|\d+\.)\s+", lines[j]) and not lines[j].startswith(("#", "|", "```")):
                j += 1
            out.append(wrap_par(" ".join(l.strip() for l in lines[i:j])[len(m.group(0).strip()) + 1:] if False else " ".join(l.strip() for l in lines[i:j]), indent, m.group(1)))
            # (the item's marker is part of the joined text: re-insert hanging indent only)
            i = j
            continue
        j = i + 1
        while j < len(lines) and lines[j].strip() and not lines[j].startswith(("#", "|", "```")) and not re.match(r"^\s*([-*]|\d+\.)\s+", lines[j]):
            j += 1
        lead = re.match(r"^\s*", ln).group(0)
        out.append(wrap_par(" ".join(l.strip() for l in lines[i:j]), lead))
        i = j
    return out


def main():
    src = open(sys.argv[1]).read().split("\n")
    starts = {}
    for k, ln in enumerate(src):
        for prefix, path, _ in SECTIONS:
            if ln.startswith(prefix):
                starts[path] = k
    order = sorted(starts.items(), key=lambda kv: kv[1])
    tops = [k for k, ln in enumerate(src) if ln.startswith("## ")] + [len(src)]
    for path, k in order:
        end = min(t for t in tops if t > k)
        title = [t for p, f, t in SECTIONS if f == path][0]
        body = convert([l + "\n" for l in src[k + 1:end]])
        text = [f"# {title}", "", f"(Moved out of DESIGN.md in round 6; what follows is the text of rounds 1-5 re-wrapped to {W} columns - figures unchanged. Current status: DESIGN.md.)", ""] + body
        open(path, "w").write("\n".join(text).rstrip() + "\n")
        longest = max(len(l) for l in text)
        print(path, len(text), "lines, longest", longest)


if __name__ == "__main__":
    main()
