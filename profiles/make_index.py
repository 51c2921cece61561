#!/usr/bin/env python3
"""Writes profiles/README.md: every script under profiles/ with the first sentence of its header, and the committed result
files (r0N_*) grouped by round.   python profiles/make_index.py"""
import ast
import glob
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def first_words(path):
    src = open(path, errors="replace").read()
    if path.endswith(".py"):
        try:
            doc = ast.get_docstring(ast.parse(src))
        except SyntaxError:
            doc = None
        if doc:
            return " ".join(doc.split())
    for line in src.splitlines()[1:8]:
        if line.startswith("#") and len(line) > 3:
            return line.lstrip("# ").strip()
    return ""


def main():
    out = ["# profiles/ — what produced what", "",
           "Scripts (run on the GPU box through `gpurun`, from the repo root) and the result files they leave here.  The tables in",
           "DESIGN.md / README.md are rendered from these files by `make_tables.py`; `gpurun_out/` is scratch.", "",
           "## Scripts", "", "| script | what it does |", "|---|---|"]
    for p in sorted(glob.glob(os.path.join(HERE, "*.py")) + glob.glob(os.path.join(HERE, "*.sh"))):
        text = first_words(p)
        text = re.split(r"(?<=[.:;])\s", text, maxsplit=1)[0] if len(text) > 220 else text
        out.append("| `%s` | %s |" % (os.path.basename(p), text[:260].replace("|", "/")))
    out += ["", "## Result files", ""]
    files = sorted(f for f in os.listdir(HERE) if re.match(r"r\d\d_", f))
    by_round = {}
    for f in files:
        by_round.setdefault(f[:3], []).append(f)
    for r in sorted(by_round, reverse=True):
        out.append("* **round %d**: %s" % (int(r[1:]), ", ".join("`%s`" % f for f in by_round[r])))
    out += ["* `traffic.json`: HBM bytes per launch from the PMC passes (`collect*.sh`), keyed by workload and tagged with the round that measured them",
            "* `micro/`: stand-alone micro-benchmarks behind single design decisions (LABNOTES cites them)", ""]
    open(os.path.join(HERE, "README.md"), "w").write("\n".join(out))
    print("wrote profiles/README.md:", len(out), "lines")


if __name__ == "__main__":
    main()
