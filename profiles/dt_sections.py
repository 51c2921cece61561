"""Static instruction counts of one delaunay_kernel instantiation (default <false, 4, true>: the 2000-point build) between the DT_MARK markers of a -DMVOSR_DT_MARKS -S build:
   (cd mvoscalerecovery_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -DMVOSR_DT_MARKS -S mvosr_delaunay.hip -o /tmp/dt_marks.s)
   python profiles/dt_sections.py [/tmp/dt_marks.s] [section to print] [mangled name prefix, e.g. _ZN5mvosr15delaunay_kernelILb0ELi8ELb0]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/dt_marks.s"
show = sys.argv[2] if len(sys.argv) > 2 else None
lines = open(path).read().split("\n")
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "_ZN5mvosr15delaunay_kernelILb0ELi4ELb1"
start = [i for i, l in enumerate(lines) if l.startswith(KERNEL)][0]
marks = []
for i in range(start, len(lines)):
    m = re.search(r"; DTMARK (\w+)", lines[i])
    if m: marks.append((i, m.group(1)))
    if lines[i].startswith(".Lfunc_end") and i > start + 100: marks.append((i, "end")); break
def is_inst(l): return l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")
shown = False
for (a, n), (b, _) in zip(marks, marks[1:]):
    ins = [l.split()[0] for l in lines[a:b] if is_inst(l)]
    v = sum(1 for x in ins if x.startswith("v_")); sa = sum(1 for x in ins if x.startswith("s_")); m = len(ins) - v - sa
    print("%-14s %4d instructions (%4d vector, %4d scalar, %3d memory)" % (n, len(ins), v, sa, m))
    if show == n and not shown:
        shown = True
        for l in lines[a:b]:
            if is_inst(l) or l.startswith(".LBB"): print("      " + l[:100])
