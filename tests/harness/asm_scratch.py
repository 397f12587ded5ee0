"""Where the scratch (register spill) accesses of one kernel instantiation lie: for every scratch_load / scratch_store of
the ISA listing made by `make asm`, the innermost loop (a backward branch and its target label) that encloses it.
Argument: a piece of the mangled kernel name, e.g. ILi12ELi14ELi6ELb0ELb1ELi0ELb0ELb0EEELb0ELb0E."""
import os, re, sys
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "raytracer_amd", "csrc", "_asm")
tag = sys.argv[1]
s = open(os.path.join(d, "rt_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r"^(_ZN\S*render_persistent_kernel\S*" + tag + r"\S*):", s, re.M)
body = s[m.start():s.index(".end_amdhsa_kernel", m.end())].split("\n")
labels = {}
for i, line in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", line)
    if mm: labels[mm.group(1)] = i
loops = []
for j, line in enumerate(body):
    mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < j:
        loops.append((labels[mm.group(1)], j))
def insts(a, b): return sum(1 for l in body[a:b] if re.match(r"^\s+[vsgdb]\w*_\w+", l))
print("kernel lines", len(body), "instructions", insts(0, len(body)), "loops", len(loops))
for a, b in sorted(loops, key=lambda t: t[1] - t[0]):
    n = sum(1 for l in body[a:b] if "scratch_" in l)
    lanes = sum(1 for l in body[a:b] if re.search(r"v_readlane_b32|v_writelane_b32", l))
    print("loop lines %d-%d: %d instructions, %d scratch accesses, %d global/ds loads, %d v_readlane/v_writelane (scalar spills)"
          % (a, b, insts(a, b), n, sum(1 for l in body[a:b] if re.search(r"global_load|ds_read", l)), lanes))
tot = [i for i, l in enumerate(body) if "scratch_" in l]
print("scratch accesses in the kernel:", len(tot))
for i in tot:
    enc = [(b - a, a, b) for a, b in loops if a <= i <= b]
    print(i, body[i].strip()[:60], "| innermost loop:", min(enc)[1:] if enc else None)
