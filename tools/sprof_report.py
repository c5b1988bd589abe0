#!/usr/bin/env python3
"""Report for tools/sprof.c: per thread, the functions by self and by inclusive
samples.  Addresses are resolved with nm (dynamic + static symbols) over the
files /proc/self/maps named.   tools/sprof_report.py /tmp/sprof.txt [more.txt ...] [top=25] [under=<symbol>]
(under=: only samples with that symbol on the stack, inclusive counts below it)."""
import bisect
import collections
import os
import subprocess
import sys


def load(path):
    maps, samples = [], []
    for line in open(path):
        if line.startswith("M "):
            f = line.split()
            if len(f) < 7:
                continue
            lo, hi = (int(x, 16) for x in f[1].split("-"))
            maps.append((lo, hi, int(f[3], 16), f[6], "x" in f[2]))
        elif line.startswith("S "):
            f = line.split()
            samples.append((int(f[1]), [0 if x == "(nil)" else int(x, 16) for x in f[2:]]))
    return maps, samples


class Symbols:
    def __init__(self, maps):
        self.maps = sorted(maps)
        self.base = {}
        for lo, hi, off, name, _ in self.maps:
            self.base[name] = min(self.base.get(name, lo - off), lo - off)
        self.tables = {}
        self.exec_lo = [m[0] for m in self.maps if m[4]]
        self.exec_hi = [m[1] for m in self.maps if m[4]]

    def is_code(self, addr):
        i = bisect.bisect_right(self.exec_lo, addr) - 1
        return i >= 0 and addr < self.exec_hi[i]

    def table(self, name):
        if name not in self.tables:
            syms = []
            for flags in (["-D"], []):
                r = subprocess.run(["nm", "--defined-only", "-n"] + flags + [name],
                                   capture_output=True, text=True)
                for l in r.stdout.split("\n"):
                    p = l.split()
                    if len(p) == 3 and p[1] in "TtWwiV":
                        syms.append((int(p[0], 16), p[2]))
            syms = sorted(set(syms))
            self.tables[name] = ([a for a, _ in syms], [s for _, s in syms])
        return self.tables[name]

    def resolve(self, addr):
        for lo, hi, off, name, _ in self.maps:
            if lo <= addr < hi:
                if not os.path.exists(name):
                    return f"[{name}]"
                v = addr - self.base[name]
                addrs, names = self.table(name)
                if name.endswith("python3.10") or "/bin/" in name:
                    v = addr if addrs and addrs[0] > 0x100000 else v
                i = bisect.bisect_right(addrs, v) - 1
                short = os.path.basename(name).split(".so")[0]
                return f"{names[i]} ({short})" if i >= 0 else f"? ({short})"
        return "?"


def main():
    paths = [a for a in sys.argv[1:] if "=" not in a]
    top, under = 25, None
    for a in sys.argv[1:]:
        if "=" not in a:
            continue
        k, v = a.split("=")
        if k == "top":
            top = int(v)
        if k == "under":
            under = v
    per_thread = collections.defaultdict(list)
    runs = []
    for path in paths:             # (one address space per run)
        maps, samples = load(path)
        sym = Symbols(maps)
        cache = {}
        for tid, pcs in samples:
            # the PC, then the stack words that point into code (probable return
            # addresses; stale ones among them -- read the inclusive list with that in mind)
            pcs = [pcs[0] + 1] + [a for a in pcs[1:] if sym.is_code(a)]
            for a in pcs:
                if a not in cache:
                    cache[a] = sym.resolve(a - 1)
            runs.append((tid if len(paths) == 1 else 0, [cache[a] for a in pcs]))
    for tid, names in runs:
        if under:
            hit = [i for i, n in enumerate(names) if n.startswith(under)]
            if not hit:
                continue
            names = names[:hit[-1] + 1]
        per_thread[tid].append(names)
    for tid, stacks in sorted(per_thread.items(), key=lambda kv: -len(kv[1])):
        print(f"== thread {tid}: {len(stacks)} samples")
        self_c, incl_c = collections.Counter(), collections.Counter()
        for names in stacks:
            if not names:
                continue
            self_c[names[0]] += 1
            for n in set(names):
                incl_c[n] += 1
        print("  -- self")
        for n, c in self_c.most_common(top):
            print(f"  {c:6d} {100.0 * c / len(stacks):5.1f}%  {n}")
        print("  -- inclusive")
        for n, c in incl_c.most_common(top):
            print(f"  {c:6d} {100.0 * c / len(stacks):5.1f}%  {n}")


if __name__ == "__main__":
    main()
