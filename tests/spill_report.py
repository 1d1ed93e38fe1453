"""Parse the AMDGPU metadata of a `hipcc --cuda-device-only -S` listing: per kernel VGPR / spill / scratch figures.
usage: python tests/spill_report.py file.s  (prints offenders; exit code 1 if any kernel spills or uses scratch)"""
import re
import sys


def kernels(text):
    out = []
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        get = lambda key: (re.search(r"\." + key + r":\s*(\S+)", blk) or [None, None])[1]
        name = get("name")
        if name is None:
            continue
        out.append(dict(name=name, vgpr=int(get("vgpr_count") or 0), spill=int(get("vgpr_spill_count") or 0),
                        sgpr_spill=int(get("sgpr_spill_count") or 0), scratch=int(get("private_segment_fixed_size") or 0),
                        lds=int(get("group_segment_fixed_size") or 0)))
    return out


def scratch_in_inner_loops(text):
    """{kernel symbol: number of scratch_* instructions inside its deepest loops}.  The AMDGPU asm printer annotates every loop
    block label with '; in Loop: Header=... Depth=D' / '; =>This (Inner) Loop Header: Depth=D' comments (possibly continued on
    comment-only lines); blocks outside any loop carry no annotation."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.splitlines()
        depth_of_line, cur = [], 0
        i = 0
        while i < len(lines):
            line = lines[i]
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
                ds = [int(d) for d in re.findall(r"Depth=(\d+)", line)]
                j = i + 1
                while j < len(lines) and re.match(r"^\s*;", lines[j]) and not re.match(r"^; %bb\.", lines[j]):
                    ds += [int(d) for d in re.findall(r"Depth=(\d+)", lines[j])]
                    j += 1
                cur = max(ds) if ds else 0
            depth_of_line.append(cur)
            i += 1
        deepest = max(depth_of_line) if depth_of_line else 0
        out[name] = sum(1 for line, d in zip(lines, depth_of_line) if deepest > 0 and d == deepest and "scratch_" in line)
    return out


if __name__ == "__main__":
    ks = kernels(open(sys.argv[1]).read())
    bad = [k for k in ks if k["spill"] or k["scratch"]]
    print(f"{len(ks)} kernels, {len(bad)} with spills/scratch")
    for k in bad:
        print(k)
    sys.exit(1 if bad else 0)
