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


if __name__ == "__main__":
    ks = kernels(open(sys.argv[1]).read())
    bad = [k for k in ks if k["spill"] or k["scratch"]]
    print(f"{len(ks)} kernels, {len(bad)} with spills/scratch")
    for k in bad:
        print(k)
    sys.exit(1 if bad else 0)
