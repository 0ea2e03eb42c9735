"""Static VALU instruction count of one kernel per source line, from a `hipcc -S -gline-tables-only` listing.
usage: isa_lines.py listing.s kernel_symbol_substring source.hpp [top]   (with `by-function` as 5th argument: sums per
enclosing `static VSL_DEV` / `VSL_DEV` function of the source file)"""
import collections
import re
import sys


def main():
    path, key, srcpath = sys.argv[1:4]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    src = open(srcpath).read().split("\n")
    srcname = srcpath.split("/")[-1]
    cur, cnt, seg, segcnt = None, collections.Counter(), 0, collections.Counter()
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        elif t.startswith("s_barrier"):
            seg += 1
        elif t.startswith("v_"):
            cnt[cur] += 1
            segcnt[(seg, cur)] += 1
    print("total valu", sum(cnt.values()))
    if len(sys.argv) > 5 and sys.argv[5] == "by-function":
        fn_at = {}
        name = "?"
        for i, l in enumerate(src, 1):
            m = re.match(r"\s*(?:static\s+)?VSL_DEV\s+[\w:<>\s\*&]+?\s+(\w+)\s*\(", l)
            if m:
                name = m.group(1)
            fn_at[i] = name
        agg = collections.Counter()
        for (f, ln), n in cnt.items():
            agg[fn_at.get(ln, "?") if f == srcname else f] += n
        for k, n in agg.most_common(top):
            print("%6d %s" % (n, k))
        return
    for (f, ln), n in cnt.most_common(top):
        text = src[ln - 1].strip()[:100] if f == srcname and 0 < ln <= len(src) else ""
        print("%5d %s:%d  %s" % (n, f, ln, text))


main()
