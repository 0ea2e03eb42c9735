"""Static instruction mix of one kernel in a hipcc -S listing, split at s_barrier.
usage: isa_mix.py listing.s kernel_symbol_substring [--top N]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    seg, segs = collections.Counter(), []
    ops = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            segs.append(seg)
            seg = collections.Counter()
            continue
        cls = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else
               "smem" if op.startswith(("s_load", "s_buffer")) else
               "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu")
        seg[cls] += 1
        seg["all"] += 1
        ops[op] += 1
    segs.append(seg)
    for i, s in enumerate(segs):
        print("segment %d: " % i + "  ".join("%s %d" % (k, s[k]) for k in ("all", "valu", "lds", "vmem", "salu", "smem", "wait")))
    tot = sum(segs, collections.Counter())
    print("total:     " + "  ".join("%s %d" % (k, tot[k]) for k in ("all", "valu", "lds", "vmem", "salu", "smem", "wait")))
    if top:
        for op, n in ops.most_common(top):
            print("%6d %s" % (n, op))


main()
