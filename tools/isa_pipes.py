"""Static split of one kernel's vector-ALU instructions by the issue restrictions measured with tools/probe/pair_probe.hip:
 S  reads a scalar register (s[N], vcc, exec as data)   -> back-to-back S instructions issue every 4 cycles, not 2
 M  compare / min / max / logic / shift / convert / 24-bit multiply-add -> back-to-back M instructions likewise
 F  fma / mul / add / sub / mov / integer add with vector-register, inline-constant or literal operands: 2 cycles
 H  float64, reciprocal-class, 32-bit integer multiply, DPP: several times longer
usage: isa_pipes.py listing.s kernel_symbol_substring"""
import collections
import re
import sys

FAST = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_cndmask_b32", "v_med3_f32", "v_accvgpr")
HEAVY = ("_f64", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_mul_lo", "v_mul_hi", "v_mad_u64", "v_div_scale", "v_div_fmas", "v_div_fixup")


def classify(t):
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    args = t[len(op):].split(";")[0]
    ops = [a.strip() for a in args.split(",")]
    srcs = ops[1:] if not base.startswith("v_cmp") else ops[1:]
    reads_s = any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|ttmp|m0)", a) for a in srcs)
    if "dpp" in op or "row_" in t or "quad_perm" in t:
        return "H"
    if any(h in base for h in HEAVY):
        return "H"
    if base.startswith("v_cmp") or not base.startswith(FAST):
        return "MS" if reads_s else "M"
    return "S" if reads_s else "F"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    seg, cnt, adj = 0, collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    prev = None
    nonv = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            seg += 1
            prev = None
            continue
        if not op.startswith("v_"):
            if not op.startswith(("s_waitcnt", "s_nop")):
                nonv[seg] += 1
                prev = None
            continue
        c = classify(t)
        cnt[seg][c] += 1
        if prev is not None:
            for r in "SM":
                if r in c and r in prev:
                    adj[seg][r + r] += 1
        prev = c
    tot = collections.Counter()
    for s in sorted(cnt):
        n = sum(cnt[s].values())
        print("segment %d: valu %4d  F %4d  S %4d  M %4d  MS %3d  H %3d | adjacent S-S %3d  M-M %3d | other issuing instructions %d" % (
            s, n, cnt[s]["F"], cnt[s]["S"], cnt[s]["M"], cnt[s]["MS"], cnt[s]["H"], adj[s]["SS"], adj[s]["MM"], nonv[s]))
        tot.update(cnt[s])
    print("total: ", dict(tot))


main()
