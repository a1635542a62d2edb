"""Static instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py lcfe.s <kernel substring> ..."""
import collections
import re
import sys


def kernel_body(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


def classify(op):
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_mov"): return "v_mov (+dpp)"
    if op.startswith(("v_min_f64", "v_max_f64")): return "min/max f64"
    if op.startswith("v_cmp"): return "v_cmp"
    if "_f64" in op: return "f64"
    if op.startswith(("v_accvgpr", "v_readlane", "v_writelane")): return "lane/acc moves"
    if op.startswith("v_"): return "valu other"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "global"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for name in sys.argv[2:]:
        body = kernel_body(lines, name)
        ops = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.strip()]
        c = collections.Counter(classify(o) for o in ops)
        print(name, len(ops), dict(c.most_common()))


if __name__ == "__main__":
    main()
