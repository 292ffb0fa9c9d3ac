"""Instruction mix of one kernel in a disassembly made by tools/isa_dump.sh: python tools/isa_count.py out.s k_traceILb0"""
import collections
import sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and l.endswith(">:")][0]
end = [i for i, l in enumerate(lines) if i > start and l.endswith(">:")][0]
c = collections.Counter()
for l in lines[start:end]:
    m = l.strip().split()
    if m:
        c[m[0].replace("_e32", "").replace("_e64", "")] += 1
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print("%s: %d instructions, VALU %d (v_perm_b32 %d, v_pk_fma_f32 %d, v_ldexp_f32 %d, v_readlane %d), SALU %d, LDS %d (ds_bpermute %d), global loads %d / stores %d, scratch %d" % (
    sys.argv[2], end - start, valu, c["v_perm_b32"], c["v_pk_fma_f32"], c["v_ldexp_f32"], c["v_readlane_b32"], sum(v for k, v in c.items() if k.startswith("s_")),
    sum(v for k, v in c.items() if k.startswith("ds_")), c["ds_bpermute_b32"], sum(v for k, v in c.items() if k.startswith("global_load")),
    sum(v for k, v in c.items() if k.startswith("global_store")), sum(v for k, v in c.items() if k.startswith("scratch"))))
