"""Static census of an ISA listing against the issue classes measured by tools/gen_probe_issue.py (profiles/r05_valu_issue_classes.txt):
how many vector instructions COULD pair (fast opcode, no SGPR / literal source, no DPP / SDWA form).  usage: isa_fast_class.py file.s ..."""
import re, sys
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32")
for path in sys.argv[1:]:
    n = fast = fast_op = two = 0
    for l in open(path):
        m = re.match(r"\s+(v_[a-z0-9_]+)\s+(.*)", l)
        if not m: continue
        op, args = m.group(1), m.group(2).split(";")[0]
        n += 1
        if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_ashr_pk_u8")): two += 1
        base = re.sub(r"_e(32|64)$", "", op)
        if base not in FAST: continue
        fast_op += 1
        srcs = args.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|m0|0x|\d{3,})", s) for s in srcs) or "dpp" in l or "sdwa" in l: continue
        fast += 1
    print(f"{path}: {n} vector instructions, {fast_op} with a fast-class opcode, {fast} of them with operands that pair ({100.0 * fast / max(n, 1):.1f} %), {two} two-turn")
