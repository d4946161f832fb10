"""Static VALU instruction census of an ISA listing between line ranges, weighted by the issue cost classes measured with
tools/probe_rate (cycles per wave-instruction per SIMD).  usage: isa_count.py file.s start:end[:name] ..."""
import re, sys
COST8 = ("v_rcp", "v_rsq", "v_sqrt", "v_ashr_pk_u8", "v_exp", "v_log", "v_sin", "v_cos")
COST25 = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32_e32",
          "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_legacy", "v_fmaak", "v_fmamk", "v_add_i32")
def cost(op):
    if op.startswith(COST8): return 8.5
    if op.startswith("v_pk_"): return 4.8
    if op.startswith(COST25) and not op.endswith(("_sdwa", "_dpp")): return 2.5
    return 4.2
def census(lines):
    n = c = 0; salu = lds = vmem = 0; ops = {}
    for l in lines:
        m = re.match(r"\s+([a-z][a-z0-9_]+)", l)
        if not m: continue
        op = m.group(1)
        if op.startswith("v_"): n += 1; c += cost(op); ops[op] = ops.get(op, 0) + 1
        elif op.startswith("s_"): salu += 1
        elif op.startswith("ds_"): lds += 1
        elif op.startswith(("global_", "buffer_", "flat_")): vmem += 1
    return n, c, salu, lds, vmem, ops
src = open(sys.argv[1]).read().split("\n")
for spec in sys.argv[2:]:
    p = spec.split(":"); a, b = int(p[0]), int(p[1]); name = p[2] if len(p) > 2 else spec
    n, c, salu, lds, vmem, ops = census(src[a - 1:b])
    top = ", ".join(f"{k} {v}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:8])
    print(f"{name:10s} valu {n:5d}  est.cycles {c:7.0f}  salu {salu:4d} lds {lds:3d} vmem {vmem:3d} | {top}")
