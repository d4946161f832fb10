#!/usr/bin/env python3
"""Generates tools/probe_issue.hip: the issue cost of gfx950 vector instructions as a function of the OPERAND PATTERN.

Round 2's probe (tools/probe_rate.hip, profiles/history/r02_valu_issue_rates.txt) found two classes -- 2.3-2.8 and 4.1-4.7
cycles per wave-instruction and SIMD -- but always with the destination as first source and one VGPR repeated in the other
slots.  This one fixes the registers by number: every kernel is a loop around 128 copies of ONE instruction whose sources are
explicit VGPRs / SGPRs / inline constants and whose destinations rotate over eight registers nobody reads (no RAW chain).

Patterns (3-source ops; 2- and 1-source ops take the applicable subset):
  vvv      three distinct VGPRs in three different banks (v4, v5, v6; bank = index mod 4)
  bank     three distinct VGPRs of ONE bank (v4, v8, v12)
  s0=s1    v4, v4, v6
  s1=s2    v4, v6, v6            (round 2's pattern, without the chain)
  chain    dst, v6, v6 with dst also src0 (round 2's pattern: RAW distance 8)
  sgpr0/1/2  an SGPR in that slot, VGPRs elsewhere
  imm0/1/2   an inline constant in that slot
  vii      VGPR, inline constant, inline constant
Run: python3 tools/gen_probe_issue.py && hipcc --offload-arch=gfx950 -O2 tools/probe_issue.hip -o tools/probe_issue
"""
import os
import re

# name -> (number of sources, template); {d} destination, {a} {b} {c} sources
OPS3 = {
    "v_fma_f32": "v_fma_f32 {d}, {a}, {b}, {c}",
    "v_mad_u32_u24": "v_mad_u32_u24 {d}, {a}, {b}, {c}",
    "v_mad_i32_i24": "v_mad_i32_i24 {d}, {a}, {b}, {c}",
    "v_dot4_u32_u8": "v_dot4_u32_u8 {d}, {a}, {b}, {c}",
    "v_dot2_u32_u16": "v_dot2_u32_u16 {d}, {a}, {b}, {c}",
    "v_perm_b32": "v_perm_b32 {d}, {a}, {b}, {c}",
    "v_lshl_add_u32": "v_lshl_add_u32 {d}, {a}, {b}, {c}",
    "v_add_lshl_u32": "v_add_lshl_u32 {d}, {a}, {b}, {c}",
    "v_lshl_or_b32": "v_lshl_or_b32 {d}, {a}, {b}, {c}",
    "v_and_or_b32": "v_and_or_b32 {d}, {a}, {b}, {c}",
    "v_or3_b32": "v_or3_b32 {d}, {a}, {b}, {c}",
    "v_add3_u32": "v_add3_u32 {d}, {a}, {b}, {c}",
    "v_bfe_u32": "v_bfe_u32 {d}, {a}, {b}, {c}",
    "v_bfi_b32": "v_bfi_b32 {d}, {a}, {b}, {c}",
    "v_alignbit_b32": "v_alignbit_b32 {d}, {a}, {b}, {c}",
    "v_alignbyte_b32": "v_alignbyte_b32 {d}, {a}, {b}, {c}",
    "v_med3_i32": "v_med3_i32 {d}, {a}, {b}, {c}",
    "v_max3_u32": "v_max3_u32 {d}, {a}, {b}, {c}",
    "v_min3_u32": "v_min3_u32 {d}, {a}, {b}, {c}",
    "v_ashr_pk_u8_i32": "v_ashr_pk_u8_i32 {d}, {a}, {b}, {c}",
    "v_sad_u8": "v_sad_u8 {d}, {a}, {b}, {c}",
    "v_lerp_u8": "v_lerp_u8 {d}, {a}, {b}, {c}",
    "v_mad_u32_u16": "v_mad_u32_u16 {d}, {a}, {b}, {c}",
    "v_pk_mad_u16": "v_pk_mad_u16 {d}, {a}, {b}, {c}",
    "v_pk_fma_f16": "v_pk_fma_f16 {d}, {a}, {b}, {c}",
    "v_cndmask_b32(sgpr)": "v_cndmask_b32_e64 {d}, {a}, {b}, s[30:31]",  # two data sources + mask
}
OPS3_PK = {  # 64-bit operands
    "v_pk_fma_f32": "v_pk_fma_f32 {d}, {a}, {b}, {c}",
}
OPS2 = {
    "v_add_u32": "v_add_u32 {d}, {a}, {b}",
    "v_sub_u32": "v_sub_u32 {d}, {a}, {b}",
    "v_mul_u32_u24": "v_mul_u32_u24 {d}, {a}, {b}",
    "v_mul_lo_u32": "v_mul_lo_u32 {d}, {a}, {b}",
    "v_and_b32": "v_and_b32 {d}, {a}, {b}",
    "v_or_b32": "v_or_b32 {d}, {a}, {b}",
    "v_xor_b32": "v_xor_b32 {d}, {a}, {b}",
    "v_lshlrev_b32": "v_lshlrev_b32 {d}, {a}, {b}",
    "v_lshrrev_b32": "v_lshrrev_b32 {d}, {a}, {b}",
    "v_ashrrev_i32": "v_ashrrev_i32 {d}, {a}, {b}",
    "v_max_i32": "v_max_i32 {d}, {a}, {b}",
    "v_min_i32": "v_min_i32 {d}, {a}, {b}",
    "v_max_u32": "v_max_u32 {d}, {a}, {b}",
    "v_min_u32": "v_min_u32 {d}, {a}, {b}",
    "v_max_f32": "v_max_f32 {d}, {a}, {b}",
    "v_add_f32": "v_add_f32 {d}, {a}, {b}",
    "v_mul_f32": "v_mul_f32 {d}, {a}, {b}",
    "v_cndmask_b32(vcc)": "v_cndmask_b32 {d}, {a}, {b}, vcc",
    "v_cmp_lt_u32(sgpr)": "v_cmp_lt_u32_e64 s[32:33], {a}, {b}",
    "v_cmp_lt_u32(vcc)": "v_cmp_lt_u32 vcc, {a}, {b}",
    "v_pk_add_u16": "v_pk_add_u16 {d}, {a}, {b}",
    "v_pk_lshrrev_b16": "v_pk_lshrrev_b16 {d}, {a}, {b}",
    "v_pk_mul_lo_u16": "v_pk_mul_lo_u16 {d}, {a}, {b}",
    "v_pk_min_u16": "v_pk_min_u16 {d}, {a}, {b}",
    "v_cvt_pk_u16_u32": "v_cvt_pk_u16_u32 {d}, {a}, {b}",
    "v_pack_b32_f16": "v_pack_b32_f16 {d}, {a}, {b}",
}
OPS2_PK = {
    "v_pk_mul_f32": "v_pk_mul_f32 {d}, {a}, {b}",
    "v_pk_add_f32": "v_pk_add_f32 {d}, {a}, {b}",
}
OPS1 = {
    "v_mov_b32": "v_mov_b32 {d}, {a}",
    "v_mov_b32_dpp(quad)": "v_mov_b32_dpp {d}, {a} quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf",
    "v_mov_b32_dpp(row_shr)": "v_mov_b32_dpp {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0 {d}, {a}",
    "v_cvt_f32_u32": "v_cvt_f32_u32 {d}, {a}",
    "v_rcp_f32": "v_rcp_f32 {d}, {a}",
    "v_sqrt_f32": "v_sqrt_f32 {d}, {a}",
    "v_readlane(s)": "v_readlane_b32 s34, {a}, 3",
    "v_readfirstlane(s)": "v_readfirstlane_b32 s34, {a}",
}
# VOP2 forms with DPP / SDWA modifiers (a shift or byte select folded into an add)
OPS_EXTRA = {
    "v_add_u32_sdwa(byte sel)": "v_add_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
    "v_add_u32_dpp(quad)": "v_add_u32_dpp {d}, {a}, {b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
    "v_and_b32_sdwa(word sel)": "v_and_b32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
}

DST = ["v%d" % (40 + k) for k in range(8)]
DST_PK = ["v[%d:%d]" % (40 + 2 * k, 41 + 2 * k) for k in range(8)]


def pat3(pk):
    V = (lambda i: "v[%d:%d]" % (i, i + 1)) if pk else (lambda i: "v%d" % i)
    S = (lambda i: "s[%d:%d]" % (i, i + 1)) if pk else (lambda i: "s%d" % i)
    imm = "1.0" if pk else "5"
    if pk:  # 64-bit operands: bank of the pair = first index mod 4
        return {
            "vvv": (V(4), V(6), V(8)), "bank": (V(4), V(8), V(12)), "s0=s1": (V(4), V(4), V(6)), "s1=s2": (V(4), V(6), V(6)),
            "chain": ("D", V(6), V(6)), "sgpr0": (S(20), V(6), V(8)), "sgpr1": (V(4), S(20), V(8)), "sgpr2": (V(4), V(6), S(20)),
            "imm1": (V(4), imm, V(8)), "imm2": (V(4), V(6), imm), "vii": (V(4), imm, imm),
        }
    return {
        "vvv": (V(4), V(5), V(6)), "bank": (V(4), V(8), V(12)), "s0=s1": (V(4), V(4), V(6)), "s1=s2": (V(4), V(6), V(6)),
        "chain": ("D", V(6), V(6)), "sgpr0": (S(20), V(5), V(6)), "sgpr1": (V(4), S(20), V(6)), "sgpr2": (V(4), V(5), S(20)),
        "imm0": (imm, V(5), V(6)), "imm1": (V(4), imm, V(6)), "imm2": (V(4), V(5), imm), "vii": (V(4), imm, imm),
    }


def pat2(pk):
    V = (lambda i: "v[%d:%d]" % (i, i + 1)) if pk else (lambda i: "v%d" % i)
    S = (lambda i: "s[%d:%d]" % (i, i + 1)) if pk else (lambda i: "s%d" % i)
    imm = "1.0" if pk else "5"
    return {
        "vv": (V(4), V(6) if pk else V(5)), "bank": (V(4), V(8)), "s0=s1": (V(4), V(4)), "chain": (V(6), "D"), "chain0": ("D", V(6)),
        "sgpr0": (S(20), V(6) if pk else V(5)), "sgpr1": (V(4), S(20)), "imm0": (imm, V(6) if pk else V(5)), "imm1": (V(4), imm),
    }


PAT1 = {"v": ("v4",), "chain": ("D",), "sgpr": ("s20",), "imm": ("5",)}

kernels = []  # (label, body)


def emit(opname, tmpl, pats, dsts):
    for pname, srcs in pats.items():
        lines = []
        for k in range(128):
            d = dsts[k % 8]
            s = [d if x == "D" else x for x in srcs]
            kw = dict(d=d, a=s[0])
            if len(s) > 1: kw["b"] = s[1]
            if len(s) > 2: kw["c"] = s[2]
            lines.append(tmpl.format(**kw))
        kernels.append(("%s|%s" % (opname, pname), lines))


for n, t in OPS3.items(): emit(n, t, pat3(False), DST)
for n, t in OPS3_PK.items(): emit(n, t, pat3(True), DST_PK)
for n, t in OPS2.items(): emit(n, t, pat2(False), DST)
for n, t in OPS2_PK.items(): emit(n, t, pat2(True), DST_PK)
for n, t in OPS1.items(): emit(n, t, PAT1, DST)
for n, t in OPS_EXTRA.items(): emit(n, t, {"vv": ("v4", "v5"), "chain0": ("D", "v6")}, DST)

# mixes: what the kernels actually interleave
F1, F2, F3 = "v_add_u32 {d}, v4, v5", "v_and_b32 {d}, v4, v5", "v_mul_f32 {d}, v4, v5"
FMA = "v_fma_f32 {d}, v4, v5, v6"
S1, S2, S3 = "v_mad_u32_u24 {d}, v4, v5, v6", "v_dot4_u32_u8 {d}, v4, v5, v6", "v_perm_b32 {d}, v4, v5, v6"
PK = "v_pk_fma_f32 {dp}, v[4:5], v[6:7], v[8:9]"
MIX = {
    "mix: F S (add, mad) alternating": [F1, S1],
    "mix: F F S S": [F1, F1, S1, S1],
    "mix: F F F F S S S S": [F1] * 4 + [S1] * 4,
    "mix: 8 F 8 S": [F1] * 8 + [S1] * 8,
    "mix: 16 F 16 S": [F1] * 16 + [S1] * 16,
    "mix: F F S": [F1, F1, S1],
    "mix: F S S": [F1, S1, S1],
    "mix: F F F S": [F1, F1, F1, S1],
    "mix: add and alternating (both fast)": [F1, F2],
    "mix: add and mul_f32 mov (all fast)": [F1, F2, F3, "v_mov_b32 {d}, v4"],
    "mix: add + fma(vvv)": [F1, FMA],
    "mix: add + add(sgpr)": [F1, "v_add_u32 {d}, s20, v5"],
    "mix: add + s_nop 0": [F1, "s_nop 0"],
    "mix: add + s_mov_b32": [F1, "s_mov_b32 s34, s20"],
    "mix: mad + s_mov_b32": [S1, "s_mov_b32 s34, s20"],
    "mix: mad + s_nop 0": [S1, "s_nop 0"],
    "mix: add + ds_read_u16": [F1, "ds_read_u16 {d}, v14"],
    "mix: mad + ds_read_u16": [S1, "ds_read_u16 {d}, v14"],
    "mix: ds_read_u16 only": ["ds_read_u16 {d}, v14"],
    "mix: ds_read2_b32 only": ["ds_read2_b32 {dp}, v14 offset1:1"],
    "mix: ds_read_b64 only": ["ds_read_b64 {dp}, v14"],
    "mix: ds_write_b8 only": ["ds_write_b8 v14, v4"],
    "mix: ds_write_b64 only": ["ds_write_b64 v14, v[4:5]"],
    "mix: ds_read_u16 + 4 v_add_u32": ["ds_read_u16 {d}, v14", F1, F1, F1, F1],
    "mix: ds_read_u16 + 2 mad": ["ds_read_u16 {d}, v14", S1, S1],
    "mix: dot4 + and alternating": [S2, F2],
    "mix: pk_fma_f32 + add alternating": [PK, F1],
    "mix: pk_fma_f32 + 2 add": [PK, F1, F1],
    "mix: v_readlane + add": ["v_readlane_b32 s34, v4, 3", F1],
    "mix: dpp mov + add": ["v_mov_b32_dpp {d}, v4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf", F1],
    "mix: s_nop 0": ["s_nop 0"],
    "mix: s_mov_b32": ["s_mov_b32 s34, s20"],
    # v_cndmask in context (the map phase selects with it): the VOP2 form reads VCC, the VOP3 form any SGPR pair
    "mix: 8 pk_fma": [PK],
    "mix: cmp vcc, 3 pk_fma, cndmask_e32 vcc, 3 pk_fma": ["v_cmp_lt_f32_e32 vcc, 1.0, v4", PK, PK, PK, "v_cndmask_b32_e32 {d}, v5, v6, vcc", PK, PK, PK],
    "mix: cmp sgpr, 3 pk_fma, cndmask_e64 sgpr, 3 pk_fma": ["v_cmp_lt_f32_e64 s[32:33], 1.0, v4", PK, PK, PK, "v_cndmask_b32_e64 {d}, v5, v6, s[32:33]", PK, PK, PK],
    "mix: cmp vcc, 3 pk_fma, cndmask_e64 vcc, 3 pk_fma": ["v_cmp_lt_f32_e32 vcc, 1.0, v4", PK, PK, PK, "v_cndmask_b32_e64 {d}, v5, v6, vcc", PK, PK, PK],
    "mix: cmp vcc, 2 cndmask_e32 vcc, 5 pk_fma": ["v_cmp_lt_f32_e32 vcc, 1.0, v4", "v_cndmask_b32_e32 {d}, v5, v6, vcc", "v_cndmask_b32_e32 {d}, v6, v5, vcc", PK, PK, PK, PK, PK],
    "mix: cndmask_e32 vcc + 7 pk_fma (vcc never written)": ["v_cndmask_b32_e32 {d}, v5, v6, vcc", PK, PK, PK, PK, PK, PK, PK],
    "mix: cndmask_e32 vcc + 7 mad": ["v_cndmask_b32_e32 {d}, v5, v6, vcc", S1, S1, S1, S1, S1, S1, S1],
}
for label, pattern in MIX.items():
    lines = []
    for k in range(128):
        lines.append(pattern[k % len(pattern)].format(d=DST[k % 8], dp=DST_PK[k % 8]))
    kernels.append((label, lines))

# drop the (op, pattern) pairs the assembler refuses (constant-bus limits, operand kinds): one llvm-mc run over the first line of each
import subprocess, tempfile
with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
    f.write("\n".join(lines[0] for _, lines in kernels) + "\n")
r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-mc", "-arch=amdgcn", "-mcpu=gfx950", "-o", "/dev/null", f.name], capture_output=True, text=True)
bad = {int(m.group(1)) - 1 for m in re.finditer(r":(\d+):\d+: error", r.stderr)}
dropped = [kernels[i][0] for i in sorted(bad)]
kernels = [k for i, k in enumerate(kernels) if i not in bad]
print("dropped (not encodable):", ", ".join(dropped))

CLOB = ", ".join('"v%d"' % i for i in list(range(4, 16)) + list(range(40, 56))) + ', "s20", "s21", "s30", "s31", "s32", "s33", "s34", "vcc", "memory"'

out = []
out.append("// GENERATED by tools/gen_probe_issue.py -- do not edit.  Issue cost of gfx950 vector instructions by operand pattern.")
out.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <cstring>\n#include <cstdlib>")
out.append("#define HIPCHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf(\"%s: %s\\n\", #e, hipGetErrorString(r_)); return 1; } } while (0)")
for i, (label, lines) in enumerate(kernels):
    body = "\\n\\t\"\n        \"".join(lines)
    out.append("__global__ void __launch_bounds__(256) k%d(unsigned long long *stamp, int iters) {" % i)
    out.append("    extern __shared__ uint32_t lds[];")
    out.append("    asm volatile(\"v_mov_b32 v4, 0x3f800001\\n v_mov_b32 v5, 0x3f800003\\n v_mov_b32 v6, 0x3f800005\\n v_mov_b32 v7, 0x3f800007\\n\"")
    out.append("                 \"v_mov_b32 v8, 0x3f800009\\n v_mov_b32 v9, 0x3f80000b\\n v_mov_b32 v10, 3\\n v_mov_b32 v11, 7\\n v_mov_b32 v12, 0x01020304\\n v_mov_b32 v13, 0x05060708\\n\"")
    out.append("                 \"v_mov_b32 v14, 64\\n v_mov_b32 v15, 0\\n s_mov_b32 s20, 0x3f800011\\n s_mov_b32 s21, 0x3f800013\\n s_mov_b64 s[30:31], 0x5555\\n s_mov_b64 vcc, 0x3333\\n\"")
    for k in range(16): out.append("                 \"v_mov_b32 v%d, v4\\n\"" % (40 + k))
    out.append("                 ::: %s);" % CLOB)
    out.append("    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();")
    out.append("    for (int it = 0; it < iters; it++) {")
    out.append("        asm volatile(\"%s\\n\\ts_waitcnt lgkmcnt(0)\" ::: %s);" % (body, CLOB))
    out.append("    }")
    out.append("    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();")
    out.append("    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t1 - t0, stamp[1] = r1 - r0;")
    out.append("    if (iters < 0) lds[threadIdx.x] = 1;")
    out.append("}")
out.append("typedef void (*kern_t)(unsigned long long *, int);")
out.append("struct Entry { const char *label; kern_t k; };")
out.append("static const Entry table[] = {")
for i, (label, _) in enumerate(kernels): out.append("    {\"%s\", k%d}," % (label, i))
out.append("};")
out.append(r"""
int main(int argc, char **argv) {
    // waves per SIMD: workgroups of 256 threads (one wave per SIMD each), LDS sized so that exactly W fit a CU
    const char *filter = argc > 2 ? argv[2] : nullptr;
    const int iters = 400;
    unsigned long long *stamp;
    HIPCHECK(hipMalloc(&stamp, 16));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    int wlist[4] = {1, 2, 4, 8}, nw = 4;
    if (argc > 1 && atoi(argv[1]) > 0) wlist[0] = atoi(argv[1]), nw = 1;
    printf("# cycles per wave-instruction and SIMD = launch duration (HIP events) x shader clock / (instructions per wave x W waves per SIMD); shader clock from s_memtime / s_memrealtime inside each kernel; 'own' = the oldest wave's own stamps per instruction at the first W\n");
    printf("%-44s", "# op|pattern");
    printf("  oldest wave");
    for (int wi = 0; wi < nw; wi++) printf("  W=%d cyc (MHz)", wlist[wi]);
    printf("\n");
    for (const Entry &e : table) {
        if (filter && !strstr(e.label, filter)) continue;
        printf("%-44s", e.label);
        for (int wi = 0; wi < nw; wi++) {
            const int W = wlist[wi];
            const size_t lds = (size_t)(160 * 1024 / W) - (W == 1 ? 0 : 1024);
            HIPCHECK(hipFuncSetAttribute((const void *)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = 256 * W;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, stamp, 20);
            HIPCHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, stamp, iters);
            HIPCHECK(hipEventRecord(e1));
            HIPCHECK(hipEventSynchronize(e1));
            float ms;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2];
            HIPCHECK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
            const double mhz = (double)h[0] / ((double)h[1] / 100.0);  // s_memrealtime ticks at 100 MHz
            // in-kernel: cycles of wave 0 of block 0 for its own iters x 128 instructions, shared with W - 1 other waves of its SIMD
            // (the OLDEST wave of a SIMD issues nearly unimpeded whatever runs beside it, so its own stamps say nothing about
            // throughput: wave 0 of block 0 takes ~4.4 cycles per instruction at every W.  Throughput = the launch's duration.)
            const double cyc = (double)ms * 1e-3 * mhz * 1e6 / ((double)iters * 128.0 * W);
            const double own = (double)h[0] / ((double)iters * 128.0);
            if (wi == 0) printf("  %5.2f own", own);
            printf("  %6.2f (%4.0f)", cyc, mhz);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
""")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_issue.hip")
open(path, "w").write("\n".join(out) + "\n")
print("wrote", path, len(kernels), "kernels")
