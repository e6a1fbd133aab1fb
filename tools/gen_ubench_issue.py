#!/usr/bin/env python3
"""Generates tools/ubench_issue.hip: hand-written gfx950 instruction streams with fixed registers, timed with
s_memtime at 1..4 waves per SIMD (one 256*W-thread workgroup per CU).  Answers, in shader cycles per
wave-instruction per SIMD: what an f64 / 32-bit / DPP / convert instruction costs to issue, what operand
sources (VGPR banks, SGPR pairs) do to it, and what ds_bpermute_b32 costs beside f64 work.

usage: python tools/gen_ubench_issue.py > tools/ubench_issue.hip
build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench_issue.hip -o tools/ubench_issue.bin
"""

def d(r):  # 64-bit register pair
    return f"v[{r}:{r + 1}]"

BODIES = {}

def body(name, lines, n_valu=None, note=""):
    BODIES[name] = (lines, len(lines) if n_valu is None else n_valu, note)

CH = [32 + 2 * i for i in range(16)]  # 16 accumulator pairs v32..v63

# f64 fma, three VGPR sources, 16 independent chains
body("fma_f64_vvv", [f"v_fma_f64 {d(c)}, {d(c)}, v[2:3], v[4:5]" for c in CH] * 4, note="x = x*b + a, b and a in VGPRs")
# fma with an SGPR pair (like the FIR taps)
body("fma_f64_svv", [f"v_fma_f64 {d(c)}, s[30:31], v[2:3], {d(c)}" for c in CH] * 4, note="acc = tap(sgpr)*x + acc")
body("add_f64", [f"v_add_f64 {d(c)}, {d(c)}, v[2:3]" for c in CH] * 4)
body("mul_f64", [f"v_mul_f64 {d(c)}, {d(c)}, v[4:5]" for c in CH] * 4)
# sources all congruent mod 4 (same VGPR banks) vs spread
body("add_f64_samebank", [f"v_add_f64 v[{64 + 4 * (i % 8)}:{65 + 4 * (i % 8)}], v[{96 + 4 * (i % 8)}:{97 + 4 * (i % 8)}], v[8:9]" for i in range(64)],
     note="dst, src0, src1 all at register index = 0 mod 4")
body("add_f64_spread", [f"v_add_f64 v[{64 + 4 * (i % 8)}:{65 + 4 * (i % 8)}], v[{98 + 4 * (i % 8)}:{99 + 4 * (i % 8)}], v[8:9]" for i in range(64)],
     note="src0 at 2 mod 4")
# a butterfly-like mix: independent adds with two different VGPR sources each
body("add_f64_2src", [f"v_add_f64 v[{64 + 2 * (i % 16)}:{65 + 2 * (i % 16)}], v[{32 + 2 * (i % 16)}:{33 + 2 * (i % 16)}], v[{32 + 2 * ((i + 5) % 16)}:{33 + 2 * ((i + 5) % 16)}]" for i in range(64)])
body("mov_b32", [f"v_mov_b32 v{64 + (i % 32)}, v{32 + (i % 32)}" for i in range(64)])
body("mov_dpp", [f"v_mov_b32_dpp v{64 + (i % 32)}, v{32 + (i % 32)} row_shr:1 row_mask:0xf bank_mask:0xf" for i in range(64)])
body("add_u32", [f"v_add_u32 v{64 + (i % 32)}, v{64 + (i % 32)}, v6" for i in range(64)])
body("sub_sdwa", [f"v_sub_u32_sdwa v{64 + (i % 32)}, v{32 + (i % 32)}, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" for i in range(64)])
body("cvt_f64_i32", [f"v_cvt_f64_i32 {d(64 + 2 * (i % 16))}, v{7 + (i % 2)}" for i in range(64)])
body("cvt_f32_f64", [f"v_cvt_f32_f64 v{64 + (i % 32)}, {d(32 + 2 * (i % 16))}" for i in range(64)])
body("cvt_f64_f32", [f"v_cvt_f64_f32 {d(64 + 2 * (i % 16))}, v{3 + 30 * (i % 2)}" for i in range(64)])
body("cndmask", [f"v_cndmask_b32 v{64 + (i % 32)}, v{32 + (i % 32)}, v6, vcc" for i in range(64)])
# mixes
mix = []
for i in range(16):
    c = CH[i]
    mix += [f"v_fma_f64 {d(c)}, {d(c)}, v[2:3], v[4:5]", f"v_add_f64 {d(64 + 2 * i)}, {d(c)}, v[2:3]",
            f"v_fma_f64 {d(c)}, s[30:31], v[2:3], {d(c)}", f"v_mov_b32_dpp v{100 + i}, v{32 + i} row_shr:1 row_mask:0xf bank_mask:0xf"]
body("mix_3f64_1dpp", mix, note="3 f64 : 1 DPP move")
mix = []
for i in range(16):
    c = CH[i]
    mix += [f"v_fma_f64 {d(c)}, {d(c)}, v[2:3], v[4:5]", f"v_add_u32 v{100 + i}, v{100 + i}, v6"]
body("mix_1f64_1u32", mix * 2, note="f64 and 32-bit integer alternating")
# ds_bpermute alone and beside f64 work (no LDS memory is touched; v9 = lane address)
body("bpermute", [f"ds_bpermute_b32 v{64 + (i % 32)}, v9, v{32 + (i % 32)}" for i in range(32)] + ["s_waitcnt lgkmcnt(0)"], n_valu=32,
     note="32 ds_bpermute_b32 then wait; 'per instr' = per bpermute")
mixb = []
for i in range(32):
    mixb.append(f"ds_bpermute_b32 v{64 + i}, v9, v{100 + (i % 16)}")
    mixb += [f"v_fma_f64 {d(CH[(2 * i) % 16])}, {d(CH[(2 * i) % 16])}, v[2:3], v[4:5]", f"v_fma_f64 {d(CH[(2 * i + 1) % 16])}, {d(CH[(2 * i + 1) % 16])}, v[2:3], v[4:5]"]
mixb.append("s_waitcnt lgkmcnt(0)")
body("fma_with_bpermute", mixb, n_valu=64, note="64 fma + 32 bpermute interleaved; per fma")
# LDS exchange beside f64 work: 8 b64 writes + 4 b128 reads per 64 fma (v10 = this lane's byte address)
mixl = []
for i in range(8):
    mixl.append(f"ds_write_b64 v10, {d(CH[i])} offset:{8192 * 0 + 512 * i}")
    mixl += [f"v_fma_f64 {d(CH[(j + 8 * i) % 16])}, {d(CH[(j + 8 * i) % 16])}, v[2:3], v[4:5]" for j in range(4)]
mixl.append("s_waitcnt lgkmcnt(0)")
for i in range(4):
    mixl.append(f"ds_read_b128 v[{64 + 4 * i}:{67 + 4 * i}], v11 offset:{1024 * i}")
    mixl += [f"v_fma_f64 {d(CH[(j + 8 * i) % 16])}, {d(CH[(j + 8 * i) % 16])}, v[2:3], v[4:5]" for j in range(8)]
mixl.append("s_waitcnt lgkmcnt(0)")
body("fma_with_lds_xchg", mixl, n_valu=64, note="64 fma + 8 ds_write_b64 + 4 ds_read_b128, two waits; per fma")

# dependent chains: what one wave's issue looks like when every instruction needs the previous result
body("dep_add_f64", ["v_add_f64 v[32:33], v[32:33], v[2:3]"] * 64, note="one dependent chain")
body("dep2_add_f64", ["v_add_f64 v[32:33], v[32:33], v[2:3]", "v_add_f64 v[34:35], v[34:35], v[2:3]"] * 32, note="two interleaved chains")
body("dep_cvt_add_cvt", ["v_cvt_f64_f32 v[32:33], v34", "v_add_f64 v[32:33], v[32:33], v[2:3]", "v_cvt_f32_f64 v34, v[32:33]"] * 21, note="the ordered sum's chain")
body("fmac_e32", [f"v_fmac_f64_e32 {d(c)}, v[2:3], v[4:5]" for c in CH] * 4, note="4-byte encoding")
def fir(width):
    # the 17-tap FIR as hipcc schedules it (width 1: add -> fmac -> add -> fmac ..., every instruction dependent on
    # the one before) and with `width` outputs interleaved; x in v64..v127 (never written), taps in s[30:31]
    out = []
    for o0 in range(0, 8, width):
        for m in range(8):
            for w in range(width):
                o = o0 + w
                p, y = 16 + 2 * w, 32 + 2 * o
                out.append(f"v_add_f64 v[{p}:{p + 1}], v[{64 + 2 * ((o + m) % 16)}:{65 + 2 * ((o + m) % 16)}], v[{96 + 2 * ((o + 15 - m) % 16)}:{97 + 2 * ((o + 15 - m) % 16)}]")
            for w in range(width):
                o = o0 + w
                p, y = 16 + 2 * w, 32 + 2 * o
                out.append(f"v_mul_f64 v[{y}:{y + 1}], s[30:31], v[{p}:{p + 1}]" if m == 0 else f"v_fmac_f64_e32 v[{y}:{y + 1}], s[30:31], v[{p}:{p + 1}]")
        for w in range(width):
            y = 32 + 2 * (o0 + w)
            out.append(f"v_fmac_f64_e32 v[{y}:{y + 1}], s[30:31], v[{80 + 2 * w}:{81 + 2 * w}]")
    return out
body("fir_serial", fir(1), note="8 outputs x 17 ops, one chain at a time (hipcc's order)")
body("fir_x2", fir(2), note="two outputs interleaved")
body("fir_x4", fir(4), note="four outputs interleaved")
body("cndmask_e64", [f"v_cndmask_b32_e64 v{64 + (i % 32)}, v{32 + (i % 32)}, v6, s[30:31]" for i in range(64)])
# the f64 matrix pipe: v_mfma_f64_16x16x4_f64 (1024 fma per instruction) alone, back to back on one accumulator, and
# with N independent v_fma_f64 behind each one — does the vector f64 pipe run beside it, from the same wave and from
# the other wave of the SIMD?  A = v[2:3] (1.0), B = v[4:5] (0.0), accumulators v[64:71] .. v[88:95].
ACC = [f"v[{64 + 8 * i}:{71 + 8 * i}]" for i in range(4)]
body("mfma_f64_indep", [f"v_mfma_f64_16x16x4_f64 {ACC[i % 4]}, v[2:3], v[4:5], {ACC[i % 4]}" for i in range(16)], note="four accumulators in turn")
body("mfma_f64_dep", [f"v_mfma_f64_16x16x4_f64 {ACC[0]}, v[2:3], v[4:5], {ACC[0]}" for i in range(16)], note="one accumulator: latency")
body("mfma_f64_4x4", [f"v_mfma_f64_4x4x4_4b_f64 v[{64 + 2 * (i % 8)}:{65 + 2 * (i % 8)}], v[2:3], v[4:5], v[{64 + 2 * (i % 8)}:{65 + 2 * (i % 8)}]" for i in range(16)], note="4 blocks of 4x4x4 (256 fma)")
for nf in (4, 8, 12, 16, 24):
    mm = []
    for i in range(8):
        mm.append(f"v_mfma_f64_16x16x4_f64 {ACC[i % 4]}, v[2:3], v[4:5], {ACC[i % 4]}")
        mm += [f"v_fma_f64 {d(CH[(j + nf * i) % 16])}, {d(CH[(j + nf * i) % 16])}, v[2:3], v[4:5]" for j in range(nf)]
    body(f"mfma_f64_fma{nf}", mm, n_valu=8, note=f"1 mfma + {nf} independent v_fma_f64; cycles per (mfma + {nf} fma)")
# the integer instructions of the statistics pass (scan_hist_word, the v_dot2 sums)
body("mad_u32_u16", [f"v_mad_u32_u16 v{64 + (i % 32)}, v{32 + (i % 32)}, 4, v6 op_sel:[{i % 2},0,0,0]" for i in range(64)], note="extract a half, * 4, + base")
body("dot2_i32_i16", [f"v_dot2_i32_i16 v{64 + (i % 32)}, v{32 + (i % 32)}, v{32 + (i % 32)}, v{64 + (i % 32)}" for i in range(64)], note="lo*lo + hi*hi + acc")
body("pk_add_u16", [f"v_pk_add_u16 v{64 + (i % 32)}, v{32 + (i % 32)}, v6" for i in range(64)])
body("pk_fma_f32", [f"v_pk_fma_f32 {d(64 + 2 * (i % 16))}, {d(32 + 2 * (i % 16))}, v[2:3], {d(64 + 2 * (i % 16))}" for i in range(64)], note="two f32 fma per lane")
body("lshl_add_u32", [f"v_lshl_add_u32 v{64 + (i % 32)}, v{32 + (i % 32)}, 2, v6" for i in range(64)])
body("bfe_u32", [f"v_bfe_u32 v{64 + (i % 32)}, v{32 + (i % 32)}, 16, 16" for i in range(64)])
body("mad_u64_u32", [f"v_mad_u64_u32 v[{64 + 2 * (i % 16)}:{65 + 2 * (i % 16)}], vcc, v{32 + (i % 32)}, v6, v[{64 + 2 * (i % 16)}:{65 + 2 * (i % 16)}]" for i in range(64)], note="32 x 32 + 64 -> 64")
body("add3_u32", [f"v_add3_u32 v{64 + (i % 32)}, v{32 + (i % 32)}, v6, v{64 + (i % 32)}" for i in range(64)])
body("addc_u64", [x for i in range(32) for x in (f"v_add_co_u32 v{64 + 2 * (i % 16)}, vcc, v{64 + 2 * (i % 16)}, v6", f"v_addc_co_u32 v{65 + 2 * (i % 16)}, vcc, 0, v{65 + 2 * (i % 16)}, vcc")], note="64-bit add as add_co + addc")
# for tools/energy_probe.py (socket power per kind of work): pure LDS streams and a stream that issues nothing
body("lds_read_b128", [f"ds_read_b128 v[{64 + 4 * (i % 8)}:{67 + 4 * (i % 8)}], v11" + (f" offset:{1024 * (i % 8)}" if i % 8 else "") for i in range(16)] + ["s_waitcnt lgkmcnt(0)"],
     n_valu=16, note="16 ds_read_b128 then wait; per read")
body("lds_write_b128", [f"ds_write_b128 v11, v[{32 + 4 * (i % 8)}:{35 + 4 * (i % 8)}]" + (f" offset:{1024 * (i % 8)}" if i % 8 else "") for i in range(16)] + ["s_waitcnt lgkmcnt(0)"],
     n_valu=16, note="16 ds_write_b128 then wait; per write")
# gfx950-only lane swaps: would they carry the half-wave hand-overs of the window kernel cheaper
# than DPP moves?  v_permlane16_swap exchanges the odd rows of vdst with the even rows of src, v_permlane32_swap the upper
# half of vdst with the lower half of src: both move ACROSS 16-lane rows, the kernel's exchanges are inside a row
body("permlane16_swap", [f"v_permlane16_swap_b32 v{64 + (i % 32)}, v{32 + (i % 32)}" for i in range(64)], note="swap odd rows of vdst with even rows of src")
body("permlane32_swap", [f"v_permlane32_swap_b32 v{64 + (i % 32)}, v{32 + (i % 32)}" for i in range(64)], note="swap upper half of vdst with lower half of src")
# plain and packed f32 arithmetic of the frequency pass (bl_fft_lavc.h: v_pk_mul / v_pk_add, fused-DPP adds were an option)
body("add_f32", [f"v_add_f32 v{64 + (i % 32)}, v{32 + (i % 32)}, v{64 + (i % 32)}" for i in range(64)])
body("pk_add_f32", [f"v_pk_add_f32 {d(64 + 2 * (i % 16))}, {d(32 + 2 * (i % 16))}, {d(64 + 2 * (i % 16))}" for i in range(64)], note="two f32 adds per lane")
body("pk_mul_f32", [f"v_pk_mul_f32 {d(64 + 2 * (i % 16))}, {d(32 + 2 * (i % 16))}, v[2:3]" for i in range(64)], note="two f32 products per lane")
body("add_f32_dpp", [f"v_add_f32_dpp v{64 + (i % 32)}, v{32 + (i % 32)}, v{64 + (i % 32)} row_ror:8 row_mask:0xf bank_mask:0xf" for i in range(64)], note="an add with its first operand from lane ^ 8")
body("idle_nop", ["s_nop 15"] * 64, note="nothing but s_nop: the clocked-but-idle baseline")
REPEAT = {k: 4 for k in BODIES}
for k in ("bpermute", "fma_with_bpermute", "fma_with_lds_xchg", "fir_serial", "fir_x2", "fir_x4"): REPEAT[k] = 2

CLOBBER = ", ".join(f'"v{i}"' for i in range(0, 128)) + ', "s20", "s21", "s22", "s23", "s24", "s25", "s30", "s31", "vcc", "memory"'

print("// GENERATED by tools/gen_ubench_issue.py — do not edit.  See that file for what is measured.")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>\n#include <vector>\n#include <algorithm>")
INIT_ACC = "".join('      "v_mov_b32 v%d, 0\\n v_mov_b32 v%d, 0x3ff00000\\n"\n' % (r, r + 1) for r in range(32, 128, 2))
# Every wave runs its stream until `dur` shader cycles have passed since the workgroup's barrier and
# reports how many passes it completed: steady-state throughput of the SIMD = passes of its waves x
# instructions / dur, with no tail in which the waves that were served first have already finished
# (a fixed amount of work per wave measures that tail as well: the first version of this tool did).
TEMPLATE = r"""
__global__ void k_@NAME@(unsigned *out, int dur, int prio_split) {
  extern __shared__ double lds[];
  unsigned t;
  if (prio_split && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);
  asm volatile(
      "v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 1\n"
      "v_mov_b32 v7, 3\n v_mov_b32 v8, 0\n"
      "v_mbcnt_lo_u32_b32 v9, -1, 0\n v_mbcnt_hi_u32_b32 v9, -1, v9\n v_xor_b32 v9, 15, v9\n v_lshlrev_b32 v9, 2, v9\n"
      "v_mbcnt_lo_u32_b32 v10, -1, 0\n v_mbcnt_hi_u32_b32 v10, -1, v10\n v_lshlrev_b32 v11, 4, v10\n v_lshlrev_b32 v10, 3, v10\n"
      "s_mov_b32 s30, 0\n s_mov_b32 s31, 0x3ff00000\n"
      "s_mov_b64 vcc, 0x5555\n"
@INIT@      "s_mov_b32 s20, %1\n"
      "s_mov_b32 s21, 0\n"
      "s_barrier\n"
      "s_memtime s[22:23]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "1:\n"
      "s_memtime s[24:25]\n"
@ASM@      "s_waitcnt lgkmcnt(0)\n"
      "s_add_u32 s21, s21, 1\n"
      "s_sub_u32 s26, s24, s22\n"
      "s_cmp_lt_u32 s26, s20\n"
      "s_cbranch_scc1 1b\n"
      "s_mov_b32 %0, s21\n"
      : "=s"(t) : "s"(dur) : @CLOBBER@);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t;
  if (dur < 0) lds[threadIdx.x] = t;
}"""
for name, (lines, nv, note) in BODIES.items():
    asm = "".join('      "%s\\n"\n' % l for l in lines * REPEAT[name])
    print(TEMPLATE.replace("@NAME@", name).replace("@INIT@", INIT_ACC).replace("@ASM@", asm).replace("@CLOBBER@", CLOBBER))

print("""
struct bench { const char *name; void (*fn)(unsigned *, int, int); int n_instr; const char *note; };
static bench benches[] = {""")
for name, (lines, nv, note) in BODIES.items():
    print('  {"%s", k_%s, %d, "%s"},' % (name, name, nv * REPEAT[name], note))
print(r"""};

int main(int argc, char **argv) {
  const int dur = 2000000, blocks = 256; /* shader cycles per run */
  unsigned *out; hipMalloc(&out, 4 * blocks * 16);
  std::vector<unsigned> h(blocks * 16);
  /* `ubench_issue.bin <name> long <waves per SIMD> <seconds>`: ONE stream on every CU for seconds on end, so that the
   * socket power and the shader clock settle and can be sampled from outside (tools/energy_probe.py); prints the
   * wave-instructions per second the whole chip retired, by the wall clock */
  if (argc > 4 && !strcmp(argv[2], "long")) {
    const int w = atoi(argv[3]);
    const double seconds = atof(argv[4]);
    for (auto &bn : benches) {
      if (strcmp(bn.name, argv[1])) continue;
      hipFuncSetAttribute(reinterpret_cast<const void *>(bn.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      bn.fn<<<blocks, 256 * w, 100 * 1024>>>(out, 20000, 0);
      hipDeviceSynchronize();
      double passes = 0, ms_total = 0;
      while (ms_total < 1e3 * seconds) {
        hipEventRecord(e0);
        bn.fn<<<blocks, 256 * w, 100 * 1024>>>(out, 1000000000, 0);   /* ~0.42 s of s_memtime ticks */
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, 4 * blocks * 4 * w, hipMemcpyDeviceToHost);
        for (int i = 0; i < blocks * 4 * w; ++i) passes += h[i];
        ms_total += ms;
      }
      printf("{\"stream\": \"%s\", \"waves_per_simd\": %d, \"seconds\": %.3f, \"wave_instr_per_s\": %.6g, "
             "\"instr_per_pass\": %d}\n", bn.name, w, ms_total / 1e3, passes * bn.n_instr / (ms_total / 1e3), bn.n_instr);
    }
    return 0;
  }
  printf("%-22s %6s %4s  %14s  %s\n", "stream", "W/SIMD", "prio", "cyc/instr/SIMD", "share of the SIMD's instructions by wave age (oldest first) | note");
  for (auto &bn : benches) {
    if (argc > 1 && !strstr(bn.name, argv[1])) continue;
    for (int w = 1; w <= 4; ++w) {
      for (int prio = 0; prio <= (w == 2 ? 1 : 0); ++prio) {
        const int threads = 256 * w;
        hipFuncSetAttribute(reinterpret_cast<const void *>(bn.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        bn.fn<<<blocks, threads, 100 * 1024>>>(out, 20000, prio);
        bn.fn<<<blocks, threads, 100 * 1024>>>(out, dur, prio);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 4 * blocks * 4 * w, hipMemcpyDeviceToHost);
        /* wave i of a block sits on SIMD i % 4; waves i, i + 4, i + 8 ... share it, in dispatch (age) order */
        double tot = 0, byage[4] = {0, 0, 0, 0};
        for (int b = 0; b < blocks; ++b)
          for (int i = 0; i < 4 * w; ++i) { tot += h[b * 4 * w + i]; byage[i / 4] += h[b * 4 * w + i]; }
        const double per_simd = tot / (blocks * 4) * bn.n_instr;
        printf("%-22s %6d %4d  %14.2f  ", bn.name, w, prio, dur / per_simd);
        for (int a = 0; a < w; ++a) printf("%.2f ", byage[a] / tot);
        printf("| %s\n", w == 1 ? bn.note : "");
      }
    }
  }
  return 0;
}""")
