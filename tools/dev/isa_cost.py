#!/usr/bin/env python3
"""tools/dev/isa_cost.py file.s first_line last_line [waves_per_simd]

Developer aid: prices a straight-line stretch of gfx950 ISA (hipcc -S) per instruction class with the
issue rates measured on this part (profiles/r04/ubench_valu_rates.txt, cycles per wave instruction per
SIMD at 1 / 2 / 4 / 8 waves per SIMD) and prints the weighted cycle count next to the plain count.
Used for the streaming loop of k_cloud_voxel (profiles/r05/voxel_instruction_mix_r05.txt)."""
import re, sys
RATES = {  # mnemonic prefix -> cycles at (1, 2, 4, 8) waves per SIMD
    "v_fma_f32": (5.60, 3.12, 2.98, 2.75), "v_fmac_f32": (4.78, 3.04, 2.81, 2.64), "v_mul_f32": (5.36, 2.92, 2.81, 2.59),
    "v_add_f32": (5.02, 2.68, 2.60, 2.53), "v_sub_f32": (4.72, 2.50, 2.41, 2.34), "v_add_u32": (4.97, 2.65, 2.57, 2.49),
    "v_sub_u32": (4.70, 2.48, 2.42, 2.38), "v_subrev_u32": (4.70, 2.48, 2.42, 2.38), "v_and_b32": (4.65, 2.41, 2.34, 2.29),
    "v_or_b32": (4.64, 2.41, 2.34, 2.28), "v_lshrrev_b32": (4.61, 2.34, 2.28, 2.26), "v_mov_b32": (4.68, 2.42, 2.35, 2.30),
    "v_pk_fma_f32": (5.81, 5.18, 4.94, 4.71), "v_pk_mul_f32": (5.49, 4.86, 4.58, 4.42), "v_pk_add_f32": (5.39, 4.78, 4.52, 4.33),
    "v_cvt_": (4.63, 4.30, 4.15, 4.10), "v_floor_f32": (4.75, 4.41, 4.26, 4.12), "v_perm_b32": (5.24, 4.66, 4.43, 4.23),
    "v_alignbit_b32": (5.11, 4.49, 4.28, 4.19), "v_bfe_": (5.08, 4.48, 4.27, 4.19), "v_cndmask_b32": (5.21, 4.64, 4.37, 4.18),
    "v_cmp_": (5.29, 4.52, 4.25, 4.11), "v_mbcnt_": (5.13, 4.50, 4.28, 4.20), "v_lshlrev_b32": (5.04, 4.44, 4.20, 4.08),
    "v_lshl_add_u64": (5.13, 4.51, 4.26, 4.17), "v_addc_co_u32": (5.29, 4.52, 4.25, 4.11), "v_readfirstlane_b32": (5.11, 4.51, 4.27, 4.17),
    "v_mov_b64": (5.11, 4.51, 4.27, 4.17),
}
DPP = (5.29, 4.65, 4.39, 4.24)
def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    w = {1: 0, 2: 1, 4: 2, 8: 3}[int(sys.argv[4]) if len(sys.argv) > 4 else 4]
    per, n, cyc, unk = {}, 0, 0.0, {}
    for ln in open(f).read().splitlines()[a - 1:b]:
        ln = ln.strip()
        m = re.match(r"(v_[a-z0-9_]+)", ln)
        if not m: continue
        op = m.group(1)
        if "dpp" in op or "row_" in ln or "wave_sh" in ln: r, cls = DPP, "dpp"
        else:
            key = next((k for k in sorted(RATES, key=len, reverse=True) if op.startswith(k)), None)
            if key is None: unk[op] = unk.get(op, 0) + 1; r, cls = (5.1, 4.5, 4.3, 4.2), "other"
            else: r, cls = RATES[key], ("fast" if RATES[key][2] < 3.2 else "slow")
        n += 1; cyc += r[w]; c = per.setdefault(cls, [0, 0.0]); c[0] += 1; c[1] += r[w]
    print(f"lines {a}-{b}: {n} vector instructions, {cyc:.0f} cycles per wave pass at {('1','2','4','8')[w]} waves per SIMD")
    for k, (c, t) in sorted(per.items()): print(f"  {k:6s} {c:4d} instr {t:7.1f} cycles")
    if unk: print("  unpriced (taken as 4.3):", unk)
if __name__ == "__main__": main()
