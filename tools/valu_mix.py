#!/usr/bin/env python3
"""Weighted VALU-issue ceiling of the extraction path.

Every vector instruction of the gfx950 code objects is put into one of the MEASURED issue classes (profiles/r02_valu_issue_rates.txt, 8 waves per
SIMD): `fast` = 2.4 cycles per wave64 instruction and SIMD (v_add/sub_u32, v_and/or/xor/not_b32, v_mov_b32, right shifts, v_fma/add/sub/mul/
fmac_f32, v_min_u16, v_sub_u16), `slow` = 4.2 (everything else: min/max, three-operand, packed, SDWA / DPP forms, compares, conversions,
v_lshlrev, multiplies), `v8` = 8.2 (v_min3_i16 and friends).  A kernel's loops are weighted with their trip counts per wave -- analytic from the
launch geometry of the 640x480 / 8-level / 1000-feature workload where they are, fitted to the kernel's own SQ counters where they depend on
the data (the number of FAST score batches per cell, ...) -- and the prediction is CLOSED against the measured SQ_INSTS_VALU of the same launch:
`closure` = predicted / measured wave-instructions (1.0 = the weights account for every instruction the hardware counted).

usage: valu_mix.py --sq profiles/r03_pmc_sq_counters.json [--fps F] [--out profiles/r03_valu_mix.json]
(compiles rumi_slam_amd/csrc/*.hip to assembly with hipcc -S; no GPU needed)"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rumi_slam_amd", "csrc")
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_fma_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_min_u16", "v_sub_u16", "v_add_u16", "v_xnor_b32"}
V8 = {"v_min3_i16", "v_max3_i16", "v_med3_i16", "v_min3_u16", "v_max3_u16"}
CYC = {"fast": 2.4, "slow": 4.2, "v8": 8.2}


def classify(op, line):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if base in V8:
        return "v8"
    if op.endswith(("_sdwa", "_dpp")) or " row_" in line or " quad_perm" in line or "dst_sel" in line:
        return "slow"
    return "fast" if base in FAST else "slow"


def assemble(src):
    out = os.path.join(tempfile.gettempdir(), "valu_mix_" + os.path.basename(src) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "--cuda-device-only", "-S", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels_of(lines):
    """name -> list of (label, loop_header_or_None, depth, [ (op, line) ])"""
    ks, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            name = m.group(1); cur = [["entry", None, 0, []]]; ks[name] = cur
            continue
        if cur is None:
            continue
        if ln.startswith("\ts_endpgm"):
            cur[-1][3].append(("s_endpgm", ln)); cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            c = m.group(2) or ""
            hdr, depth = None, 0
            mh = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", c)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            if mh:
                hdr, depth = m.group(1)[1:] if False else m.group(1).lstrip("."), int(mh.group(1))
                hdr = hdr[1:] if hdr.startswith("L") else hdr           # LBB3_8 -> BB3_8
            elif mi:
                hdr, depth = mi.group(1), int(mi.group(2))
            cur.append([m.group(1), hdr, depth, []])
            continue
        mc = re.match(r"^\s+;\s*=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", ln)      # continuation comment lines of a label
        if mc and not cur[-1][3]:
            lab = cur[-1][0].lstrip(".")
            cur[-1][1] = lab[1:] if lab.startswith("L") else lab
            cur[-1][2] = int(mc.group(1))
            continue
        m = re.match(r"^\s+([a-z]\w+)", ln)
        if m and not ln.lstrip().startswith((";", ".")):
            cur[-1][3].append((m.group(1), ln))
    return ks


def count(block_ops):
    c = collections.Counter()
    for op, ln in block_ops:
        if op.startswith("v_"):
            c[classify(op, ln)] += 1; c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_"):
            if op in ("s_waitcnt", "s_nop", "s_endpgm") or op.startswith(("s_load", "s_buffer_load")):
                if op.startswith(("s_load", "s_buffer")): c["smem"] += 1
            else:
                c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    return c


def loops_of(blocks):
    """innermost-loop regions: header -> Counter, plus 'straight' for blocks outside loops; also the signature ops per region"""
    reg, sig = collections.defaultdict(collections.Counter), collections.defaultdict(set)
    for lab, hdr, depth, ops in blocks:
        key = hdr if hdr else "straight"
        reg[key] += count(ops)
        for op, _ in ops:
            sig[key].add(re.sub(r"_(e32|e64|sdwa|dpp)$", "", op))
    return reg, sig


if __name__ == "__main__" and "--dump" in sys.argv:
    for f in sys.argv[sys.argv.index("--dump") + 1:]:
        ks = kernels_of(assemble(os.path.join(CSRC, f)))
        for name, blocks in ks.items():
            if "rumi" not in name: continue
            reg, sig = loops_of(blocks)
            tot = sum((r for r in reg.values()), collections.Counter())
            print(f"== {name[:70]}  VALU {tot['valu']} (fast {tot['fast']} slow {tot['slow']}) LDS {tot['lds']} SALU {tot['salu']}")
            for k, c in reg.items():
                marks = [s for s in ("v_bfi_b32", "v_pk_minimum3_f16", "ds_read_u8_d16_hi", "v_cmp_gt_u16", "global_store_dword", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_perm_b32", "v_bcnt_u32_b32", "v_mfma", "v_mbcnt_lo_u32_b32", "v_mul_hi_u32", "v_cvt_f32_ubyte0", "v_fma_f32", "v_sad_u8", "ds_bpermute_b32") if s in sig[k]]
                print(f"   {k:12s} VALU {c['valu']:5d} fast {c['fast']:5d} slow {c['slow']:5d} v8 {c['v8']:3d} LDS {c['lds']:4d} SALU {c['salu']:4d} VMEM {c['vmem']:3d}  {marks}")


# ---------------------------------------------------------------------------------------------------------------------------------------
# The workload's launch geometry (640x480, 8 levels, scale 1.2, W = 35: R/lib_src/ORBextractor.cc:410-438, 729-763), restated here so that
# the tool runs without the library
def level_sizes(w=640, h=480, n=8, sf=1.2):
    import numpy as np
    out, sc = [], np.float32(1.0)
    for l in range(n):
        if l:
            sc = np.float32(np.float64(sc) * np.float64(np.float32(sf)))
        inv = np.float32(1.0) / sc
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def fast_cells(w=640, h=480):
    """(dw, dh) of every FAST cell of a frame"""
    cells = []
    for lw, lh in level_sizes(w, h):
        minB, maxBX, maxBY = 16, lw - 16, lh - 16
        W_, H_ = maxBX - minB, maxBY - minB
        nCols, nRows = W_ // 35, H_ // 35
        wCell, hCell = -(-W_ // nCols), -(-H_ // nRows)
        for i in range(nRows):
            iniY = minB + i * hCell
            maxY = min(iniY + hCell + 6, maxBY)
            if iniY >= maxBY - 3:
                continue
            for j in range(nCols):
                iniX = minB + j * wCell
                maxX = min(iniX + wCell + 6, maxBX)
                if iniX >= maxBX - 6 or maxX - iniX < 7 or maxY - iniY < 7:
                    continue
                cells.append((maxX - iniX - 6, maxY - iniY - 6))
    return cells


def region_table(blocks):
    reg, sig = loops_of(blocks)
    return {k: dict(c, sig=sorted(sig[k])) for k, c in reg.items()}


def pick(regs, *ops):
    """the loop regions whose instructions include all of `ops`"""
    return [k for k, r in regs.items() if k != "straight" and all(o in r["sig"] for o in ops)]


def mix_of(parts):
    """parts: list of (Counter-like region, executions per wave) -> totals per wave"""
    t = collections.Counter()
    for r, n in parts:
        for k in ("valu", "fast", "slow", "v8", "lds", "salu"):
            t[k] += r.get(k, 0) * n
    return t


def solve2(a11, a12, a21, a22, b1, b2):
    det = a11 * a22 - a12 * a21
    return ((b1 * a22 - a12 * b2) / det, (a11 * b2 - a21 * b1) / det) if abs(det) > 1e-9 else (0.0, 0.0)


def analyse(sq, fps=None):
    asm = {f: kernels_of(assemble(os.path.join(CSRC, f))) for f in ("orb_kernels.hip", "orb_octree_kernel.hip", "match.hip")}

    def kernel(f, needle):
        for name, blocks in asm[f].items():
            if needle in name:
                return region_table(blocks)
        raise KeyError(needle)
    out = {}
    # ---- k_fast_cells: prologue once per wave (= cell), the quick-test loop once per 64 four-pixel groups (analytic from the cell grid), the exact
    # batches and the NMS sweeps data-dependent: fitted to the kernel's own VALU and LDS counts (two unknowns, two equations)
    c = sq["k_fast_cells"]
    waves = c["SQ_WAVES"]
    regs = kernel("orb_kernels.hip", "k_fast_cellsILi48")
    cells = fast_cells()
    steps = sum(-(-(-(-dw // 4) * dh) // 64) for dw, dh in cells) / len(cells)
    quick, = pick(regs, "v_bfi_b32")
    batch = pick(regs, "v_pk_minimum3_f16")
    nms, = pick(regs, "global_store_dword", "v_mbcnt_lo_u32_b32")
    bavg = {k: sum(regs[b].get(k, 0) for b in batch) / len(batch) for k in ("valu", "fast", "slow", "v8", "lds", "salu")}
    fixed = mix_of([(regs["straight"], 1), (regs[quick], steps)])
    vw, lw = c["SQ_INSTS_VALU"] / waves, c["SQ_INSTS_LDS"] / waves
    nb, nw = solve2(bavg["valu"], regs[nms]["valu"], bavg["lds"], regs[nms]["lds"], vw - fixed["valu"], lw - fixed["lds"])
    tot = mix_of([(regs["straight"], 1), (regs[quick], steps), (bavg, nb), (regs[nms], nw)])
    out["k_fast_cells"] = dict(waves=waves, per_wave={"prologue": 1, "quick-test steps (analytic: cell grid)": round(steps, 3), "exact-score batches (fitted)": round(nb, 3),
                                                      "NMS + emission sweeps (fitted)": round(nw, 3)},
                               closure_salu=round(tot["salu"] / (c["SQ_INSTS_SALU"] / waves), 3), mix=tot, fit="VALU and LDS counts closed by construction (2 unknowns); SALU is the check")
    # ---- kernels whose hot code is one loop: its trips per wave fitted to VALU, LDS as the check
    for name, f, needle, ops in (("k_blur", "orb_kernels.hip", "k_blurILi0ELi64", ("v_dot4_u32_u8",)),):
        c = sq[name]; waves = c["SQ_WAVES"]
        regs = kernel(f, needle)
        loop, = pick(regs, *ops)
        trips = (c["SQ_INSTS_VALU"] / waves - regs["straight"]["valu"]) / regs[loop]["valu"]
        tot = mix_of([(regs["straight"], 1), (regs[loop], trips)])
        out[name] = dict(waves=waves, per_wave={"prologue": 1, "strip loop, 7 source rows per trip (fitted to VALU)": round(trips, 3)},
                         closure_lds=round(tot["lds"] / max(c.get("SQ_INSTS_LDS", 0) / waves, 1e-9), 3) if c.get("SQ_INSTS_LDS") else None, mix=tot)
    # ---- the rest: the static mix of the whole kernel (straight-line or fully unrolled hot code; the alternative paths have the same classes)
    for name, f, needle in (("k_resize", "orb_kernels.hip", "k_resize"), ("k_orient_desc", "orb_kernels.hip", "k_orient_descILi2"), ("k_compact", "orb_kernels.hip", "k_compact"),
                            ("k_octree", "orb_octree_kernel.hip", "k_octreeILb0ELi256"), ("k_assemble", "orb_octree_kernel.hip", "k_assemble"), ("k_bruteforce", "match.hip", "k_bruteforce")):
        if name not in sq:
            continue
        c = sq[name]; waves = c["SQ_WAVES"]
        try:
            regs = kernel(f, needle)
        except KeyError:
            regs = kernel("orb_kernels.hip" if f != "orb_kernels.hip" else "orb_octree_kernel.hip", needle)
        tot = mix_of([(r, 1) for r in regs.values()])
        scale = (c["SQ_INSTS_VALU"] / waves) / max(tot["valu"], 1)
        loops = {k: r for k, r in regs.items() if k != "straight" and r.get("valu", 0) > 0}
        if scale > 1.5 and loops:
            # the count is dominated by a loop: every region once, the loop with the most vector instructions as often as the count needs
            hot = max(loops, key=lambda k: loops[k]["valu"])
            trips = max((c["SQ_INSTS_VALU"] / waves - (tot["valu"] - regs[hot]["valu"])) / regs[hot]["valu"], 1.0)
            tot = mix_of([(r, trips if k == hot else 1) for k, r in regs.items()])
            out[name] = dict(waves=waves, per_wave={"every region once; hottest loop (%d vector instructions) trips fitted to VALU" % regs[hot]["valu"]: round(trips, 2)},
                             closure_lds=round(tot["lds"] / (c["SQ_INSTS_LDS"] / waves), 3) if c.get("SQ_INSTS_LDS") else None, mix=tot)
        else:
            out[name] = dict(waves=waves, per_wave={"whole kernel, static mix; executed / static instructions": round(scale, 3)}, mix={k: v * scale for k, v in tot.items()})
    # ---- totals
    res = {"classes_cycles": CYC, "class_source": "profiles/r02_valu_issue_rates.txt (8 waves per SIMD)", "kernels": {}}
    total_cycles = total_insts = 0.0
    for name, o in out.items():
        m, waves = o["mix"], o["waves"]
        launches = 7 if name == "k_resize" else 1
        v = max(m["valu"], 1e-9)
        ff, fs, f8 = m["fast"] / v, m["slow"] / v, m["v8"] / v
        insts = sq[name]["SQ_INSTS_VALU"] * launches
        avg = ff * CYC["fast"] + fs * CYC["slow"] + f8 * CYC["v8"]
        res["kernels"][name] = {"valu_wave_insts_per_256_frames": int(insts), "fast_frac": round(ff, 4), "slow_frac": round(fs, 4), "v8_frac": round(f8, 4),
                                "avg_cycles_per_inst": round(avg, 3), "weighted_cycles": int(insts * avg), "per_wave": o["per_wave"],
                                **{k: o[k] for k in ("closure_salu", "closure_lds", "fit") if k in o and o[k] is not None}}
        total_cycles += insts * avg; total_insts += insts
    res["valu_wave_insts_per_256_frames"] = int(total_insts)
    res["weighted_cycles_per_256_frames"] = int(total_cycles)
    res["avg_cycles_per_inst"] = round(total_cycles / total_insts, 3)
    if fps:
        simd_cycles = 1024 * 2.4e9 * (256.0 / fps)
        res["fps"] = fps
        res["frac_of_weighted_ceiling"] = round(total_cycles / simd_cycles, 4)
    return res


if __name__ == "__main__" and "--dump" not in sys.argv:
    a = sys.argv
    sq = json.load(open(a[a.index("--sq") + 1]))["kernels"]
    fps = float(a[a.index("--fps") + 1]) if "--fps" in a else None
    r = analyse(sq, fps)
    txt = json.dumps(r, indent=1)
    if "--out" in a:
        open(a[a.index("--out") + 1], "w").write(txt + "\n")
    for k, v in r["kernels"].items():
        print(f"{k:15s} VALU {v['valu_wave_insts_per_256_frames']/1e6:7.1f} M  fast {v['fast_frac']:.3f} slow {v['slow_frac']:.3f} v8 {v['v8_frac']:.3f}  avg {v['avg_cycles_per_inst']:.2f} cyc  {v['per_wave']}  " +
              " ".join(f"{c}={v[c]}" for c in ("closure_salu", "closure_lds") if c in v))
    print("total", r["valu_wave_insts_per_256_frames"] / 1e6, "M instructions,", r["weighted_cycles_per_256_frames"] / 1e6, "M weighted SIMD-cycles, average", r["avg_cycles_per_inst"], "cycles;",
          "fraction of the weighted ceiling:", r.get("frac_of_weighted_ceiling"))
