#!/usr/bin/env python3
"""k_mlp_grad_v5's LDS address maps (csrc/qenv_grad5.hpp), checked on the CPU before any launch:

  1. LOGIC: every lane address formula of the kernel is replayed on a model of the LDS (a dict byte address -> (row, position)
     tag written by the store formulas) and of gfx950's ds_read_b64_tr_b16 (per 16-lane group: lane 4q + p supplies the address
     of row q, columns 4p .. 4p + 3; lane i receives column i of the four rows): the fragment a lane ends up with must be the
     MFMA operand element the kernel's comments claim (v_mfma_f32_16x16x32_bf16: A lane (i = l % 16, k = 8 (l / 16) + e), B lane
     (j = l % 16, same k), D lane (j = l % 16, i = 4 (l / 16) + r)).
  2. BANKS: the worst multiplicity per instruction under MI355X_MICROARCH.md's LDS table (ds_read_b128: four fixed 16-lane
     groups, bank (a / 4) mod 64; ds_read_b64_tr_b16: two 32-lane halves, mod 64; ds_write_b64: 4 x 16 contiguous lanes, mod 32;
     ds_write_b128: 8 x 8 contiguous, mod 32).

Exit code 0 = every map is the claimed operand and only the accumulator fragments' 8-byte stores conflict (two-way).
"""
import sys
from collections import Counter

ROW, TERM, PTERM = 256, 8192, 1408


def swz(row):
    return (row & 15) ^ (12 if row & 1 else 0)


def img_off(row, ch):
    return ROW * row + 16 * (ch ^ swz(row))


def p_off(row, slot):
    return 32 * row + 128 * (row >> 3) + 8 * (slot ^ ((row >> 2) & 3))


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in grp] for grp in B128_GROUPS]


def worst(groups, addrs, nbytes, mod):
    """Largest number of distinct dwords on one bank within a lane group (identical addresses broadcast)."""
    w = 1
    for grp in groups:
        per_bank = {}
        for l in grp:
            for i in range(nbytes // 4):
                dw = addrs[l] // 4 + i
                per_bank.setdefault(dw % mod, set()).add(dw)
        w = max(w, max(len(v) for v in per_bank.values()))
    return w


def lane_parts(lane, wave):
    c16, g = lane & 15, lane >> 4
    q, p = c16 >> 2, c16 & 3
    fc = swz(c16)
    d = dict(c16=c16, g=g, q=q, p=p)
    d["rr_base"] = ROW * c16 + 16 * (g ^ fc)
    d["aw_base"] = ROW * c16 + 16 * ((2 * wave + (g >> 1)) ^ fc) + 8 * (g & 1)
    d["tr0"] = ROW * (8 * g + q) + 16 * ((p >> 1) ^ swz(8 * g + q)) + 8 * (p & 1)
    d["tr1"] = ROW * (8 * g + q + 4) + 16 * ((p >> 1) ^ swz(8 * g + q + 4)) + 8 * (p & 1)
    d["pw_base"] = p_off(c16, g)
    d["pt0"], d["pt1"] = p_off(8 * g + q, p), p_off(8 * g + q + 4, p)
    return d


def tr_read(mem, addrs):
    """addrs[lane] -> per lane the 4 tags it receives (ds_read_b64_tr_b16)."""
    out = [None] * 64
    for grp in range(4):
        rows = []
        for qq in range(4):
            row = []
            for pp in range(4):
                a = addrs[16 * grp + 4 * qq + pp]
                row += [mem[a + 2 * i] for i in range(4)]
            rows.append(row)          # 16 columns of block row qq
        for i in range(16):
            out[16 * grp + i] = [rows[qq][i] for qq in range(4)]
    return out


def main():
    bad = 0
    # ---- a shared image written as accumulator fragments by 8 waves: tag = (sample, hidden)
    mem = {}
    wconf = 1
    for wave in range(8):
        for n in range(2):
            addrs = {}
            for lane in range(64):
                L = lane_parts(lane, wave)
                a = 4096 * n + L["aw_base"]
                addrs[lane] = a
                for r in range(4):
                    key = a + 2 * r
                    assert key not in mem, "two stores to one element"
                    mem[key] = (16 * n + L["c16"], 16 * wave + 4 * L["g"] + r)
            wconf = max(wconf, worst([list(range(16 * k, 16 * k + 16)) for k in range(4)], addrs, 8, 32))
    assert len(mem) == 32 * 128
    print(f"accumulator-fragment stores (ds_write_b64): {wconf}-way")
    # row reads: B operand of a chain, k-step ks, half n: lane (j = c16 -> sample 16 n + c16, k = 32 ks + 8 g + e)
    rconf = 1
    for ks in range(4):
        for n in range(2):
            addrs = {}
            for lane in range(64):
                L = lane_parts(lane, 0)
                a = 4096 * n + (L["rr_base"] ^ (64 * ks))
                addrs[lane] = a
                for e in range(8):
                    if mem[a + 2 * e] != (16 * n + L["c16"], 32 * ks + 8 * L["g"] + e):
                        bad += 1
            rconf = max(rconf, worst(B128_GROUPS, addrs, 16, 64))
    print(f"row reads (ds_read_b128): {rconf}-way, wrong elements {bad}")
    # transposing reads of column block t: lane (i = 16 t + c16, k = sample 8 g + e)
    tconf = 1
    for t in range(8):
        a0, a1 = {}, {}
        for lane in range(64):
            L = lane_parts(lane, 0)
            a0[lane] = L["tr0"] ^ (32 * t)
            a1[lane] = L["tr1"] ^ (32 * t)
        lo, hi = tr_read(mem, a0), tr_read(mem, a1)
        for lane in range(64):
            c16, g = lane & 15, lane >> 4
            frag = lo[lane] + hi[lane]
            for e in range(8):
                if frag[e] != (8 * g + e, 16 * t + c16):
                    bad += 1
        halves = [list(range(32)), list(range(32, 64))]
        tconf = max(tconf, worst(halves, a0, 8, 64), worst(halves, a1, 8, 64))
    print(f"transposing reads (ds_read_b64_tr_b16): {tconf}-way, wrong elements so far {bad}")
    # the observations' stores: waves 0..4, lane (sample = lane & 31, chunk 2 w + (lane >> 5)), 16 bytes: positions 8 chunk + i
    xmem, xconf = {}, 1
    for wave in range(5):
        addrs = {}
        for lane in range(64):
            xs, xh = lane & 31, lane >> 5
            a = img_off(xs, 2 * wave + xh)
            addrs[lane] = a
            for i in range(8):
                assert a + 2 * i not in xmem
                xmem[a + 2 * i] = (xs, 8 * (2 * wave + xh) + i)
        xconf = max(xconf, worst([list(range(8 * k, 8 * k + 8)) for k in range(8)], addrs, 16, 32))
    for key, tag in xmem.items():
        if mem[key] != tag:      # the same (row, position) map as the accumulator-fragment image
            bad += 1
    print(f"observation stores (ds_write_b128): {xconf}-way")
    # ---- the 32-byte-row images: written as accumulator fragments of ONE wave (tag = (sample, position 4 g + r))
    pm, pconf = {}, 1
    for n in range(2):
        addrs = {}
        for lane in range(64):
            L = lane_parts(lane, 0)
            a = 768 * n + L["pw_base"]
            assert a == p_off(16 * n + L["c16"], L["g"])
            addrs[lane] = a
            for r in range(4):
                assert a + 2 * r not in pm
                pm[a + 2 * r] = (16 * n + L["c16"], 4 * L["g"] + r)
        pconf = max(pconf, worst([list(range(16 * k, 16 * k + 16)) for k in range(4)], addrs, 8, 32))
    assert max(pm) < PTERM
    a0 = {lane: lane_parts(lane, 0)["pt0"] for lane in range(64)}
    a1 = {lane: lane_parts(lane, 0)["pt1"] for lane in range(64)}
    lo, hi = tr_read(pm, a0), tr_read(pm, a1)
    for lane in range(64):
        c16, g = lane & 15, lane >> 4
        frag = lo[lane] + hi[lane]
        for e in range(8):
            if frag[e] != (8 * g + e, c16):
                bad += 1
    halves = [list(range(32)), list(range(32, 64))]
    ptc = max(worst(halves, a0, 8, 64), worst(halves, a1, 8, 64))
    print(f"32-byte-row image: stores {pconf}-way, transposing reads {ptc}-way, wrong elements in total {bad}")
    ok = bad == 0 and rconf == 1 and tconf == 1 and xconf == 1 and pconf == 1 and ptc == 1 and wconf <= 2
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
