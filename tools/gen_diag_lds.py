#!/usr/bin/env python3
"""Writes algebra_amd/csrc/malcs_diag_lds.inc: the instruction stream of the window kernel's two-limb mode with the
match masks looked up in LDS (one asm statement for all blocks of a piece, k_llcs_diag / diag_halves_lds).

Why generated: the stream is software-pipelined over block boundaries (the LDS read of a column is issued LOOKAHEAD
columns before its use, into a ring of register pairs) and every s_waitcnt carries the exact number of LDS operations
issued after the one it waits for; that bookkeeping is easier to get right in a script than in nested C macros.

    python tools/gen_diag_lds.py            # rewrites the .inc; the library's Makefile does not run it

The stream, per block of 32 columns (i = block, e = i % 2; the loop body is an even and an odd block):
  start   masks of the limb that enters (w[i+2]; four symbols) from its planes; loads of the rows of block i + 1;
          the masks 16 rows further down (alignbit with the masks of w[i+1]) -> LDS, shifted slot (i + 1) % 2
  half 0  16 columns on rows (w[i], w[i+1])                      reads: aligned slots i % 2, (i + 1) % 2
  middle  the window moves down 16 rows; masks of w[i+2] -> LDS, aligned slot i % 2 (w[i] is not read any more)
  half 1  16 columns on rows ((w[i+1]:w[i]) >> 16, (w[i+2]:w[i+1]) >> 16)   reads: shifted slots i % 2, (i + 1) % 2
  end     the window moves down 16 rows
A column: v_and_b32_sdwa (the column's 2-bit symbol straight into byte 1 of the LDS address) + ds_read2st64_b32 (both
limbs' masks) for the column LOOKAHEAD ahead, then per limb v_and_b32, v_add(c)_co_u32, v_bitop3_b32; limb 1 lags one column
behind limb 0 as in the plane-folding stream, so the carry crosses between steps.
Every block starts by setting the wave's priority to the number of blocks it has left, mod 4 (see set_priority).
LDS of a wave: 16 rows of 64 dwords: row (ring * 2 + slot) * 4 + symbol, ring 0 = aligned, 1 = shifted by 16.
"""
import os

LOOKAHEAD = int(os.environ.get("GEN_LOOKAHEAD", "6"))
RING = int(os.environ.get("GEN_RING", "8"))     # register pairs v[64:65] ... (v[78:79] for 8)
SDWA = int(os.environ.get("GEN_SDWA", "1"))     # the symbol goes straight into byte 1 of the LDS address (one SDWA instruction per column)
NT = " nt" if int(os.environ.get("GEN_NT", "1")) else ""
EXP = os.environ.get("GEN_EXP", "")             # timing experiments (wrong results): noread / nowrite / noaddr / noload / nowaitvm

ADD64 = int(os.environ.get("GEN_ADD64", "0"))   # experiment: one v_lshl_add_u64 per column (both limbs on the same column) instead of two carry-chained additions; measured 23.2 against 22.2 us per C2 launch, not adopted
if ADD64:
    V0, V1 = "v20", "v21"         # even-aligned pairs: V v[20:21], U v[22:23], S v[26:27]
    U0, U1, S0, S1 = "v22", "v23", "v26", "v27"
else:
    V0, V1 = "v21", "v24"
    U0, S0, U1, S1 = "v22", "v23", "v26", "v27"
BASE = "v30"                  # lane * 4 (+ the table's LDS offset, a multiple of 1024)
ADDR = ["v31", "v35"]
TEXT = [("v32", "v36"), ("v33", "v37")]     # 2-bit symbols of columns 0-15 / 16-31 of an even / odd block
PL0, PL1 = "v28", "v29"       # planes of the entering limb
MASKS = [["v40", "v41", "v42", "v43"], ["v44", "v45", "v46", "v47"]]
SHIFTED = ["v48", "v49", "v50", "v51"]
SHT = ["v38", "v39", "v53"]    # the text word moved down by 2, 4, 6 bits
THREE = "v54"
CARRY = ["s[40:41]", "s[44:45]"]
CARRY_TOP = "s[48:49]"


JOINPRE = int(os.environ.get("GEN_JOINPRE", "1"))      # the other part's join word is asked for when this many blocks are left (0: never)
PRIO = int(os.environ.get("GEN_PRIO", "1"))            # the wave's priority follows the blocks it has left
ALIGN4 = int(os.environ.get("GEN_ALIGN4", "0"))       # keep 64-bit encodings at addresses = 4 (mod 8), see DESIGN section 4
ALIGNED_KINDS = ("v_", "ds_")


def insn_size(s):
    """Bytes of an instruction of this stream (gfx950 encodings)."""
    op = s.split()[0]
    if op.endswith(":"):
        return 0
    if op.startswith("s_"):
        return 4
    if op in ("v_and_b32", "v_xor_b32", "v_add_u32", "v_bcnt_u32_b32"):
        return 8 if "0x" in s else 4          # VOP2, + 4 with a literal
    return 8                                  # VOP3, DS, GLOBAL


def ring(col):
    assert 32 % RING == 0 and RING >= LOOKAHEAD + 2     # the columns of a block are numbered from 0 again
    k = col % RING
    return f"v[{64 + 2 * k}:{65 + 2 * k}]", f"v{64 + 2 * k}", f"v{65 + 2 * k}"


class Stream:
    def __init__(self):
        self.lines = []
        self.ds_issued = 0
        self.read_at = {}         # column (absolute, block-relative numbering below) -> index of its ds_read
        self.addr_turn = 0
        self.pos = 0              # bytes from the last .p2align 3
        self.nops = 0

    def emit(self, s):
        size = insn_size(s)
        if ALIGN4 and size == 8 and self.pos % 8 == 0 and s.startswith(ALIGNED_KINDS):
            self.lines.append("s_nop 0")
            self.pos += 4
            self.nops += 1
        self.lines.append(s)
        self.pos += size

    def ds(self, s):
        if s.startswith("ds_write") and EXP in ("nowrite", "noread"):
            if EXP == "noread":
                self.emit(s)          # not counted: nothing waits for it
            return
        self.emit(s)
        self.ds_issued += 1

    def wait_read(self, key):
        later = self.ds_issued - self.read_at.pop(key) - 1
        assert 0 <= later <= 15, later
        if EXP != "noread":
            self.emit(f"s_waitcnt lgkmcnt({later})")


def text_of(e, G):
    """Register and column (0 ... 15) inside it of column G (0 ... 31 this block of parity e, 32 ... the next block)."""
    if G < 16:
        return TEXT[e][0], G
    if G < 32:
        return TEXT[e][1], G - 16
    return TEXT[1 - e][0], G - 32


def read_column(st, e, G, key):
    """Issue the LDS read of column G (0 ... 31 this block of parity e, 32 ... the next block)."""
    if G < 16:
        lo, hi = 4 * e, 4 * (1 - e)             # aligned ring, slots i % 2 and (i + 1) % 2
    elif G < 32:
        lo, hi = 8 + 4 * e, 8 + 4 * (1 - e)     # shifted ring
    else:
        lo, hi = 4 * (1 - e), 4 * e
    a = ADDR[st.addr_turn]
    st.addr_turn ^= 1
    pair, _, _ = ring(G)
    text, k = text_of(e, G)
    if EXP == "noaddr":
        a = BASE
    elif SDWA:
        # Address = lane * 4 (byte 0, set once) | symbol << 8 (byte 1): the symbol of column k sits in the low bits of
        # byte k / 4 of the text word moved down by 2 (k % 4) bits; three moved copies per word, made when its first
        # column comes up.  One instruction per column where v_bfe_u32 + v_lshl_or_b32 are two.
        if k == 0:
            for j in (1, 2, 3):
                st.emit(f"v_lshrrev_b32 {SHT[j - 1]}, {2 * j}, {text}")
        src = text if k % 4 == 0 else SHT[k % 4 - 1]
        st.emit(f"v_and_b32_sdwa {a}, {src}, {THREE} dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{k // 4} src1_sel:DWORD")
    else:
        st.emit(f"v_bfe_u32 {a}, {text}, {2 * k}, 2")
        st.emit(f"v_lshl_or_b32 {a}, {a}, 8, {BASE}")
    st.read_at[key] = st.ds_issued
    if EXP == "noread":
        st.ds_issued += 1
    else:
        st.ds(f"ds_read2st64_b32 {pair}, {a} offset0:{lo} offset1:{hi}")


def window_down(st):
    st.emit(f"v_and_b32 {U0}, 0xffff, {V0}")
    st.emit(f"v_bcnt_u32_b32 %[bank], {U0}, %[bank]")
    st.emit(f"v_alignbit_b32 {V0}, {V1}, {V0}, 16")
    st.emit(f"v_alignbit_b32 {V1}, -1, {V1}, 16")


def set_priority(st, e):
    """Priority of the wave for this block = blocks it still has to do, mod 4: of the waves that share a SIMD the one
    that is behind is served first (three blocks out of four when they are one block apart), so they finish together.
    Left alone, the SIMD serves its oldest wave first, which then finishes a third ahead of the youngest and leaves it
    to run the rest alone, at the rate of one wave (tools/diag_timeline.py)."""
    st.emit("s_and_b32 s42, %[cnt], 3")
    for k in (0, 1, 2):
        st.emit(f"s_cmp_eq_u32 s42, {k}")
        st.emit(f"s_cbranch_scc0 Ldiag_p{k}{e}%=")
        st.emit(f"s_setprio {k}")
        st.emit(f"s_branch Ldiag_pd{e}%=")
        st.emit(f"Ldiag_p{k}{e}%=:")
    st.emit("s_setprio 3")
    st.emit(f"Ldiag_pd{e}%=:")


def block(st, e, row_offset):
    new, prev = MASKS[e], MASKS[1 - e]
    if PRIO:
        set_priority(st, e)
    if JOINPRE:
        # A part of a cut tile, JOINPRE blocks before its end: ask for the other part's word in the join buffer (agent
        # scope).  If that part is done already, the word is here when the loop ends and this part joins without its own
        # trip to memory (the swap); if not, it reads zero and swaps as before.  %[trig] = 0 for a piece that is no part.
        st.emit("s_cmp_eq_u32 %[cnt], %[trig]")
        st.emit(f"s_cbranch_scc0 Ldiag_nj{e}%=")
        st.emit("global_load_dwordx2 %[pre], %[vj], %[sj] sc1")
        st.emit(f"Ldiag_nj{e}%=:")
    # masks of the entering limb: symbol = plane 0 bit | plane 1 bit << 1
    st.emit(f"v_and_b32 {new[3]}, {PL0}, {PL1}")
    st.emit(f"v_bitop3_b32 {new[0]}, {PL0}, {PL1}, {PL1} bitop3:0x03")
    st.emit(f"v_xor_b32 {new[1]}, {PL0}, {new[3]}")
    st.emit(f"v_xor_b32 {new[2]}, {PL1}, {new[3]}")
    nt = TEXT[1 - e]
    # rows are read once per launch: marked non-temporal, so that they do not push the lines a wave starts with out of L2
    if EXP != "noload":
        st.emit(f"global_load_dword {PL0}, %[voff], %[sbase] offset:{row_offset}{NT}")
        st.emit(f"global_load_dword {PL1}, %[voff], %[sbase] offset:{row_offset + 256}{NT}")
        st.emit(f"global_load_dword {nt[0]}, %[voff], %[sbase] offset:{row_offset + 512}{NT}")
        st.emit(f"global_load_dword {nt[1]}, %[voff], %[sbase] offset:{row_offset + 768}{NT}")
    for k in range(4):
        st.emit(f"v_alignbit_b32 {SHIFTED[k]}, {new[k]}, {prev[k]}, 16")
    slot = 8 + 4 * (1 - e)
    st.ds(f"ds_write2st64_b32 {BASE}, {SHIFTED[0]}, {SHIFTED[1]} offset0:{slot} offset1:{slot + 1}")
    st.ds(f"ds_write2st64_b32 {BASE}, {SHIFTED[2]}, {SHIFTED[3]} offset0:{slot + 2} offset1:{slot + 3}")
    for h in range(2):
        for s in range(16 if ADD64 else 0):
            # both limbs on column g: U = V & mask, S = V + U as one 64-bit addition (the carry between the limbs is
            # inside it, the one out of the top is dropped), V' = S | (V ^ U)
            g = 16 * h + s
            if h == 1 and s == 15 - LOOKAHEAD:
                st.emit("s_waitcnt vmcnt(0)")
            read_column(st, e, g + LOOKAHEAD, ("n", g + LOOKAHEAD - 32) if g + LOOKAHEAD >= 32 else ("c", g + LOOKAHEAD))
            st.wait_read(("c", g))
            st.emit(f"v_and_b32 {U0}, {V0}, {ring(g)[1]}")
            st.emit(f"v_and_b32 {U1}, {V1}, {ring(g)[2]}")
            st.emit(f"v_lshl_add_u64 v[{S0[1:]}:{S1[1:]}], v[{V0[1:]}:{V1[1:]}], 0, v[{U0[1:]}:{U1[1:]}]")
            st.emit(f"v_bitop3_b32 {V0}, {U0}, {S0}, {V0} bitop3:0xde")
            st.emit(f"v_bitop3_b32 {V1}, {U1}, {S1}, {V1} bitop3:0xde")
        for s in range(0 if ADD64 else 17):
            g = 16 * h + s
            if s >= 1:
                st.emit(f"v_and_b32 {U1}, {V1}, {ring(g - 1)[2]}")
            if s < 16:
                if h == 1 and s == 15 - LOOKAHEAD and EXP != "nowaitvm":
                    st.emit("s_waitcnt vmcnt(0)")           # the next block's text is read from here on (the shift for its first pair)
                read_column(st, e, g + LOOKAHEAD, ("n", g + LOOKAHEAD - 32) if g + LOOKAHEAD >= 32 else ("c", g + LOOKAHEAD))
            if s < 16:
                st.wait_read(("c", g))
                st.emit(f"v_and_b32 {U0}, {V0}, {ring(g)[1]}")
            if s >= 1:
                st.emit(f"v_addc_co_u32_e64 {S1}, {CARRY_TOP}, {V1}, {U1}, {CARRY[(s - 1) % 2]}")
            if s < 16:
                st.emit(f"v_add_co_u32_e64 {S0}, {CARRY[s % 2]}, {V0}, {U0}")
            if s >= 1:
                st.emit(f"v_bitop3_b32 {V1}, {U1}, {S1}, {V1} bitop3:0xde")
            if s < 16:
                st.emit(f"v_bitop3_b32 {V0}, {U0}, {S0}, {V0} bitop3:0xde")
        window_down(st)
        if h == 0:
            slot = 4 * e
            st.ds(f"ds_write2st64_b32 {BASE}, {new[0]}, {new[1]} offset0:{slot} offset1:{slot + 1}")
            st.ds(f"ds_write2st64_b32 {BASE}, {new[2]}, {new[3]} offset0:{slot + 2} offset1:{slot + 3}")
    # what is in flight now belongs to the next block
    assert all(k[0] == "n" for k in st.read_at), st.read_at
    st.read_at = {("c", k[1]): v for k, v in st.read_at.items()}


def generate():
    st = Stream()
    if SDWA:
        st.emit(f"v_mov_b32 {THREE}, 3")
        st.emit(f"v_mov_b32 {ADDR[0]}, {BASE}")
        st.emit(f"v_mov_b32 {ADDR[1]}, {BASE}")
    for g in range(LOOKAHEAD):
        read_column(st, 0, g, ("c", g))
    entry = {k: st.ds_issued - v for k, v in st.read_at.items()}
    turn = st.addr_turn
    st.lines.append(".p2align 3")
    st.pos = 0
    st.emit("Ldiag_lds_loop%=:")
    block(st, 0, 1024)
    st.emit("s_sub_u32 %[cnt], %[cnt], 1")
    st.emit("s_cmp_eq_u32 %[cnt], 0")
    st.emit("s_cbranch_scc1 Ldiag_lds_done%=")
    block(st, 1, 2048)
    st.emit("v_add_u32 %[voff], 0x800, %[voff]")
    st.emit("s_sub_u32 %[cnt], %[cnt], 1")
    st.emit("s_cmp_eq_u32 %[cnt], 0")
    st.emit("s_cbranch_scc0 Ldiag_lds_loop%=")
    # the reads in flight at the end of either block were issued back to back, like the ones of the preamble
    assert {k: st.ds_issued - v for k, v in st.read_at.items()} == entry
    assert st.addr_turn == turn
    st.emit("Ldiag_lds_done%=:")
    if PRIO:
        st.emit("s_setprio 0")
    st.emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return st.lines


# ---- three limbs (T = 64): the same idea without half steps -------------------------------------------------------------
# Block i runs on the limbs w[i], w[i + 1], w[i + 2] and moves down one limb at its end.  LDS: four slots of four symbol
# rows, limb j in slot j % 4 -- the limb that enters block i + 1 (w[i + 3]: its planes were asked for at the start of
# block i) is written while block i still reads the other three, so the loop body is four blocks (the slots are offsets
# of the instructions).  A column: v_and_b32_sdwa (address) + ds_read2st64_b32 (limbs 0, 1) + ds_read_b32 (limb 2) for the
# column LOOKAHEAD3 ahead, then per limb v_and_b32, v_add(c)_co_u32, v_bitop3_b32: 10 VALU instructions where the
# plane-folding stream (TG_STEP in malcs_diag.hip) has 14.  Limb k lags k columns behind limb 0, so the carries cross
# between steps; a block is 34 steps.  The planes of the window's limbs follow it (PA, PB, PC) for whatever follows
# the stream.  The loop can be left after any block.
LOOKAHEAD3 = int(os.environ.get("GEN_LOOKAHEAD3", "5"))
RING3 = 8
V3 = ["v20", "v21", "v24"]
U3 = ["v22", "v26", "v55"]
S3 = ["v23", "v27", "v56"]
PA, PB, PC = ("v57", "v58"), ("v59", "v60"), ("v61", "v62")      # planes of the limbs w[i], w[i + 1], w[i + 2] of block i
NEW3 = ["v40", "v41", "v42", "v43"]
C01 = ["s[40:41]", "s[44:45]"]       # limb 0 -> limb 1
C12 = ["s[46:47]", "s[50:51]"]       # limb 1 -> limb 2
WAITCOL3 = 32 - LOOKAHEAD3 - 2       # from this step on the next block's rows are there (its text is needed LOOKAHEAD3 columns before its start)


def ring3(col):
    assert 32 % RING3 == 0 and RING3 >= LOOKAHEAD3 + 3
    k = col % RING3
    return f"v[{64 + 4 * k}:{65 + 4 * k}]", f"v{64 + 4 * k}", f"v{65 + 4 * k}", f"v{66 + 4 * k}"     # (register pairs start at even registers)


def read_column3(st, ph, G, key):
    """LDS reads of column G (0 ... 31: this block, of phase ph = i % 4; 32 ...: the next block)."""
    first = ph if G < 32 else ph + 1
    slots = [(first + k) % 4 for k in range(3)]
    a = ADDR[st.addr_turn]
    st.addr_turn ^= 1
    pair, _, _, third = ring3(G)
    text, k = text_of(ph % 2, G)
    if k == 0:
        for j in (1, 2, 3):
            st.emit(f"v_lshrrev_b32 {SHT[j - 1]}, {2 * j}, {text}")
    src = text if k % 4 == 0 else SHT[k % 4 - 1]
    st.emit(f"v_and_b32_sdwa {a}, {src}, {THREE} dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_{k // 4} src1_sel:DWORD")
    st.ds(f"ds_read2st64_b32 {pair}, {a} offset0:{4 * slots[0]} offset1:{4 * slots[1]}")
    st.read_at[key] = st.ds_issued            # (a column is waited for as a whole: the drain steps wait for nothing else)
    st.ds(f"ds_read_b32 {third}, {a} offset:{4 * slots[2] * 256}")


def block3(st, ph, row_offset):
    e = ph % 2
    if PRIO:
        set_priority(st, f"3{ph}")
    nt = TEXT[1 - e]
    # the rows of the next block: the planes of the limb that enters there and its text (2-bit symbols)
    st.emit(f"global_load_dword {PL0}, %[voff], %[sbase] offset:{row_offset}{NT}")
    st.emit(f"global_load_dword {PL1}, %[voff], %[sbase] offset:{row_offset + 256}{NT}")
    st.emit(f"global_load_dword {nt[0]}, %[voff], %[sbase] offset:{row_offset + 512}{NT}")
    st.emit(f"global_load_dword {nt[1]}, %[voff], %[sbase] offset:{row_offset + 768}{NT}")
    for s in range(34):
        if s == WAITCOL3:
            st.emit("s_waitcnt vmcnt(0)")
            st.emit(f"v_and_b32 {NEW3[3]}, {PL0}, {PL1}")
            st.emit(f"v_bitop3_b32 {NEW3[0]}, {PL0}, {PL1}, {PL1} bitop3:0x03")
            st.emit(f"v_xor_b32 {NEW3[1]}, {PL0}, {NEW3[3]}")
            st.emit(f"v_xor_b32 {NEW3[2]}, {PL1}, {NEW3[3]}")
            slot = 4 * ((ph + 3) % 4)
            st.ds(f"ds_write2st64_b32 {BASE}, {NEW3[0]}, {NEW3[1]} offset0:{slot} offset1:{slot + 1}")
            st.ds(f"ds_write2st64_b32 {BASE}, {NEW3[2]}, {NEW3[3]} offset0:{slot + 2} offset1:{slot + 3}")
        if s < 32 and s + LOOKAHEAD3 < 32 + LOOKAHEAD3:
            G = s + LOOKAHEAD3
            read_column3(st, ph, G, ("n", G - 32) if G >= 32 else ("c", G))
        if s < 32:
            st.wait_read(("c", s))
        if s >= 2:
            st.emit(f"v_and_b32 {U3[2]}, {V3[2]}, {ring3(s - 2)[3]}")
        if 1 <= s <= 32:
            st.emit(f"v_and_b32 {U3[1]}, {V3[1]}, {ring3(s - 1)[2]}")
        if s < 32:
            st.emit(f"v_and_b32 {U3[0]}, {V3[0]}, {ring3(s)[1]}")
        if s >= 2:
            st.emit(f"v_addc_co_u32_e64 {S3[2]}, {CARRY_TOP}, {V3[2]}, {U3[2]}, {C12[(s - 1) % 2]}")
        if 1 <= s <= 32:
            st.emit(f"v_addc_co_u32_e64 {S3[1]}, {C12[s % 2]}, {V3[1]}, {U3[1]}, {C01[(s - 1) % 2]}")
        if s < 32:
            st.emit(f"v_add_co_u32_e64 {S3[0]}, {C01[s % 2]}, {V3[0]}, {U3[0]}")
        if s >= 2:
            st.emit(f"v_bitop3_b32 {V3[2]}, {U3[2]}, {S3[2]}, {V3[2]} bitop3:0xde")
        if 1 <= s <= 32:
            st.emit(f"v_bitop3_b32 {V3[1]}, {U3[1]}, {S3[1]}, {V3[1]} bitop3:0xde")
        if s < 32:
            st.emit(f"v_bitop3_b32 {V3[0]}, {U3[0]}, {S3[0]}, {V3[0]} bitop3:0xde")
    # the window moves down one limb: the ones of the limb that leaves are counted, all ones enter on top; the planes
    # of the limbs that stay follow (the next block's loads overwrite PL)
    st.emit(f"v_bcnt_u32_b32 %[bank], {V3[0]}, %[bank]")
    st.emit(f"v_mov_b32 {V3[0]}, {V3[1]}")
    st.emit(f"v_mov_b32 {V3[1]}, {V3[2]}")
    st.emit(f"v_mov_b32 {V3[2]}, -1")
    st.emit(f"v_mov_b32 {PA[0]}, {PB[0]}")
    st.emit(f"v_mov_b32 {PA[1]}, {PB[1]}")
    st.emit(f"v_mov_b32 {PB[0]}, {PC[0]}")
    st.emit(f"v_mov_b32 {PB[1]}, {PC[1]}")
    st.emit(f"v_mov_b32 {PC[0]}, {PL0}")
    st.emit(f"v_mov_b32 {PC[1]}, {PL1}")
    assert all(k[0] == "n" for k in st.read_at), st.read_at
    st.read_at = {("c", k[1]): v for k, v in st.read_at.items()}


def generate3():
    st = Stream()
    st.emit(f"v_mov_b32 {THREE}, 3")
    st.emit(f"v_mov_b32 {ADDR[0]}, {BASE}")
    st.emit(f"v_mov_b32 {ADDR[1]}, {BASE}")
    for g in range(LOOKAHEAD3):
        read_column3(st, 0, g, ("c", g))
    entry = {k: st.ds_issued - v for k, v in st.read_at.items()}
    turn = st.addr_turn
    st.lines.append(".p2align 3")
    st.pos = 0
    st.emit("Ldiag_lds3_loop%=:")
    for ph in range(4):
        block3(st, ph, 1024 if ph % 2 == 0 else 2048)
        if ph % 2 == 1:
            st.emit("v_add_u32 %[voff], 0x800, %[voff]")
        st.emit("s_sub_u32 %[cnt], %[cnt], 1")
        st.emit("s_cmp_eq_u32 %[cnt], 0")
        st.emit("s_cbranch_scc1 Ldiag_lds3_done%=" if ph < 3 else "s_cbranch_scc0 Ldiag_lds3_loop%=")
        assert {k: st.ds_issued - v for k, v in st.read_at.items()} == entry, (ph, st.read_at)
    assert st.addr_turn == turn
    st.emit("Ldiag_lds3_done%=:")
    if PRIO:
        st.emit("s_setprio 0")
    st.emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return st.lines


def write_macro(f, name, lines):
    f.write(f"#define {name} \\\n")
    for i, line in enumerate(lines):
        end = "\\n\\t" if i + 1 < len(lines) else ""
        cont = " \\" if i + 1 < len(lines) else ""
        f.write(f'    "{line}{end}"{cont}\n')


def main():
    lines = generate()
    out = os.environ.get("GEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "algebra_amd", "csrc", "malcs_diag_lds.inc")
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_diag_lds.py -- do not edit.  The two-limb window stream with match masks from LDS\n")
        f.write(f"// (lookahead {LOOKAHEAD} columns, ring of {RING} register pairs); registers and operands: see diag_halves_lds.\n")
        f.write(f'#define DIAG_LDS_V0 "{V0}"\n#define DIAG_LDS_V1 "{V1}"\n#define DIAG_LDS_JOINPRE {JOINPRE}\n')
        write_macro(f, "DIAG_LDS_STREAM", lines)
        lines3 = generate3()
        f.write(f"// The three-limb window stream with match masks from LDS (lookahead {LOOKAHEAD3} columns, ring of {RING3} register triples);\n")
        f.write("// registers and operands: see diag_thirds_lds.\n")
        write_macro(f, "DIAG_LDS3_STREAM", lines3)
        n3 = sum(1 for l in lines3 if l.startswith("v_"))
        print(f"three limbs: {len(lines3)} lines, {n3} VALU instructions for four blocks + preamble -> {n3 / 128:.2f} per column")
    n_valu = sum(1 for l in lines if l.startswith("v_"))
    print(f"{len(lines)} lines, {n_valu} VALU instructions for two blocks + preamble -> {n_valu / 64:.2f} per column, "
          f"{sum(1 for l in lines if l == 's_nop 0')} s_nop")


if __name__ == "__main__":
    main()
