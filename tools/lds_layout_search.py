#!/usr/bin/env python
"""Enumerates LDS images of the row GEMM's operand chunk (ao_amd/csrc/gemm.hip) and counts their bank conflicts with the
gfx950 rules of guides/MI355X_MICROARCH.md (section LDS):

  ds_read_b128   four NON-contiguous 16-lane groups, bank = (addr / 4) % 64
  ds_write_b128  eight contiguous 8-lane groups,     bank = (addr / 4) % 32
  ds_write_b32   two 32-lane groups,                 bank = (addr / 4) % 32

Tile: rows x KC floats.  Operand read: lane (i = lane % 16, s = lane / 16) reads the float4 slots s * KC/16 + j of row i.
Staging store: thread t stores float4 slot t % (KC/4) of row t / (KC/4) (coalesced global order); the (k,n)-major weight
tile is stored transposed with scalar stores (thread q: k = q % KC, columns 4 (q / KC) .. + 3).
Candidates: row pitch KC + pad, slot index XORed with (row >> shift) & mask.  Prints the conflict-free ones (extra LDS cycles
per wave-instruction summed over the instructions of one chunk); `python tools/lds_layout_search.py` -> the image used:
pitch KC + 8, slot ^= row & 1 (KC = 32) / row & 3 (KC = 64); the rounds-1/2 image (pitch KC + 4, no swizzle) scores 8 / 16."""
RG = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
RG += [[l + 32 for l in g] for g in RG]


def extra_cycles(groups, addr, slots, width):
    tot = 0
    for g in groups:
        cnt = {}
        for l in g:
            a = addr(l)
            cnt.setdefault((a // width) % slots, set()).add(a)
        tot += max(len(v) for v in cnt.values()) - 1
    return tot


def score(KC, pitch, shift, mask):
    KQ = KC // 4

    def phys(r, k4):
        return r * pitch + 4 * (k4 ^ ((r >> shift) & mask))

    if any((k4 ^ m) >= KQ for k4 in range(KQ) for m in range(mask + 1)):
        return None
    rd = sum(extra_cycles(RG, lambda l: phys(l % 16, (l // 16) * (KQ // 4) + j), 16, 4) for j in range(KQ // 4))
    w128 = sum(extra_cycles([list(range(g, g + 8)) for g in range(0, 64, 8)],
                            lambda l: phys((w * 64 + l) // KQ, (w * 64 + l) % KQ), 8, 4) for w in range(4))
    w32 = 0
    for w in range(4):
        for e in range(4):
            def a(l):
                q = w * 64 + l
                kk, cq = q % KC, (q // KC) * 4
                return phys(cq + e, kk >> 2) + (kk & 3)
            w32 += extra_cycles([list(range(0, 32)), list(range(32, 64))], a, 32, 1)
    return rd, w128, w32


if __name__ == "__main__":
    for KC in (32, 64):
        print("KC = %d   (reads, b128 stores, transposing b32 stores) extra cycles per chunk" % KC)
        print("   rounds 1-2: pitch %d, no swizzle -> %s" % (KC + 4, score(KC, KC + 4, 0, 0)))
        for pad in (0, 4, 8, 12, 16):
            for shift in (0, 1, 2, 3):
                for mask in (0, 1, 3, 7, 15):
                    if mask == 0 and shift:
                        continue
                    sc = score(KC, KC + pad, shift, mask)
                    if sc is not None and sum(sc) == 0:
                        print("   conflict-free: pitch %d, slot ^= (row >> %d) & %d" % (KC + pad, shift, mask))
