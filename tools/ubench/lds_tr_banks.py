#!/usr/bin/env python3
"""Bank-conflict check of conv_wgrad.hip's LDS image for ds_read_b64_tr_b16 (MI355X_MICROARCH.md §LDS: 64 banks of
4 bytes, bank = (addr/4) % 64, conflicts counted per 32-lane half; each lane reads 8 bytes = 2 banks).
Image: [pixel row r][8 chunks of 16 B], slot s of row r holds source chunk s ^ key(r)."""


def key(r):
    return (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2)


def addr(lane, cb, ks, hh):
    g, q, pp = lane >> 4, (lane & 15) >> 2, lane & 3
    r = ks * 32 + g * 8 + hh * 4 + q
    c = 2 * cb + (pp >> 1)
    return r * 128 + ((c ^ key(r)) << 4) + 8 * (pp & 1)


worst = 0
for cb in range(4):
    for ks in range(2):
        for hh in range(2):
            for half in range(2):
                banks = {}
                for lane in range(half * 32, half * 32 + 32):
                    a = addr(lane, cb, ks, hh)
                    for b in ((a // 4) % 64, (a // 4 + 1) % 64):
                        banks.setdefault(b, set()).add(a)
                worst = max(worst, max(len(v) for v in banks.values()))
print("worst distinct addresses per bank within a 32-lane half:", worst, "(1 = conflict-free)")
assert worst == 1
