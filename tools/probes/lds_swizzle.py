"""Brute-force check of the LDS swizzles of the decoder's conv kernels against the gfx950 ds_read_b128 lane groups
(MI355X_MICROARCH.md, LDS table): rows of 64 B = 4 chunks of 16 B, lane (r16, q) reads chunk q of pixel P0 + r16 at halo
column tx + r16.  A swizzle is conflict-free when the 16 lanes of every group touch 16 distinct 16-byte slots of the 256-byte
bank line, for every alignment P0 and column shift tx.  CPU only:  python tools/probes/lds_swizzle.py
(tests/test_lds_swizzle.py holds the kernels' two swizzles to zero conflicts.)"""
_G0 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = _G0 + [[l + 32 for l in g] for g in _G0]


def conflicts(sw):
    """Number of (alignment, column shift, lane group) cases in which two lanes of a group hit the same 16-byte slot."""
    bad = 0
    for P0 in range(8):
        for tx in range(4):
            for g in GROUPS:
                slots = {((P0 + (l & 15)) % 4) * 4 + ((l >> 4) ^ sw(P0 + (l & 15), tx + (l & 15))) for l in g}
                bad += len(slots) != 16
    return bad


def pixel_swizzle(P, x):  # weight rows (sB) and the z tile: chunk ^ ((row >> 1) & 3)
    return (P >> 1) & 3


def column_swizzle(P, x):  # halo tiles (sA): chunk ^ ((halo column >> 1) & 3), so that fragment rows differ by a constant
    return (x >> 1) & 3


if __name__ == "__main__":
    print("no swizzle                :", conflicts(lambda P, x: 0), "conflicting (alignment, shift, group) cases")
    print("(pixel >> 1) & 3          :", conflicts(pixel_swizzle))
    print("(halo column >> 1) & 3    :", conflicts(column_swizzle))
