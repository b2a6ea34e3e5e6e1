"""Bank-conflict check of the padded LDS images of csrc/bconv.hip: enumerates every ds_read_b128 an MFMA fragment read
issues (all taps, all k-steps, all wave positions) and counts, per 16-lane service group of MI355X_MICROARCH.md (LDS
table), how many lanes share a 16-byte bank group (address / 16 mod 16).  A conflict-free layout prints max ways = 1.
Pure host arithmetic - mirrors BConvCfg in bconv.hip; run it after changing a stride there."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def cfg(C, TW, TH, x3):
    HW = TW + 2
    PS = 2 * C + 16
    raw = HW * PS
    RP = ((raw + 255) // 256) * 256 if TW == 16 else ((raw - 64 + 255) // 256) * 256 + 64
    KC = 32 if (x3 and C == 128 and TW == 16) else 64
    return HW, PS, RP, KC, KC * 2 + 16


def ways(addrs):
    worst = 1
    for g in GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def main():
    for (C, TW, TH, x3) in [(128, 16, 16, 0), (64, 16, 32, 0), (128, 4, 16, 0), (64, 4, 16, 0), (128, 16, 8, 1), (64, 16, 16, 1)]:
        HW, PS, RP, KC, BROW = cfg(C, TW, TH, x3)
        wa = 1
        for blk in range(TH * TW // 32):
            for dr, dc in itertools.product(range(3), range(3)):
                for c0 in range(0, C, 16):
                    addrs = []
                    for lane in range(64):
                        n, kh = lane & 31, lane >> 5
                        q = blk * 32 + n
                        pr, pc = q // TW, q % TW
                        addrs.append((pr + dr) * RP + (pc + dc) * PS + 2 * c0 + 16 * kh)
                    wa = max(wa, ways(addrs))
        wb = 1
        for nb in range(C // 32):
            for ks in range(KC // 16):
                addrs = [(nb * 32 + (l & 31)) * BROW + 32 * ks + 16 * (l >> 5) for l in range(64)]
                wb = max(wb, ways(addrs))
        print(f"C={C:3d} W={TW:2d} TH={TH:2d} x3={x3}: pixel stride {PS} B, row pitch {RP} B, chunk row {BROW} B -> "
              f"A fragment reads max {wa}-way, B fragment reads max {wb}-way")
        assert wa == 1 and wb == 1


if __name__ == "__main__":
    main()
