"""Slot keys for convt_whole's x tile ([512 voxels][256 B], 16 slots per row) under 16x16x32 fragment reads."""
import itertools
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
def cost(keyf):
    tot = 0; worst = 0
    for dh in (-1, 0, 1):
        for dw in (-1, 0, 1):
            for ct in range(4):
                for k32 in range(4):
                    for g in GROUPS:
                        banks = {}
                        for l in g:
                            r, q = l & 15, l >> 4
                            zh, zw = 2 * ct + (r >> 3) + dh, (r & 7) + dw
                            pos = ((4 * k32 + q) ^ keyf(zh, zw)) & 15
                            banks[pos] = banks.get(pos, 0) + 1
                        c = max(banks.values()); worst = max(worst, c); tot += c - 1
    return worst, tot
print('current (8zh+zw)&15:', cost(lambda zh, zw: (8 * zh + zw) & 15))
res = []
for a, b in itertools.product(range(16), repeat=2):
    for sh in (0, 1, 2):
        res.append((cost(lambda zh, zw, a=a, b=b, sh=sh: ((a * zw + b * zh) >> sh) & 15), a, b, sh))
res.sort()
print(res[:8])
