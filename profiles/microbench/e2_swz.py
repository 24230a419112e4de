"""Brute-force search of a slot key for conv_direct's phase tile that makes 16x16x32 fragment reads conflict-free."""
import itertools
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

def conflicts(keyf):
    worst, total = 0, 0
    for zd in range(5):
        for ah in range(2):
            for aw in range(2):
                for ct in range(4):
                    for s32 in range(2):
                        for g in GROUPS:
                            banks = {}
                            for l in g:
                                r, q = l & 15, l >> 4
                                jh, jw = 2 * ct + (r >> 3) + ah, (r & 7) + aw
                                rl = zd * 81 + jh * 9 + jw
                                pos = ((rl & 1) << 3) | (((4 * s32 + q) ^ keyf(zd, jh, jw)) & 7)
                                banks.setdefault(pos, set()).add(rl * 8 + ((4 * s32 + q) ^ keyf(zd, jh, jw)))
                            c = max(len(v) for v in banks.values())
                            worst = max(worst, c); total += c - 1
    return worst, total

print('current jw&7:', conflicts(lambda zd, jh, jw: jw & 7))
best = []
for a, b, c in itertools.product(range(8), repeat=3):
    for sh in (0, 1):
        f = lambda zd, jh, jw, a=a, b=b, c=c, sh=sh: ((a * jw + b * jh + c * zd) >> sh) & 7
        w, t = conflicts(f)
        best.append((t, w, a, b, c, sh))
best.sort()
print(best[:10])
