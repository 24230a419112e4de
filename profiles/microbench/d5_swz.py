"""Exhaustive search for the LDS slot key of the last layer's P rows (CPU only).

The sweep kernel publishes P[halo cell][32 taps] rows (8 quads of 16 B) and every output thread gathers whole quads of three
neighbouring cells with ds_read_b128.  A wave's ds_read_b128 is serviced in four fixed groups of 16 lanes
(/opt/skills/guides/MI355X_MICROARCH.md, LDS table); a group is conflict-free when its 16 quads fall into 16 different
4-bank groups.  Searched: row pitch 32 / 36 / 40 floats x slot keys (a*zh + b*zw) >> s applied as quad ^ key.
Result: pitch 32 (dense rows) with key zw & 7 is conflict-free for every (wave, ah, cell offset); the linear pitch 36 of
rounds 1-2 is 2-way for quads (and was 4-way for its dword gathers)."""
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
groups += [[l + 32 for l in g] for g in groups]


def worst(PP, f):
    tot = mx = 0
    for wave in range(4):
        for ah in range(2):
            for co in range(3):
                for g in groups:
                    banks = {}
                    for lane in g:
                        tid = wave * 64 + lane
                        mw, ohh, sl = tid & 7, (tid >> 3) & 15, tid >> 7
                        mh, ph = ohh >> 1, ohh & 1
                        zh, th = mh + ph - ah + 1, 1 - ph + 2 * ah
                        zw = mw + co
                        addr = (zh * 10 + zw) * PP * 4 + (((sl * 4 + th) ^ f(zh, zw)) & 7) * 16
                        banks.setdefault((addr // 16) % 16, set()).add(addr)
                    c = max(len(v) for v in banks.values())
                    tot += c
                    mx = max(mx, c)
    return mx, tot


if __name__ == '__main__':
    res = []
    for PP in (32, 36, 40):
        res.append((worst(PP, lambda zh, zw: 0), PP, 'no key'))
        for a in range(8):
            for b in range(8):
                for sh in (0, 1):
                    res.append((worst(PP, lambda zh, zw, a=a, b=b, sh=sh: (a * zh + b * zw) >> sh), PP, '(%d*zh + %d*zw) >> %d' % (a, b, sh)))
    res.sort()
    for r in res[:8]:
        print('max %d-way, %3d group-cycles: pitch %d, key %s' % (r[0][0], r[0][1], r[1], r[2]))
    print('...')
    for r in res:
        if r[2] == 'no key':
            print('max %d-way, %3d group-cycles: pitch %d, %s' % (r[0][0], r[0][1], r[1], r[2]))
