"""LDS bank model of final_bce_sweepw_kernel's three access kinds (CPU only; lane groups and bank functions from
/opt/skills/guides/MI355X_MICROARCH.md, LDS table) and a search over slot keys.
  A reads : ds_read_b128 of the staged plane, 16x16x32 fragment shape: lane (c16, kq), rows centre / left / right of tile T
  publish : ds_write_b128 of Q rows (8 x 8 contiguous lanes, banks mod 32)
  gather  : ds_read_b64 of Q granules (2 x 32 lanes, banks mod 64)"""
import itertools
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 += [[l + 32 for l in g] for g in G128]
G64 = [list(range(32)), list(range(32, 64))]
GW128 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(groups, addr_of, unit, nbanks):
    """sum over groups of the worst number of distinct addresses per bank-unit (unit bytes wide, nbanks units)"""
    tot = mx = 0
    for g in groups:
        b = {}
        for l in g:
            a = addr_of(l)
            if a is None:
                continue
            b.setdefault((a // unit) % nbanks, set()).add(a)
        c = max((len(v) for v in b.values()), default=0)
        tot += c
        mx = max(mx, c)
    return tot, mx


def a_reads(xkey):
    tot = mx = n = 0
    for wv in range(4):
        for t in range(2 if wv == 0 else 1):
            for shift in (0, -1, 1):
                for hf in range(2):
                    def addr(l):
                        c16, kq = l & 15, l >> 4
                        row = (2 * (wv + 4 * t) + (c16 >> 3)) * 10 + 1 + (c16 & 7) + shift
                        return row * 128 + (((hf * 4 + kq) ^ xkey(row)) & 7) * 16
                    c, m = cycles(G128, addr, 16, 16)
                    tot += c; mx = max(mx, m); n += 4
    return tot, n, mx


def publish(qkey):
    tot = mx = n = 0
    for wv in range(4):
        for t in range(2 if wv == 0 else 1):
            for half in range(2):
                def addr(l):
                    c16, kq = l & 15, l >> 4
                    pr = (wv + 4 * t) * 16 + c16
                    return pr * 128 + (((4 * kq + 2 * half) ^ qkey(pr)) << 3)
                c, m = cycles(GW128, addr, 16, 8)
                tot += c; mx = max(mx, m); n += 8
    return tot, n, mx


def gather(qkey):
    tot = mx = n = 0
    for wave in range(4):
        for ah in range(2):
            for tdsel in range(2):
                def addr(l):
                    tid = wave * 64 + l
                    mw, ohh, sl = tid & 7, (tid >> 3) & 15, tid >> 7
                    mh, ph = ohh >> 1, ohh & 1
                    zh, th = mh + ph - ah + 1, 1 - ph + 2 * ah
                    pr = zh * 8 + mw
                    g = (2 * tdsel + sl) * 4 + th
                    return pr * 128 + ((g ^ qkey(pr)) << 3)
                c, m = cycles(G64, addr, 8, 32)
                tot += c; mx = max(mx, m); n += 2
    return tot, n, mx


if __name__ == '__main__':
    xk0 = lambda row: (row >> 1) & 7
    qk0 = lambda pr: (((pr & 7) >> 1) << 1) | (((pr >> 3) & 1) << 3)
    print('tree: A reads %s  publish %s  gather %s   (LDS cycles, conflict-free cycles, worst way)' % (a_reads(xk0), publish(qk0), gather(qk0)))
    # X tile keys: (a * zh + b * zw + c * (row >> 1)) & 7 and the like, applied as slot ^ key
    best = []
    for a, b, sh in itertools.product(range(8), range(8), (0, 1, 2)):
        f = lambda row, a=a, b=b, sh=sh: ((a * (row // 10) + b * (row % 10)) >> sh) & 7
        best.append((a_reads(f), 'x: ((%d*zh + %d*zw) >> %d) & 7' % (a, b, sh)))
    for a, sh in itertools.product(range(1, 8), (0, 1, 2)):
        f = lambda row, a=a, sh=sh: ((a * row) >> sh) & 7
        best.append((a_reads(f), 'x: ((%d*row) >> %d) & 7' % (a, sh)))
    best.sort()
    for r in best[:6]:
        print(r)
    # Q keys: 4-bit granule keys with bit 0 clear (the two pw stay adjacent), from zh, zw-1
    bq = []
    for a, b, c, d in itertools.product(range(8), range(8), range(8), range(8)):
        f = lambda pr, a=a, b=b, c=c, d=d: ((((a * (pr & 7) + b * (pr >> 3)) & 7) ^ ((c * (pr & 7) + d * (pr >> 3)) >> 1 & 7)) << 1) & 14
        g, p = gather(f), publish(f)
        bq.append((g[0] + p[0], g, p, 'q: (((%d*c + %d*r) & 7) ^ (((%d*c + %d*r) >> 1) & 7)) << 1' % (a, b, c, d)))
    bq.sort()
    for r in bq[:6]:
        print(r)
