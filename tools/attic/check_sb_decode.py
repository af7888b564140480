"""Host check of k_solve_sb's arithmetic gather decode against an independent table construction (the table version was validated on the GPU)."""
NPOSE, ND, NR, NCH = 66, 75, 76, 10
OFF_P = 0; OFF_D = NR * (NR + 1) // 2; OFF_E = OFF_D + NCH * 81; OFF_BAND = OFF_E + (NCH - 1) * 81
BSTR = lambda a: 28 if a == 1 else 20
BOFF = lambda a: 0 if a == 1 else 252 + 180 * (a - 2)
prow = lambda r: r * (r + 1) // 2
ref = {}
def put(dst, r, c, srcs):
    assert dst not in ref; ref[dst] = (r, c, tuple(sorted(x for x in srcs if x is not None)))
for r in range(NPOSE):
    for c in range(r + 1):
        A, l1, B, l2 = r // 6, r % 6, c // 6, c % 6
        src = [("vis", 36 * (A * (A + 1) // 2 + B) + 6 * l1 + l2)]
        if A == B:
            if A >= 1: src += [("imu", 900 * (A - 1) + 30 * (15 + l1) + 15 + l2), ("lid", 144 * (A - 1) + 12 * (6 + l1) + 6 + l2)]
            if A <= 9: src += [("imu", 900 * A + 30 * l1 + l2), ("lid", 144 * A + 12 * l1 + l2)]
        elif A == B + 1: src += [("imu", 900 * B + 30 * (15 + l1) + l2), ("lid", 144 * B + 12 * (6 + l1) + l2)]
        put(OFF_P + prow(r) + c, r, c, src)
for i in range(9):
    for c in range(NPOSE + i + 1):
        r = NPOSE + i; s = None
        if c < NPOSE:
            B, l2 = c // 6, c % 6
            if B <= 1: s = ("imu", 30 * (6 + i) + (l2 if B == 0 else 15 + l2))
        else: s = ("imu", 30 * (6 + i) + 6 + (c - NPOSE))
        put(OFF_P + prow(r) + c, r, c, [s])
for a in range(1, NCH + 1):
    ra = NPOSE + 9 * a
    for i in range(9):
        for j in range(i + 1): put(OFF_D + 81 * (a - 1) + 9 * i + j, ra + i, ra + j, [("imu", 900 * (a - 1) + 30 * (21 + i) + 21 + j), ("imu", 900 * a + 30 * (6 + i) + 6 + j) if a <= 9 else None])
    if a <= 9:
        for i in range(9):
            for j in range(9): put(OFF_E + 81 * (a - 1) + 9 * i + j, ra + 9 + i, ra + j, [("imu", 900 * a + 30 * (21 + i) + 6 + j)])
    for i in range(9):
        dst = OFF_BAND + BOFF(a) + i * BSTR(a)
        for m in range(6):
            put(dst + m, ra + i, 6 * (a - 1) + m, [("imu", 900 * (a - 1) + 30 * (21 + i) + m)])
            put(dst + 6 + m, ra + i, 6 * a + m, [("imu", 900 * (a - 1) + 30 * (21 + i) + 15 + m), ("imu", 900 * a + 30 * (6 + i) + m) if a <= 9 else None])
            if a <= 9: put(dst + 12 + m, ra + i, 6 * (a + 1) + m, [("imu", 900 * a + 30 * (6 + i) + 15 + m)])
        if a == 1:
            for j in range(9): put(dst + 18 + j, ra + i, NPOSE + j, [("imu", 30 * (21 + i) + 6 + j)])

got = {}
def gput(dst, r, c, srcs):
    assert dst not in got, dst; got[dst] = (r, c, tuple(sorted(x for x in srcs if x is not None)))
import math
# --- kernel decode, transcribed ---
for ln in range(64):
    for u in range(13):
        if ln + 64 * u >= 810: continue
        q = ln + 64 * u; a1 = q // 81; rem = q - 81 * a1; i = rem // 9; j = rem - 9 * i
        if j > i: continue
        gput(OFF_D + q, NPOSE + 9 + 9 * a1 + i, NPOSE + 9 + 9 * a1 + j, [("imu", 900 * a1 + 30 * (21 + i) + 21 + j), ("imu", 900 * (a1 + 1) + 30 * (6 + i) + 6 + j) if a1 + 1 <= 9 else None])
    for u in range(12):
        if ln + 64 * u >= 729: continue
        q = ln + 64 * u; a1 = q // 81; rem = q - 81 * a1; i = rem // 9; j = rem - 9 * i
        gput(OFF_E + q, NPOSE + 18 + 9 * a1 + i, NPOSE + 9 + 9 * a1 + j, [("imu", 900 * (a1 + 1) + 30 * (21 + i) + 6 + j)])
for td in range(192):
    for u in range(4):
        if td + 192 * u >= 756: continue
        q = td + 192 * u; nb = q // 36; e = q - 36 * nb; l1 = e // 6; l2 = e - 6 * l1
        dg = nb < 11
        A = nb if dg else nb - 10; B = nb if dg else nb - 11; r = 6 * A + l1; c = 6 * B + l2
        i0 = 900 * (A - 1) + 30 * (15 + l1) + 15 + l2 if dg else 900 * B + 30 * (15 + l1) + l2; i1 = 900 * A + 30 * l1 + l2
        j0 = 144 * (A - 1) + 12 * (6 + l1) + 6 + l2 if dg else 144 * B + 12 * (6 + l1) + l2; j1 = 144 * A + 12 * l1 + l2
        h0 = (not dg) or A >= 1; h1 = dg and A <= 9
        if c > r: continue
        gput(OFF_P + prow(r) + c, r, c, [("vis", 36 * (A * (A + 1) // 2 + B) + e), ("imu", i0) if h0 else None, ("imu", i1) if h1 else None, ("lid", j0) if h0 else None, ("lid", j1) if h1 else None])
    for u in range(9):
        if td + 192 * u >= 1620: continue
        q = td + 192 * u; fb = q // 36; e = q - 36 * fb; l1 = e // 6; l2 = e - 6 * l1
        A2 = int((math.sqrt(8.0 * fb + 1.0) - 1.0) * 0.5)
        B = fb - A2 * (A2 + 1) // 2; A = A2 + 2; r = 6 * A + l1; c = 6 * B + l2
        gput(OFF_P + prow(r) + c, r, c, [("vis", 36 * (A * (A + 1) // 2 + B) + e)])
    for u in range(4):
        if td + 192 * u >= 675: continue
        q = td + 192 * u; i = q // 75; c = q - 75 * i; r = NPOSE + i; B = c // 6; l2 = c - 6 * B
        src = 30 * (6 + i) + (l2 if B == 0 else 15 + l2) if c < NPOSE else 30 * (6 + i) + 6 + (c - NPOSE)
        hs = c >= NPOSE or B <= 1
        if c > r: continue
        gput(OFF_P + prow(r) + c, r, c, [("imu", src) if hs else None])
    for u in range(13):
        if td + 192 * u >= 2430: continue
        q = td + 192 * u; a1 = q // 243; rem = q - 243 * a1; i = rem // 27; pos = rem - 27 * i; a = a1 + 1
        seg = pos // 6; m = pos - 6 * seg
        on = seg < 2 or (seg == 2 and a <= 9) or (seg >= 3 and a == 1)
        if not on: continue
        s1 = None
        if seg == 0: s0 = 900 * a1 + 30 * (21 + i) + m; c165 = 6 * a1 + m
        elif seg == 1: s0 = 900 * a1 + 30 * (21 + i) + 15 + m; s1 = 900 * a + 30 * (6 + i) + m if a <= 9 else None; c165 = 6 * a + m
        elif seg == 2: s0 = 900 * min(a, 9) + 30 * (6 + i) + 15 + m; c165 = 6 * (a + 1) + m
        else: s0 = 30 * (21 + i) + 6 + (pos - 18); c165 = NPOSE + (pos - 18)
        gput(OFF_BAND + BOFF(a) + i * BSTR(a) + pos, NPOSE + 9 * a + i, c165, [("imu", s0), ("imu", s1) if s1 is not None else None])
print(len(ref), len(got))
bad = [k for k in ref if got.get(k) != ref[k]]
print("missing/different:", len(bad), bad[:5], [(ref[k], got.get(k)) for k in bad[:3]])
print("extra:", [k for k in got if k not in ref][:5])
