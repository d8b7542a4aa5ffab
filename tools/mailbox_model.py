"""Exhaustive interleaving check (one pixel, frames 1..N, sequentially consistent atomics) of the mailbox hand-over that was designed for
small shares in round 4 and NOT adopted (docs/kernels.md, CHANGELOG.md): every decision between a waiting result and the resolve it waits
for is one read-modify-write on a 64-bit word per pixel, S = R << 32 | M (R = frame + 1 of the newest resolve that reported, M = frame + 1
of the result waiting in the mailbox).  claim(F): compare-and-swap (R = F - 1, M = 0) -> (R = F - 1, M = F + 1), then store the data in
slot F & 1.  report(F): fetch-max with (F + 1) << 32 — records the resolve AND clears M in one step; an M = F + 2 in the old value is the
next frame's result: read its slot until the data carries that frame, fold it, report again.  Checked: every frame folded exactly once,
in order, from every reachable state.  (Two earlier variants fail: with M in the high bits a report made while a mail is waiting is
swallowed by the fetch-max, and a result can later be deposited for a resolver that has already passed.)
    python tools/mailbox_model.py N R0     (R0 = 1: frame 0 has reported, 0: the state words were zeroed)"""
import sys
N = int(sys.argv[1]); BATCH = N + 1; R0 = int(sys.argv[2])  # R = newest reported frame + 1 (0 = none).  frame 0 resolved; R0 = 1 if it reported, 0 if state was zeroed
def step(s, t):
    pix, folded, R, M, d0, d1, pcs, curs = s
    pcs = list(pcs); curs = list(curs); j = t + 1; pc = pcs[t]; data = [d0, d1]
    if pc == 0: pcs[t] = 10 if pix == j - 1 else 1
    elif pc == 1:                       # claim(F=j): S == (R=j-1, M=0) -> M=j+1
        if M == 0 and R == j - 1: M = j + 1; pcs[t] = 2
        else: pcs[t] = 0
    elif pc == 2: data[j & 1] = j; pcs[t] = 99
    elif pc == 10:
        if pix != j - 1: return ('ERR', 'fold on changed pixel', s, t)
        pix = j; folded += (j,); curs[t] = j; pcs[t] = 11
    elif pc == 11:                      # report(F=cur): fetch_max((F+1)<<32)
        F = curs[t]
        if R < F + 1:
            took = M; R = F + 1; M = 0
            if took != 0 and took != F + 2: return ('ERR', 'took foreign mail', s, t)
            pcs[t] = 12 if took == F + 2 else 99
        else: pcs[t] = 99
    elif pc == 12:
        k = curs[t] + 1
        if data[k & 1] == k: pcs[t] = 13
    elif pc == 13:
        k = curs[t] + 1
        if pix != k - 1: return ('ERR', 'relay on wrong pixel', s, t)
        pix = k; folded += (k,); curs[t] = k; pcs[t] = 11
    return (pix, folded, R, M, data[0], data[1], tuple(pcs), tuple(curs))
init = (0, (), R0, 0, 0, 0, tuple([0] * N), tuple([0] * N))
seen = set(); stack = [init]; bad = []
while stack:
    s = stack.pop()
    if s in seen: continue
    seen.add(s)
    for t in range(N):
        if s[6][t] != 99:
            n = step(s, t)
            if n[0] == 'ERR': bad.append(n)
            elif n != s: stack.append(n)
good = {s for s in seen if all(p == 99 for p in s[6]) and s[1] == tuple(range(1, N + 1))}
fin_bad = [s for s in seen if all(p == 99 for p in s[6]) and s not in good]
pred = {}
for s in seen:
    for t in range(N):
        if s[6][t] != 99:
            n = step(s, t)
            if n[0] != 'ERR' and n != s: pred.setdefault(n, []).append(s)
reach = set(good); front = list(good)
while front:
    x = front.pop()
    for p in pred.get(x, []):
        if p not in reach: reach.add(p); front.append(p)
print('N', N, 'R0', R0, 'states', len(seen), 'errors', len(bad), 'finished wrong', len(fin_bad), 'cannot finish', len([s for s in seen if s not in reach]))
for b in (bad + fin_bad)[:3]: print(b)
