"""CPU model of one superstep of the speculative segmented inverse (64 lanes x K pixels, plain rule for kScanAfter passes, then the
long-correction scan): which lanes are still wrong after each pass, on a frame of a given content.  For looking at the
supersteps that need dozens of passes.  python tools/exp/inv_chain_model.py [sines|plasma] [optimist 0/1]"""
import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
from oracle.oracle import Oracle
XS, YS, K = 1280, 800, 20
kind = sys.argv[1] if len(sys.argv) > 1 else "sines"
MEMO_FROM = int(os.environ.get('MEMO_FROM', '4'))   # the memo is kept from this pass on (4: the long-correction path only)
optimist = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # 1: lanes without evidence are shifts, 0: constants (dynamic evidence either way), -1: no scan (the plain rule), -2: static shifts only (round 3)
if kind == "sines":
    yy, xx = np.mgrid[0:YS, 0:XS].astype(np.float32)
    def sines(k): return 30000 + 9000 * np.sin(xx / (310.0 + 7 * k) + 0.3 * k) + 7000 * np.sin(yy / (270.0 - 5 * k) + 0.2 * k) + 5000 * np.sin((xx + yy) / (530.0 + 11 * k))
    base = np.stack([np.clip(sines(k), 0, 65535) for k in range(2)]).astype(np.uint16); shift = 0
elif kind.startswith("sines+"):    # sines + n bits of noise
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:YS, 0:XS].astype(np.float32)
    def sines(k): return 30000 + 9000 * np.sin(xx / (310.0 + 7 * k) + 0.3 * k) + 7000 * np.sin(yy / (270.0 - 5 * k) + 0.2 * k) + 5000 * np.sin((xx + yy) / (530.0 + 11 * k))
    base = np.stack([np.clip(sines(k) + rng.integers(0, 1 << int(kind[6:]), (YS, XS)), 0, 65535) for k in range(2)]).astype(np.uint16); shift = 0
elif kind.startswith("dark") and len(kind) > 4:   # dark<sigma>
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:YS, 0:XS].astype(np.float32)
    blobs = 3000 * np.exp(-((xx - 400) ** 2 + (yy - 300) ** 2) / 5000.0) + 1500 * np.exp(-((xx - 900) ** 2 + (yy - 500) ** 2) / 20000.0)
    base = np.stack([np.clip(100 + blobs * (k > 0) + rng.normal(0, float(kind[4:]), (YS, XS)), 0, 4095) for k in range(2)]).astype(np.uint16); shift = 4
elif kind == "dark":
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:YS, 0:XS].astype(np.float32)
    blobs = 3000 * np.exp(-((xx - 400) ** 2 + (yy - 300) ** 2) / 5000.0) + 1500 * np.exp(-((xx - 900) ** 2 + (yy - 500) ** 2) / 20000.0)
    base = np.stack([np.clip(100 + blobs * (k > 0) + rng.normal(0, 4, (YS, XS)), 0, 4095) for k in range(2)]).astype(np.uint16); shift = 4
else:
    base = synth.plasma_frames(2, XS, YS, bits=12, seed=7); shift = 4
orc = Oracle()
high, low, pv, flags = orc.predict(base[1], XS, YS, shift, False, base[0])
assert flags & 2, flags
R = np.asarray(high, dtype=np.int64).reshape(-1)            # residual bytes of the high plane (after delta)
# the truth: the plane the chain reconstructs (before delta^-1): x[i] = r[i] + CG(x[i-xs], x[i-1], x[i-xs-1]) mod 256
x = np.zeros(XS * YS, dtype=np.int64)
x[:XS + 1] = R[:XS + 1]
for i in range(XS + 1, XS * YS):
    n, w, nw = x[i - XS], x[i - 1], x[i - XS - 1]
    g = n + w - nw
    x[i] = (R[i] + max(min(n, w), min(max(n, w), g))) & 255
head = XS + 4
S = 64 * K
def seg(nv, dv, rv, win):           # one pass of every lane: win (64,), nv/dv/rv (64, K) -> outputs (64, K)
    out = np.zeros_like(nv); w = win.copy()
    for k in range(K):
        g = w + dv[:, k]
        lo, hi = np.minimum(nv[:, k], w), np.maximum(nv[:, k], w)
        w = (rv[:, k] + np.maximum(lo, np.minimum(hi, g))) & 255
        out[:, k] = w
    return out
worst = []
nwrong_first = []
for t in range((XS * YS - head) // S):
    s0 = head + t * S
    idx = s0 + np.arange(S).reshape(64, K)
    nv, nwv, rv = x[idx - XS], x[idx - XS - 1], R[idx]
    dv = nv - nwv
    wprev = x[s0 - 1]
    inp = x[idx[:, 0] - 1 - XS].copy()                       # guess: the pixel above the west neighbour
    if os.environ.get("GUESS") == "nr": inp = (inp + R[idx[:, 0] - 1]) & 255   # ... + the west neighbour's own residual
    inp[0] = wprev
    nwrong_first.append(int((inp != np.concatenate([[wprev], x[idx[:-1, K - 1]]])).sum()))
    static = (dv == 0).all(axis=1); static[0] = False
    log, errs, npass = [], [], 0
    hist = np.zeros(64, bool); evid = np.zeros(64, bool); dyn = np.full(64, optimist in (1, 4)); memo = optimist in (3, 4); min_ = np.zeros(64, np.int64); mout = np.zeros(64, np.int64); mhave = np.zeros(64, bool); prev_in = None; prev_out = None; pin = np.zeros(64, np.int64); pout = np.zeros(64, np.int64)
    X = None
    while True:
        Xn = seg(nv, dv, rv, inp)
        if X is None: X = Xn
        else: X = Xn
        npass += 1
        last = X[:, K - 1]
        if memo and npass > MEMO_FROM and prev_in is not None:       # the pair a lane had before its input last changed
            chg = inp != prev_in
            min_ = np.where(chg, prev_in, min_); mout = np.where(chg, prev_out, mout); mhave |= chg
        prev_in, prev_out = inp.copy(), last.copy()
        if npass > 4:
            ch = hist & (inp != pin)
            dyn = np.where(ch, ((last - pout) & 255) == ((inp - pin) & 255), dyn); evid |= ch
            upd = (~hist) | (inp != pin)
            pin = np.where(upd, inp, pin); pout = np.where(upd, last, pout); hist |= True
        pred = np.concatenate([[wprev], last[:-1]])
        bad = pred != inp
        tin = np.concatenate([[x[s0 - 1]], x[idx[:-1, K - 1]]])
        log.append(np.flatnonzero(bad))
        errs.append([(int(j), int(((inp[j] - tin[j] + 128) & 255) - 128), int(((last[j] - x[idx[j, K - 1]] + 128) & 255) - 128)) for j in np.flatnonzero((inp != tin) | (last != x[idx[:, K - 1]]))])
        if not bad.any() or npass > 70: break
        nxt = pred.copy()
        if memo and MEMO_FROM < 4 and npass > MEMO_FROM and npass < (3 if os.environ.get('MEMO_ONCE') else 4):   # the memo alone behind the early passes (no scan yet); MEMO_ONCE: behind the second pass only
            hitm = mhave & (nxt == min_); hitm[0] = False
            est = np.where(hitm, mout, last)
            nxt2 = np.concatenate([[wprev], est[:-1]])
            nxt = np.where(np.concatenate([[False], hitm[:-1]]), nxt2, nxt)
        if optimist != -1 and npass >= 4 and npass < 28:
            if optimist == 2:    # momentum: a lane without evidence believes what the nearest lane to its left with evidence showed
                ev = evid.copy(); bel = dyn.copy()
                for j in range(1, 64):
                    if not ev[j]: bel[j] = bel[j - 1] if j > 1 else False
                sh = static | bel
            else:
                sh = static.copy() if optimist == -2 else (static | dyn)
            sh[0] = False
            isc = (~sh).astype(np.int64); v = np.where(sh, (last - inp) & 255, last)
            # inclusive segmented scan, sequential model
            for j in range(1, 64):
                if not isc[j]:
                    v[j] = (v[j] + v[j - 1]) & 255; isc[j] = isc[j - 1]
            nxt = np.concatenate([[wprev], v[:-1]])
            if memo:   # a lane proposed the input it had before answers with the output it had then; its right neighbour is proposed that
                for _ in range(int(os.environ.get("MEMO_ITERS", "1"))):   # (iterated: a memo hit's own right neighbour may hit its memo in turn)
                    hitm = mhave & (nxt == min_)
                    est = np.where(hitm, mout, np.where(nxt == inp, last, v))
                    nxt2 = np.concatenate([[wprev], est[:-1]])
                    nxt = np.where(np.concatenate([[False], hitm[:-1]]), nxt2, nxt)
        inp = nxt
    worst.append((npass, t, log, errs))
worst.sort(key=lambda a: -a[0])
hist_p = np.bincount([w[0] for w in worst])
print("wrong first guesses per superstep (of 63):", round(float(np.mean(nwrong_first)), 2)); print("share of supersteps in the long-correction path (5 passes and more):", round(sum(1 for w in worst if w[0] >= 5) / len(worst), 3)); print(kind, "optimist", optimist, "total passes", sum(w[0] for w in worst), "passes histogram", {i: int(c) for i, c in enumerate(hist_p) if c})
for npass, t, log, errs in worst[:1]:
    print(f"superstep {t}: {npass} passes; wrong lanes after each pass:")
    for p, l in enumerate(log[:45]): print(f"  pass {p + 1:2d}: {len(l):2d} wrong: {l.tolist()[:24]}   (lane, input error, output error): {errs[p][:12]}")
# the front of the worst superstep, lane by lane: what a lane's input / output were before and after the correction reached it
npass, t, log, errs = worst[0]
s0 = head + t * S
idx = s0 + np.arange(S).reshape(64, K)
nv, nwv, rv = x[idx - XS], x[idx - XS - 1], R[idx]
dv = nv - nwv
truth_in = np.concatenate([[x[s0 - 1]], x[idx[:-1, K - 1]]])
guess = x[idx[:, 0] - 1 - XS].copy(); guess[0] = x[s0 - 1]
print("lane: guess - true input, output(guess) - output(true), number of nonzero north differences, sum |r|")
for j in range(20, 64):
    og = seg(nv[j:j + 1], dv[j:j + 1], rv[j:j + 1], guess[j:j + 1])[0, K - 1]
    ot = seg(nv[j:j + 1], dv[j:j + 1], rv[j:j + 1], truth_in[j:j + 1])[0, K - 1]
    outs = [int(seg(nv[j:j + 1], dv[j:j + 1], rv[j:j + 1], np.array([(truth_in[j] + e) & 255]))[0, K - 1] - ot) for e in (-3, -2, -1, 0, 1, 2, 3)]
    print(f"  {j:2d}: {int(guess[j] - truth_in[j]):+d}  {int(og - ot):+d}  D!=0: {int((dv[j] != 0).sum())}  sum|r| {int(np.minimum(rv[j], 256 - rv[j]).sum())}   out(true+e)-out(true), e=-3..3: {outs}")
j = 26
w0 = truth_in[j]
for e in (0, 1):
    w = (w0 + e) & 255; tr = []
    for k in range(K):
        g = w + dv[j, k]; lo, hi = min(nv[j, k], w), max(nv[j, k], w)
        w = (rv[j, k] + max(lo, min(hi, g))) & 255; tr.append(int(w))
    print("lane 26, input", int((w0 + e) & 255), "->", tr)
print("n :", nv[j].tolist()); print("d :", dv[j].tolist()); print("r :", rv[j].tolist())
