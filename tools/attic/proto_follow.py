"""Prototype: envelope-follower states by Newton / policy iteration over 256-sample chunks with an affine scan between iterations."""
import sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import synth
SR = 48000
x = synth.speech_like(30.0, SR, seed=5).astype(np.float64)
a = x * x
att, rel = 1 / 60.0, 1 / 2400.0
L = 256
n = (a.size // L) * L; a = a[:n]; nch = n // L
A = a.reshape(nch, L)
# reference sequential
def seq(a):
    s = 0.0; out = np.empty(nch + 1); out[0] = 0
    for c in range(nch):
        row = A[c]
        for t in range(L):
            v = row[t]; s += (v - s) * (att if v > s else rel)
        out[c + 1] = s
    return out
import time
t0 = time.time(); ref = seq(a); print("seq", time.time() - t0)
S = np.zeros(nch)
for it in range(12):
    s = S.copy(); M = np.ones(nch)
    for t in range(L):
        v = A[:, t]; c = np.where(v > s, att, rel)
        s = s + (v - s) * c; M *= (1 - c)
    E = s; B = E - M * S
    Sn = np.zeros(nch); st = 0.0
    for c in range(nch):
        Sn[c] = st; st = M[c] * st + B[c]
    chg = np.max(np.abs(Sn - S) / np.maximum(np.abs(Sn), 1e-300))
    err = np.max(np.abs(Sn - ref[:nch]) / np.maximum(ref[:nch], 1e-300))
    print(it, "max rel change %.3e" % chg, "max rel err vs sequential %.3e" % err)
    S = Sn
    if chg < 1e-13: break

def iterate(S, label):
    for it in range(14):
        s = S.copy(); M = np.ones(nch)
        for t in range(L):
            v = A[:, t]; c = np.where(v > s, att, rel)
            s = s + (v - s) * c; M *= (1 - c)
        E = s; B = E - M * S
        Sn = np.zeros(nch); st = 0.0
        for c in range(nch):
            Sn[c] = st; st = M[c] * st + B[c]
        chg = np.max(np.abs(Sn - S) / np.maximum(np.abs(Sn), 1e-300))
        S = Sn
        if chg < 1e-13:
            print(label, "converged after", it + 1, "iterations"); return
    print(label, "not converged")
# (b) pure-release linear average as the start
s = 0.0; S0 = np.zeros(nch)
for c in range(nch):
    S0[c] = s
    for t in range(L): s += (A[c, t] - s) * rel
iterate(S0.copy(), "release-average start")
# (c) pure-attack linear average
s = 0.0; S1 = np.zeros(nch)
for c in range(nch):
    S1[c] = s
    for t in range(L): s += (A[c, t] - s) * att
iterate(S1.copy(), "attack-average start")
iterate(np.maximum(S0, S1), "max(release, attack) start")
# (d) chunk maxima decayed: upper-ish bound
mx = A.max(axis=1); s = 0.0; S2 = np.zeros(nch)
for c in range(nch):
    S2[c] = s; s = max(s * (1 - rel) ** L, mx[c])
iterate(S2.copy(), "decayed chunk max start")
