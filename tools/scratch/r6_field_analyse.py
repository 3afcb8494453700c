#!/usr/bin/env python3
"""Offline: per-k histories (FI_FIELD_TRACE) and true errors of tools/scratch/r6_field_trace.py -> how candidate estimators would
have stopped.  usage: r6_field_analyse.py <trace file> <tol> [print]"""
import sys, re, math
import numpy as np
path, tol = sys.argv[1], float(sys.argv[2])
hist, true = {}, {}
cur = {}
for line in open(path):
    m = re.match(r"field trace (\d+) r (\S+) s (\S+) t2 (\S+) est (\S+)", line)
    if m:
        k = int(m.group(1)); cur[k] = (float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)))
        continue
    m = re.match(r"RUN (\d+) true (\S+)", line)
    if m:
        k = int(m.group(1)); true[k] = float(m.group(2)); hist[k] = dict(cur); cur = {}
K = max(true)
H = hist[K]                       # the longest run's history (a consistent sequence if the runs repeat each other)
# consistency: does run k's last entry equal the longest run's entry k?
bad = sum(1 for k in true if k in H and k in hist[k] and abs(hist[k][k][0] - H[k][0]) > 1e-9 * abs(H[k][0]))
r = {k: H[k][0] for k in H}; s = {k: H[k][1] for k in H}; t2 = {k: H[k][2] for k in H}; shipped = {k: H[k][3] for k in H}
print("%s: %d iterations, %d runs inconsistent with the longest" % (path, K, bad))

def est_shipped(k):
    return shipped[k]

def rate(series, k, lags=(1, 2, 4, 8, 16), whole=True, power=1.0):
    sig = 0.0
    for lag in lags:
        if k - lag >= 1 and series[k - lag] > 0:
            sig = max(sig, (series[k] / series[k - lag]) ** (power / lag))
    return sig

def est_t(k, margin=2.0):
    """sigma from the A-norm steps (t2: squared), step carried"""
    if k < 2: return -1
    sig = rate(t2, k, power=0.5)
    sig = max(sig, rate(r, k))
    if not (0 < sig < 0.95): return -1
    step = s[k]; f = sig
    for j in range(1, 4):
        if k - j >= 1: step = max(step, s[k - j] * f); f *= sig
    return margin * step * sig / (1 - sig)

def est_delay(k, d=2, margin=1.0):
    """the error at k - d as the sum of the d steps since, plus the geometric tail of the last; accepted for x_k"""
    if k <= d: return -1
    sig = max(rate(r, k), rate(t2, k, power=0.5))
    if not (0 < sig < 0.97): return -1
    tail = s[k] * sig / (1 - sig)
    return margin * (sum(s[j] for j in range(k - d + 1, k + 1)) + tail)

def est_carry(k, W=16, margin=2.0, fast=0.5):
    """the shipped rule with the step carried over W iterations"""
    if k < 1: return -1
    sig = rate(r, k)
    if k > 2: sig = max(sig, (r[k] / r0) ** (1.0 / k))
    if not (0 < sig < 0.95): return -1
    step = s[k]
    if sig < fast:
        if k - 1 >= 1: sig = r[k] / r[k - 1]
        else: sig = r[k] / r0
    else:
        f = sig
        for j in range(1, W + 1):
            if k - j >= 1: step = max(step, s[k - j] * f); f *= sig
    return margin * step * sig / (1 - sig)

def first_stop(est):
    for k in range(2, K + 1):
        e = est(k)
        if 0 <= e <= tol and k in true: return k, e
    return None, None

r0 = r[1]
for name, est in (("shipped", est_shipped), ("carry 3", lambda k: est_carry(k, 3)), ("carry 8", lambda k: est_carry(k, 8)), ("carry 16", lambda k: est_carry(k, 16)), ("carry 30", lambda k: est_carry(k, 30)), ("t-rate", est_t), ("delay 1", lambda k: est_delay(k, 1)), ("delay 2", lambda k: est_delay(k, 2)), ("delay 3", lambda k: est_delay(k, 3))):
    k, e = first_stop(est)
    if k is None: print("  %-8s never stops" % name); continue
    kmin = min((kk for kk in true if all(true[j] <= tol for j in true if j >= kk)), default=None)
    print("  %-8s stops at %4d (estimate %.2e, true %.2e = %.2f x tol); first k from which the true error stays below tol: %s" % (name, k, e, true[k], true[k] / tol, kmin))
if len(sys.argv) > 3:
    for k in sorted(true):
        if k in r: print("%4d r %.2e s %.2e sqrt(t2) %.2e shipped %.2e t-rate %.2e delay2 %.2e TRUE %.2e" % (k, r[k], s[k], math.sqrt(max(t2[k], 0)), shipped[k], est_t(k), est_delay(k, 2), true[k]))
