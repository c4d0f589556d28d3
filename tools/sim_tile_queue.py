#!/usr/bin/env python3
"""Discrete-event model of a two-round GEMM launch ("store pattern"): 256 CUs pull tiles of width 96 * npn from the launch
queue in order; a tile = prologue + 12 K tiles + epilogue VALU, then its stores, which share the chip write rate (6 TB/s measured by
tools/ubench/store_pattern.hip) with every other CU storing at that moment.  Prints the modelled launch time of queue orders that de-phase the
rounds with mixed tile widths.  Measurement aid only."""
import sys
# discrete-time simulation (dt = 0.05 us) of 256 CUs pulling tiles from an ordered queue
T_ITER = {1: 0.77, 2: 1.28, 3: 1.83}
def run(queue, nk=12, pro=2.0, valu_per_unit=2.5, bytes_per_unit=98304*2, chip_bw=6.0e6, cu_bw=0.10e6, ncu=256, dt=0.02):
    # bw in bytes/us
    q = list(queue); qi = 0
    # state per CU: (phase, remaining)
    cus = [None]*ncu
    t = 0.0; done = 0; n = len(q)
    while done < n:
        storing = [c for c in range(ncu) if cus[c] and cus[c][0] == 's']
        share = min(cu_bw, chip_bw/len(storing)) if storing else 0
        for c in range(ncu):
            s = cus[c]
            if s is None:
                if qi < n:
                    npn = q[qi]; qi += 1
                    cus[c] = ['c', pro + nk*T_ITER[npn] + valu_per_unit*npn, npn]
                continue
            if s[0] == 'c':
                s[1] -= dt
                if s[1] <= 0: cus[c] = ['s', bytes_per_unit*s[2], s[2]]
            else:
                s[1] -= share*dt
                if s[1] <= 0: cus[c] = None; done += 1
        t += dt
    return t
def q_uniform2(): return [2]*512
def q_mix121():   # per XCD-agnostic: first 128 x2 + 128 x1 interleaved, then 256 x2, then 128 x1
    a = []
    for i in range(128): a += [2, 1]
    return a + [2]*256 + [1]*128
def q_mix31():
    a = []
    for i in range(128): a += [3, 1]
    return a + [3]*128 + [1]*128
def q_mix3_121():   # 3,1 | 1... 
    a = []
    for i in range(128): a += [3, 1]
    return a + [2]*128 + [1]*128+[1]*128
for name, f in [("uniform [2,2]", q_uniform2), ("[2,2]/[1,2,1]", q_mix121), ("[3,1]/[1,3]", q_mix31), ("[3,1]/[1,2,1]", q_mix3_121)]:
    for valu, bpu, lab in [(2.5, 49152*2, "GELU_DG 2 outs"), (0.3, 49152, "bias->bf16 1 out"), (0.6, 49152*2, "MUL: 1 out + 1 in")]:
        print("%-16s %-20s %.1f us" % (name, lab, run(f(), valu_per_unit=valu, bytes_per_unit=bpu)))
print("---- more orders (GELU_DG)")
def interleave(first, rest): 
    a = []
    for i in range(256): a.append(first[i % len(first)])
    return a + rest
cands = {
 "thirds 3/2/1 then ...": interleave([3,2,1,2], [1]*64 + [2]*128 + [3]*64 + [1]*0),
 "[1,3],[3,1] + [2,2]":   interleave([3,1,2,2], [3]*64 + [2]*128 + [1]*64),
 "4 groups": interleave([1,2,2,1], [2]*128 + [1]*128 + [2]*0 + [1]*0+[2]*64+[1]*0),
}
for k, q in cands.items():
    units = sum(q)
    print("%-24s units %d  %.1f us" % (k, units, run(q, valu_per_unit=2.5, bytes_per_unit=49152*2)))
# ideal: no store cost
print("no stores uniform", run([2]*512, valu_per_unit=2.5, bytes_per_unit=1))
