#!/usr/bin/env python3
"""Static LDS-DMA schedule of the ping-pong GEMM (carel_vae_amd/csrc/gemm_pp.hip) -- generator AND checker.

The kernel runs 8 waves as two groups of four (group = wave >> 2) that alternate between a "load" segment L(q)
(fragment ds_reads of phase q, issue of this phase's DMA units, one counted s_waitcnt vmcnt) and a "matrix" segment
M(q) (12 MFMAs), separated by workgroup barriers; group 1 runs one barrier behind group 0:

    barrier index   ... B(2q-1) |   B(2q)   | B(2q+1) ...
    group 0             L(q)    |   M(q)    |  L(q+1)
    group 1             M(q-1)  |   L(q)    |  M(q)

A K tile (64 deep) is NP = 2 * NPN phases: (M half h of each wave's 64 rows) x (B part j of 96 columns), walked in
serpentine order.  DMA units, all 2 global_load_lds_dwordx4 per wave: A0, A1 (the two M halves, 16 KiB each) and
B0..B{NPN-1} (12 KiB ROW images / 16 KiB COL images).  S LDS stages (tile t lives in stage t % S).

Rules (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"; derived for this barrier
pattern in DESIGN.md section 4):
  RAW  a unit first read in phase n must have been waited for (vmcnt) by EVERY wave in L(n-1) or earlier.
  WAR  a unit may be issued in L(i) only if the previous occupant of its LDS slot was last read in phase <= i-2.
vmcnt retires in issue order, so "wait for unit X" = s_waitcnt vmcnt(2 * units issued after X so far).

Output: per phase p the units issued (type, tile delta) and the vmcnt immediates for the steady state and for the
last tiles (R = tiles remaining, counting the current one), plus the prologue issue list.  `check()` replays the
schedule event by event for many tiles and asserts both rules; tests/test_gemm_sched.py runs it and compares the
tables with the constexpr copies in gemm_pp.hip.
"""
import sys


def phases(npn, wide=False):
    """(h, j) order of a K tile: serpentine over (M half, B part); wide: one phase per B part, both M halves (h = -1),
    the A fragments of both halves read once in phase 0 and kept in registers"""
    if wide:
        return [(-1, j) for j in range(npn)]
    out = []
    h = 0
    for j in range(npn):
        out.append((h, j))
        h ^= 1
        out.append((h, j))
    return out


def phase_reads(npn, wide=False):
    """the units whose LDS image phase p reads"""
    if wide:
        return [(["A0", "A1"] if j == 0 else []) + ["B%d" % j] for j in range(npn)]
    return [["A%d" % h, "B%d" % j] for (h, j) in phases(npn)]


def unit_span(npn, wide=False):
    """unit -> (first phase, last phase) inside a tile"""
    span = {}
    for p, us in enumerate(phase_reads(npn, wide)):
        for u in us:
            f, l = span.get(u, (p, p))
            span[u] = (min(f, p), max(l, p))
    return span


def make(npn, stages, max_lead=None, wide=False, war=2):
    """war: phases between a unit's last read and the issue that overwrites its slot.  2 for the fine schedule (group 1's
    reads of phase q are only known complete at the start of its M(q), one barrier after group 0's L(q+1)); 1 when every
    wave drains its LDS reads (lgkmcnt(0)) BEFORE the barrier that ends its load segment (the wide schedule does)."""
    NP = len(phases(npn, wide))
    span = unit_span(npn, wide)
    units = ["A0", "A1"] + ["B%d" % j for j in range(npn)]
    lead = {}
    for u in units:
        first, last = span[u]
        d = stages * NP - last + first - war         # WAR: n - d >= (t - S) * NP + last + war
        if max_lead is not None:
            d = min(d, max_lead)
        if d < 2:
            raise ValueError("no legal issue phase for %s (npn=%d stages=%d)" % (u, npn, stages))     # RAW needs one phase of flight
        lead[u] = d
    # issue[p] = [(unit, tile delta)], ordered by need phase
    issue = [[] for _ in range(NP)]
    for u in units:
        first, _ = span[u]
        i = first - lead[u]                           # relative to tile t's phase 0
        delta = 0
        while i < 0:
            i += NP
            delta += 1
        issue[i].append((u, delta, first))
    for p in range(NP):
        issue[p].sort(key=lambda x: (x[1] * NP + x[2], x[0]))
        issue[p] = [(u, d) for (u, d, _) in issue[p]]
    return dict(npn=npn, stages=stages, NP=NP, span=span, units=units, lead=lead, issue=issue, wide=wide, war=war)


def program(s, nk):
    """Event list of ONE wave: ('issue', unit, tile) / ('read', unit, tile, q) / ('wait', q) in program order, with the
    prologue first.  The prologue issues every (unit, tile) the steady-state schedule would have issued in a phase < 0."""
    NP = s["NP"]
    span = s["span"]
    ev = []
    pro = []
    for p in range(NP):
        for (u, d) in s["issue"][p]:
            # steady state: phase (t, p) issues (u, t + d).  Tiles t + d with t < 0:
            for t in range(-d, 0):
                if 0 <= t + d < nk:
                    pro.append(((t * NP + p), (t + d) * NP + span[u][0], u, t + d))
    pro.sort()
    for (_, _, u, tt) in pro:
        ev.append(("issue", u, tt))
    reads = phase_reads(s["npn"], s["wide"])
    for t in range(nk):
        for p in range(NP):
            q = t * NP + p
            for u in reads[p]:
                ev.append(("read", u, t, q))
            for (u, d) in s["issue"][p]:
                if t + d < nk:
                    ev.append(("issue", u, t + d))
            ev.append(("wait", q))
    return ev


def wait_counts(s, nk):
    """vmcnt immediate of every wait: 2 * (units issued after the last unit whose FIRST read is in phase q+1)."""
    NP = s["NP"]
    span = s["span"]
    ev = program(s, nk)
    issued = []        # program-order list of (unit, tile)
    waits = {}
    for e in ev:
        if e[0] == "issue":
            issued.append((e[1], e[2]))
        elif e[0] == "wait":
            q = e[1]
            need = [k for k, (u, tt) in enumerate(issued) if tt * NP + span[u][0] == q + 1]
            if q + 1 >= nk * NP:
                waits[q] = None
            elif not need:
                waits[q] = None
            else:
                waits[q] = 2 * (len(issued) - 1 - max(need))
    return waits


def tables(s):
    """steady-state and tail vmcnt tables: W[R][p], R = 0 steady, 1 = last tile, 2 = second to last ..."""
    NP = s["NP"]
    nk = 12
    w = wait_counts(s, nk)
    maxd = max(d for p in range(NP) for (_, d) in s["issue"][p]) if any(s["issue"]) else 0
    tabs = {}
    t_mid = nk // 2
    tabs[0] = [w[t_mid * NP + p] for p in range(NP)]
    for R in range(1, maxd + 2):
        t = nk - R
        tabs[R] = [w[t * NP + p] for p in range(NP)]
    # the steady table must hold for every tile that is not in a tail table
    for t in range(0, nk - (maxd + 1)):
        assert [w[t * NP + p] for p in range(NP)] == tabs[0], (t, [w[t * NP + p] for p in range(NP)], tabs[0])
    return tabs, maxd + 1


def check(s, nk):
    """Replay both groups against the barrier pattern and assert RAW / WAR."""
    NP = s["NP"]
    S = s["stages"]
    span = s["span"]
    ev = program(s, nk)
    w = wait_counts(s, nk)
    # time stamps in half-slots: group g's L(q) lies in barrier interval 2q - 1 + g, M(q) in 2q + g.
    # "retired[u,t]" = barrier interval after which EVERY wave has waited for it = (2q + 1) with q the wait phase
    issued = []
    retired_at = {}         # (unit, tile) -> phase q of the wait that retires it (same for both groups)
    issue_phase = {}        # (unit, tile) -> phase in whose L segment it is issued (-1 = prologue)
    cur_q = -1
    for e in ev:
        if e[0] == "issue":
            issued.append((e[1], e[2]))
            issue_phase[(e[1], e[2])] = cur_q if cur_q >= 0 else -1
        elif e[0] == "read":
            cur_q = e[3]
        elif e[0] == "wait":
            q = e[1]
            n = w[q]
            if n is not None:
                done = len(issued) - n // 2
                for k in range(done):
                    retired_at.setdefault(issued[k], q)
    last_read = {}
    for e in ev:
        if e[0] == "read":
            _, u, t, q = e
            key = (u, t)
            # RAW: retired in a phase <= q - 1 (prologue units of phase 0 are drained by the prologue's vmcnt(0) + barrier)
            if q == 0:
                continue
            assert key in retired_at and retired_at[key] <= q - 1, ("RAW", key, q, retired_at.get(key))
            last_read[key] = q
    for e in ev:
        if e[0] == "read":
            last_read[(e[1], e[2])] = e[3]
    for (u, t), i in issue_phase.items():
        prev = (u, t - S)
        if prev in last_read:
            assert i >= last_read[prev] + s["war"], ("WAR", (u, t), i, last_read[prev])
    # in-flight bound: vmcnt is a 6-bit counter
    assert max([x for x in w.values() if x is not None] + [0]) <= 63
    return True


def first_tile_waits_target_prologue(s, nk=12, ne=12):
    """gemm_pp.hip requests its epilogue's inputs (ne plain load instructions) between the prologue's DMA units and K tile 0's, and adds ne
    to the prologue wait and to every counted wait of K tile 0.  vmcnt retires in issue order, so that is exact iff each of those waits
    targets a unit issued in the PROLOGUE (older than the loads): then "everything but the newest w + ne instructions" still retires the
    target and nothing newer.  Replayed here at instruction level: with the loads in the stream and the widened immediates, every unit is
    retired by the same wait as without them (so RAW / WAR of check() carry over), and no wait of tile 0 retires a load."""
    NP = s["NP"]
    span = s["span"]
    ev = program(s, nk)
    w = wait_counts(s, nk)
    first_read = next(k for k, e in enumerate(ev) if e[0] == "read")
    stream = []            # instruction-level issue order: (kind, key); a unit is 2 instructions
    retired_plain, retired_with = {}, {}
    for with_loads in (False, True):
        stream, retired = [], (retired_with if with_loads else retired_plain)
        for k, e in enumerate(ev):
            if k == first_read and with_loads:
                stream += [("load", i) for i in range(ne)]
            if e[0] == "issue":
                stream += [("unit", (e[1], e[2]))] * 2
            elif e[0] == "wait":
                q = e[1]
                n = w[q]
                if n is None:
                    continue
                if with_loads and q < NP:
                    n += ne
                for item in stream[:max(0, len(stream) - n)]:
                    retired.setdefault(item, q)
    for key, q in retired_plain.items():
        assert retired_with.get(key) == q, ("unit retires at a different wait with the input loads in flight", key, q, retired_with.get(key))
    loads = [q for (kind, _), q in retired_with.items() if kind == "load"]
    assert loads and min(loads) >= NP, ("a wait of tile 0 retires an input load", loads)
    return True


def describe(npn, stages, max_lead=None, wide=False, war=2):
    s = make(npn, stages, max_lead, wide, war)
    for nk in (4, 5, 6, 7, 12, 13, 36, 48):
        check(s, nk)
    tabs, ntail = tables(s)
    return s, tabs, ntail


CONFIGS = {1: 3, 2: 2, 3: 2}     # NPN -> LDS stages
LEADS = {1: 4, 2: 4, 3: 5}       # NPN -> cap on the issue lead in phases (about 3-4 units = 45-60 KiB in flight per CU)
WIDE_CONFIGS = {1: 3, 2: 2, 3: 2}             # wide phases (24 MFMAs each): NPN -> LDS stages; war = 1, leads uncapped
UNIT_ID = {"A0": 0, "A1": 1, "B0": 2, "B1": 3, "B2": 4}


def prologue_of(s, nk=12):
    ev = program(s, nk)
    first_read = next(k for k, e in enumerate(ev) if e[0] == "read")
    pro = [(e[1], e[2]) for e in ev[:first_read]]
    span = s["span"]
    need0 = [k for k, (u, t) in enumerate(pro) if t * s["NP"] + span[u][0] == 0]
    return pro, 2 * (len(pro) - 1 - max(need0))


def header():
    """C++ tables included by carel_vae_amd/csrc/gemm_pp.hip (committed as gemm_pp_sched.inc; test_gemm_sched.py diffs it)"""
    out = ["// generated by tools/gemm_sched.py (python tools/gemm_sched.py --header) -- do not edit by hand.",
           "// Units: 0 = A0, 1 = A1, 2 + j = Bj.  wait[R][p]: vmcnt immediate at the end of L(p); R = 0 steady state,",
           "// R = r: r tiles remain including the current one; -1 = no wait.",
           "template <int NPN> struct PPSched;"]
    out.append("template <int NPN> struct PPSchedW;     // wide phases: phase_h = -1 (both M halves), A fragments read in phase 0 only")
    todo = [("PPSched", npn, st, LEADS[npn], False, 2) for npn, st in CONFIGS.items()]
    todo += [("PPSchedW", npn, st, None, True, 1) for npn, st in WIDE_CONFIGS.items()]
    for (name, npn, st, cap, wide, war) in todo:
        s, tabs, ntail = describe(npn, st, cap, wide, war)
        while ntail > 1 and tabs[ntail] == tabs[0]:
            ntail -= 1
        NP = s["NP"]
        maxi = max(1, max(len(x) for x in s["issue"]))
        pro, pro_wait = prologue_of(s)
        ph = phases(npn, wide)
        def arr(rows):
            return "{" + ", ".join("{" + ", ".join(str(v) for v in r) + "}" for r in rows) + "}"
        iu = [[UNIT_ID[x[0]] for x in s["issue"][p]] + [-1] * (maxi - len(s["issue"][p])) for p in range(NP)]
        idl = [[x[1] for x in s["issue"][p]] + [0] * (maxi - len(s["issue"][p])) for p in range(NP)]
        wt = [[(-1 if v is None else v) for v in tabs[R]] for R in range(ntail + 1)]
        out += ["template <> struct %s<%d> {" % (name, npn),
                "  static constexpr int NP = %d, STAGES = %d, NTAIL = %d, MAXI = %d, NPRO = %d, PRO_WAIT = %d;" % (NP, st, ntail, maxi, len(pro), pro_wait),
                "  static constexpr int phase_h[NP] = {%s};" % ", ".join(str(h) for h, _ in ph),
                "  static constexpr int phase_j[NP] = {%s};" % ", ".join(str(j) for _, j in ph),
                "  static constexpr int n_issue[NP] = {%s};" % ", ".join(str(len(x)) for x in s["issue"]),
                "  static constexpr int issue_unit[NP][MAXI] = %s;" % arr(iu),
                "  static constexpr int issue_delta[NP][MAXI] = %s;" % arr(idl),
                "  static constexpr int wait[NTAIL + 1][NP] = %s;" % arr(wt),
                "  static constexpr int pro_unit[NPRO] = {%s};" % ", ".join(str(UNIT_ID[u]) for u, _ in pro),
                "  static constexpr int pro_tile[NPRO] = {%s};" % ", ".join(str(t) for _, t in pro),
                "};"]
    return "\n".join(out) + "\n"


def emit():
    out = []
    todo = [(npn, st, LEADS[npn], False, 2) for npn, st in CONFIGS.items()] + [(npn, st, None, True, 1) for npn, st in WIDE_CONFIGS.items()]
    for (npn, st, cap, wide, war) in todo:
        s, tabs, ntail = describe(npn, st, cap, wide, war)
        out.append("NPN=%d stages=%d%s phases=%s" % (npn, st, " WIDE" if wide else "", phases(npn, wide)))
        out.append("  lead   %s" % s["lead"])
        for p in range(s["NP"]):
            out.append("  phase %d issue %s" % (p, s["issue"][p]))
        for R in sorted(tabs):
            out.append("  vmcnt R=%d %s" % (R, tabs[R]))
        pro = [e for e in program(s, 12) if e[0] == "issue"]
        first_read = next(k for k, e in enumerate(program(s, 12)) if e[0] == "read")
        out.append("  prologue %s" % [(e[1], e[2]) for e in program(s, 12)[:first_read]])
    return "\n".join(out)


if __name__ == "__main__":
    if "--header" in sys.argv:
        sys.stdout.write(header())
    else:
        print(emit())
