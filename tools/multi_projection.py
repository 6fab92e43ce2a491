"""8-GPU critical-path projection of the sharded factorisation from per-panel timings measured with P loopback ranks on
one GPU (tools/gpu_multi_profile.py -> JSON; round-4 layout: per panel factor_ms, lookahead_update_ms, panel_bytes, then
per rank near_a_ms, near_b_ms, far_ms -- csrc/multi.hip explains the three update classes).

Every GPU is ONE server: its launches run one at a time, in the order the streams allow, the panel stream (look-ahead
update + factorisation) first, then the near stream, then the update stream -- a launch of a higher stream that becomes
ready while a lower one runs pushes the lower one's completion back by its own duration (dispatch priority; `contend`
additionally stretches the look-ahead work while the GPU has other work, 1.3 = what the single-GPU look-ahead measured).
This is more conservative than the round-3 model, which let the look-ahead overlap the owner's updates for free.
Dependencies as csrc/multi.hip enqueues them (G = panels per group, g = J // G):

  panel J is factored, sent and applied to panel J+1 in nsub[J] sub-panels (the pipeline of csrc/multi.hip: factor): sub-panel
  q is final at start + (q + 1) fac / nsub, lands transport / nsub later (the pieces queue on the links), and the look-ahead
  update with it (la / nsub) starts when it has landed and the previous one is done -- only the last piece's transport and
  update sit between the end of J's factorisation and the start of J+1's
  look-ahead(J+1) on owner(J+1): needs near_a(J-1) there (and near_b(J-1) for the first two panels of a group)
  recv[J+1][i] = the last sub-panel of J+1 has landed          (i != owner)
  near_a(J)[i]: needs recv[J][i] (+ near_b(J-1)[i] at the first step of a group)       -- the current group's panels
  near_b(J)[i]: needs recv[J][i], after everything earlier on the update stream          -- the next group's panels
  far(g)[i]   : at the last step of group g, after near_b of that step                   -- groups >= g + 2, K = G panels

transport(bytes): scatter + all-gather over point-to-point xGMI links of `link` GB/s per direction: 2 * bytes / ((P - 1) *
link); direct copy: bytes / link.
Round 6: the bracketed "phases pipelined" figure lets the all-gather of sub-panel q run under the scatter of sub-panel q + 1 (they
use different links and the library issues them on different streams); the headline figure keeps the round-4 / 5 assumption
(a piece occupies "the links" for both phases) so that rounds stay comparable.  Pieces are priced with their MEASURED times
(profile field piece_ms) and widths (piece_frac).
Round 6 (profile field fuse_la): the look-ahead update with the LAST sub-panel of J rides in the first panel launch of J+1, so
it is part of fac[J+1] (priced as equal pieces -- the first launch is in fact the heaviest, so the first sub-panel leaves a
little later and the last a little earlier than modelled; the end of the factorisation, which is what the chain waits for, is
exact).
usage: python tools/multi_projection.py profile.json [link_GBps=77] [contend=1.0]"""
import json
import sys


def project(prof, P, link_gbps, form, contend, free_overlap=False, phases=False):
    rows = prof["per_panel"]
    G = int(prof.get("group", 1))
    npan = len(rows)
    fac = [r[0] for r in rows]
    la = [r[1] for r in rows]
    byt = [r[2] for r in rows]
    cls = [[r[3 + 3 * i: 6 + 3 * i] for i in range(P)] for r in rows]   # [J][i] = (near_a, near_b, far)

    def transport(b):
        if P == 1:
            return 0.0
        if form == "allgather" and P > 2:
            return 2.0 * b / ((P - 1) * link_gbps * 1e9) * 1e3
        return b / (link_gbps * 1e9) * 1e3

    owners = prof.get("owners")       # round 5: the ownership table of the run (own_table.h); older profiles: the cyclic deal
    owner = (lambda J: owners[J]) if owners else (lambda J: J % P)
    t_upd = [0.0] * P      # the update stream of rank i is free
    t_near = [0.0] * P     # the near stream
    t_pan = [0.0] * P      # the panel stream
    ev_a = [0.0] * P       # completion of the latest near-A launch
    ev_b = [0.0] * P       # ... near-B
    busy = [0.0] * P

    def run(i, stream, ready, dur):
        """a launch of `dur` ms on `stream` of GPU i that may start at `ready`: returns its end; lower streams that are busy
        at that time are pushed back by `dur` (dispatch priority), idle ones cannot start before it ends"""
        nonlocal t_upd, t_near, t_pan
        busy[i] += dur
        if stream == "panel":
            start = max(ready, t_pan[i])
            end = start + dur
            t_pan[i] = end
            if free_overlap:      # the round-3 model: the look-ahead work costs the owner's other streams nothing
                return end
            for t in (t_near, t_upd):
                t[i] = t[i] + dur if t[i] > start else max(t[i], end)
            return end
        if stream == "near":
            start = max(ready, t_near[i], 0.0 if free_overlap else t_pan[i])
            end = start + dur
            t_near[i] = end
            t_upd[i] = t_upd[i] + dur if t_upd[i] > start else max(t_upd[i], end)
            return end
        start = max(ready, t_upd[i], t_near[i], 0.0 if free_overlap else t_pan[i])
        end = start + dur
        t_upd[i] = end
        return end

    nsub = prof.get("nsub") or [1] * npan
    fused_la = bool(prof.get("fuse_la"))
    piece_ms = prof.get("piece_ms")
    piece_frac = prof.get("piece_frac")

    def factor_and_send(J, start_ready, scale):
        """factor panel J on its owner from `start_ready` on; returns (end of the factorisation, arrival time of every
        sub-panel at the other ranks)"""
        o = owner(J)
        ns = nsub[J]
        # time share of every piece: measured (round 6: profile field piece_ms) or equal; byte share: its width (uneven pieces)
        pm = piece_ms[J][:ns] if piece_ms else None
        tfrac = [v / sum(pm) for v in pm] if pm and sum(pm) > 0 else [1.0 / ns] * ns
        bfrac = piece_frac if (piece_frac and len(piece_frac) == ns) else [1.0 / ns] * ns
        arr, t, link_free, p1_free, p2_free = [], start_ready, 0.0, 0.0, 0.0
        for q in range(ns):
            t = run(o, "panel", t, fac[J] * scale * tfrac[q])
            if phases and form == "allgather" and P > 2:
                # round 6: the two phases of scatter + all-gather use DIFFERENT links (owner -> peer, then peer -> peer: xGMI is
                # a full mesh of point-to-point links) and the library issues them on different streams (one incoming stream
                # per source rank), so the gather of piece q runs under the scatter of piece q + 1
                ph = byt[J] * bfrac[q] / ((P - 1) * link_gbps * 1e9) * 1e3
                e1 = max(t, p1_free) + ph
                p1_free = e1
                e2 = max(e1, p2_free) + ph
                p2_free = e2
                arr.append(e2)
            else:
                link_free = max(t, link_free) + transport(byt[J]) * bfrac[q]
                arr.append(link_free)
        return t, arr

    fact_done, arr = factor_and_send(0, 0.0, 1.0)
    recv = [fact_done if i == owner(0) else arr[-1] for i in range(P)]
    waited = 0.0
    for J in range(npan):
        nxt = J + 1
        new_recv = None
        if nxt < npan:
            o = owner(nxt)
            ready = max(ev_a[o], ev_b[o] if (nxt % G == 0 or J % G == 0) else 0.0)
            loaded = t_upd[o] > recv[o] or t_near[o] > recv[o]
            c = contend if loaded else 1.0
            t = ready
            nq = nsub[J]
            for q in range(nq):       # the look-ahead update follows the sub-panels of J as they land
                landed = arr[q] if o != owner(J) else fact_done
                if fused_la:
                    # round 6: the LAST piece rides in nxt's first panel launch (its cost is part of fac[nxt]); la[nxt] is the
                    # sum of the nq - 1 earlier pieces
                    if q == nq - 1:
                        t = max(t, landed)
                    else:
                        bf = piece_frac if (piece_frac and len(piece_frac) == nq) else [1.0 / nq] * nq
                        t = run(o, "panel", max(t, landed), la[nxt] * c * bf[q] / sum(bf[:nq - 1]))
                else:
                    t = run(o, "panel", max(t, landed), la[nxt] * c / nq)
            fd, new_arr = factor_and_send(nxt, t, c)
            new_recv = [fd if i == o else new_arr[-1] for i in range(P)]
        for i in range(P):
            a, b, f = cls[J][i]
            if a > 0:
                ev_a[i] = run(i, "near", max(recv[i], ev_b[i] if J % G == 0 else 0.0), a)
            idle_until = max(t_upd[i], t_near[i], t_pan[i])
            if b > 0 or f > 0:
                waited += max(0.0, recv[i] - idle_until)
            if b > 0:
                ev_b[i] = run(i, "upd", recv[i], b)
            else:
                ev_b[i] = max(ev_b[i], min(t_upd[i], recv[i]))
            if f > 0:
                run(i, "upd", recv[i], f)
        if new_recv is not None:
            recv = new_recv
            fact_done = fd
            arr = new_arr
    total = max(max(t_upd), max(t_near), max(t_pan), fact_done)
    return total, waited, max(busy)


def project_rowchain(prof, P, link_gbps, ineff=1.25, latency_ms=0.02):
    """WHAT-IF, not a schedule the library has (docs/08_scope_and_next.md, section 9.1): the panel chain distributed over the
    ranks BY ROWS.  From a profile of the unfused chain (fuse_la = 0: factor_ms and lookahead_update_ms measured apart):
      top[J]   = the factorisation of the panel's w x w top block on its owner -- priced with the LAST panel's factor_ms (that
                 panel is its top block and the bordered rows);
      slice[J] = (factor_ms[J] - top + lookahead_update_ms[J]) / P * ineff -- every rank updates and solves 1 / P of the rows
                 below the top block (smaller launches: `ineff`);
      the owner scatters the raw rows ahead of time (same bytes as today's scatter of solved slabs: off the chain), broadcasts the
      factored top block (w x w doubles + latency), every rank sends its solved slice to the P - 1 others (bytes / P per link).
    Panel J+1's chain starts when panel J has been gathered everywhere; a rank's chain work runs on its panel stream ahead of
    its updates (the GPU-as-one-server model of project()).  Updates: as measured, released by the gather of their panel."""
    rows = prof["per_panel"]
    G = int(prof.get("group", 1))
    npan = len(rows)
    fac = [r[0] for r in rows]
    la = [r[1] for r in rows]
    byt = [r[2] for r in rows]
    cls = [[r[3 + 3 * i: 6 + 3 * i] for i in range(P)] for r in rows]
    owners = prof.get("owners")
    owner = (lambda J: owners[J]) if owners else (lambda J: J % P)
    top = min(fac[-1], min(fac))
    W = int(prof["panel_width"])
    bc = 8.0 * W * W / (link_gbps * 1e9) * 1e3 + latency_ms
    t_upd, t_near, t_pan = [0.0] * P, [0.0] * P, [0.0] * P
    ev_a, ev_b, busy = [0.0] * P, [0.0] * P, [0.0] * P

    def run(i, stream, ready, dur):
        busy[i] += dur
        if stream == "panel":
            start = max(ready, t_pan[i])
            end = start + dur
            t_pan[i] = end
            for t in (t_near, t_upd):
                t[i] = t[i] + dur if t[i] > start else max(t[i], end)
            return end
        if stream == "near":
            start = max(ready, t_near[i], t_pan[i])
            end = start + dur
            t_near[i] = end
            t_upd[i] = t_upd[i] + dur if t_upd[i] > start else max(t_upd[i], end)
            return end
        start = max(ready, t_upd[i], t_near[i], t_pan[i])
        end = start + dur
        t_upd[i] = end
        return end

    def chain(J, ready):
        """panel J: every rank holds its raw row slice and (J > 0) the whole of panel J - 1 at `ready`; returns the time panel J
        has been gathered everywhere"""
        o = owner(J)
        work = max(0.0, fac[J] - top) + (la[J] if J > 0 else 0.0)
        sl = work / P * ineff
        share_la = (la[J] / max(work, 1e-12)) if J > 0 else 0.0
        t_top = run(o, "panel", ready, top) + bc                     # the owner: its top block first
        done = 0.0
        for i in range(P):
            t = run(i, "panel", ready, sl * share_la)               # the slice's look-ahead update needs nothing from the owner
            t = run(i, "panel", max(t, t_top), sl * (1.0 - share_la))   # its solve needs the factored top block
            done = max(done, t)
        return done + byt[J] / P / (link_gbps * 1e9) * 1e3 + latency_ms

    recv_t = chain(0, 0.0)
    for J in range(npan):
        nxt = J + 1
        new_recv = None
        if nxt < npan:
            # the raw slices of nxt leave its owner when its updates through J - 1 are in (near A of the previous step)
            o = owner(nxt)
            raw = max(ev_a[o], ev_b[o] if (nxt % G == 0 or J % G == 0) else 0.0) + byt[nxt] / P / (link_gbps * 1e9) * 1e3
            new_recv = chain(nxt, max(recv_t, raw))
        for i in range(P):
            a, b, f = cls[J][i]
            if a > 0:
                ev_a[i] = run(i, "near", max(recv_t, ev_b[i] if J % G == 0 else 0.0), a)
            if b > 0:
                ev_b[i] = run(i, "upd", recv_t, b)
            else:
                ev_b[i] = max(ev_b[i], min(t_upd[i], recv_t))
            if f > 0:
                run(i, "upd", recv_t, f)
        if new_recv is not None:
            recv_t = new_recv
    return max(max(t_upd), max(t_near), max(t_pan), recv_t), max(busy), top


def main():
    prof = json.load(open(sys.argv[1]))
    link = float(sys.argv[2]) if len(sys.argv) > 2 else 77.0
    contend = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    P = prof["ranks"]
    rows = prof["per_panel"]
    if len(rows[0]) != 3 + 3 * P:
        raise SystemExit("this is a round-3 profile (3 + P columns): use the tool of that round (git show 14a0e1b:tools/multi_projection.py)")
    one_gpu = sum(r[0] + r[1] + sum(r[3:]) for r in rows)
    per_rank = [[sum(r[3 + 3 * i + c] for r in rows) for c in range(3)] for i in range(P)]
    print(f"config {prof['config']}  N {prof['N']}  ranks {P}  panels {prof['panels']} (first width {prof['panel_width']}), "
          f"{prof.get('group', 1)} panels per update group")
    print(f"sum of all serialised kernel groups (= one GPU doing everything, no overlap): {one_gpu:.1f} ms")
    print(f"panel factorisations {sum(r[0] for r in rows):.1f} ms, look-ahead updates {sum(r[1] for r in rows):.1f} ms, "
          f"bytes per receiver {sum(r[2] for r in rows) / 1e9:.2f} GB")
    print("updates per rank, ms (near a / near b / far): " + "  ".join("%.1f/%.1f/%.1f" % tuple(v) for v in per_rank))
    owners = prof.get("owners") or [J % P for J in range(len(rows))]
    tot = [sum(per_rank[i]) + sum(r[0] + r[1] for J, r in enumerate(rows) if owners[J] == i) for i in range(P)]
    upd = [sum(v) for v in per_rank]
    mean_u, mean_t = sum(upd) / P, sum(tot) / P
    print(f"panel ownership: {prof.get('ownership', 'cyclic')}; update sums per rank {min(upd):.1f} .. {max(upd):.1f} ms "
          f"({100 * (min(upd) / mean_u - 1):+.1f} % / {100 * (max(upd) / mean_u - 1):+.1f} % of their mean {mean_u:.1f}); all work per "
          f"rank (updates + its panels' factorisations and look-ahead updates) {min(tot):.1f} .. {max(tot):.1f} ms "
          f"({100 * (min(tot) / mean_t - 1):+.1f} % / {100 * (max(tot) / mean_t - 1):+.1f} %)")
    for form in ("allgather", "direct"):
        for c in sorted({contend, 1.0, 1.3}):
            t, w, b = project(prof, P, link, form, c)
            t3, _, _ = project(prof, P, link, form, c, free_overlap=True)
            tp, _, _ = project(prof, P, link, form, c, phases=True)
            print(f"  {form:9s} link {link:.0f} GB/s contend {c:.2f}: projected {t:8.1f} ms  (busiest GPU {b:6.1f} ms of work; "
                  f"update streams idle waiting for panels: {w:6.1f} ms summed over ranks)   "
                  f"[round-3 model, look-ahead overlaps the owner's updates for free: {t3:.1f} ms]"
                  + (f"   [scatter / gather phases of successive pieces pipelined: {tp:.1f} ms]" if form == "allgather" else ""))
    t_inf, _, _ = project(prof, P, 1e9, "direct", 1.0)
    print(f"  infinite link bandwidth: {t_inf:.1f} ms  -> compute / critical-path bound of this schedule")
    if not prof.get("fuse_la") and P > 2:
        for ineff in (1.0, 1.25, 1.5):
            t, b, top = project_rowchain(prof, P, link, ineff)
            print(f"  WHAT-IF (not built, docs/08 section 9.1) panel chain distributed by rows, slice launches {ineff:.2f} x the full-height "
                  f"cost: {t:8.1f} ms  (busiest GPU {b:6.1f} ms of work; top block {top:.2f} ms per panel)")


if __name__ == "__main__":
    main()
