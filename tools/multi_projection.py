"""8-GPU critical-path projection of the sharded factorisation from per-panel timings measured with P loopback ranks on
one GPU (tools/gpu_multi_profile.py -> JSON).  Model (one-panel look-ahead, as csrc/multi.hip schedules it):

  fact_done[J+1] = max(recv[J] at owner(J+1), upd_done[J-1] at owner(J+1)) + lookahead_update[J+1] + factor[J+1]
  recv[J+1][i]   = fact_done[J+1] + transport(bytes[J+1])                (i != owner; the owner has it at once)
  upd_done[J][i] = max(recv[J][i], upd_done[J-1][i], [owner(J+1): its panel stream shares the GPU]) + rest_update[J][i]

transport(bytes): scatter + all-gather over point-to-point xGMI links of `link` GB/s per direction -- two phases of
bytes / (P - 1) each over disjoint links: 2 * bytes / ((P - 1) * link); direct copy: bytes / link (every receiver
pulls the whole panel over its one link to the owner, the owner's P - 1 egress links run in parallel).
On the owner of the next panel the look-ahead update + factorisation share the GPU with its rest updates; `contend`
multiplies both while they overlap (1.0 = perfect overlap, the measured single-GPU figure is ~1.3 for the updates).
usage: python tools/multi_projection.py profile.json [link_GBps=77] [contend=1.0]"""
import json
import sys


def project(prof, P, link_gbps, form, contend):
    rows = prof["per_panel"]
    npan = len(rows)
    fac = [r[0] for r in rows]
    la = [r[1] for r in rows]
    byt = [r[2] for r in rows]
    rest = [r[3:3 + P] for r in rows]

    def transport(b):
        if P == 1:
            return 0.0
        if form == "allgather" and P > 2:
            return 2.0 * b / ((P - 1) * link_gbps * 1e9) * 1e3
        return b / (link_gbps * 1e9) * 1e3

    owner = lambda J: J % P
    upd_done = [0.0] * P                  # end of each rank's update stream
    fact_done = fac[0]
    recv = [fact_done + (0.0 if i == owner(0) else transport(byt[0])) for i in range(P)]
    exposed_transport = 0.0
    panel_busy = [0.0] * P
    for J in range(npan):
        nxt = J + 1
        if nxt < npan:
            o = owner(nxt)
            start = max(recv[o], upd_done[o] if J > 0 else 0.0)
            fd = start + (la[nxt] + fac[nxt]) * contend
            panel_busy[o] += (la[nxt] + fac[nxt]) * contend
            new_recv = [fd + (0.0 if i == o else transport(byt[nxt])) for i in range(P)]
        for i in range(P):
            s = max(recv[i], upd_done[i])
            c = contend if (nxt < npan and i == owner(nxt)) else 1.0
            upd_done[i] = s + rest[J][i] * c
        if nxt < npan:
            # how long the slowest rank waited for the panel beyond the end of its own work
            exposed_transport += max(0.0, max(new_recv[i] - upd_done[i] for i in range(P)))
            recv = new_recv
            fact_done = fd
    total = max(max(upd_done), fact_done)
    return total, exposed_transport


def main():
    prof = json.load(open(sys.argv[1]))
    link = float(sys.argv[2]) if len(sys.argv) > 2 else 77.0
    contend = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    P = prof["ranks"]
    rows = prof["per_panel"]
    one_gpu = sum(r[0] + r[1] + sum(r[3:3 + P]) for r in rows)
    print(f"config {prof['config']}  N {prof['N']}  ranks {P}  panels {prof['panels']} x {prof['panel_width']}")
    print(f"sum of all serialised kernel groups (= one GPU doing everything, no overlap): {one_gpu:.1f} ms")
    print(f"panel factorisations {sum(r[0] for r in rows):.1f} ms, look-ahead updates {sum(r[1] for r in rows):.1f} ms, "
          f"rest updates per rank {[round(sum(r[3 + i] for r in rows), 1) for i in range(P)]} ms, "
          f"bytes per receiver {sum(r[2] for r in rows) / 1e9:.2f} GB")
    for form in ("allgather", "direct"):
        for c in sorted({contend, 1.0, 1.3}):
            t, ex = project(prof, P, link, form, c)
            print(f"  {form:9s} link {link:.0f} GB/s contend {c:.2f}: projected {t:8.1f} ms  (waiting for panels: {ex:6.1f} ms)")
    t_inf, _ = project(prof, P, 1e9, "direct", 1.0)
    print(f"  infinite link bandwidth: {t_inf:.1f} ms  -> compute/critical-path bound of this schedule")


if __name__ == "__main__":
    main()
