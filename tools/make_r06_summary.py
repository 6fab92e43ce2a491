"""gpurun_out/r06final (tools/collect_r06.sh, ONE gpurun call) -> profiles/r06_*: bench lines, rocprofv3 kernel stats, the
HBM-side traffic of every line's dominant kernel (profiles/r06_traffic.json -- what bench.py prints as roofline.traffic), the
multi-GPU profiles / projections, and profiles/r06_summary.md.  Refuses a collection whose pieces were not produced by ONE
build of libsthenomi.so (every record carries the library's sha1)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06final")
DST = os.path.join(ROOT, "profiles")
TAG = "r06"
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (update_bytes_avg: the algorithmic bytes of the trailing updates)

sha = open(os.path.join(SRC, "lib_sha1.txt")).read().strip()


def jload(name):
    p = os.path.join(SRC, name)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        return None
    d = json.load(open(p))
    if d.get("lib_sha1") != sha:
        raise SystemExit(f"{name}: produced by a different build of libsthenomi.so ({d.get('lib_sha1')} != {sha}) -- re-collect")
    return d


DOMINANT = {   # config -> substring of the dominant kernel's name
    "c5": "gemm_nt_dma_potrf_kernel<1", "target": "gemm_nt_dma_potrf_kernel<1", "n4k": "chol_dataflow", "c2": "chol_dataflow", "c3": "chol_dataflow",
    "n32k": "chol_dataflow", "c4": "gemm_nt_dma_kernel<0>",
}
lines, traffic = {}, {}
for c in ("default", "c1", "n4k", "c2", "c3", "n32k", "c4", "target", "c3_dense", "target_dense", "target_multi8_loopback", "c5_f32", "c5_multi8_loopback", "c5_multi8_loopback_sub0",
          "c4_multi2_loopback", "c5_dist1", "c5_nohybrid", "target_nohybrid", "n32k_nohybrid", "c3_nohybrid",
          "c5_multi8_loopback_r5chain", "target_multi8_loopback_r5chain"):
    d = jload(f"bench_{c}.json")
    if d:
        lines[c] = d
for c, pat in DOMINANT.items():
    f_, w_ = jload(f"pmc_{c}_FETCH_SIZE.json"), jload(f"pmc_{c}_WRITE_SIZE.json")
    if not f_ or not w_:
        continue
    sched = ((lines.get("default" if c == "c5" else c) or {}).get("roofline") or {}).get("schedule")
    if sched == "hybrid":      # the trailing updates of the hybrid schedule are plain launches (the panels: chol_dataflow_fat_kernel)
        pat = DOMINANT[c] = "gemm_nt_dma_kernel<1>"
    ks = [k for k in f_["kernels"] if pat in k]
    if not ks:
        continue
    fk = sum(f_["kernels"][k]["sum_KiB"] for k in ks)
    n = sum(f_["kernels"][k]["launches"] for k in ks)
    wk = sum(w_["kernels"].get(k, {"sum_KiB": 0.0})["sum_KiB"] for k in ks)
    line = lines.get("default" if c == "c5" else c)
    N = line["config"]["N"] if line else None
    per_step = c == "c4"                      # many launch shapes: the figure is per STEP of the bound, not per launch
    div = 1 if per_step else max(1, n)
    hbm = (2.0 * fk + wk) * 1024.0 / div
    if c == "c5" or sched == "hybrid" or c == "target":
        # (structured models: the launches skip the structurally dead tiles -- their share of the dense launches' bytes)
        alg = bench.update_bytes_avg(N, sched) * float(line.get("executed_work_fraction") or 1.0)
    elif c == "c4":
        alg = line["roofline"]["hbm_stage"]["algorithmic_bytes"] * 4.0    # K(x,z) written, read, A written, read
    else:
        alg = 8.0 * N * (N + 1)
    traffic[c] = {"kernel": ks if len(ks) > 1 else ks[0], "launches_profiled": n, "per": "step" if per_step else "launch",
                  "FETCH_SIZE_KiB": fk / div, "WRITE_SIZE_KiB": wk / div, "hbm_bytes": hbm,
                  "hbm_bytes_note": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per the gfx950 correction for 16 B / lane "
                                    "streaming reads (MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted: an upper bound",
                  "algorithmic_bytes": alg, "ratio_to_algorithmic": hbm / alg, "lib_sha1": sha,
                  "schedule": (line or {}).get("roofline", {}).get("schedule"),
                  "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --config {c} --steps 1 --warmup 0 (tools/collect_r06.sh)"}
json.dump(traffic, open(os.path.join(DST, f"{TAG}_traffic.json"), "w"), indent=1)
for c in ("c5", "target", "c3"):
    p = os.path.join(SRC, f"pmc_{c}_MFMA.json")
    if os.path.exists(p) and os.path.getsize(p):
        d = json.load(open(p))
        assert d["lib_sha1"] == sha
        out = {}
        for k, v in d["kernels"].items():
            if (DOMINANT[c] in k or "chol_dataflow" in k) and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                out[k] = {"launches": v["GRBM_GUI_ACTIVE"]["launches"],
                          "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (v["GRBM_GUI_ACTIVE"]["sum"] * 128.0)}
        json.dump({"lib_sha1": sha, "note": "SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 1024) -- the round-3 normalisation", "kernels": out},
                  open(os.path.join(DST, f"{TAG}_mfma_{c}.json"), "w"), indent=1)
for c, d in lines.items():
    json.dump(d, open(os.path.join(DST, f"{TAG}_bench_{c}.json"), "w"), indent=1)
for c in ("c5", "target", "c3", "c2", "n4k", "c1", "c4", "c5_f32", "grad_c2"):
    st = glob.glob(os.path.join(SRC, f"prof_{c}", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(DST, f"{TAG}_bench_{c}_kernel_stats.csv"))
for f in (glob.glob(os.path.join(SRC, "multi_profile_*.json")) + glob.glob(os.path.join(SRC, "projection_*.txt")) +
          glob.glob(os.path.join(SRC, "trace_busy_*.json")) + glob.glob(os.path.join(SRC, "batch_time.json")) +
          glob.glob(os.path.join(SRC, "midn_sweep.txt")) + glob.glob(os.path.join(SRC, "f32_sz_time.json"))):
    shutil.copy(f, os.path.join(DST, f"{TAG}_" + os.path.basename(f)))

# ---- summary
L = [f"# Round 6 -- collected evidence (one `gpurun` call, `tools/collect_r06.sh`; libsthenomi.so sha1 `{sha[:12]}`)", ""]
py = open(os.path.join(SRC, "pytest_gpu.log")).read().strip().splitlines()
L += ["* `pytest -m gpu`: " + next((ln for ln in reversed(py) if "passed" in ln or "failed" in ln), "?"),
      "* smoke: " + open(os.path.join(SRC, "smoke.log")).read().strip().splitlines()[-1], ""]
L += ["| line (`bench.py --config`) | ms / step | whole step, fraction of the fp64 MFMA peak | dominant kernel in situ (`roofline.frac`) | "
      "fabric traffic of that kernel (counters) vs algorithmic | parity vs CPU golden |", "|---|---|---|---|---|---|"]


def row(name, d, key):
    rf = d.get("roofline") or {}
    tr = traffic.get(key)
    trs = "--" if not tr else f"{tr['hbm_bytes'] / 1e9:.1f} GB per {tr['per']} = {tr['ratio_to_algorithmic']:.1f} x"
    fr = d.get("cholesky_frac_of_fp64_matrix_peak")
    wf = d.get("executed_work_fraction")
    if wf is not None and wf < 0.999:   # structural zeros: the fractions are on the flops that ran
        name = f"{name} [structural zeros: {wf:.3f} of the dense tile products ran; dense-equivalent {d.get('dense_equivalent_tflops', 0):.1f} TFLOP/s]"
    extra = ""
    if rf.get("schedule") == "hybrid":      # in situ / flops over the union of the launch intervals / the same launches alone
        wb = rf.get("achieved_while_busy")
        un = (rf.get("uncontended") or {}).get("frac")
        extra = (" [hybrid: in situ" + (f"; while busy {wb / rf['peak']:.3f}" if wb else "") + (f"; uncontended {un:.3f}" if un else "") + "]")
    return (f"| {name} | {d['ms_per_step']:.2f} | {fr:.3f} | {rf.get('frac', float('nan')):.3f}{extra} ({str(rf.get('kernel', ''))[:60]}...) | {trs} | "
            f"{d['parity_rel']:.1e} |" if fr is not None else
            f"| {name} | {d['ms_per_step']:.2f} | -- | {rf.get('frac', float('nan')):.3f} | {trs} | {d['parity_rel']:.1e} |")


if "default" in lines:
    d = lines["default"]
    L.append(row("c5 (the default line)", d, "c5"))
    ns = d.get("north_star_target")
    if ns:
        if ns.get("executed_work_fraction", 1.0) < 0.999:
            L.append(f"| ... (next row: structural zeros, {ns['executed_work_fraction']:.3f} of the dense tile products ran; the fraction is on those; "
                     f"dense-equivalent {ns.get('dense_equivalent_tflops', 0):.1f} TFLOP/s) | | | | | |")
        L.append(f"| ... its `north_star_target` extra (3-process @gppp, N = 65536, host-buffer entry point) | {ns['ms_per_step']:.2f} | "
                 f"{ns['frac']:.3f} | | | {ns['parity_rel']:.1e} |")
    r4 = (d.get("roofline") or {}).get("round4_schedule_same_box")
    if r4:
        L.append(f"| ... its `roofline.round4_schedule_same_box` leg (SGP_HYBRID=0, second context, same buffers; same bits: {r4['same_bits']}) | "
                 f"{r4['ms_per_step']:.2f} | | hybrid / this = {r4['hybrid_over_this']:.3f} | | |")
    for k, v in (d.get("sizes") or {}).items():
        if "members" in v:
            L.append(f"| ... its `sizes.{k}` extra: sgp_logpdf_batch, {v['members']} members, one task pool | {v['ms_per_call']:.3f} per call = "
                     f"{v['ms_per_member']:.3f} per member | {v['frac']:.3f} aggregate | every member bit-equal to its own call: "
                     f"{v['every_member_bit_equal_to_its_own_call']} | | {v['member0_parity_rel']:.1e} (member 0) |")
        else:
            L.append(f"| ... its `sizes.{k}` extra | {v['ms_per_step']:.3f} | {v['frac']:.3f} | ({v['schedule']}) | | {v['parity_rel']:.1e} |")
for c in ("c5_nohybrid", "target", "target_nohybrid", "target_dense", "c3", "c3_nohybrid", "c3_dense", "n32k", "n32k_nohybrid", "c2", "n4k", "c1",
          "c4", "c5_f32"):
    if c in lines:
        L.append(row(c + (" (SGP_STRUCT_ZEROS=0: the dense schedule, same bits)" if c.endswith("_dense") else
                          " (SGP_HYBRID=0: the round-4 schedule on the same box, same bits)" if c.endswith("_nohybrid") else ""), lines[c],
                     c if not (c.endswith("_dense") or c.endswith("_nohybrid")) else "none"))
L.append("")
d0 = lines.get("default")
if d0 and d0.get("grad"):
    L += ["## The gradient (`grad` extra of the default line: `sgp_logpdf_grad`, host buffers; goldens: tests/golden/grad_configs.json)", "",
          "| config | ms / call | x logpdf | fraction of the fp64 MFMA peak on N^3 flops | executed share of the bordered factorisation | d/d sigma^2 vs golden | d/d inscale vs golden | logpdf vs golden |",
          "|---|---|---|---|---|---|---|---|"]
    for k, v in d0["grad"].items():
        f = lambda x: "--" if x is None else f"{x:.1e}"
        frs = f"{v['frac']:.3f}" if v.get('frac') is not None else f"(dense-equivalent {v.get('dense_equivalent_frac', 0):.3f})"
        L.append(f"| {k} (N = {v['N']}) | {v['ms_per_call']:.2f} | {v['ratio_to_logpdf']:.2f} | {frs} | "
                 f"{v.get('executed_work_fraction_of_the_bordered_factorisation', 1.0):.3f} | {f(v.get('parity_rel_d_sigma2'))} | "
                 f"{f(v.get('parity_rel_d_inscale'))} | {f(v.get('parity_rel_logpdf'))} |")
    L.append("")
if d0 and d0.get("cpu_baseline"):
    cb = d0["cpu_baseline"]
    L += ["## CPU baseline of the default line (a restatement, NOT Julia; a reported number, not credit)", "",
          f"* {cb.get('sample')}", f"* value {cb['value']:.5f} {cb['unit']} on {cb['cores']} threads of {cb.get('host_cores')} logical / "
          f"{cb.get('physical_cores_visible')} physical cores; Cholesky sweep (GFLOP/s at n = {cb.get('cholesky_sweep_n')}): `{json.dumps(cb.get('cholesky_sweep_gflops'))}`", ""]
for c in ("c5", "target"):
    for suffix, what in (("", "update launches (gemm_nt_dma_kernel<1>)"), ("_panel_kernel", "panel launches (chol_dataflow_fat_kernel)")):
        p = os.path.join(SRC, f"trace_busy_{c}{suffix}.json")
        if os.path.exists(p) and os.path.getsize(p):
            tb = json.load(open(p))
            ms_ = tb.get("median_step") or {}
            if ms_:
                fr_ = tb.get("achieved_while_busy_frac_of_78.6")
                L.append(f"* rocprofv3 kernel trace of `{c}`, {what}: {ms_['dispatches_of_kernel']} dispatches per step, union of their intervals "
                         f"{ms_['busy_ms']:.1f} ms, sum of their durations {ms_['sum_of_durations_ms']:.1f} ms, step span {ms_['step_span_ms']:.1f} ms"
                         + (f" => {fr_:.3f} of the fp64 MFMA peak while the kernel is on the chip" if fr_ and not suffix else ""))
L.append("")
for c in ("c5_multi8_loopback", "c5_multi8_loopback_r5chain", "c5_multi8_loopback_sub0", "target_multi8_loopback", "target_multi8_loopback_r5chain",
          "c4_multi2_loopback", "c5_dist1"):
    if c in lines:
        d = lines[c]
        mg = d.get("multi_gpu") or {}
        L.append(f"* `{c}`: {d['ms_per_step']:.1f} ms per step, ranks {d.get('ranks')}, physical GPUs {d['n_gpus']}, parity {d['parity_rel']:.1e}"
                 + (f", host enqueue {mg.get('host_enqueue_ms', 0):.1f} ms of {mg.get('last_call_ms', 0):.1f} ms" if mg else ""))
L += ["", "## 8-GPU projections from the serialised loopback profiles (`tools/multi_projection.py`)", ""]
for f in sorted(glob.glob(os.path.join(SRC, "projection_*.txt"))):
    L += [f"`{os.path.basename(f)}`", "", "```"] + open(f).read().rstrip().splitlines() + ["```", ""]
open(os.path.join(DST, f"{TAG}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:40]))
